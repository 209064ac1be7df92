"""Vectorised packer / unpacker for the Marlin/QQQ int4 layout (torch; runs on CPU or GPU).

Closed form of what QuantLinear.pack produces with python loops (qlinear_marlin.py:147-176, :228-248):
word B[kt][128*ng + 16*c + 4*kq + jt] holds k = 16*kt + 4*kq + r, n = 64*ng + 16*jt + 8*b + c;
nibble p of the word is (b, r) = (1-(p&1), p>>1) per-channel, ((p&3)>>1, 2*(p&1)+(p>>2)) per-group.
"""
from __future__ import annotations

import torch

_B_OF_P = {False: [1, 0, 1, 0, 1, 0, 1, 0], True: [0, 0, 1, 1, 0, 0, 1, 1]}
_R_OF_P = {False: [0, 0, 1, 1, 2, 2, 3, 3], True: [0, 2, 0, 2, 1, 3, 1, 3]}


def pack_codes(codes: torch.Tensor, grouped: bool) -> torch.Tensor:
    """codes [K,N] integer (signed int4 per-channel / unsigned u per-group) -> int32 [K/16, 2N]."""
    K, N = codes.shape
    assert K % 16 == 0 and N % 64 == 0
    dev = codes.device
    t = (codes.to(torch.int64) & 0xF).reshape(K // 16, 4, 4, N // 64, 4, 2, 8)  # kt,kq,r,ng,jt,b,c
    t = t.permute(0, 3, 6, 1, 4, 5, 2)  # kt,ng,c,kq,jt,b,r
    bi = torch.tensor(_B_OF_P[grouped], device=dev)
    ri = torch.tensor(_R_OF_P[grouped], device=dev)
    nib = t[..., bi, ri]  # kt,ng,c,kq,jt,p
    shifts = 4 * torch.arange(8, device=dev, dtype=torch.int64)
    w = (nib << shifts).sum(-1)
    w = torch.where(w >= 2**31, w - 2**32, w).to(torch.int32)
    return w.reshape(K // 16, 2 * N).contiguous()


def unpack_codes(B: torch.Tensor, grouped: bool) -> torch.Tensor:
    """int32 [K/16, 2N] -> int8 codes [K,N]."""
    KT, W = B.shape
    N = W // 2
    dev = B.device
    w = B.to(torch.int64) & 0xFFFFFFFF
    shifts = 4 * torch.arange(8, device=dev, dtype=torch.int64)
    nib = ((w.reshape(KT, N // 64, 8, 4, 4, 1) >> shifts) & 0xF)  # kt,ng,c,kq,jt,p
    out = torch.empty((KT, N // 64, 8, 4, 4, 2, 4), dtype=torch.int64, device=dev)  # kt,ng,c,kq,jt,b,r
    bi = torch.tensor(_B_OF_P[grouped], device=dev)
    ri = torch.tensor(_R_OF_P[grouped], device=dev)
    out[..., bi, ri] = nib
    if not grouped:
        out = torch.where(out >= 8, out - 16, out)
    out = out.permute(0, 3, 6, 1, 4, 5, 2)  # kt,kq,r,ng,jt,b,c
    return out.reshape(KT * 16, N).to(torch.int8).contiguous()


def s_channel_stored_index(n: torch.Tensor) -> torch.Tensor:
    w = n % 32
    return (n // 32) * 32 + 8 * ((w % 8) // 2) + 2 * (w // 8) + (w % 2)


def s_group_stored_index(n: torch.Tensor) -> torch.Tensor:
    return (n // 64) * 64 + 8 * (n % 8) + (n % 64) // 8


def permute_s_channel(s_logical: torch.Tensor) -> torch.Tensor:
    """logical [.., N] -> stored order (applies `_scale_perm_single`, qlinear_marlin.py:173-175)."""
    N = s_logical.shape[-1]
    out = torch.empty_like(s_logical)
    out[..., s_channel_stored_index(torch.arange(N, device=s_logical.device))] = s_logical
    return out


def permute_s_group(s_logical: torch.Tensor) -> torch.Tensor:
    """logical [G, N] -> stored order (applies `_scale_perm`, qlinear_marlin.py:170-172)."""
    N = s_logical.shape[-1]
    out = torch.empty_like(s_logical)
    out[..., s_group_stored_index(torch.arange(N, device=s_logical.device))] = s_logical
    return out
