"""Packer / unpacker for the Marlin/QQQ int4 layout (native: HIP kernel on device tensors, threaded C++ on host tensors).

Closed form of what QuantLinear.pack produces with python loops (qlinear_marlin.py:147-176, :228-248):
word B[kt][128*ng + 16*c + 4*kq + jt] holds k = 16*kt + 4*kq + r, n = 64*ng + 16*jt + 8*b + c;
nibble p of the word is (b, r) = (1-(p&1), p>>1) per-channel, ((p&3)>>1, 2*(p&1)+(p>>2)) per-group.
"""
from __future__ import annotations

import torch

from . import _lib


def _native(fn_name: str, src: torch.Tensor, dst: torch.Tensor, K: int, N: int, grouped: bool) -> None:
    L = _lib.lib()
    on_dev = src.is_cuda
    if src.data_ptr() % 8:  # a view at an odd storage offset: the native entry points want 8-byte aligned buffers
        src = src.clone()
    stream = torch.cuda.current_stream(src.device).cuda_stream if on_dev else None
    rc = getattr(L, fn_name)(src.data_ptr(), dst.data_ptr(), K, N, int(grouped), int(on_dev),
                             (src.device.index or 0) if on_dev else 0, stream)
    if rc:
        raise RuntimeError(f"qqq_amd: {fn_name} error {rc}: {_lib.last_error()}")


def pack_codes(codes: torch.Tensor, grouped: bool) -> torch.Tensor:
    """codes [K,N] integer (signed int4 per-channel / unsigned u per-group) -> int32 [K/16, 2N], on codes' device.
    Native: HIP kernel for device tensors, threaded C++ for host tensors (qqq_pack_int4, include/qqq_amd.h)."""
    K, N = codes.shape
    if K % 16 or N % 64:
        raise ValueError("pack_codes: K must be a multiple of 16 and N of 64")
    src = codes.to(torch.int8).contiguous()
    out = torch.empty((K // 16, 2 * N), dtype=torch.int32, device=codes.device)
    if K and N:
        _native("qqq_pack_int4", src, out, K, N, grouped)
    return out


def unpack_codes(B: torch.Tensor, grouped: bool) -> torch.Tensor:
    """int32 [K/16, 2N] -> int8 codes [K,N] (inverse of pack_codes; qqq_unpack_int4)."""
    KT, W = B.shape
    K, N = KT * 16, W // 2
    if W % 128:
        raise ValueError("unpack_codes: B must be [K/16, 2N] with N a multiple of 64")
    src = B.to(torch.int32).contiguous()
    out = torch.empty((K, N), dtype=torch.int8, device=B.device)
    if K and N:
        _native("qqq_unpack_int4", src, out, K, N, grouped)
    return out


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
