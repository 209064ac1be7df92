"""QuantLinear: the MI355X counterpart of QQQ/gptq/qlinear/qlinear_marlin.py:48-288.

Same constructor arguments, same registered buffer names / shapes / dtypes (B, s_channel, s_group,
bias persistent; workspace, reduce_buffer non-persistent) so a QQQ checkpoint's state-dict loads
unchanged, same pack() arguments, same forward() semantics.  Differences, all deliberate:
  * no CUDA-only constructor guards (qlinear_marlin.py:56-63 reject ROCm);
  * pack() uses the native packer (qqq_amd/pack.py -> qqq_pack_int4) instead of python loops, on any device;
  * forward() uses one fused HIP kernel for dynamic_quant instead of ~8 torch launches;
  * expand_for_prefill() (opt-in, default off): a per-group layer may keep a second, NON-persistent copy of its weights
    re-quantised to int8 at load time, which the large-m kernel reads instead of re-quantising inside its loop.
"""
from __future__ import annotations

from logging import getLogger

import torch
import torch.nn as nn

from . import ops, pack as _pack

logger = getLogger(__name__)


class QuantLinear(nn.Module):
    QUANT_TYPE = "marlin"

    def __init__(self, bits, group_size, infeatures, outfeatures, bias, trainable=False, **kwargs):
        super().__init__()
        # (thread_k, thread_n) -- the reference's shape admission rule (qlinear_marlin.py:65-77)
        self.thread_config = [(64, 256), (128, 128), (128, 64), (64, 128)]
        if not any(infeatures % tk == 0 and outfeatures % tn == 0 for tk, tn in self.thread_config):
            raise ValueError("Not supported `infeatures`: {} and `outfeatures`: {}.".format(infeatures, outfeatures))
        if bits not in [4]:
            raise NotImplementedError("Only 4 bits are supported.")
        if group_size not in [-1, 128] and group_size != infeatures:
            raise ValueError("Only group_size -1 and 128 are supported.")
        if trainable:
            raise NotImplementedError("Marlin does not support train.")

        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.group_size = group_size if group_size != -1 else infeatures
        if self.infeatures % self.group_size != 0:
            raise ValueError("`infeatures` must be divisible by `group_size`.")
        self.bits = bits
        self.tile = 16
        self.maxq = 2**self.bits - 1 if self.group_size != self.infeatures else 2 ** (self.bits - 1) - 1
        self.max_par = 16
        self.register_buffer("B", torch.empty((self.infeatures // 16, self.outfeatures * 16 // 8), dtype=torch.int32))
        self.register_buffer("s_channel", torch.empty((1, self.outfeatures), dtype=torch.float32))
        if self.group_size != self.infeatures:
            self.register_buffer(
                "s_group", torch.empty((self.infeatures // self.group_size, self.outfeatures), dtype=torch.half)
            )
        else:
            self.register_buffer("s_group", torch.tensor([], dtype=torch.half))
        self.register_buffer("workspace", torch.zeros(self.outfeatures // 128 * 16, dtype=torch.int32), persistent=False)
        self.register_buffer(
            "reduce_buffer", torch.zeros((self.max_par * 16 * 4, self.outfeatures), dtype=torch.int), persistent=False
        )
        if bias:
            self.register_buffer("bias", torch.zeros((outfeatures), dtype=torch.half))
        else:
            self.bias = None
        self.W8 = None  # expand_for_prefill(): the expanded int8 weights (non-persistent buffer), None = not expanded

    def _apply(self, fn, recurse=True):
        # Keep scale dtypes pinned across .half()/.to() (qlinear_marlin.py:141-145) -- and the bias: the fused epilogue
        # reads it as fp16 bits, and the reference's own `prepare_for_inference` does model.to(bf16/fp32).  Unlike the
        # reference's pin (cast there and back: fp32 -> bf16 -> fp32 rounds the scales to 8 bits), the stored VALUES are
        # preserved: the pinned tensors only follow device moves.
        keep = {"s_group": self.s_group, "s_channel": self.s_channel}
        if self.bias is not None:
            keep["bias"] = self.bias
        super()._apply(fn, recurse=recurse)
        for name, old in keep.items():
            cur = getattr(self, name)
            if cur.dtype != old.dtype:  # a dtype cast: take the original values to wherever the module now lives
                setattr(self, name, old.to(device=cur.device))
        return self

    def post_init(self):
        pass

    @torch.no_grad()
    def pack(self, linear, scales, s_extra=None):
        """Same contract as qlinear_marlin.py:181-262: `linear` fake-quantised fp16 nn.Linear,
        `scales` [outfeatures, groups], `s_extra` [outfeatures, 1] (per-group only)."""
        grouped = self.group_size != self.infeatures
        if grouped:
            assert s_extra is not None, "s_extra is needed"
        if linear.weight.dtype != torch.half:
            logger.warning("The dtype of weights is %s, while the w4a8 GEMM's output is torch.half.", linear.weight.dtype)
        s = scales.t()
        w = linear.weight.data.t()  # [K, N]
        K, N = self.infeatures, self.outfeatures
        if grouped:
            G = K // self.group_size
            s_full = s.reshape(G, 1, N).expand(G, self.group_size, N).reshape(K, N)
            codes = torch.clamp(torch.round(w / s_full).int() + (self.maxq + 1) // 2, 0, self.maxq)
            se = s_extra.reshape(1, -1).to(dtype=torch.float32)
            s_group = (s.reshape(-1, N) / se).to(dtype=torch.half)
            self.s_group[:, :] = _pack.permute_s_group(s_group).to(self.s_group.device)
            self.s_channel[:, :] = _pack.permute_s_channel(se).to(self.s_channel.device)
        else:
            codes = torch.clamp(torch.round(w / s).int(), -self.maxq, self.maxq)
            sc = (s / (2 ** (8 - self.bits))).reshape(1, N)
            self.s_group = torch.tensor([], dtype=torch.half, device=self.s_channel.device)
            self.s_channel[:, :] = _pack.permute_s_channel(sc).to(dtype=torch.float32, device=self.s_channel.device)
        self.B[:, :] = _pack.pack_codes(codes, grouped).to(self.B.device)
        if linear.bias is not None:
            if self.bias is not None:
                self.bias[:] = linear.bias.data.to(self.bias.device).to(torch.half)
            else:
                # a layer built with bias=False that is packed from a biased linear: the reference assigns a plain
                # attribute (qlinear_marlin.py:259-262) that .to(device) does not move; register it as a buffer
                del self.bias
                self.register_buffer("bias", linear.bias.data.clone().to(device=self.B.device, dtype=torch.half))

    @torch.no_grad()
    def expand_for_prefill(self, per_channel: bool = False):
        """Opt-in: keep the weights ALSO as int8 in the large-m kernel's operand order -- `infeatures * outfeatures` bytes of
        device memory, twice the packed tensor, non-persistent (never in a state-dict; call again after loading new weights).
        Calls of a few hundred tokens and more then skip the in-loop re-quantisation of a per-group layer
        (csrc/qqq_gemm.cu:167-210 runs once per weight here, at load time, bit for bit: -21 ... -25 % at 1024 ... 8192 tokens);
        results are bit-identical, small-m calls keep reading `B`.  A per-channel layer has no re-quantiser in its loop, only
        the int4 unpack, and gains a few per cent: it is expanded only with `per_channel=True`.  Returns self."""
        grouped = self.group_size != self.infeatures and self.s_group.numel() != 0
        if not grouped and not per_channel:
            return self
        if not self.B.is_cuda:
            raise RuntimeError("expand_for_prefill: move the layer to the GPU first (there is no CPU path)")
        w8 = ops.expand_int8(self.B, self.s_group)
        if "W8" in self._buffers:
            self._buffers["W8"] = w8
        else:
            if "W8" in self.__dict__:
                del self.W8
            self.register_buffer("W8", w8, persistent=False)
        return self

    def drop_expanded(self):
        """Release the expanded weights of expand_for_prefill()."""
        if "W8" in self._buffers:
            del self._buffers["W8"]
        self.__dict__["W8"] = None
        return self

    def dynamic_quant(self, x: torch.Tensor):
        """Per-token int8 quantisation (qlinear_marlin.py:265-268), one fused HIP kernel."""
        return ops.dynamic_quant(x)

    def forward(self, A):
        # reference: dynamic_quant (:265-268) -> mul (:28-45) -> `D + self.bias` (:287); here one binding call that
        # launches the fused quantiser and the GEMM with the bias in its epilogue (fp16 add after the fp16 round)
        if A.dim() == 2 and A.dtype == torch.half and A.is_contiguous():
            x, out_shape = A, None  # the common serving case: nothing to reshape or cast
        else:
            out_shape = A.shape[:-1] + (self.outfeatures,)
            x = A.reshape(-1, A.shape[-1]).half().contiguous()
        D = ops.quantlinear_forward(x, self.B, self.reduce_buffer, self.s_channel, self.s_group, self.workspace,
                                    self.bias, max_par=self.max_par, W8=self.W8)
        return D if out_shape is None else D.reshape(out_shape)


def fuse_quant_linears(layers) -> QuantLinear:
    """Concatenate QuantLinear layers that share their input along the output dimension (q/k/v, gate/up:
    the reference wraps them as separate modules, gptq/models/llama.py:202-229, :275-283) into ONE layer,
    so one W4A8 GEMM with N = sum(N_i) replaces several (what vLLM does with merged projections).

    The packed layout is made of independent 64-column groups (word index 128*ng + ...), s_channel of
    independent 32-column blocks and s_group of independent 64-column blocks, so concatenating the stored
    tensors along the column axis IS the packing of the concatenated weight: no repacking.
    """
    layers = list(layers)
    first = layers[0]
    for l in layers[1:]:
        if l.infeatures != first.infeatures or l.group_size != first.group_size or l.bits != first.bits:
            raise ValueError("fuse_quant_linears: layers must share infeatures / group_size / bits")
        if (l.bias is None) != (first.bias is None):
            raise ValueError("fuse_quant_linears: either all or none of the layers carry a bias")
    n_total = sum(l.outfeatures for l in layers)
    gs = -1 if first.group_size == first.infeatures else first.group_size
    fused = QuantLinear(first.bits, gs, first.infeatures, n_total, bias=first.bias is not None)
    fused = fused.to(first.B.device)
    fused.B.copy_(torch.cat([l.B for l in layers], dim=1))
    fused.s_channel.copy_(torch.cat([l.s_channel for l in layers], dim=1))
    if first.s_group.numel():
        fused.s_group.copy_(torch.cat([l.s_group for l in layers], dim=1))
    else:
        fused.s_group = torch.tensor([], dtype=torch.half, device=first.B.device)
    if first.bias is not None:
        fused.bias.copy_(torch.cat([l.bias for l in layers], dim=0))
    return fused


__all__ = ["QuantLinear", "fuse_quant_linears"]
