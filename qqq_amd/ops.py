"""Operator layer: the reference's `qqq_gemm` signature on top of the C-ABI (include/qqq_amd.h).

`qqq_gemm` is positionally identical to `QQQ._CUDA.qqq_gemm` (csrc/pybind.cpp:3-5,
csrc/qqq_gemm.cu:1048-1106, qqq_gemm.h:23-36) and raises RuntimeError in the same situations with
the same messages.  It is also registered as the torch custom op `qqq_amd::qqq_gemm`
(torch.library), so it does not graph-break under torch.compile.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

ERR_PROB_SHAPE = 1
ERR_KERN_SHAPE = 2
_FORCE_DISPATCHER = os.environ.get("QQQ_AMD_FORCE_DISPATCHER", "0") == "1"


def _load_torch_ext():
    """The compiled binding (csrc/qqq_torch.cpp -> qqq_amd/_torch_ext*.so): the eager fast path -- same C-ABI calls, same checks
    and messages, ~2 us of host time per call instead of ~7.5 through ctypes.  Absent (not built) -> the ctypes binding below
    serves; both end in libqqq_amd.so.  Not used with a QQQ_AMD_LIB override (the module binds the default library)."""
    if os.environ.get("QQQ_AMD_LIB") or os.environ.get("QQQ_AMD_NO_TORCH_EXT") == "1":
        return None
    try:
        _lib.lib()  # the operator library first: the module's NEEDED entry then resolves to the same mapping
        from . import _torch_ext  # type: ignore

        return _torch_ext if _torch_ext.abi_version() == _lib.ABI_VERSION else None
    except Exception:
        return None


_EXT = None
_EXT_TRIED = False


def _ext():
    global _EXT, _EXT_TRIED
    if not _EXT_TRIED:
        _EXT_TRIED = True
        _EXT = _load_torch_ext()
    return _EXT


def _ptr(t: Optional[torch.Tensor]):
    # plain ints: the argtypes declared in _lib.py convert them to void* (cheaper than building c_void_p objects)
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_for(t: torch.Tensor):
    # the raw hipStream_t of t's device; the private fast path saves ~1.5 us of Stream-object construction per call
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_common(A, B, C, D, s1, s2, s3, workspace, max_par):
    # the reference's own checks (csrc/qqq_gemm.cu:1062-1075) ...
    prob_m, prob_n, prob_k = A.size(0), C.size(1), A.size(1)
    groupsize = -1 if s3.numel() == 0 else prob_k // s3.size(0)
    if groupsize != -1 and groupsize * s3.size(0) != prob_k:
        raise RuntimeError(f"k={prob_k} not compatible with {s3.size(0)} groups.")
    if workspace.numel() < prob_n // 128 * max_par:
        raise RuntimeError(f"workspace must be of size at least {prob_n // 128 * max_par}.")
    if s1.dtype != torch.float32:
        raise RuntimeError(f"s1 dtype must be float32, but got {s1.dtype}.")
    if s2.dtype != torch.float32:
        raise RuntimeError(f"s2 dtype must be float32, but got {s2.dtype}.")
    if s3.dtype != torch.float16:
        raise RuntimeError(f"s3 dtype must be float16, but got {s3.dtype}.")
    # ... plus the ones the reference leaves as undefined behaviour (SURVEY 8b "Errors")
    if A.dtype != torch.int8 or B.dtype != torch.int32 or D.dtype != torch.float16 or C.dtype != torch.int32:
        raise RuntimeError("qqq_gemm: expected A int8, B int32, C int32, D float16")
    if workspace.dtype != torch.int32:
        raise RuntimeError("qqq_gemm: workspace must be int32")
    for name, t in (("A", A), ("B", B), ("C", C), ("D", D), ("s1", s1), ("s2", s2), ("workspace", workspace)):
        if not t.is_contiguous():
            raise RuntimeError(f"qqq_gemm: {name} must be contiguous")
        if not t.is_cuda or t.device != A.device:
            raise RuntimeError(f"qqq_gemm: {name} must live on the same GPU as A (there is no CPU path)")
    if s3.numel() and (not s3.is_contiguous() or s3.device != A.device):
        raise RuntimeError("qqq_gemm: s3 must be contiguous and on A's device")
    if B.numel() != (prob_k // 16) * (prob_n * 2) or D.numel() != prob_m * prob_n:
        raise RuntimeError("qqq_gemm: B must be [k/16, 2n] and D [m, n]")
    if s1.numel() != prob_m or s2.numel() != prob_n:
        raise RuntimeError("qqq_gemm: s1 must have m and s2 n elements")
    if C.size(0) < max_par * 64:
        raise RuntimeError(f"qqq_gemm: C must have at least max_par*64={max_par * 64} rows")
    return prob_m, prob_n, prob_k, groupsize


def _raise_for(err, prob_m, prob_n, prob_k, thread_k, thread_n, groupsize):
    if err == 0:
        return
    if err == ERR_PROB_SHAPE:  # csrc/qqq_gemm.cu:1096-1100
        raise RuntimeError(
            f"Problem (m={prob_m}, n={prob_n}, k={prob_k}) not compatible with thread_k={thread_k}, thread_n={thread_n}."
        )
    if err == ERR_KERN_SHAPE:  # csrc/qqq_gemm.cu:1101-1105
        raise RuntimeError(
            f"No kernel implementation for thread_k={thread_k}, thread_n={thread_n}, groupsize={groupsize}."
        )
    raise RuntimeError(f"qqq_amd: error {err}: {_lib.last_error()}")


def _check_w8(W8, prob_k, prob_n, groupsize, device):
    if W8 is None or W8.numel() == 0:
        return None
    if (W8.dtype != torch.int8 or W8.numel() != prob_k * prob_n or not W8.is_contiguous() or W8.device != device
            or groupsize not in (128, -1)):
        raise RuntimeError("W8 must be the contiguous int8 [k * n] tensor of expand_int8 on A's device")
    return W8


def qqq_gemm_ex(A, B, C, D, s1, s2, s3, workspace, thread_k=-1, thread_n=-1, sms=-1, max_par=8,
                tune: Optional[dict] = None, acc_out: Optional[torch.Tensor] = None,
                bias: Optional[torch.Tensor] = None, W8: Optional[torch.Tensor] = None) -> None:
    """qqq_gemm with tuning / debug hooks (tests, bench), the fused fp16 bias epilogue and (per-group layers, opt-in) the
    layer's expanded int8 weights `W8` (expand_int8; used where the plan is the wide kernel's, bit-identical results).
    `tune` keys: kernel, ksplit, waves, fused, bm, glds, pf, stages, mt, pw, split_m, skew, w8 (include/qqq_amd.h)."""
    L = _lib.lib()
    prob_m, prob_n, prob_k, groupsize = _check_common(A, B, C, D, s1, s2, s3, workspace, max_par)
    tn = None
    if tune:
        tn = _lib.QQQTune()
        for k, v in tune.items():
            setattr(tn, k, int(v))
    if acc_out is not None:
        if acc_out.dtype != torch.int32 or acc_out.numel() != prob_m * prob_n or not acc_out.is_contiguous():
            raise RuntimeError("acc_out must be a contiguous int32 [m, n] tensor")
    if bias is not None:
        if bias.dtype != torch.float16 or bias.numel() != prob_n or not bias.is_contiguous() or bias.device != A.device:
            raise RuntimeError("bias must be a contiguous fp16 [n] tensor on A's device")
    W8 = _check_w8(W8, prob_k, prob_n, groupsize, A.device)
    err = L.qqq_w4a8_gemm_ex2(
        _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(s1), _ptr(s2), _ptr(s3), prob_m, prob_n, prob_k,
        _ptr(workspace), groupsize, A.device.index if A.device.index is not None else 0, _stream_for(A),
        thread_k, thread_n, sms, max_par, ctypes.byref(tn) if tn is not None else None, _ptr(acc_out),
        _ptr(bias), _ptr(W8),
    )
    _raise_for(err, prob_m, prob_n, prob_k, thread_k, thread_n, groupsize)


def _qqq_gemm_impl(A, B, C, D, s1, s2, s3, workspace, thread_k, thread_n, sms, max_par) -> None:
    L = _lib.lib()
    prob_m, prob_n, prob_k, groupsize = _check_common(A, B, C, D, s1, s2, s3, workspace, max_par)
    err = L.qqq_w4a8_gemm(
        _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(s1), _ptr(s2), _ptr(s3), prob_m, prob_n, prob_k,
        _ptr(workspace), groupsize, A.device.index if A.device.index is not None else 0, _stream_for(A),
        thread_k, thread_n, sms, max_par,
    )
    _raise_for(err, prob_m, prob_n, prob_k, thread_k, thread_n, groupsize)


@torch.library.custom_op("qqq_amd::qqq_gemm", mutates_args=("C", "D", "workspace"))
def _qqq_gemm_op(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, D: torch.Tensor, s1: torch.Tensor,
                 s2: torch.Tensor, s3: torch.Tensor, workspace: torch.Tensor, thread_k: int, thread_n: int,
                 sms: int, max_par: int) -> None:
    _qqq_gemm_impl(A, B, C, D, s1, s2, s3, workspace, thread_k, thread_n, sms, max_par)


@torch.library.custom_op("qqq_amd::qqq_gemm_bias", mutates_args=("C", "D", "workspace"))
def _qqq_gemm_bias_op(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, D: torch.Tensor, s1: torch.Tensor,
                      s2: torch.Tensor, s3: torch.Tensor, workspace: torch.Tensor, bias: torch.Tensor,
                      max_par: int) -> None:
    qqq_gemm_ex(A, B, C, D, s1, s2, s3, workspace, -1, -1, -1, max_par, bias=bias)


@torch.library.custom_op("qqq_amd::qqq_gemm_w8", mutates_args=("C", "D", "workspace"))
def _qqq_gemm_w8_op(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, D: torch.Tensor, s1: torch.Tensor,
                    s2: torch.Tensor, s3: torch.Tensor, workspace: torch.Tensor, bias: Optional[torch.Tensor],
                    W8: Optional[torch.Tensor], max_par: int) -> None:
    qqq_gemm_ex(A, B, C, D, s1, s2, s3, workspace, -1, -1, -1, max_par, bias=bias, W8=W8)


def _expand_int8_impl(B: torch.Tensor, s_group: torch.Tensor) -> torch.Tensor:
    if B.dtype != torch.int32 or not B.is_cuda or not B.is_contiguous() or B.dim() != 2:
        raise RuntimeError("expand_int8: B must be the packed int32 [k/16, 2n] weight on the GPU (there is no CPU path)")
    k, n = B.size(0) * 16, B.size(1) // 2
    grouped = s_group.numel() != 0
    if grouped and (s_group.dtype != torch.float16 or not s_group.is_contiguous() or s_group.device != B.device or s_group.dim() != 2
                    or s_group.size(1) != n or s_group.size(0) * 128 != k):
        raise RuntimeError("expand_int8: s_group must be the contiguous fp16 [k/128, n] tensor of a per-group layer on B's device (or empty: per-channel)")
    W8 = torch.empty(k * n, dtype=torch.int8, device=B.device)
    err = _lib.lib().qqq_expand_int8(_ptr(B), _ptr(s_group), _ptr(W8), k, n, 128 if grouped else -1, B.device.index or 0, _stream_for(B))
    if err:
        raise RuntimeError(f"qqq_amd: expand_int8 error {err}: {_lib.last_error()}")
    return W8


@torch.library.custom_op("qqq_amd::expand_int8", mutates_args=())
def _expand_int8_op(B: torch.Tensor, s_group: torch.Tensor) -> torch.Tensor:
    return _expand_int8_impl(B, s_group)


@_expand_int8_op.register_fake
def _(B, s_group):
    return B.new_empty(B.shape[0] * 16 * (B.shape[1] // 2), dtype=torch.int8)


def expand_int8(B: torch.Tensor, s_group: torch.Tensor) -> torch.Tensor:
    """Opt-in load-time re-layout of a layer (beside QuantLinear.pack, qlinear_marlin.py:181-262): the int4 weights as
    the int8 operand the reference kernel forms inside its loop -- per-group re-quantised ONCE, bit for bit dequant_per_group
    (csrc/qqq_gemm.cu:167-210); per-channel (`s_group` empty) 16 * w4 (:146-151) -- stored in the wide kernel's MFMA operand order
    (include/qqq_amd.h: qqq_expand_int8).  Returns int8 [k * n]."""
    if _compiling(B, s_group):
        return _expand_int8_op(B, s_group)
    return _EXT.expand_int8(B, s_group) if _ext() is not None else _expand_int8_impl(B, s_group)


def qqq_gemm_w8(A, B, C, D, s1, s2, s3, workspace, bias=None, W8=None, max_par=16) -> None:
    """qqq_gemm with the optional fused bias and the optional expanded int8 weights of the layer (expand_int8)."""
    if _compiling(A, B, C, D, s1, s2, s3, workspace, bias, W8):
        _qqq_gemm_w8_op(A, B, C, D, s1, s2, s3, workspace, bias, W8, max_par)
    elif _ext() is not None:
        _EXT.qqq_gemm_w8(A, B, C, D, s1, s2, s3, workspace, bias, W8, max_par)
    else:
        qqq_gemm_ex(A, B, C, D, s1, s2, s3, workspace, -1, -1, -1, max_par, bias=bias, W8=W8)


_PLAIN = (torch.Tensor, torch.nn.Parameter)


def _compiling(*tensors) -> bool:
    # Under torch.compile / make_fx / FakeTensorMode / torch.export / any TorchDispatchMode, or with tensor
    # subclasses among the arguments, the calls must go through the registered custom ops (no graph break,
    # fake-tensor propagation, visible to dispatch-based tooling).  For plain eager tensors the dispatcher round
    # trip of a Python custom op costs ~9 us per call -- more than a decode GEMM on a 4096x4096 layer takes on
    # the GPU (tools/host_overhead.py) -- so those calls go straight to the ctypes binding.
    # QQQ_AMD_FORCE_DISPATCHER=1 forces the dispatcher path everywhere.
    if torch.compiler.is_compiling() or _FORCE_DISPATCHER or _dispatch_modes_active():
        return True
    for t in tensors:
        if t is not None and type(t) not in _PLAIN:
            return True
    return False


_len_dispatch_stack = getattr(torch._C, "_len_torch_dispatch_stack", None)  # private: absent -> always the dispatcher path


def _dispatch_modes_active() -> bool:
    return True if _len_dispatch_stack is None else _len_dispatch_stack() > 0


def qqq_gemm_bias(A, B, C, D, s1, s2, s3, workspace, bias, max_par=16) -> None:
    """qqq_gemm + the reference's `D + self.bias` (qlinear_marlin.py:287) fused into the epilogue."""
    if _compiling(A, B, C, D, s1, s2, s3, workspace, bias):
        _qqq_gemm_bias_op(A, B, C, D, s1, s2, s3, workspace, bias, max_par)
    elif _ext() is not None:
        _EXT.qqq_gemm_bias(A, B, C, D, s1, s2, s3, workspace, bias, max_par)
    else:
        qqq_gemm_ex(A, B, C, D, s1, s2, s3, workspace, -1, -1, -1, max_par, bias=bias)


def qqq_gemm(A, B, C, D, s1, s2, s3, workspace, thread_k=-1, thread_n=-1, sms=-1, max_par=8) -> None:
    """Drop-in for `QQQ._CUDA.qqq_gemm` (qqq_gemm.h:23-36): writes fp16 `D` in place, returns None."""
    if _compiling(A, B, C, D, s1, s2, s3, workspace):
        _qqq_gemm_op(A, B, C, D, s1, s2, s3, workspace, thread_k, thread_n, sms, max_par)
    elif _ext() is not None:
        _EXT.qqq_gemm(A, B, C, D, s1, s2, s3, workspace, thread_k, thread_n, sms, max_par)
    else:
        _qqq_gemm_impl(A, B, C, D, s1, s2, s3, workspace, thread_k, thread_n, sms, max_par)


def mul(A, B, C, D, s1, s2, s3, workspace, thread_k=-1, thread_n=-1, sms=-1, max_par=16):
    """Drop-in for qlinear_marlin.mul (qlinear_marlin.py:28-45)."""
    qqq_gemm(A, B, C, D, s1, s2, s3, workspace, thread_k, thread_n, sms, max_par)


def marlin_qqq_gemm(a, b_q_weight, s_tok, s_ch, s_group, workspace, size_m, size_n, size_k):
    """vLLM-style wrapper (`ops.marlin_qqq_gemm`, external to the reference tree): allocates the
    int32 reduce buffer and the fp16 output itself and returns the output."""
    max_par = 16
    C = torch.empty((max_par * 64, size_n), dtype=torch.int32, device=a.device)
    D = torch.empty((size_m, size_n), dtype=torch.float16, device=a.device)
    if s_group is None:
        s_group = torch.empty(0, dtype=torch.float16, device=a.device)
    qqq_gemm(a, b_q_weight, C, D, s_tok, s_ch, s_group, workspace, -1, -1, -1, max_par)
    return D


def _dynamic_quant_impl(x: torch.Tensor):
    L = _lib.lib()
    if x.dtype != torch.float16 or not x.is_cuda or x.dim() < 1:
        raise RuntimeError("dynamic_quant: expected an fp16 tensor on the GPU (there is no CPU path)")
    # any rank, like the reference method (max over the last dim with keepdim, qlinear_marlin.py:265-268)
    k = x.shape[-1]
    x2 = x.reshape(-1, k).contiguous()
    m = x2.shape[0]
    xq = torch.empty((m, k), dtype=torch.int8, device=x.device)
    s1 = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    err = L.qqq_dynamic_quant(_ptr(x2), _ptr(xq), _ptr(s1), m, k, x.device.index or 0, _stream_for(x))
    if err:
        raise RuntimeError(f"qqq_amd: dynamic_quant error {err}: {_lib.last_error()}")
    return xq.reshape(x.shape), s1.reshape(x.shape[:-1] + (1,))


@torch.library.custom_op("qqq_amd::dynamic_quant", mutates_args=())
def _dynamic_quant_op(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    return _dynamic_quant_impl(x)


@_dynamic_quant_op.register_fake
def _(x):
    return x.new_empty(x.shape, dtype=torch.int8), x.new_empty(x.shape[:-1] + (1,), dtype=torch.float32)


def dynamic_quant(x: torch.Tensor):
    """Fused replacement of QuantLinear.dynamic_quant (qlinear_marlin.py:265-268) for an fp16 tensor of any rank:
    (int8 x.shape, f32 x.shape[:-1] + (1,)).  Deviations from the reference expression: an all-zero row quantises to
    0 with scale 0 (reference: 0/0 = NaN -> int8, undefined), and the scale is the torch-GPU evaluation
    fp16(amax * (1/127)) (a CPU run of the reference differs by one fp16 ulp of the scale on a few rows)."""
    if _compiling(x):
        return _dynamic_quant_op(x)
    return _EXT.dynamic_quant(x) if _ext() is not None else _dynamic_quant_impl(x)


def quantlinear_forward(x: torch.Tensor, B, C, s2, s3, workspace, bias=None, max_par: int = 16, W8=None) -> torch.Tensor:
    """QuantLinear.forward (qlinear_marlin.py:270-288) for a 2-D fp16 input in ONE binding call: fused dynamic int8
    quantisation + W4A8 GEMM (+ fp16 bias).  The buffers are a module's own (qlinear.QuantLinear): only the cheap
    checks are made here.  Under torch.compile the two registered custom ops are used instead."""
    if _compiling(x, B, C, s2, s3, workspace, bias, W8):
        xq, s1 = _dynamic_quant_op(x)
        D = torch.empty((x.shape[0], C.size(1)), dtype=torch.float16, device=x.device)
        if W8 is not None:
            _qqq_gemm_w8_op(xq, B, C, D, s1, s2, s3, workspace, bias, W8, max_par)
        elif bias is not None:
            _qqq_gemm_bias_op(xq, B, C, D, s1, s2, s3, workspace, bias, max_par)
        else:
            _qqq_gemm_op(xq, B, C, D, s1, s2, s3, workspace, -1, -1, -1, max_par)
        return D
    if _ext() is not None:
        return _EXT.quantlinear_forward(x, B, C, s2, s3, workspace, bias, max_par, W8)
    if x.dtype != torch.float16 or not x.is_cuda or x.dim() != 2 or not x.is_contiguous():
        raise RuntimeError("quantlinear_forward: expected a contiguous 2-D fp16 tensor on the GPU (there is no CPU path)")
    m, k = x.shape
    n = C.size(1)
    dev = x.device
    if B.size(0) * 16 != k or B.device != dev:
        raise RuntimeError("quantlinear_forward: B must be the packed [k/16, 2n] weight on x's device")
    groupsize = -1 if s3.numel() == 0 else k // s3.size(0)
    # everything the kernels would read as raw bits
    if B.numel() != (k // 16) * (n * 2):
        raise RuntimeError("quantlinear_forward: B must be the packed [k/16, 2n] weight on x's device")
    if (B.dtype != torch.int32 or C.dtype != torch.int32 or workspace.dtype != torch.int32 or s2.dtype != torch.float32
            or C.device != dev or s2.device != dev or workspace.device != dev
            or not (B.is_contiguous() and C.is_contiguous() and s2.is_contiguous() and workspace.is_contiguous())):
        raise RuntimeError("quantlinear_forward: expected contiguous int32 B / C / workspace and float32 s2 on x's device")
    if s2.numel() != n or C.size(0) < max_par * 64 or workspace.numel() < n // 128 * max_par:
        raise RuntimeError(f"quantlinear_forward: s2 needs n={n} elements, C max_par*64={max_par * 64} rows, "
                           f"workspace at least {n // 128 * max_par} entries")
    if s3.numel() and (s3.dtype != torch.float16 or s3.device != dev or not s3.is_contiguous()
                       or groupsize * s3.size(0) != k or s3.numel() != s3.size(0) * n):
        raise RuntimeError("quantlinear_forward: s3 must be a contiguous fp16 [k/groupsize, n] tensor on x's device")
    if bias is not None and (bias.dtype != torch.float16 or bias.numel() != n or bias.device != dev
                             or not bias.is_contiguous()):
        raise RuntimeError(f"quantlinear_forward: bias must be a contiguous fp16 [n] tensor on x's device "
                           f"(got {bias.dtype}, {tuple(bias.shape)}, {bias.device})")
    xq = torch.empty((m, k), dtype=torch.int8, device=x.device)
    s1 = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    D = torch.empty((m, n), dtype=torch.float16, device=x.device)
    if m == 0:
        return D
    W8 = _check_w8(W8, k, n, groupsize, dev)
    err = _lib.lib().qqq_quantlinear_forward2(
        x.data_ptr(), xq.data_ptr(), s1.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr(), s2.data_ptr(),
        _ptr(s3), m, n, k, workspace.data_ptr(), groupsize, x.device.index or 0, _stream_for(x), max_par, _ptr(bias), _ptr(W8))
    _raise_for(err, m, n, k, -1, -1, groupsize)
    return D
