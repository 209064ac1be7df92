"""qqq_amd -- MI355X-native (gfx950) W4A8 GEMM behind QQQ's `qqq_gemm` operator.

Public surface (mirrors the reference's hot path, HandH1998/QQQ):
    qqq_gemm(A, B, C, D, s1, s2, s3, workspace, thread_k, thread_n, sms, max_par)   # QQQ._CUDA.qqq_gemm
    mul(...)                                                                       # qlinear_marlin.mul
    QuantLinear                                                                    # qlinear_marlin.QuantLinear
    marlin_qqq_gemm(...)                                                           # vLLM-style wrapper
    dynamic_quant(x)                                                               # fused per-token int8 quant
    quantlinear_forward(x, B, C, s2, s3, workspace, bias)                          # QuantLinear.forward, one call
    expand_int8(B, s_group) / QuantLinear.expand_for_prefill()                     # opt-in load-time int8 expansion
"""
from .ops import (  # noqa: F401
    dynamic_quant,
    expand_int8,
    marlin_qqq_gemm,
    mul,
    qqq_gemm,
    qqq_gemm_bias,
    qqq_gemm_ex,
    qqq_gemm_w8,
    quantlinear_forward,
)
from .qlinear import QuantLinear, fuse_quant_linears  # noqa: F401

__all__ = ["qqq_gemm", "qqq_gemm_bias", "qqq_gemm_ex", "qqq_gemm_w8", "expand_int8", "mul", "marlin_qqq_gemm", "dynamic_quant", "quantlinear_forward",
           "QuantLinear", "fuse_quant_linears"]
