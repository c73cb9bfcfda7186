"""Quantised linears: CPU restatement (test infrastructure).

Parity unpinned: the reference contains only the offline recipes
(``hf_7B_model/quant_7B.py:32-36`` -- gptqmodel ``QuantizeConfig(bits=8, group_size=64)``;
``hf_model/quant_w8a8.py:121-132`` -- llmcompressor ``GPTQModifier(scheme="W8A8")``) and the
module lists (``hf_7B_model/gla_swa_gptq.py:18-24``, ``hf_model/hymeta_gptq.py:22-35``); the
arithmetic runs inside vLLM 0.10.0 and gptqmodel 2.2.0 (third-party, not vendored).  Restated
here from their published tensor formats:

* GPTQ v1: ``qweight int32 [K/pack, N]``, ``qzeros int32 [K/group, N/pack]`` (stored minus 1),
  ``scales [K/group, N]``; ``w = scales * (q - (z + 1))``.
* W8A8 (compressed-tensors int8, channel-wise weights, dynamic per-token activations):
  ``s_x = max|x| / 127``, ``xq = clamp(round(x / s_x))``, ``y = (xq @ wq^T) * s_x * s_w``.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch


def gptq_pack_ref(w_int: torch.Tensor, zeros_int: torch.Tensor, bits: int,
                  zero_offset: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pack integer weights ``[K, N]`` (values in [0, 2^bits)) and zeros ``[G, N]`` into the GPTQ
    int32 tensors (test helper; the inverse of :func:`gptq_dequant_ref`)."""
    pack = 32 // bits
    K, N = w_int.shape
    qw = torch.zeros(K // pack, N, dtype=torch.int64)
    for j in range(pack):
        qw |= w_int[j::pack].to(torch.int64) << (bits * j)
    z = (zeros_int.to(torch.int64) - zero_offset) & ((1 << bits) - 1)
    qz = torch.zeros(z.shape[0], N // pack, dtype=torch.int64)
    for j in range(pack):
        qz |= z[:, j::pack] << (bits * j)

    def to_i32(t):
        return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)

    return to_i32(qw), to_i32(qz)


def gptq_dequant_ref(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, bits: int,
                     group_size: int, zero_offset: int = 1) -> torch.Tensor:
    """GPTQ tensors -> fp64 weight ``[K, N]``."""
    pack = 32 // bits
    mask = (1 << bits) - 1
    qw = qweight.to(torch.int64) & 0xFFFFFFFF
    K = qw.shape[0] * pack
    N = qw.shape[1]
    q = torch.stack([(qw >> (bits * j)) & mask for j in range(pack)], dim=1).reshape(K, N)
    qz = qzeros.to(torch.int64) & 0xFFFFFFFF
    z = torch.stack([(qz >> (bits * j)) & mask for j in range(pack)], dim=2).reshape(qz.shape[0], N)
    z = z + zero_offset
    g = torch.arange(K) // group_size
    return scales.double()[g] * (q - z[g]).double()


def gptq_gemm_ref(x: torch.Tensor, qweight, qzeros, scales, bits: int, group_size: int,
                  bias: Optional[torch.Tensor] = None, zero_offset: int = 1) -> torch.Tensor:
    w = gptq_dequant_ref(qweight, qzeros, scales, bits, group_size, zero_offset)
    y = x.double() @ w
    return y if bias is None else y + bias.double()


def w8a8_quant_act_ref(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dynamic per-token symmetric int8 quantisation: ``(xq int8 [M,K], scale fp32 [M])``."""
    xf = x.float()
    scale = xf.abs().amax(dim=-1).clamp_min(1e-10) / 127.0
    xq = torch.clamp(torch.round(xf / scale[:, None]), -128, 127).to(torch.int8)
    return xq, scale


def w8a8_gemm_ref(x: torch.Tensor, qweight: torch.Tensor, w_scale: torch.Tensor,
                  bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16 ``[M,K]``, qweight int8 ``[N,K]``, w_scale fp32 ``[N]`` -> fp64 ``[M,N]``."""
    xq, xs = w8a8_quant_act_ref(x)
    acc = xq.double() @ qweight.double().t()
    y = acc * xs.double()[:, None] * w_scale.double()[None, :]
    return y if bias is None else y + bias.double()
