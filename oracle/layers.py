"""Layer-level glue around the GLA / SWA kernels: CPU restatement (test infrastructure).

Parity unpinned (no vectors in the reference).  Follows, per function, the
eager modules ``hf_7B_model`` / ``hf_model`` and their vLLM twins.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- gates
def gate_7b_ref(q: torch.Tensor, k: torch.Tensor, gk: torch.Tensor,
                normalizer: float = 16.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """7B gate: ``q, k = relu``; ``g = logsigmoid(gk) / 16`` evaluated in the
    activation dtype (``vllm_hymeta/model_for_7B/gla_attention.py:251-262``;
    eager: ``hf_7B_model/gla_attention.py:134-145``)."""
    return F.relu(q), F.relu(k), F.logsigmoid(gk) / normalizer


def gate_70b_ref(q: torch.Tensor, k: torch.Tensor, lower_bound: Optional[torch.Tensor],
                 first_layer: bool, clamp_max: float = 0.95
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """70B gate (``vllm_hymeta/models/modeling_hymeta.py:611-629`` ==
    ``hf_model/linear_attention.py:143-157``).  q ``[T,H,D]`` and k ``[T,H,D]``
    (already repeated to q-heads); ``lower_bound [H,D]`` fp32.  Returns
    ``(relu(q), k in k's dtype, g fp32)``."""
    dt = k.dtype
    q = F.relu(q)
    kf = k.float().sigmoid()
    if lower_bound is None or first_layer:
        kf = torch.clamp(kf, max=clamp_max)
        g = (1 - kf).log()
    else:
        lb = lower_bound.float()[None]
        g = 1 - (1 - lb) * kf
        kf, g = 1 - g, g.log()
    return q, kf.to(dt), g


def lower_bounds_ref(lower_bounds: torch.Tensor) -> torch.Tensor:
    """``softmax(lower_bounds, 0).cumsum(0) - row_0``
    (``hf_model/modeling_hymeta.py:339-341``); ``[L, H*D]`` in, same shape out."""
    lb = lower_bounds.float().softmax(0)
    return lb.cumsum(0) - lb[0]


# --------------------------------------------------------------------------- norms
def rms_norm_ref(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float = 1e-6) -> torch.Tensor:
    """RMS norm over the last dim as vLLM's ``RMSNorm`` applies it to bf16:
    fp32 statistics, normalised value rounded to the activation dtype, then
    multiplied by the weight in that dtype (g_norm of
    ``vllm_hymeta/model_for_7B/gla_attention.py:210,233``; eager semantics:
    ``hf_7B_model/layernorm.py:39-52``)."""
    dt = x.dtype
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    y = (xf * torch.rsqrt(var + eps)).to(dt)
    return y if weight is None else y * weight.to(dt)


def rms_norm_tp_ref(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
                    variance: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``HymetaRMSNormTP._forward`` (``vllm_hymeta/models/modeling_hymeta.py:151-173``):
    fp32 variance over the (global) hidden dim -- ``variance`` overrides the local
    mean of squares with the all-reduced one -- result cast to x's dtype, then
    times the fp32 weight (promoting to fp32)."""
    dt = x.dtype
    xf = x.float()
    if variance is None:
        variance = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(variance + eps)
    return xf.to(dt) * weight.float()


def hybrid_combine_ref(o_attn: torch.Tensor, o_lin: torch.Tensor, w1: torch.Tensor,
                       w2: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """``(norm1(o_attn) + norm2(o_lin)) / 2 -> bf16``
    (``vllm_hymeta/models/modeling_hymeta.py:909-919``)."""
    y = (rms_norm_tp_ref(o_attn, w1, eps) + rms_norm_tp_ref(o_lin, w2, eps)) / 2
    return y.to(torch.bfloat16)


# --------------------------------------------------------------------------- RoPE
def rope_cos_sin_cache_ref(head_dim: int, max_pos: int, base: float,
                           dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """``HymetaRotaryEmbedding._compute_cos_sin_cache``
    (``vllm_hymeta/models/modeling_hymeta.py:206-218``): ``[max_pos, head_dim]``,
    cos in the first half, sin in the second, rounded to ``dtype`` (``:203``)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_pos, dtype=torch.float)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)


def rope_neox_ref(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                  cos_sin_cache: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """NeoX (rotate-half) RoPE as vLLM's ``rotary_embedding`` op applies it in
    place (call site ``vllm_hymeta/models/modeling_hymeta.py:230-231``; eager
    form ``hf_7B_model/window_attention.py:49-79``): per pair ``(x, y) =
    (v[i], v[i + D/2])``: ``x*cos - y*sin``, ``y*cos + x*sin``.  The op is
    instantiated for the activation dtype, so each product and the sum are
    rounded to bf16 -- reproduced here by doing the arithmetic on bf16 tensors.
    q ``[T,H,D]``, k ``[T,Hkv,D]``; ``cos_sin_cache [max_pos, D]`` (cos | sin)."""
    D = q.shape[-1]
    cs = cos_sin_cache[positions.long()].to(q.dtype)
    cos, sin = cs[:, None, : D // 2], cs[:, None, D // 2:]

    def rot(x):
        x1, x2 = x[..., : D // 2], x[..., D // 2:]
        return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)

    return rot(q), rot(k)
