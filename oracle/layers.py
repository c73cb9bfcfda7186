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


def add_rms_norm_ref(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor,
                     eps: float = 1e-6) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """The blocks' ``attn_norm(x)`` / ``mlp_norm(x, residual)``
    (``vllm_hymeta/models/modeling_hymeta.py:1024,1039``,
    ``model_for_7B/modeling_gla_swa.py:294,302``): vLLM's ``RMSNorm`` with an optional
    residual.  As the device op computes it (vllm==0.10.0 ``fused_add_rms_norm``, third
    party): the sum is rounded to the activation dtype, stored as the new residual, and
    normed.  Returns ``(normed, new_residual)``."""
    if residual is not None:
        x = (x.float() + residual.float()).to(x.dtype)
        residual = x
    return rms_norm_ref(x, weight, eps), residual


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


# --------------------------------------------------------------------------- whole layers (eager restatement)
def _linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 linear the way an eager bf16 ``nn.Linear`` computes it: fp32 accumulation, one
    rounding to the activation dtype."""
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    return y.to(x.dtype)


def gla_layer_7b_ref(x: torch.Tensor, p: dict, num_heads: int, num_kv_heads: int,
                     initial_state: Optional[torch.Tensor] = None,
                     norm_eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    """One 7B GLA attention layer on one sequence (``hf_7B_model/gla_attention.py:129-173``):
    q/k/v projections with bias, rank-16 gate projection, relu feature map, ``logsigmoid/16``,
    GQA, the GLA recurrence, per-head RMS norm (``g_norm``), output projection.

    x ``[T, hidden]`` bf16; ``p`` holds ``q_proj.weight/bias``, ``k_proj.*``, ``v_proj.*``,
    ``gk_proj.0.weight``, ``gk_proj.1.weight/bias``, ``g_norm.weight``, ``o_proj.weight``.
    Returns ``(out [T, hidden] bf16, final_state [H, D, D] fp64)``."""
    from .gla import gla_recurrent_ref

    T = x.shape[0]
    D = p["q_proj.weight"].shape[0] // num_heads
    q = _linear(x, p["q_proj.weight"], p["q_proj.bias"]).view(T, num_heads, D)
    k = _linear(x, p["k_proj.weight"], p["k_proj.bias"]).view(T, num_kv_heads, D)
    v = _linear(x, p["v_proj.weight"], p["v_proj.bias"]).view(T, num_kv_heads, D)
    gk = _linear(_linear(x, p["gk_proj.0.weight"]), p["gk_proj.1.weight"], p["gk_proj.1.bias"])
    q, k, g = gate_7b_ref(q, k, gk.view(T, num_kv_heads, D))
    o, state = gla_recurrent_ref(q, k, v, g, initial_state)
    o = rms_norm_ref(o.to(x.dtype), p["g_norm.weight"], norm_eps)
    return _linear(o.reshape(T, num_heads * D), p["o_proj.weight"]), state


def swa_layer_7b_ref(x: torch.Tensor, positions: torch.Tensor, p: dict, num_heads: int,
                     num_kv_heads: int, sliding_window: int, rope_theta: float = 1e6,
                     max_position: int = 131072) -> torch.Tensor:
    """One 7B sliding-window attention layer on one sequence, prefill form
    (``hf_7B_model/window_attention.py:151-221``): projections with bias, NeoX RoPE, causal
    attention over the last ``sliding_window + 1`` keys (``window_size=(W, 0)``, ``:215``),
    output projection.  RoPE uses the vLLM path's bf16 cos/sin table."""
    from .meta_swa import meta_swa_dense_ref

    T = x.shape[0]
    D = p["q_proj.weight"].shape[0] // num_heads
    q = _linear(x, p["q_proj.weight"], p["q_proj.bias"]).view(T, num_heads, D)
    k = _linear(x, p["k_proj.weight"], p["k_proj.bias"]).view(T, num_kv_heads, D)
    v = _linear(x, p["v_proj.weight"], p["v_proj.bias"]).view(T, num_kv_heads, D)
    cache = rope_cos_sin_cache_ref(D, max_position, rope_theta)
    q, k = rope_neox_ref(positions, q, k, cache)
    o, _ = meta_swa_dense_ref(q, k, v, None, None, None, sliding_window)
    return _linear(o.to(x.dtype).reshape(T, num_heads * D), p["o_proj.weight"])


def xavier_linear_params_ref(gen: torch.Generator, out_features: int, in_features: int, bias: bool,
                             gain: float = 2 ** -2.5, dtype=torch.bfloat16) -> dict:
    """Random init of the eager layers (``hf_7B_model/gla_attention.py:93-100``):
    ``xavier_uniform_(gain=2**-2.5)`` weights, zero biases."""
    bound = gain * (6.0 / (in_features + out_features)) ** 0.5
    w = (torch.rand(out_features, in_features, generator=gen) * 2 - 1) * bound
    out = {"weight": w.to(dtype)}
    if bias:
        out["bias"] = torch.zeros(out_features, dtype=dtype)
    return out


def hybrid_layer_70b_ref(x: torch.Tensor, meta_x: torch.Tensor, positions: torch.Tensor, p: dict,
                         num_heads: int, num_kv_heads: int, sliding_window: Optional[int],
                         lower_bound: Optional[torch.Tensor], first_layer: bool,
                         rope_theta: float = 1e6, max_position: int = 131072,
                         norm_eps: float = 1e-6, return_meta: bool = False, linears: Optional[dict] = None) -> torch.Tensor:
    """One 70B intra-hybrid attention block on ONE sequence, prefill form, with the meta tokens
    already "cached": the softmax branch sees the roped K/V of ``meta_x`` (positions 0..M-1) as
    always-visible keys and ropes the real tokens at ``positions + M``; the GLA branch starts from
    the state left by the meta tokens.  (``hf_model/modeling_hymeta.py:89-130``,
    ``hf_model/attention.py:155-263``, ``hf_model/linear_attention.py:124-181``; vLLM path
    ``vllm_hymeta/models/modeling_hymeta.py:580-659,747-815,865-933``.)

    ``p``: ``vanilla_attn.{q,k,v}_proj.{weight,bias}``, ``linear_attn.{q,k,v}_proj.{weight,bias}``,
    ``norm1.weight``, ``norm2.weight`` (fp32), ``out_proj.weight``.  ``sliding_window`` = W keys
    including the query (None: full attention).  Returns ``[T, hidden]`` bf16; with ``return_meta``
    ``[M + T, hidden]``, the meta tokens' own rows first (their queries see every meta key and no
    real key; their GLA output is the pass from the zero state).

    ``linears``: quantised projections.  Callables ``{"vanilla_attn.qkv", "linear_attn.qkv", "out_proj"}`` taking the
    bf16 input rows and returning the projection (fused q | k | v columns for the two qkv entries) replace the bf16
    weights of ``p`` (which then only needs the norm weights); each stands for the restated arithmetic of a GPTQ or
    W8A8 linear (``oracle/quant.py``)."""
    from .gla import gla_recurrent_ref
    from .meta_swa import meta_swa_dense_ref

    T, M = x.shape[0], meta_x.shape[0]
    D = p["norm1.weight"].shape[0] // num_heads
    rep = num_heads // num_kv_heads
    cache = rope_cos_sin_cache_ref(D, max_position, rope_theta)

    def qkv(prefix, h):
        n = h.shape[0]
        if linears is not None:
            y = linears[f"{prefix}.qkv"](h).to(h.dtype)
            return (y[:, : num_heads * D].reshape(n, num_heads, D),
                    y[:, num_heads * D:(num_heads + num_kv_heads) * D].reshape(n, num_kv_heads, D),
                    y[:, (num_heads + num_kv_heads) * D:].reshape(n, num_kv_heads, D))
        return (_linear(h, p[f"{prefix}.q_proj.weight"], p[f"{prefix}.q_proj.bias"]).view(n, num_heads, D),
                _linear(h, p[f"{prefix}.k_proj.weight"], p[f"{prefix}.k_proj.bias"]).view(n, num_kv_heads, D),
                _linear(h, p[f"{prefix}.v_proj.weight"], p[f"{prefix}.v_proj.bias"]).view(n, num_kv_heads, D))

    # softmax branch
    q2, k2, v2 = qkv("vanilla_attn", meta_x)
    q2, k2 = rope_neox_ref(torch.arange(M), q2, k2, cache)
    q, k, v = qkv("vanilla_attn", x)
    q, k = rope_neox_ref(positions + M, q, k, cache)
    wl = -1 if not sliding_window else sliding_window - 1
    o_attn, _ = meta_swa_dense_ref(q, k, v, k2, v2, q2 if return_meta else None, wl)
    o_attn = o_attn.to(x.dtype).reshape(-1, num_heads * D)

    # GLA branch: meta tokens first (zero state), then the sequence from the meta state
    def gla(h, init):
        ql, kl, vl = qkv("linear_attn", h)
        qg, kg, gg = gate_70b_ref(ql, kl.repeat_interleave(rep, dim=1), lower_bound, first_layer)
        return gla_recurrent_ref(qg, kg, vl.repeat_interleave(rep, dim=1), gg, init)

    o_meta, meta_state = gla(meta_x, None)
    o_lin, _ = gla(x, meta_state.to(x.dtype))          # the shared meta state is stored in bf16
    if return_meta:
        o_lin = torch.cat([o_meta, o_lin])
    o_lin = o_lin.to(x.dtype).reshape(-1, num_heads * D)
    y = hybrid_combine_ref(o_attn, o_lin, p["norm1.weight"], p["norm2.weight"], norm_eps)
    if linears is not None:
        return linears["out_proj"](y).to(x.dtype)
    return _linear(y, p["out_proj.weight"])


# --------------------------------------------------------------------------- decoder block
def decoder_block_ref(x: torch.Tensor, attn, mlp, attn_norm_w: torch.Tensor, mlp_norm_w: torch.Tensor,
                      eps: float = 1e-6) -> torch.Tensor:
    """``HybridBlock.forward`` / ``GLAswaBlock.forward``
    (``vllm_hymeta/models/modeling_hymeta.py:1021-1075``, ``model_for_7B/modeling_gla_swa.py:292-306``):
    ``r = x; h = attn(attn_norm(x)); n, r = mlp_norm(h, r); return r + mlp(n)``; ``attn`` and ``mlp``
    are callables on bf16 ``[T, hidden]``."""
    h, _ = add_rms_norm_ref(x, None, attn_norm_w, eps)
    h = attn(h)
    n, r = add_rms_norm_ref(h, x, mlp_norm_w, eps)
    return (r.float() + mlp(n).float()).to(x.dtype)
