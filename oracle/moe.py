"""CPU restatement of the 70B sparse MoE block and the dense MLP.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
path.  Parity unpinned: the expert arithmetic of the drop-in target lives in vLLM 0.10.0
(``FusedMoE``), the reference holds no vectors for it; the eager module
(``hf_model/moe.py:34-101``) is restated here, with ``renormalize`` selecting between its
behaviour (top-k weights renormalised to 1, ``:73-75``) and the vLLM path's
(``renormalize=False``, ``vllm_hymeta/models/modeling_hymeta.py:319``)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def silu_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    """``silu(x[..., :I]) * x[..., I:]`` (vLLM ``SiluAndMul``, ``modeling_hymeta.py:263-270``):
    the product in fp32, one rounding to the input dtype."""
    inter = x.shape[-1] // 2
    g, u = x[..., :inter].float(), x[..., inter:].float()
    return (F.silu(g) * u).to(x.dtype)


def mlp_ref(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """Dense gated MLP (``HymetaMLP``, ``modeling_hymeta.py:236-271``): ``w13 [2I, H]`` = gate rows
    then up rows, ``w2 [H, I]``; every linear rounds its output to the activation dtype."""
    h = (x.double() @ w13.double().t()).to(x.dtype)
    return (silu_and_mul_ref(h).double() @ w2.double().t()).to(x.dtype)


def moe_route_ref(router_logits: torch.Tensor, top_k: int, renormalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """Softmax over the experts in fp32, top-k probabilities (``hf_model/moe.py:71-75``; vLLM
    ``fused_topk``).  Returns ``(ids [T,k] int64, weights [T,k] fp32)``."""
    probs = F.softmax(router_logits.float(), dim=-1)
    # ties (frequent with bf16 logits) go to the lower expert index, as in vLLM's topk_softmax
    w, ids = torch.sort(probs, dim=-1, descending=True, stable=True)
    w, ids = w[:, :top_k], ids[:, :top_k]
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return ids, w


def moe_ref(x: torch.Tensor, router_logits: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, top_k: int = 1,
            renormalize: bool = False, shared: Optional[torch.Tensor] = None,
            local_experts: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """``x [T,H]``, ``router_logits [T,E]``, ``w13 [E,2I,H]``, ``w2 [E,H,I]``.  Per (token, choice):
    ``y = bf16(w * (silu_and_mul(bf16(x W13^T)) W2^T))`` (routing weight applied to the fp32 sums, as in
    the fused expert kernel), summed over the choices in the activation dtype, plus the shared
    expert's output when given (``modeling_hymeta.py:1048-1058``).  ``local_experts = (lo, hi)``
    keeps only the pairs of those experts (one rank's share under expert parallelism; the ranks'
    results add up to the full one)."""
    T, H = x.shape
    ids, w = moe_route_ref(router_logits, top_k, renormalize)
    out = torch.zeros(T, H, dtype=torch.float64)
    lo, hi = local_experts if local_experts is not None else (0, w13.shape[0])
    for t in range(T):
        for j in range(top_k):
            e = int(ids[t, j])
            if not lo <= e < hi:
                continue
            h = (x[t].double() @ w13[e].double().t()).to(x.dtype)
            a = silu_and_mul_ref(h)
            y = (float(w[t, j]) * (a.double() @ w2[e].double().t())).to(x.dtype)
            out[t] = (out[t] + y.double()).to(x.dtype).double() if j else y.double()
    out = out.to(x.dtype)
    if shared is not None:
        out = (out.float() + shared.float()).to(x.dtype)
    return out


def moe_w8a8_ref(x: torch.Tensor, router_logits: torch.Tensor, w13: torch.Tensor, w13_scale: torch.Tensor, w2: torch.Tensor,
                 w2_scale: torch.Tensor, top_k: int = 1, renormalize: bool = False, shared: Optional[torch.Tensor] = None,
                 local_experts: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """The block with W8A8 experts (a compressed-tensors W8A8 checkpoint quantises the experts' Linear layers too,
    ``hf_model/quant_w8a8.py:121-132``; vLLM runs them through its int8 ``fused_moe``, third party: parity unpinned):
    ``w13 int8 [E,2I,H]`` with ``w13_scale fp32 [E,2I]``, ``w2 int8 [E,H,I]`` with ``w2_scale fp32 [E,H]``.  Per (token,
    choice): the token is quantised per row, ``h = bf16((xq W13q^T) s_x s_w13)``, ``a = silu_and_mul(h)`` is quantised per
    row again, ``y = bf16(w * ((aq W2q^T) s_a s_w2))``; the rest as ``moe_ref``."""
    from .quant import w8a8_gemm_ref

    T, H = x.shape
    ids, w = moe_route_ref(router_logits, top_k, renormalize)
    out = torch.zeros(T, H, dtype=torch.float64)
    lo, hi = local_experts if local_experts is not None else (0, w13.shape[0])
    for t in range(T):
        for j in range(top_k):
            e = int(ids[t, j])
            if not lo <= e < hi:
                continue
            h = w8a8_gemm_ref(x[t:t + 1], w13[e], w13_scale[e]).to(x.dtype)
            a = silu_and_mul_ref(h)
            y = (float(w[t, j]) * w8a8_gemm_ref(a, w2[e], w2_scale[e]))[0].to(x.dtype)
            out[t] = (out[t] + y.double()).to(x.dtype).double() if j else y.double()
    out = out.to(x.dtype)
    if shared is not None:
        out = (out.float() + shared.float()).to(x.dtype)
    return out


def moe_gptq_ref(x: torch.Tensor, router_logits: torch.Tensor, w13, w2, bits: int = 4, group_size: int = 128, top_k: int = 1,
                 renormalize: bool = False, shared: Optional[torch.Tensor] = None,
                 local_experts: Optional[Tuple[int, int]] = None, zero_offset: int = 1) -> torch.Tensor:
    """The block with GPTQ experts (``hf_model/hymeta_gptq.py:30-32`` lists the routed experts' projections among the
    quantised modules; vLLM runs them through its quantised ``fused_moe``, third party: parity unpinned).  ``w13`` /
    ``w2``: per-expert lists of ``(qweight, qzeros, scales)`` standing for ``[2I, H]`` / ``[H, I]`` weights; the arithmetic
    is ``moe_ref``'s on the exactly de-quantised weights ``s (q - z)``."""
    from .quant import gptq_dequant_ref

    d13 = torch.stack([gptq_dequant_ref(*t, bits, group_size, zero_offset).t() for t in w13])      # [E, 2I, H] fp64
    d2 = torch.stack([gptq_dequant_ref(*t, bits, group_size, zero_offset).t() for t in w2])
    T, H = x.shape
    ids, w = moe_route_ref(router_logits, top_k, renormalize)
    out = torch.zeros(T, H, dtype=torch.float64)
    lo, hi = local_experts if local_experts is not None else (0, d13.shape[0])
    for t in range(T):
        for j in range(top_k):
            e = int(ids[t, j])
            if not lo <= e < hi:
                continue
            h = (x[t].double() @ d13[e].t()).to(x.dtype)
            a = silu_and_mul_ref(h)
            y = (float(w[t, j]) * (a.double() @ d2[e].t())).to(x.dtype)
            out[t] = (out[t] + y.double()).to(x.dtype).double() if j else y.double()
    out = out.to(x.dtype)
    if shared is not None:
        out = (out.float() + shared.float()).to(x.dtype)
    return out
