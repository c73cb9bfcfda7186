"""7B gated-linear-attention layer on the HIP kernels.

Reference: ``vllm_hymeta/model_for_7B/gla_attention.py:90-283`` (vLLM layer) and
``hf_7B_model/gla_attention.py:36-173`` (eager).  Differences in mechanism, not in result: the
gate (relu, logsigmoid/16), GQA head mapping and per-head ``g_norm`` run inside the GLA kernels;
all prefill sequences go through one launch; nothing is copied, transposed or repeated.
"""

from __future__ import annotations

from typing import Optional

import torch

from .. import _C, hip_ops
from ..linear import HymetaLinear
from .. import parallel
from ..parallel import HeadShard, all_reduce_sum, shard_heads


class GatedLinearAttention(torch.nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, num_key_value_heads: int,
                 qkv_proj: HymetaLinear, gk_proj0: HymetaLinear, gk_proj1: HymetaLinear,
                 o_proj: HymetaLinear, g_norm_weight: torch.Tensor, norm_eps: float = 1e-6,
                 gate_logit_normalizer: int = 16, shard: Optional[HeadShard] = None, tp_group=None):
        """``qkv_proj`` maps hidden -> [q | k | v] of THIS rank's heads (with bias); ``gk_proj0``
        hidden -> 16 (no bias) and ``gk_proj1`` 16 -> kv_heads * head_dim of this rank (with bias)
        form the low-rank gate projection, which the reference leaves unquantised
        (``gla_attention.py:157-166``); ``o_proj`` maps this rank's heads back to hidden (no bias)."""
        super().__init__()
        self.shard = shard or shard_heads(num_heads, num_key_value_heads, 1, 0)
        self.head_dim = hidden_size // num_heads
        self.qkv_proj, self.gk_proj0, self.gk_proj1, self.o_proj = qkv_proj, gk_proj0, gk_proj1, o_proj
        self.register_buffer("g_norm_weight", g_norm_weight)
        self.norm_eps = norm_eps
        self.gate_logit_normalizer = float(gate_logit_normalizer)
        self.tp_group = tp_group
        # A bf16 checkpoint: the gate's first linear (16 rows over the same input) rides in the qkv projection as 16 more
        # output rows (padded to the kernels' 64-row granularity), and its second linear (16 -> Hkv*D) is applied inside
        # the recurrence kernel: a decode step then has no launch for the gate at all.
        self.qkv_features = (self.shard.num_heads + 2 * self.shard.num_kv_heads) * self.head_dim
        p0, p1 = gk_proj0, gk_proj1
        self.gate_in_qkv = (qkv_proj.kind == "bf16" and p0.kind == "bf16" and p1.kind == "bf16" and p0.bias is None
                            and p0.weight.shape == (16, hidden_size) and p1.weight.shape[1] == 16 and p1.weight.is_contiguous()
                            and qkv_proj.weight.shape == (self.qkv_features, hidden_size) and self.qkv_features % 64 == 0)
        if self.gate_in_qkv:
            w = qkv_proj.weight
            pad = torch.zeros(48, hidden_size, dtype=w.dtype, device=w.device)
            bias = qkv_proj.bias if qkv_proj.bias is not None else torch.zeros(self.qkv_features, dtype=w.dtype, device=w.device)
            self.qkv_proj = HymetaLinear.from_weight(torch.cat([w, p0.weight.to(w.device), pad]),
                                                     torch.cat([bias, torch.zeros(64, dtype=bias.dtype, device=bias.device)]))

    can_defer_out = True     # forward(defer_out=True): o_proj may leave its K slices' sum to the fused add + norm that follows

    def forward(self, hidden_states: torch.Tensor, gla_cache: torch.Tensor,
                state_indices_tensor: torch.Tensor, attn_metadata, defer_out: bool = False):
        """hidden_states ``[T, hidden]`` (prefill tokens first, then one token per decode sequence);
        ``gla_cache [slots, H_local, D, D]`` of this layer; ``state_indices_tensor`` one slot per
        sequence in batch order.  Returns ``[T, hidden]``."""
        H, Hkv, D = self.shard.num_heads, self.shard.num_kv_heads, self.head_dim
        T = hidden_states.shape[0]
        p0, p1 = self.gk_proj0, self.gk_proj1
        n_pre = getattr(attn_metadata, "num_prefills", 0)
        t_pre = getattr(attn_metadata, "num_prefill_tokens", 0)
        gate = dict(gate_mode=_C.GATE_7B, gate_normalizer=self.gate_logit_normalizer,
                    out_norm_weight=self.g_norm_weight, out_norm_eps=self.norm_eps)
        # a pure decode step with the gate in the projection: the recurrence kernel reads q, k, v and the gate's rank-16
        # input straight from the projection's K-slice sums (no reduce launch for the projection)
        # a quantised checkpoint (its gate projection is not quantised): on a pure decode step with a state per key/value
        # head the recurrence kernel applies both of the gate's linears to the layer input itself
        gate_in_kernel = (not self.gate_in_qkv and n_pre == 0 and T > 0 and gla_cache.shape[1] == Hkv != H
                          and p0.kind == "bf16" and p1.kind == "bf16" and p0.bias is None and p0.weight.shape[0] == 16
                          and p0.weight.is_contiguous() and p1.weight.is_contiguous() and hidden_states.stride(0) % 8 == 0)
        qkv = self.qkv_proj(hidden_states, defer=(self.gate_in_qkv or gate_in_kernel) and n_pre == 0)
        if isinstance(qkv, hip_ops.PartialSums):
            QKV = self.qkv_features
            if self.gate_in_qkv:
                out = hip_ops.gla_decode((T, H, D), (T, Hkv, D), None, None, gla_cache, state_indices_tensor, g_w1=p1.weight,
                                         g_bias=p1.bias, qkv_partials=qkv, qkv_cols=(0, H * D, (H + Hkv) * D, QKV), **gate)
            else:       # quantised projection: q, k, v from its partial sums, the gate from the layer input
                out = hip_ops.gla_decode((T, H, D), (T, Hkv, D), None, hidden_states, gla_cache, state_indices_tensor,
                                         g_w0=p0.weight, g_w1=p1.weight, g_bias=p1.bias, qkv_partials=qkv,
                                         qkv_cols=(0, H * D, (H + Hkv) * D, 0), **gate)
            reduce = parallel._active(self.tp_group)
            y = self.o_proj(out.view(T, H * D), defer=defer_out and not reduce)
            return all_reduce_sum(y, self.tp_group) if reduce else y
        gk = rank16 = None
        if gate_in_kernel:
            q = qkv[:, : H * D].view(T, H, D)
            k = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
            v = qkv[:, (H + Hkv) * D:(H + 2 * Hkv) * D].view(T, Hkv, D)
            out = hip_ops.gla_decode(q, k, v, hidden_states, gla_cache, state_indices_tensor, g_w0=p0.weight, g_w1=p1.weight,
                                     g_bias=p1.bias, **gate)
            reduce = parallel._active(self.tp_group)
            y = self.o_proj(out.view(T, H * D), defer=defer_out and not reduce)
            return all_reduce_sum(y, self.tp_group) if reduce else y
        if self.gate_in_qkv:
            rank16 = qkv[:, self.qkv_features:self.qkv_features + 16]       # the gate's first linear, bf16 [T, 16]
            if t_pre > 0:
                gk = torch.nn.functional.linear(rank16[:t_pre], p1.weight, p1.bias).view(t_pre, Hkv, D)
        elif (T <= hip_ops.LOWRANK_GATE_ROWS and p0.kind == "bf16" and p1.kind == "bf16" and p0.bias is None
                and p0.weight.shape[0] == 16 and p0.weight.is_contiguous() and p1.weight.is_contiguous()):
            gk = hip_ops.lowrank_gate(hidden_states, p0.weight, p1.weight, p1.bias).view(T, Hkv, D)   # one launch
        else:
            gk = p1(p0(hidden_states)).view(T, Hkv, D)
        q = qkv[:, : H * D].view(T, H, D)
        k = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
        v = qkv[:, (H + Hkv) * D:(H + 2 * Hkv) * D].view(T, Hkv, D)
        out = torch.empty(T, H, D, dtype=hidden_states.dtype, device=hidden_states.device)
        if n_pre > 0:
            hip_ops.gla_prefill_varlen(
                q[:t_pre], k[:t_pre], v[:t_pre], gk[:t_pre], gla_cache, state_indices_tensor[:n_pre],
                attn_metadata.query_start_loc[: n_pre + 1], attn_metadata.max_query_len or t_pre,
                context_lens=attn_metadata.context_lens_tensor[:n_pre], out=out[:t_pre], **gate)
        if T > t_pre:
            if rank16 is not None:
                hip_ops.gla_decode(q[t_pre:], k[t_pre:], v[t_pre:], rank16[t_pre:], gla_cache, state_indices_tensor[n_pre:],
                                   out=out[t_pre:], g_w1=p1.weight, g_bias=p1.bias, **gate)
            else:
                hip_ops.gla_decode(q[t_pre:], k[t_pre:], v[t_pre:], gk[t_pre:], gla_cache,
                                   state_indices_tensor[n_pre:], out=out[t_pre:], **gate)
        reduce = parallel._active(self.tp_group)
        y = self.o_proj(out.view(T, H * D), defer=defer_out and not reduce)
        return all_reduce_sum(y, self.tp_group) if reduce else y
