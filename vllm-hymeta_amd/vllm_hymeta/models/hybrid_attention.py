"""70B intra-layer hybrid attention on the HIP kernels: a softmax branch (sliding-window or full
attention plus 128 always-visible meta tokens) and a GLA branch run on the same input, their
outputs RMS-normed over the full hidden size, averaged and projected.

Reference: ``HLinearAttention`` (``vllm_hymeta/models/modeling_hymeta.py:366-659``),
``FlashAttention`` (``:669-815``), ``IntraHybridAttention`` (``:817-933``); eager:
``hf_model/linear_attention.py:95-181``, ``hf_model/attention.py:146-263``,
``hf_model/modeling_hymeta.py:89-130``.

What changed in mechanism: the gate math (``:611-629``) runs inside the GLA kernels; every prefill
sequence goes through one launch with no host synchronisation (the reference loops in Python with
``.isnan().any()`` / ``== 0`` device reads, ``:470-529``); the two norms share one ``[T, 2]``
variance all-reduce; the meta-token K/V and the meta GLA state are plain tensors the caller fills
once (``prepare_meta_state``) instead of a warm-up run counter (``:1160-1164``).
"""

from __future__ import annotations

from typing import Optional

import torch

from .. import _C, hip_ops, parallel
from .attention import MetaAttention, bind_kv_cache, current_attn_metadata, forward_context, standalone_prefix
from ..linear import HymetaLinear
from ..parallel import HeadShard, all_reduce_sum, all_reduce_variance, shard_heads


class HLinearAttention(torch.nn.Module):
    """GLA branch: qkv projection, key-derived decay gate, chunked prefill / recurrent decode."""

    def __init__(self, hidden_size: int, num_heads: int, num_key_value_heads: int, qkv_proj: HymetaLinear,
                 layer_idx: int, num_meta_tokens: int = 128, clamp_max: float = 0.95,
                 shard: Optional[HeadShard] = None):
        super().__init__()
        self.shard = shard or shard_heads(num_heads, num_key_value_heads, 1, 0)
        self.head_dim = hidden_size // num_heads
        self.qkv_proj = qkv_proj
        self.layer_idx = layer_idx
        self.num_meta_tokens = num_meta_tokens
        self.clamp_max = clamp_max

    def _gate(self, lower_bound: Optional[torch.Tensor]):
        # "first" layers (every 7th in the vLLM path, modeling_hymeta.py:616) clamp; the others use
        # this rank's slice of the per-layer lower bound (:622-628)
        if lower_bound is None or self.layer_idx % 7 == 0:
            return dict(gate_mode=_C.GATE_70B_CLAMP, gate_clamp=self.clamp_max)
        H, D = self.shard.num_heads, self.head_dim
        lb = lower_bound.view(-1, D)[self.shard.head_start:self.shard.head_start + H].contiguous().float()
        return dict(gate_mode=_C.GATE_70B_LOWER_BOUND, lower_bound=lb)

    def _qkv(self, hidden_states: torch.Tensor, qkv: Optional[torch.Tensor] = None):
        H, Hkv, D = self.shard.num_heads, self.shard.num_kv_heads, self.head_dim
        T = hidden_states.shape[0]
        if qkv is None:
            qkv = self.qkv_proj(hidden_states)
        return (qkv[:, : H * D].view(T, H, D), qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D),
                qkv[:, (H + Hkv) * D:].view(T, Hkv, D))

    def prepare_meta_state(self, meta_hidden: torch.Tensor, meta_linear_cache: torch.Tensor,
                           lower_bound: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Run the ``num_meta_tokens`` meta tokens from a zero state and keep the final state in
        ``meta_linear_cache [H_local, D, D]`` (reference ``:445-467``).  Returns their outputs."""
        q, k, v = self._qkv(meta_hidden)
        M = meta_hidden.shape[0]
        state = meta_linear_cache.unsqueeze(0)
        dev = meta_hidden.device
        return hip_ops.gla_prefill_varlen(
            q, k, v, None, state, torch.zeros(1, dtype=torch.int32, device=dev),
            torch.tensor([0, M], dtype=torch.int32, device=dev), M,
            context_lens=torch.zeros(1, dtype=torch.int32, device=dev), **self._gate(lower_bound))

    def forward(self, hidden_states: torch.Tensor, hymeta_cache: torch.Tensor, meta_linear_cache: torch.Tensor,
                state_indices_tensor: torch.Tensor, attn_metadata,
                lower_bound: Optional[torch.Tensor] = None, qkv: Optional[torch.Tensor] = None) -> torch.Tensor:
        """hidden_states ``[T, hidden]`` real tokens (prefills first, then decodes); returns
        ``[T, H_local * D]`` (no norm, no projection: those belong to ``IntraHybridAttention``).
        ``qkv``: this branch's projection of ``hidden_states`` when the caller has computed it already."""
        q, k, v = self._qkv(hidden_states, qkv)
        T, H, D = q.shape
        out = torch.empty(T, H, D, dtype=hidden_states.dtype, device=hidden_states.device)
        gate = self._gate(lower_bound)
        n_pre = getattr(attn_metadata, "num_prefills", 0)
        t_pre = getattr(attn_metadata, "num_prefill_tokens", 0)
        if n_pre > 0:
            hip_ops.gla_prefill_varlen(
                q[:t_pre], k[:t_pre], v[:t_pre], None, hymeta_cache, state_indices_tensor[:n_pre],
                attn_metadata.query_start_loc[: n_pre + 1], attn_metadata.max_query_len or t_pre,
                context_lens=attn_metadata.context_lens_tensor[:n_pre], meta_state=meta_linear_cache,
                out=out[:t_pre], **gate)
        if T > t_pre:
            hip_ops.gla_decode(q[t_pre:], k[t_pre:], v[t_pre:], None, hymeta_cache,
                               state_indices_tensor[n_pre:], out=out[t_pre:], **gate)
        return out.view(T, H * D)


class MetaFlashAttention(torch.nn.Module):
    """Softmax branch: qkv projection, RoPE at positions shifted by the meta tokens, paged KV write,
    window (or full) attention over the sequence plus the meta keys.  (``FlashAttention`` in the
    reference; renamed here because the 7B family has a class of the same name.)"""

    def __init__(self, hidden_size: int, num_heads: int, num_key_value_heads: int, qkv_proj: HymetaLinear,
                 cos_sin_cache: torch.Tensor, sliding_window: Optional[int], num_meta_tokens: int = 128,
                 shard: Optional[HeadShard] = None, attn: Optional[MetaAttention] = None):
        """``sliding_window``: W keys including the query itself; ``None`` / 0 = full attention (the
        ``interleaved_sliding_window`` entry 0 of the config).

        ``attn``: the registered attention layer that owns this layer's paged KV cache (reference
        ``self.attn = MetaAttention(...)``, ``:734-744``).  The model classes create it when they are
        constructed, so that the engine finds it by name and binds the cache; a layer built on its own makes
        one under a private name."""
        super().__init__()
        self.shard = shard or shard_heads(num_heads, num_key_value_heads, 1, 0)
        self.head_dim = hidden_size // num_heads
        self.qkv_proj = qkv_proj
        self.register_buffer("cos_sin_cache", cos_sin_cache)
        self.num_meta_tokens = num_meta_tokens
        if attn is None:
            attn = MetaAttention(self.shard.num_heads, self.head_dim, self.head_dim ** -0.5, self.shard.num_kv_heads,
                                 per_layer_sliding_window=sliding_window or None,
                                 prefix=standalone_prefix("vanilla_attn.attn"), num_meta_tokens=num_meta_tokens)
        if (attn.num_heads, attn.num_kv_heads, attn.head_size) != (self.shard.num_heads, self.shard.num_kv_heads, self.head_dim):
            raise ValueError("attention layer and head shard disagree on the head geometry")
        self.attn = attn

    def _qkv(self, hidden_states, positions, offset, qkv=None, roped=False):
        H, Hkv, D = self.shard.num_heads, self.shard.num_kv_heads, self.head_dim
        T = hidden_states.shape[0]
        if qkv is None:
            qkv = self.qkv_proj(hidden_states)
        q = qkv[:, : H * D].view(T, H, D)
        k = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
        v = qkv[:, (H + Hkv) * D:].view(T, Hkv, D)
        if not roped:
            hip_ops.rope_neox_(positions, q, k, self.cos_sin_cache, position_offset=offset)
        return q, k, v

    def prepare_meta_state(self, meta_hidden: torch.Tensor, meta_fattn_cache: torch.Tensor) -> torch.Tensor:
        """Roped K/V of the meta tokens (positions 0..M-1) into ``meta_fattn_cache [2, M, Hkv_local, D]``
        (reference ``:776-792``); returns the meta-query rows' attention output ``[M, H_local*D]``."""
        M = meta_hidden.shape[0]
        pos = torch.arange(M, dtype=torch.int64, device=meta_hidden.device)
        q2, k2, v2 = self._qkv(meta_hidden, pos, 0)
        meta_fattn_cache[0].copy_(k2)
        meta_fattn_cache[1].copy_(v2)
        cu = torch.zeros(1, dtype=torch.int32, device=meta_hidden.device)
        out = hip_ops.meta_swa_prefill(q2[:0], cu, 0, key=k2[:0], value=v2[:0], cu_seqlens_k=cu, meta_query=q2,
                                       meta_key=meta_fattn_cache[0], meta_value=meta_fattn_cache[1])
        return out.view(M, -1)

    def forward(self, hidden_states: torch.Tensor, positions: torch.Tensor, kv_cache: Optional[torch.Tensor],
                meta_fattn_cache: torch.Tensor, attn_metadata=None, qkv: Optional[torch.Tensor] = None,
                qkv_roped: bool = False) -> torch.Tensor:
        """positions are the sequence positions of the real tokens; the meta tokens occupy 0..M-1, so
        real token p is roped at p + M (reference ``modeling_hymeta.py:1314-1323``).

        The attention itself runs through the registered layer ``self.attn`` and the opaque op
        (reference ``:795``), which reads the paged cache bound to that layer and the step's metadata from
        the forward context.  ``kv_cache`` / ``attn_metadata`` given explicitly (callers without an engine)
        are bound / set for the call.  ``qkv``: this branch's projection of ``hidden_states`` when the caller has
        computed it already (roped in place here, unless ``qkv_roped`` says its q and k heads are rotated already)."""
        q, k, v = self._qkv(hidden_states, positions, self.num_meta_tokens, qkv, qkv_roped)
        T, H, D = q.shape
        if kv_cache is not None:
            bind_kv_cache(self.attn, kv_cache)
        k2, v2 = (meta_fattn_cache[0], meta_fattn_cache[1]) if self.num_meta_tokens > 0 else (None, None)
        q2d, k2d, v2d = q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)     # views of the fused buffer
        if attn_metadata is not None and current_attn_metadata() is not attn_metadata:
            with forward_context(attn_metadata):
                return self.attn(q2d, k2d, v2d, query2=None, key2=k2, value2=v2)
        return self.attn(q2d, k2d, v2d, query2=None, key2=k2, value2=v2)


class IntraHybridAttention(torch.nn.Module):
    def __init__(self, vanilla_attn: MetaFlashAttention, linear_attn: HLinearAttention, out_proj: HymetaLinear,
                 norm1_weight: torch.Tensor, norm2_weight: torch.Tensor, norm_eps: float = 1e-6, tp_group=None,
                 merged_qkv: Optional[HymetaLinear] = None):
        """``norm1_weight`` / ``norm2_weight``: this rank's slice (fp32 ``[hidden / tp]``) of the two
        full-hidden RMS-norm weights (reference ``HymetaRMSNormTP.weight_loader``, ``:138-149``).

        ``merged_qkv``: the two branches' qkv projections as one linear (output = the softmax branch's columns, then
        the GLA branch's; the branches then hold ``LinearColumns`` views of it): both read the same normed input, so
        the layer runs one GEMM over one weight stream and hands each branch its columns."""
        super().__init__()
        self.vanilla_attn, self.linear_attn, self.out_proj = vanilla_attn, linear_attn, out_proj
        self.merged_qkv = merged_qkv
        self.register_buffer("norm1_weight", norm1_weight.float().contiguous())
        self.register_buffer("norm2_weight", norm2_weight.float().contiguous())
        self.norm_eps = norm_eps
        self.tp_group = tp_group

    def combine(self, out_attn: torch.Tensor, out_linear: torch.Tensor) -> torch.Tensor:
        stats = None
        if self.linear_attn.shard.tp_size > 1 or parallel.SINGLE_RANK_COLLECTIVES:      # the statistics span the full hidden size: every rank's share
            stats = all_reduce_variance(hip_ops.rms_sumsq2(out_attn, out_linear), self.tp_group)
        return hip_ops.hybrid_combine(out_attn, out_linear, self.norm1_weight, self.norm2_weight,
                                      self.norm_eps, stats=stats)

    def prepare_meta_state(self, meta_hidden: torch.Tensor, meta_linear_cache: torch.Tensor,
                           meta_fattn_cache: torch.Tensor, lower_bound=None) -> torch.Tensor:
        """Fill this layer's shared meta caches from the meta tokens' (normed) hidden states and
        return the block output of the meta rows, which feeds the next layer's meta pass
        (reference: the first forward passes, ``:776-788`` and ``:445-467``)."""
        out_attn = self.vanilla_attn.prepare_meta_state(meta_hidden, meta_fattn_cache)
        out_linear = self.linear_attn.prepare_meta_state(meta_hidden, meta_linear_cache, lower_bound)
        out_linear = out_linear.reshape(meta_hidden.shape[0], -1)
        return all_reduce_sum(self.out_proj(self.combine(out_attn, out_linear)), self.tp_group)

    can_defer_out = True     # forward(defer_out=True): out_proj may leave its K slices' sum to the fused add + norm that follows

    def forward(self, hidden_states, positions, kv_cache, hymeta_cache, meta_linear_cache, meta_fattn_cache,
                state_indices_tensor, attn_metadata=None, lower_bound=None, defer_out: bool = False):
        if attn_metadata is None:
            attn_metadata = current_attn_metadata()
        qkv_attn = qkv_linear = None
        roped = False
        if self.merged_qkv is not None:
            both = self.merged_qkv(hidden_states, defer=True)
            if isinstance(both, hip_ops.PartialSums):
                # a K-sliced projection: the launch that adds its slices also rotates the softmax branch's q and k heads
                # (the first columns; everything behind them is written as it is)
                va = self.vanilla_attn
                both = hip_ops.rope_from_partials(both, positions, va.cos_sin_cache, va.shard.num_heads, va.shard.num_kv_heads,
                                                  position_offset=va.num_meta_tokens)
                roped = True
            half = both.shape[1] // 2
            qkv_attn, qkv_linear = both[:, :half], both[:, half:]
        out_attn = self.vanilla_attn(hidden_states, positions, kv_cache, meta_fattn_cache, attn_metadata, qkv=qkv_attn,
                                     qkv_roped=roped)
        out_linear = self.linear_attn(hidden_states, hymeta_cache, meta_linear_cache, state_indices_tensor,
                                      attn_metadata, lower_bound, qkv=qkv_linear)
        reduce = parallel._active(self.tp_group)
        # the K slices' sum is left to whoever reads the result next: the fused add + norm (no all-reduce), or the
        # group's one-shot exchange
        defer = parallel.defers_into_all_reduce(self.tp_group) if reduce else defer_out
        y = self.out_proj(self.combine(out_attn, out_linear), defer=defer)
        return all_reduce_sum(y, self.tp_group) if reduce else y
