"""7B sliding-window softmax attention layer on the HIP kernels.

Reference: ``FlashAttention`` in ``vllm_hymeta/model_for_7B/modeling_gla_swa.py:133-218`` (qkv ->
RoPE -> stock vLLM attention with ``per_layer_sliding_window = W + 1`` -> o_proj) and the eager
``hf_7B_model/window_attention.py:94-221`` (``window_size=(W, 0)``): a query sees itself and the
``W`` keys before it.  No meta tokens on the 7B path.
"""

from __future__ import annotations

from typing import Optional

import torch

from .. import hip_ops
from ..linear import HymetaLinear
from ..models.attention import MetaAttention, bind_kv_cache, current_attn_metadata, forward_context, standalone_prefix
from .. import parallel
from ..parallel import HeadShard, all_reduce_sum, shard_heads


def rope_cos_sin_cache(head_dim: int, max_position: int, base: float, device=None,
                       dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """``[max_position, head_dim]`` table, cos in the first half and sin in the second, rounded to
    the activation dtype (reference ``HymetaRotaryEmbedding._compute_cos_sin_cache``,
    ``vllm_hymeta/models/modeling_hymeta.py:203-218``)."""
    # Built on the host so the table does not depend on the device's cos/sin rounding, then moved.
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.outer(t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype).contiguous().to(device)


class SlidingWindowAttention(torch.nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, num_key_value_heads: int, sliding_window: int,
                 qkv_proj: HymetaLinear, o_proj: HymetaLinear, cos_sin_cache: torch.Tensor,
                 shard: Optional[HeadShard] = None, tp_group=None, attn: Optional[MetaAttention] = None):
        """``attn``: the registered attention layer that owns this layer's paged KV cache (the reference's
        ``self.attn = Attention(..., per_layer_sliding_window=W + 1)``, ``modeling_gla_swa.py:193-202``; here the
        meta-token layer with no meta tokens).  The model class creates it at construction so that the engine
        finds it by name; a layer built on its own makes one under a private name."""
        super().__init__()
        self.shard = shard or shard_heads(num_heads, num_key_value_heads, 1, 0)
        self.head_dim = hidden_size // num_heads
        self.window_left = sliding_window          # W keys before the query + the query itself
        self.qkv_proj, self.o_proj = qkv_proj, o_proj
        self.register_buffer("cos_sin_cache", cos_sin_cache)
        self.tp_group = tp_group
        if attn is None:
            attn = swa_attention_layer(self.shard, self.head_dim, sliding_window, standalone_prefix("attn.attn"))
        if (attn.num_heads, attn.num_kv_heads, attn.head_size) != (self.shard.num_heads, self.shard.num_kv_heads, self.head_dim):
            raise ValueError("attention layer and head shard disagree on the head geometry")
        self.attn = attn

    can_defer_out = True     # forward(defer_out=True): o_proj may leave its K slices' sum to the fused add + norm that follows

    def forward(self, hidden_states: torch.Tensor, positions: torch.Tensor, kv_cache: Optional[torch.Tensor] = None,
                attn_metadata=None, defer_out: bool = False):
        """The attention runs through the registered layer and the opaque op, which read the paged cache bound to
        the layer (numel 0: profiling run) and the step's metadata from the forward context; ``kv_cache
        [2, nblk, bs, Hkv_local, D]`` / ``attn_metadata`` given explicitly are bound / set for the call."""
        H, Hkv, D = self.shard.num_heads, self.shard.num_kv_heads, self.head_dim
        T = hidden_states.shape[0]
        qkv = self.qkv_proj(hidden_states, defer=True)
        if isinstance(qkv, hip_ops.PartialSums):      # K-sliced projection: its slice sum and the rotation in one launch
            qkv = hip_ops.rope_from_partials(qkv, positions, self.cos_sin_cache, H, Hkv)
            q, k, v = qkv[:, : H * D], qkv[:, H * D:(H + Hkv) * D], qkv[:, (H + Hkv) * D:]
        else:
            q, k, v = qkv[:, : H * D], qkv[:, H * D:(H + Hkv) * D], qkv[:, (H + Hkv) * D:]
            hip_ops.rope_neox_(positions, q.view(T, H, D), k.view(T, Hkv, D), self.cos_sin_cache)
        if kv_cache is not None:
            bind_kv_cache(self.attn, kv_cache)
        if attn_metadata is not None and current_attn_metadata() is not attn_metadata:
            with forward_context(attn_metadata):
                out = self.attn(q, k, v)
        else:
            out = self.attn(q, k, v)
        reduce = parallel._active(self.tp_group)
        y = self.o_proj(out, defer=defer_out and not reduce)
        return all_reduce_sum(y, self.tp_group) if reduce else y


def swa_attention_layer(shard: HeadShard, head_dim: int, sliding_window: int, prefix: str, cache_config=None) -> MetaAttention:
    """The 7B sliding-window layer's attention object: ``sliding_window`` keys before the query plus the query
    itself (``per_layer_sliding_window = W + 1``, reference ``modeling_gla_swa.py:200``), no meta tokens."""
    return MetaAttention(shard.num_heads, head_dim, head_dim ** -0.5, shard.num_kv_heads, cache_config=cache_config,
                         per_layer_sliding_window=sliding_window + 1, prefix=prefix, num_meta_tokens=0)
