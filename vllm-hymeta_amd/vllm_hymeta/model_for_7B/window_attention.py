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
                 shard: Optional[HeadShard] = None, tp_group=None):
        super().__init__()
        self.shard = shard or shard_heads(num_heads, num_key_value_heads, 1, 0)
        self.head_dim = hidden_size // num_heads
        self.window_left = sliding_window          # W keys before the query + the query itself
        self.qkv_proj, self.o_proj = qkv_proj, o_proj
        self.register_buffer("cos_sin_cache", cos_sin_cache)
        self.tp_group = tp_group

    def forward(self, hidden_states: torch.Tensor, positions: torch.Tensor, kv_cache: torch.Tensor,
                attn_metadata) -> torch.Tensor:
        """kv_cache ``[2, nblk, bs, Hkv_local, D]`` of this layer (numel 0: profiling run)."""
        H, Hkv, D = self.shard.num_heads, self.shard.num_kv_heads, self.head_dim
        T = hidden_states.shape[0]
        qkv = self.qkv_proj(hidden_states)
        q = qkv[:, : H * D].view(T, H, D)
        k = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
        v = qkv[:, (H + Hkv) * D:].view(T, Hkv, D)
        hip_ops.rope_neox_(positions, q, k, self.cos_sin_cache)
        have_cache = kv_cache.numel() > 0
        if have_cache:
            hip_ops.reshape_and_cache(k, v, kv_cache[0], kv_cache[1], attn_metadata.slot_mapping)
        out = torch.empty(T, H, D, dtype=hidden_states.dtype, device=hidden_states.device)
        n_pre = getattr(attn_metadata, "num_prefills", 0)
        t_pre = getattr(attn_metadata, "num_prefill_tokens", 0)
        if n_pre > 0:
            pre = attn_metadata.prefill_metadata
            paged = have_cache and pre.block_tables is not None and pre.block_tables.numel() > 0
            if paged:
                hip_ops.meta_swa_prefill(q[:t_pre], pre.query_start_loc, pre.max_query_len,
                                         key_cache=kv_cache[0], value_cache=kv_cache[1],
                                         block_tables=pre.block_tables, seq_lens_k=pre.seq_lens_tensor,
                                         window_left=self.window_left, out=out[:t_pre],
                                         max_seqlen_k=pre.max_prefill_seq_len or 0)
            else:
                hip_ops.meta_swa_prefill(q[:t_pre], pre.seq_start_loc, pre.max_prefill_seq_len,
                                         key=k[:t_pre], value=v[:t_pre], cu_seqlens_k=pre.seq_start_loc,
                                         window_left=self.window_left, out=out[:t_pre],
                                         max_seqlen_k=pre.max_prefill_seq_len or 0)
        if T > t_pre:
            dec = attn_metadata.decode_metadata
            hip_ops.meta_swa_paged_decode(q[t_pre:], kv_cache[0], kv_cache[1], dec.seq_lens_tensor,
                                          dec.block_tables, window_left=self.window_left,
                                          max_seqlen_k=dec.max_decode_seq_len or 0, out=out[t_pre:])
        y = self.o_proj(out.view(T, H * D))
        return all_reduce_sum(y, self.tp_group)
