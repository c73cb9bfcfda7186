"""Metadata builder of the META backend for vLLM's V1 engine.

The reference serves through the V0 engine only (``SupportsV0Only``, ``vllm_hymeta/models/modeling_hymeta.py:1378``;
its builder, ``vllm_hymeta/attention/meta_attention.py:411-608``, consumes the V0 model runner's per-sequence-group
data).  The V1 engine hands every attention backend one ``CommonAttentionMetadata`` per step instead (third party,
vllm==0.10.0 ``vllm/v1/attention/backends/utils.py``: ``query_start_loc`` / ``seq_lens`` on device and host,
``num_computed_tokens_cpu``, ``num_reqs``, ``num_actual_tokens``, ``max_query_len``, ``block_table_tensor``,
``slot_mapping``) and asks the builder to order the batch.  This builder turns that into the SAME
``MetaAttentionMetadata`` the kernels' dispatch (``MetaAttentionImpl.forward``) reads, so nothing below the metadata
changes between the engines:

* the kernels take the step's prefill rows first, then one row per decoding sequence: ``reorder_batch`` moves requests
  that are scheduled more than one token in front of the others (V1's own helper puts decodes first);
* slot mapping and block table arrive ready-made on the device from the V1 block-table manager, so the host loop over
  tokens of the V0 builder (``compute_slot_mapping``) has no counterpart here;
* sequence lengths and query lengths are taken from the host copies: no device read.
"""

from __future__ import annotations

from typing import List, Optional

import torch

from .meta_attention import MetaAttentionMetadata


def split_prefills_and_decodes(query_lens: List[int]) -> int:
    """Number of leading prefill requests of a batch ordered prefills-first; raises if a request scheduled for more
    than one token follows a decode (the kernels index decode rows after all prefill rows)."""
    n = 0
    while n < len(query_lens) and query_lens[n] > 1:
        n += 1
    if any(q != 1 for q in query_lens[n:]):
        raise ValueError("batch is not ordered prefills first, then single-token decodes: call reorder_batch()")
    return n


class MetaAttentionMetadataBuilderV1:
    """``AttentionMetadataBuilder`` of the V1 engine for the META backend (``__init__(kv_cache_spec, layer_names,
    vllm_config, device)``, ``reorder_batch``, ``build``, ``build_for_cudagraph_capture``, ``use_cascade_attention``)."""

    # full-graph capture of pure-decode batches is supported (static metadata buffers, device-advanced)
    full_cudagraph_supported = True
    reorder_batch_threshold = 1

    def __init__(self, kv_cache_spec=None, layer_names: Optional[List[str]] = None, vllm_config=None, device=None):
        self.kv_cache_spec = kv_cache_spec
        self.layer_names = list(layer_names or [])
        self.vllm_config = vllm_config
        self.device = torch.device(device) if device is not None else None
        self.block_size = getattr(kv_cache_spec, "block_size", None)

    # ------------------------------------------------------------------ batch order
    def reorder_batch(self, input_batch, scheduler_output) -> bool:
        """Put the requests scheduled for more than one token in front (stable on both sides).  ``input_batch`` has
        ``req_ids`` and ``swap_states(i, j)``; ``scheduler_output.num_scheduled_tokens`` maps request id -> tokens of
        this step.  Returns whether anything moved."""
        ids = list(input_batch.req_ids)
        is_prefill = [scheduler_output.num_scheduled_tokens[r] > 1 for r in ids]
        moved = False
        front = 0                       # first position not yet holding a prefill
        for i in range(len(ids)):
            if not is_prefill[i]:
                continue
            j = i
            while j > front:            # bubble the prefill down to `front`: keeps both groups in their order
                input_batch.swap_states(j - 1, j)
                is_prefill[j - 1], is_prefill[j] = is_prefill[j], is_prefill[j - 1]
                j -= 1
                moved = True
            front += 1
        return moved

    # ------------------------------------------------------------------ metadata
    def build(self, common_prefix_len: int, common_attn_metadata, fast_build: bool = False) -> MetaAttentionMetadata:
        m = common_attn_metadata
        num_reqs, num_tokens = int(m.num_reqs), int(m.num_actual_tokens)
        qsl_cpu = m.query_start_loc_cpu[: num_reqs + 1]
        query_lens = (qsl_cpu[1:] - qsl_cpu[:-1]).tolist()
        seq_lens = m.seq_lens_cpu[:num_reqs].tolist()
        n_pre = split_prefills_and_decodes(query_lens)
        t_pre = int(qsl_cpu[n_pre])
        device = m.query_start_loc.device
        seq_lens_dev = m.seq_lens[:num_reqs].to(torch.int32)
        qsl_dev = m.query_start_loc[: num_reqs + 1].to(torch.int32)
        context = seq_lens_dev - (qsl_dev[1:] - qsl_dev[:-1])
        seq_start = torch.zeros(num_reqs + 1, dtype=torch.int32, device=device)
        seq_start[1:] = torch.cumsum(seq_lens_dev, 0)
        return MetaAttentionMetadata(
            num_prefills=n_pre, num_prefill_tokens=t_pre, num_decode_tokens=num_tokens - t_pre,
            slot_mapping=m.slot_mapping[:num_tokens].to(torch.long),
            multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=False,
            seq_lens=seq_lens, seq_lens_tensor=seq_lens_dev,
            max_prefill_seq_len=max(seq_lens[:n_pre], default=0), max_decode_seq_len=max(seq_lens[n_pre:], default=0),
            context_lens_tensor=context, block_tables=m.block_table_tensor[:num_reqs],
            use_cuda_graph=False, max_query_len=int(m.max_query_len), max_decode_query_len=1,
            query_start_loc=qsl_dev, seq_start_loc=seq_start)

    def build_for_cudagraph_capture(self, common_attn_metadata) -> MetaAttentionMetadata:
        """Pure-decode batch with every tensor a view of the runner's persistent buffers (their contents change
        between replays, their addresses do not).  ``max_decode_seq_len`` is what the graph's kernels size their key
        splits for: the longest sequence the buffers can describe."""
        md = self.build(0, common_attn_metadata)
        if md.num_prefills != 0:
            raise ValueError("only pure-decode batches are captured")
        md.use_cuda_graph = True
        md.seq_lens = None                                   # host copy would go stale between replays
        if self.block_size is not None:
            md.max_decode_seq_len = int(md.block_tables.shape[1]) * int(self.block_size)
        return md

    def use_cascade_attention(self, *args, **kwargs) -> bool:
        return False
