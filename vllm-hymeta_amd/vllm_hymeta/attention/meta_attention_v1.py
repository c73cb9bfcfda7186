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


def is_prefill_request(query_len: int, num_computed_tokens: int) -> bool:
    """A request runs through the prefill kernels when it is scheduled for more than one token OR has nothing computed
    yet: a brand-new prompt of ONE token is a prompt, as the reference's V0 path treats it (``is_prompt``,
    ``vllm_hymeta/attention/meta_attention.py:435-470``) -- its recurrent-state slot is reset by the prefill pass
    (``context_lens == 0``), which a decode step would skip."""
    return query_len > 1 or num_computed_tokens == 0


def split_prefills_and_decodes(query_lens: List[int], computed: Optional[List[int]] = None) -> int:
    """Number of leading prefill requests (``is_prefill_request``) of a batch ordered prefills-first; raises if a prefill
    follows a decode (the kernels index decode rows after all prefill rows).  ``computed``: tokens already computed per
    request; without it only the query lengths decide."""
    if computed is None:
        computed = [1] * len(query_lens)
    flags = [is_prefill_request(q, c) for q, c in zip(query_lens, computed)]
    n = 0
    while n < len(flags) and flags[n]:
        n += 1
    if any(flags[n:]) or any(q != 1 for q in query_lens[n:]):
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
        """Put the prefill requests (``is_prefill_request``) in front (stable on both sides).  ``input_batch`` has
        ``req_ids``, ``swap_states(i, j)`` and, per request, ``num_computed_tokens_cpu``;
        ``scheduler_output.num_scheduled_tokens`` maps request id -> tokens of this step.  Returns whether anything moved."""
        ids = list(input_batch.req_ids)
        computed = getattr(input_batch, "num_computed_tokens_cpu", None)
        is_prefill = [is_prefill_request(scheduler_output.num_scheduled_tokens[r], int(computed[i]) if computed is not None else 1)
                      for i, r in enumerate(ids)]
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
        computed = getattr(m, "num_computed_tokens_cpu", None)
        computed = [int(c) for c in computed[:num_reqs]] if computed is not None else [s - q for s, q in zip(seq_lens, query_lens)]
        n_pre = split_prefills_and_decodes(query_lens, computed)
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
        """Pure-decode batch for a graph.  The captured kernels (fused KV write + single-token paged decode, GLA decode)
        read three tensors of the metadata: ``seq_lens_tensor``, ``block_tables`` and ``slot_mapping``.  Each must BE the
        runner's persistent buffer (a view: contents change between replays, the address does not), so a buffer whose
        dtype would need a converted copy is refused.  Everything derived at build time -- context lengths, start
        offsets, the host list of lengths -- would be stale on the first replay and is left out (``None``): a decode
        kernel that reached for one would fail at capture, not read old numbers.  ``max_decode_seq_len`` is what the
        graph's kernels size their key splits for: the longest sequence the buffers can describe."""
        m = common_attn_metadata
        num_reqs, num_tokens = int(m.num_reqs), int(m.num_actual_tokens)
        if num_tokens != num_reqs:
            raise ValueError("only pure-decode batches are captured")
        if m.seq_lens.dtype != torch.int32 or m.slot_mapping.dtype != torch.long or m.block_table_tensor.dtype != torch.int32:
            raise TypeError("graph capture needs the runner's buffers as they are: seq_lens int32, slot_mapping int64, "
                            "block_table int32 (a dtype conversion would capture a copy)")
        block_tables = m.block_table_tensor[:num_reqs]
        longest = int(block_tables.shape[1]) * int(self.block_size) if self.block_size is not None else int(m.seq_lens_cpu[:num_reqs].max())
        return MetaAttentionMetadata(
            num_prefills=0, num_prefill_tokens=0, num_decode_tokens=num_tokens,
            slot_mapping=m.slot_mapping[:num_tokens],
            multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=False,
            seq_lens=None, seq_lens_tensor=m.seq_lens[:num_reqs],
            max_prefill_seq_len=0, max_decode_seq_len=longest,
            context_lens_tensor=None, block_tables=block_tables,
            use_cuda_graph=True, max_query_len=1, max_decode_query_len=1,
            query_start_loc=None, seq_start_loc=None)

    def use_cascade_attention(self, *args, **kwargs) -> bool:
        return False
