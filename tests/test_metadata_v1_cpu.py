"""The V1-engine metadata builder on CPU: from a duck-typed CommonAttentionMetadata to the MetaAttentionMetadata the
kernels' dispatch reads, equal to what the V0 builder produces for the same step; batch reordering."""
from types import SimpleNamespace

import pytest
import torch

from vllm_hymeta.attention.meta_attention_v1 import MetaAttentionMetadataBuilderV1, split_prefills_and_decodes


def _common(query_lens, seq_lens, block_size=16, max_blocks=8):
    n = len(query_lens)
    qsl = torch.tensor([0] + list(torch.tensor(query_lens).cumsum(0)), dtype=torch.int32)
    bt = torch.arange(n * max_blocks, dtype=torch.int32).view(n, max_blocks).flip(0)      # request i owns row n-1-i's numbers
    slots = []
    for i, (q, s) in enumerate(zip(query_lens, seq_lens)):
        for pos in range(s - q, s):
            slots.append(int(bt[i, pos // block_size]) * block_size + pos % block_size)
    return SimpleNamespace(query_start_loc=qsl, query_start_loc_cpu=qsl.clone(), seq_lens=torch.tensor(seq_lens, dtype=torch.int32),
                           seq_lens_cpu=torch.tensor(seq_lens, dtype=torch.int32),
                           num_computed_tokens_cpu=torch.tensor([s - q for q, s in zip(query_lens, seq_lens)]),
                           num_reqs=n, num_actual_tokens=sum(query_lens), max_query_len=max(query_lens),
                           block_table_tensor=bt, slot_mapping=torch.tensor(slots, dtype=torch.int64)), bt


def test_build_mixed_step_matches_the_v0_fields():
    from oracle.paged import slot_mapping_ref

    query_lens, seq_lens = [40, 7, 1, 1], [40, 30, 55, 9]        # a whole prompt, a chunk after 23 cached tokens, two decodes
    common, bt = _common(query_lens, seq_lens)
    b = MetaAttentionMetadataBuilderV1(SimpleNamespace(block_size=16), ["model.layers.1.attn.attn"], None, "cpu")
    md = b.build(0, common)
    assert (md.num_prefills, md.num_prefill_tokens, md.num_decode_tokens) == (2, 47, 2)
    assert md.seq_lens == seq_lens and md.seq_lens_tensor.tolist() == seq_lens
    assert md.context_lens_tensor.tolist() == [0, 23, 54, 8]
    assert md.query_start_loc.tolist() == [0, 40, 47, 48, 49] and md.seq_start_loc.tolist() == [0, 40, 70, 125, 134]
    assert (md.max_query_len, md.max_prefill_seq_len, md.max_decode_seq_len, md.max_decode_query_len) == (40, 40, 55, 1)
    assert torch.equal(md.block_tables, bt) and md.slot_mapping.dtype == torch.long
    # the slots are the ones the reference's host loop computes for the same tables
    ref = []
    for i, (q, s) in enumerate(zip(query_lens, seq_lens)):
        ref += slot_mapping_ref(bt[i].tolist(), s, s - q, 16, q > 1, q)
    assert md.slot_mapping.tolist() == ref
    # the views the dispatch takes
    pre, dec = md.prefill_metadata, md.decode_metadata
    assert pre.num_prefill_tokens == 47 and pre.context_lens_tensor.tolist() == [0, 23] and pre.block_tables.shape[0] == 2
    assert dec.num_decode_tokens == 2 and dec.seq_lens_tensor.tolist() == [55, 9] and dec.query_start_loc.tolist() == [0, 1, 2]


def test_decodes_before_prefills_are_refused_and_reordered():
    with pytest.raises(ValueError):
        split_prefills_and_decodes([1, 5, 1])
    assert split_prefills_and_decodes([5, 3, 1, 1]) == 2 and split_prefills_and_decodes([1, 1]) == 0

    class Batch:
        def __init__(self, ids):
            self.req_ids = list(ids)

        def swap_states(self, i, j):
            self.req_ids[i], self.req_ids[j] = self.req_ids[j], self.req_ids[i]

    sched = SimpleNamespace(num_scheduled_tokens={"d0": 1, "p0": 9, "d1": 1, "p1": 2, "d2": 1})
    batch = Batch(["d0", "p0", "d1", "p1", "d2"])
    b = MetaAttentionMetadataBuilderV1()
    assert b.reorder_batch(batch, sched) is True
    assert batch.req_ids == ["p0", "p1", "d0", "d1", "d2"]          # both groups keep their order
    assert b.reorder_batch(batch, sched) is False


def test_a_new_prompt_of_one_token_is_a_prefill():
    """Scheduled for one token with nothing computed yet: a prompt (the reference's V0 builder goes by is_prompt), so
    that the prefill pass resets its recurrent-state slot; a decode has tokens behind it."""
    assert split_prefills_and_decodes([1, 1], [0, 7]) == 1
    with pytest.raises(ValueError):
        split_prefills_and_decodes([1, 1], [7, 0])            # the new prompt behind a decode: reorder first
    common, _ = _common([1, 1, 1], [1, 30, 9])                  # request 0: a one-token prompt
    md = MetaAttentionMetadataBuilderV1(SimpleNamespace(block_size=16)).build(0, common)
    assert (md.num_prefills, md.num_prefill_tokens, md.num_decode_tokens) == (1, 1, 2)
    assert md.prefill_metadata.context_lens_tensor.tolist() == [0] and md.decode_metadata.seq_lens_tensor.tolist() == [30, 9]

    class Batch:
        def __init__(self, ids, computed):
            self.req_ids, self.num_computed_tokens_cpu = list(ids), list(computed)

        def swap_states(self, i, j):
            self.req_ids[i], self.req_ids[j] = self.req_ids[j], self.req_ids[i]
            self.num_computed_tokens_cpu[i], self.num_computed_tokens_cpu[j] = self.num_computed_tokens_cpu[j], self.num_computed_tokens_cpu[i]

    sched = SimpleNamespace(num_scheduled_tokens={"d0": 1, "new": 1, "d1": 1})
    batch = Batch(["d0", "new", "d1"], [12, 0, 5])
    assert MetaAttentionMetadataBuilderV1().reorder_batch(batch, sched) is True
    assert batch.req_ids == ["new", "d0", "d1"]


def test_capture_metadata_is_pure_decode_over_the_whole_table():
    common, _ = _common([1, 1, 1], [17, 90, 3])
    b = MetaAttentionMetadataBuilderV1(SimpleNamespace(block_size=16))
    md = b.build_for_cudagraph_capture(common)
    assert md.use_cuda_graph and md.num_decode_tokens == 3 and md.seq_lens is None
    assert md.max_decode_seq_len == 8 * 16
    # what a captured kernel reads IS the runner's buffer; what was derived at build time is absent, not stale
    same = lambda a, b_: a.data_ptr() == b_.data_ptr() and a.dtype == b_.dtype
    assert same(md.seq_lens_tensor, common.seq_lens) and same(md.slot_mapping, common.slot_mapping) and same(md.block_tables, common.block_table_tensor)
    assert md.context_lens_tensor is None and md.seq_start_loc is None and md.query_start_loc is None
    dec = md.decode_metadata
    assert same(dec.seq_lens_tensor, common.seq_lens) and dec.num_decode_tokens == 3
    converted, _ = _common([1, 1, 1], [17, 90, 3])
    converted.seq_lens = converted.seq_lens.long()
    with pytest.raises(TypeError):
        b.build_for_cudagraph_capture(converted)
    mixed, _ = _common([4, 1], [4, 20])
    with pytest.raises(ValueError):
        b.build_for_cudagraph_capture(mixed)
    assert b.use_cascade_attention() is False
