"""CPU tests of the host-side mirror of the reference's plugin surface (no GPU, no vLLM)."""
import json
import os
import types

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_plugin_entry_points():
    import vllm_hymeta
    from vllm_hymeta.platform import META_BACKEND, STOCK_BACKEND, HymetaRocmPlatform

    assert vllm_hymeta.register() == "vllm_hymeta.platform.HymetaRocmPlatform"
    pick = HymetaRocmPlatform.get_attn_backend_cls
    assert pick(None, 128, torch.bfloat16, "auto", 256, False, False) == META_BACKEND
    assert pick("XFORMERS", 128, torch.bfloat16, "auto", 256, False, False) == STOCK_BACKEND
    # the dotted path resolves to the backend class
    mod, cls = META_BACKEND.rsplit(".", 1)
    backend = getattr(__import__(mod, fromlist=[cls]), cls)
    assert backend.get_name() == "META_FLASH_ATTN"
    assert backend.accept_output_buffer is True


def test_setup_declares_the_reference_entry_points():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "vllm-hymeta_amd", "setup.py")).read()
    assert "vllm.platform_plugins" in text and "hymeta = vllm_hymeta:register" in text
    assert "vllm.general_plugins" in text and "hymeta_model = vllm_hymeta:register_model" in text


def test_config_defaults_match_reference_fixture():
    """Fixture captured from the reference's own config classes (tests/golden/make_config_fixture.py)."""
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig

    ref = json.load(open(os.path.join(GOLD, "config_defaults.json")))
    for name, cls in (("HymetaConfig", HymetaConfig), ("GLAswaConfig", GLAswaConfig)):
        cfg = cls()
        for key, val in ref[name].items():
            assert getattr(cfg, key) == val, f"{name}.{key}: {getattr(cfg, key)!r} != {val!r}"
    assert HymetaConfig(quantization_config={"bits": 4}).quantization_config == {"bits": 4}


def test_kv_cache_shape_contract():
    from vllm_hymeta.attention.meta_attention import MetaAttentionBackend as B

    assert B.get_kv_cache_shape(7, 256, 4, 128) == (2, 7, 256, 4, 128)
    with pytest.raises(ValueError):
        B.get_kv_cache_shape(7, 8, 4, 128)
    with pytest.raises(ValueError):
        B.get_kv_cache_shape(7, 40, 4, 128)


def test_impl_constructor_error_behaviour():
    from vllm_hymeta.attention.meta_attention import MetaAttentionImpl as Impl

    ok = Impl(28, 128, 128 ** -0.5, 4, None, 4096, "auto")
    assert ok.sliding_window == (4095, 0) and ok.num_queries_per_kv == 7
    assert Impl(28, 128, 1.0, 4, None, None, "auto").sliding_window == (-1, -1)
    with pytest.raises(NotImplementedError):
        Impl(28, 128, 1.0, 4, None, None, "auto", kv_sharing_target_layer_name="x")
    with pytest.raises(ValueError):
        Impl(28, 128, 1.0, 4, None, None, "auto", blocksparse_params={})
    with pytest.raises(ValueError):
        Impl(28, 80, 1.0, 4, None, None, "auto")
    with pytest.raises(NotImplementedError):
        Impl(28, 128, 1.0, 4, None, None, "fp8")


def test_slot_allocator_lifecycle():
    from vllm_hymeta.state_cache import SlotAllocator

    copies = []
    alloc = SlotAllocator(4, lambda s, d: copies.append((s, d)))
    a = alloc.assign({"r0": [0], "r1": [1]}, [])
    assert len(set(a)) == 2 and all(0 <= s < 4 for s in a)
    # same request keeps its slot; a forked sequence gets a copy of its sibling's state
    b = alloc.assign({"r0": [0, 5], "r1": [1]}, [])
    assert b[0] == a[0] and b[2] == a[1] and b[1] not in a
    assert copies == [(a[0], b[1])]
    # finished request: its slots are released before assignment; scheduled-and-finished -> PAD
    c = alloc.assign({"r1": [1], "r2": [2], "r0": [0]}, ["r0"])
    assert c[0] == a[1] and c[2] == -1 and alloc.slot_of("r0", 0) is None
    assert alloc.num_free == 4 - 2
    with pytest.raises(RuntimeError):
        alloc.assign({"r3": [3], "r4": [4], "r5": [5]}, [])


def test_state_cache_manager_tensors_and_graph_buffers():
    from vllm_hymeta.model_for_7B.gla_cache import GLACacheManager, GLACacheParams
    from vllm_hymeta.models.hymeta_cache import HymetaCacheManager, HymetaCacheParams

    m = GLACacheManager(torch.bfloat16, (3, 4, 2, 8, 8), device="cpu")
    cache, idx = m.current_run_tensors(request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])
    assert cache.shape == (3, 4, 2, 8, 8) and idx.dtype == torch.int32 and idx.numel() == 2
    p = GLACacheParams(cache, idx).at_layer_idx(1)
    assert p.gla_cache.shape == (4, 2, 8, 8) and p.state_indices_tensor is idx
    bufs = {"seqlen_agnostic_capture_inputs": m.get_seqlen_agnostic_capture_inputs(4)}
    assert bufs["seqlen_agnostic_capture_inputs"][1].tolist() == [-1] * 4
    m.copy_inputs_before_cuda_graphs(bufs, request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])
    assert bufs["seqlen_agnostic_capture_inputs"][1].tolist() == idx.tolist() + [-1, -1]
    # fork copies the state of the sibling sequence along the slot axis
    cache[:, idx[0]] = 3.0
    _, idx2 = m.current_run_tensors(request_ids_to_seq_ids={"a": [0, 9]}, finished_requests_ids=["b"])
    assert (cache[:, idx2[1]] == 3.0).all()

    h = HymetaCacheManager(torch.bfloat16, (2, 3, 7, 8, 8), (2, 7, 8, 8), (2, 2, 16, 1, 8), device="cpu")
    st, meta_lin, meta_kv, idx = h.current_run_tensors(request_ids_to_seq_ids={"a": [0]}, finished_requests_ids=[])
    assert st.shape == (2, 3, 7, 8, 8) and meta_lin.shape == (2, 7, 8, 8) and meta_kv.shape == (2, 2, 16, 1, 8)
    q = HymetaCacheParams(st, meta_lin, meta_kv, idx).at_layer_idx(1)
    assert q.hymeta_cache.shape == (3, 7, 8, 8) and q.meta_fattn_cache.shape == (2, 16, 1, 8)


def _inter(is_prompt, seq_id, tokens, seq_len, query_len, context_len, table, window_blocks=0, hit=False):
    return types.SimpleNamespace(
        is_prompt=is_prompt, block_tables={seq_id: table} if table is not None else None, seq_ids=[seq_id],
        input_tokens=[tokens], orig_seq_lens=[seq_len], seq_lens=[seq_len], query_lens=[query_len],
        context_lens=[context_len], curr_sliding_window_blocks=[window_blocks], prefix_cache_hit=hit,
        multi_modal_placeholder_maps=None)


def test_metadata_builder_matches_oracle_slot_mapping():
    """Mixed step: one whole prompt, one chunked-prefill continuation, two decodes (bit-exact ints)."""
    from oracle import slot_mapping_ref
    from vllm_hymeta.attention.meta_attention import MetaAttentionMetadataBuilder

    bs = 16
    tables = {0: [5, 2], 1: [7, 1, 3], 2: [4], 3: [6, 0, 8, 9]}
    seqs = [  # (is_prompt, seq_id, query_len, context_len)
        (True, 0, 20, 0), (True, 1, 10, 30), (False, 2, 1, 12), (False, 3, 1, 50)]
    inter, seq_lens, query_lens, want_slots = [], [], [], []
    for is_prompt, sid, ql, ctx in seqs:
        L = ctx + ql
        inter.append(_inter(is_prompt, sid, list(range(ql)), L, ql, ctx, tables[sid]))
        seq_lens.append(L)
        query_lens.append(ql)
        want_slots += slot_mapping_ref(tables[sid], L, ctx, bs, is_prompt, ql)
    runner = types.SimpleNamespace(device=torch.device("cpu"), pin_memory=False)
    ib = types.SimpleNamespace(runner=runner, sliding_window=None, block_size=bs, inter_data_list=inter,
                               chunked_prefill_enabled=True)
    b = MetaAttentionMetadataBuilder(ib)
    b.prepare()
    md = b.build(seq_lens, query_lens, cuda_graph_pad_size=-1, batch_size=len(seqs))
    assert md.slot_mapping.dtype == torch.int64 and md.slot_mapping.tolist() == want_slots
    assert md.num_prefills == 2 and md.num_prefill_tokens == 30 and md.num_decode_tokens == 2
    assert md.query_start_loc.tolist() == [0, 20, 30, 31, 32] and md.query_start_loc.dtype == torch.int32
    assert md.seq_start_loc.tolist() == [0, 20, 60, 73, 124]
    assert md.context_lens_tensor.tolist() == [0, 30, 12, 50]
    assert md.block_tables.dtype == torch.int32
    assert md.block_tables.tolist() == [[5, 2, 0, 0], [7, 1, 3, 0], [4, 0, 0, 0], [6, 0, 8, 9]]
    assert md.max_prefill_seq_len == 40 and md.max_decode_seq_len == 51 and md.max_query_len == 20
    pre, dec = md.prefill_metadata, md.decode_metadata
    assert pre.num_prefill_tokens == 30 and pre.slot_mapping.numel() == 30 and pre.block_tables.shape[0] == 2
    assert pre.query_start_loc.tolist() == [0, 20, 30] and pre.seq_lens == [20, 40]
    assert dec.num_decode_tokens == 2 and dec.query_start_loc.tolist() == [0, 1, 2]
    assert dec.seq_lens_tensor.tolist() == [13, 51] and dec.block_tables.tolist() == [[4, 0, 0, 0], [6, 0, 8, 9]]


def test_metadata_builder_profile_run_and_window_trim():
    from vllm_hymeta.attention.meta_attention import MetaAttentionMetadataBuilder

    runner = types.SimpleNamespace(device=torch.device("cpu"), pin_memory=False)
    # profile run: no block tables -> every slot is PAD, block table rows empty
    ib = types.SimpleNamespace(runner=runner, sliding_window=None, block_size=16, chunked_prefill_enabled=False,
                               inter_data_list=[_inter(True, 0, [1, 2, 3], 3, 3, 0, None)])
    b = MetaAttentionMetadataBuilder(ib)
    b.prepare()
    md = b.build([3], [3], -1, 1)
    assert md.slot_mapping.tolist() == [-1, -1, -1] and md.block_tables.numel() == 0
    # decode with a trimmed window: only the last `window_blocks` blocks are handed to the kernel
    ib = types.SimpleNamespace(runner=runner, sliding_window=32, block_size=16, chunked_prefill_enabled=False,
                               inter_data_list=[_inter(False, 0, [9], 70, 1, 69, [3, 1, 4, 1, 5], window_blocks=3)])
    b = MetaAttentionMetadataBuilder(ib)
    b.prepare()
    md = b.build([38], [1], -1, 1)
    assert md.block_tables.tolist() == [[4, 1, 5]]
    assert md.slot_mapping.tolist() == [5 * 16 + 69 % 16]


def test_advance_step_updates_device_metadata():
    from vllm_hymeta.attention.meta_attention import MetaAttentionMetadata

    bs = 4
    bt = torch.tensor([[3, 1, 2], [0, 5, 4]], dtype=torch.int32)
    md = MetaAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=2, slot_mapping=torch.tensor([3 * 4 + 3, 5 * 4 + 0]),
        multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=True, seq_lens=[4, 5],
        seq_lens_tensor=torch.tensor([4, 5], dtype=torch.int32), max_prefill_seq_len=0, max_decode_seq_len=5,
        context_lens_tensor=torch.tensor([3, 4], dtype=torch.int32), block_tables=bt, use_cuda_graph=False,
        max_query_len=1, max_decode_query_len=1, query_start_loc=torch.tensor([0, 1, 2], dtype=torch.int32),
        seq_start_loc=torch.tensor([0, 4, 9], dtype=torch.int32))
    mi = types.SimpleNamespace(input_tokens=torch.zeros(2, dtype=torch.long), input_positions=torch.zeros(2, dtype=torch.long))
    md.advance_step(mi, torch.tensor([[11], [12]]), bs, num_seqs=2, num_queries=2)
    assert md.seq_lens == [5, 6] and md.seq_lens_tensor.tolist() == [5, 6] and md.max_decode_seq_len == 6
    assert mi.input_tokens.tolist() == [11, 12] and mi.input_positions.tolist() == [4, 5]
    assert md.slot_mapping.tolist() == [1 * 4 + 0, 5 * 4 + 1]
