"""The reference-named Python surface on the GPU: MetaAttentionImpl.forward over a mixed
prefill + decode step, and the kernel-level functions with the reference's signatures."""
import types

import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128


def _inter(is_prompt, seq_id, ql, ctx, table):
    return types.SimpleNamespace(
        is_prompt=is_prompt, block_tables={seq_id: table}, seq_ids=[seq_id], input_tokens=[[0] * ql],
        orig_seq_lens=[ctx + ql], seq_lens=[ctx + ql], query_lens=[ql], context_lens=[ctx],
        curr_sliding_window_blocks=[0], prefix_cache_hit=False, multi_modal_placeholder_maps=None)


def test_impl_forward_mixed_chunked_prefill_and_decode():
    from oracle import meta_swa_paged_decode_ref, reshape_and_cache_ref
    from oracle.meta_swa import meta_swa_paged_prefill_ref
    from vllm_hymeta.attention.meta_attention import (MetaAttentionBackend, MetaAttentionImpl,
                                                      MetaAttentionMetadataBuilder)

    H, Hkv, bs, M, W = 7, 1, 16, 32, 48
    g_ = gen(0)
    # sequences: (is_prompt, query_len, context_len)
    seqs = [(True, 40, 0), (True, 25, 60), (False, 1, 90), (False, 1, 5)]
    nblk = 32
    perm = torch.randperm(nblk, generator=g_).tolist()
    tables, off = {}, 0
    for sid, (_, ql, ctx) in enumerate(seqs):
        nb = (ctx + ql + bs - 1) // bs
        tables[sid] = perm[off:off + nb]
        off += nb
    kv = randn_bf16(g_, *MetaAttentionBackend.get_kv_cache_shape(nblk, bs, Hkv, D))
    T = sum(ql for _, ql, _ in seqs)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    k2, v2 = randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)

    dev = torch.device("cuda")
    runner = types.SimpleNamespace(device=dev, pin_memory=False)
    ib = types.SimpleNamespace(runner=runner, sliding_window=None, block_size=bs, chunked_prefill_enabled=True,
                               inter_data_list=[_inter(p, sid, ql, ctx, tables[sid])
                                                for sid, (p, ql, ctx) in enumerate(seqs)])
    b = MetaAttentionMetadataBuilder(ib)
    b.prepare()
    md = b.build([ctx + ql for _, ql, ctx in seqs], [ql for _, ql, _ in seqs], -1, len(seqs))

    impl = MetaAttentionImpl(H, D, D ** -0.5, Hkv, None, W, "auto", num_meta_tokens=M)
    kv_dev = kv.to(dev)
    out = torch.zeros(T, H, D, dtype=torch.bfloat16, device=dev)
    impl.forward(None, q.to(dev), k.to(dev), v.to(dev), None, k2.to(dev), v2.to(dev), kv_dev, md, out)
    torch.cuda.synchronize()

    # oracle: write the step's KV, then prefills through the block table, then decodes
    kv_ref = kv.clone()
    reshape_and_cache_ref(k, v, kv_ref[0], kv_ref[1], md.slot_mapping.cpu())
    assert torch.equal(kv_dev.cpu().view(torch.int16), kv_ref.view(torch.int16))
    bt = md.block_tables.cpu()
    pre = meta_swa_paged_prefill_ref(q[:65], kv_ref[0], kv_ref[1], [0, 40, 65], [40, 85], bt[:2], k2, v2, W - 1)
    dec = meta_swa_paged_decode_ref(q[65:], kv_ref[0], kv_ref[1], [91, 6], bt[2:], k2, v2, W - 1)
    assert_bf16_close(out.cpu()[:65], pre, "impl.forward prefill rows")
    assert_bf16_close(out.cpu()[65:], dec, "impl.forward decode rows")


def test_impl_forward_warmup_with_meta_queries_no_cache():
    """Profiling / warm-up call: empty kv_cache, meta-token queries present, output rows = M + T."""
    from oracle import meta_swa_varlen_ref
    from vllm_hymeta.attention.meta_attention import MetaAttentionImpl, MetaAttentionMetadata

    H, Hkv, M, W = 7, 1, 32, 20
    lens = [33, 9]
    T = sum(lens)
    g_ = gen(1)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    q2, k2, v2 = randn_bf16(g_, M, H, D), randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    dev = torch.device("cuda")
    cu = torch.tensor([0, 33, 42], dtype=torch.int32, device=dev)
    md = MetaAttentionMetadata(
        num_prefills=2, num_prefill_tokens=T, num_decode_tokens=0,
        slot_mapping=torch.full((T,), -1, dtype=torch.long, device=dev), multi_modal_placeholder_index_maps=None,
        enable_kv_scales_calculation=True, seq_lens=lens, seq_lens_tensor=torch.tensor(lens, dtype=torch.int32, device=dev),
        max_prefill_seq_len=33, max_decode_seq_len=0, context_lens_tensor=torch.zeros(2, dtype=torch.int32, device=dev),
        block_tables=torch.zeros(2, 0, dtype=torch.int32, device=dev), use_cuda_graph=False, max_query_len=33,
        max_decode_query_len=1, query_start_loc=cu, seq_start_loc=cu)
    impl = MetaAttentionImpl(H, D, D ** -0.5, Hkv, None, W, "auto", num_meta_tokens=M)
    out = torch.zeros(M + T, H, D, dtype=torch.bfloat16, device=dev)
    impl.forward(None, q.to(dev), k.to(dev), v.to(dev), q2.to(dev), k2.to(dev), v2.to(dev),
                 torch.empty(0, device=dev), md, out)
    torch.cuda.synchronize()
    ref = meta_swa_varlen_ref(q, k, v, [0, 33, 42], [0, 33, 42], k2, v2, q2, W - 1)
    assert_bf16_close(out.cpu(), ref, "warm-up forward with meta queries")


def test_reference_named_kernel_functions():
    from oracle import gla_decode_step_ref, meta_swa_paged_decode_ref, meta_swa_varlen_ref
    from vllm_hymeta.attention.fa_kernel_metatoken import (metatoken_flash_attn_varlen_func,
                                                           metatoken_flash_attn_with_kvcache)
    from vllm_hymeta.model_for_7B.my_fused_recurrent import my_fused_recurrent_gla as gla_7b
    from vllm_hymeta.models.my_fused_recurrent_gla import my_fused_recurrent_gla

    dev = torch.device("cuda")
    g_ = gen(2)
    # GLA decode: [B, H, 1, D] in, [B, H*D] out (70B) / [B, H, D] (7B)
    B, H = 3, 4
    q, k, v = (randn_bf16(g_, B, H, 1, D) for _ in range(3))
    g = torch.nn.functional.logsigmoid(torch.randn(B, H, 1, D, generator=g_)) / 16
    st = randn_bf16(g_, 5, H, D, D, scale=0.1)
    slot = torch.tensor([4, 0, 2], dtype=torch.int32)
    st_ref = st.clone()
    o_ref = gla_decode_step_ref(q[:, :, 0], k[:, :, 0], v[:, :, 0], g[:, :, 0], st_ref, slot)
    sd = st.to(dev)
    o = my_fused_recurrent_gla(q.to(dev), k.to(dev), v.to(dev), g.to(dev), kv_caches=sd, slot_idx=slot.to(dev))
    assert o.shape == (B, H * D)
    assert_bf16_close(o.cpu().view(B, H, D), o_ref, "my_fused_recurrent_gla (70B)")
    o7 = gla_7b(q.to(dev), k.to(dev), v.to(dev), g.to(torch.bfloat16).to(dev), kv_caches=st.to(dev), slot_idx=slot.to(dev))
    assert o7.shape == (B, H, D)

    # varlen prefill: [1, M + T, H, D], meta rows first
    Hq, Hkv, M, T = 4, 2, 32, 50
    q1, k1, v1 = randn_bf16(g_, T, Hq, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    q2, k2, v2 = randn_bf16(g_, M, Hq, D), randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    cu = torch.tensor([0, T], dtype=torch.int32)
    out = metatoken_flash_attn_varlen_func(q1.to(dev), q2.to(dev), k1.to(dev), k2.to(dev), v1.to(dev), v2.to(dev),
                                           cu_seqlens_q=cu.to(dev), cu_seqlens_k=cu.to(dev), max_seqlen_q=T,
                                           max_seqlen_k=T, num_meta_tokens=M, causal=True, window_size=(15, 0))
    assert out.shape == (1, M + T, Hq, D)
    assert_bf16_close(out[0].cpu(), meta_swa_varlen_ref(q1, k1, v1, cu, cu, k2, v2, q2, 15), "varlen func")
    with pytest.raises(AssertionError):
        metatoken_flash_attn_varlen_func(q1.to(dev), None, k1.to(dev), k2.to(dev), v1.to(dev), v2.to(dev),
                                         cu_seqlens_q=cu.to(dev), cu_seqlens_k=cu.to(dev), causal=False)
    with pytest.raises(AssertionError):
        metatoken_flash_attn_varlen_func(q1.to(dev), None, k1.to(dev), k2.to(dev), v1.to(dev), v2.to(dev),
                                         cu_seqlens_q=cu.to(dev), cu_seqlens_k=cu.to(dev), causal=True,
                                         num_meta_tokens=M + 1)

    # paged decode: [B, 1, H, D]
    bs = 16
    kc, vc = randn_bf16(g_, 8, bs, Hkv, D), randn_bf16(g_, 8, bs, Hkv, D)
    bt = torch.tensor([[3, 5, 1], [0, 2, 7]], dtype=torch.int32)
    lens = torch.tensor([40, 17], dtype=torch.int32)
    qd = randn_bf16(g_, 2, Hq, D)
    o = metatoken_flash_attn_with_kvcache(qd.to(dev), None, k2.to(dev), v2.to(dev), key_cache=kc.to(dev),
                                          value_cache=vc.to(dev), cache_seqlens=lens.to(dev), num_meta_tokens=M,
                                          causal=True, window_size=(15, -1), block_tables=bt.to(dev))
    assert o.shape == (2, 1, Hq, D)
    ref = meta_swa_paged_decode_ref(qd, kc, vc, [40, 17], bt, k2, v2, 15)
    assert_bf16_close(o[:, 0].cpu(), ref, "with_kvcache func")
