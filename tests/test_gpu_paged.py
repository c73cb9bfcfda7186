"""Paged KV write (bit-exact) and paged sliding-window + meta-token decode, through the C ABI."""
import pytest
import torch

from _util import assert_bf16_close, assert_bf16_ulp, gen, randn_bf16

pytestmark = pytest.mark.gpu

D = 128


def _paged_problem(seed, seq_lens, Hkv, bs, extra_blocks=3):
    """Random caches with every sequence's blocks scattered through a permuted pool."""
    g_ = gen(seed)
    nb_per = [(L + bs - 1) // bs for L in seq_lens]
    max_blocks = max(max(nb_per), 1)
    nblk = sum(nb_per) + extra_blocks
    perm = torch.randperm(nblk, generator=g_)
    bt = torch.zeros(len(seq_lens), max_blocks, dtype=torch.int32)
    off = 0
    for i, n in enumerate(nb_per):
        bt[i, :n] = perm[off:off + n].to(torch.int32)
        off += n
    kc = randn_bf16(g_, nblk, bs, Hkv, D)
    vc = randn_bf16(g_, nblk, bs, Hkv, D)
    return kc, vc, bt


@pytest.mark.parametrize("Hkv,bs,T", [(4, 256, 37), (1, 16, 300), (4, 64, 1)])
def test_reshape_and_cache_bit_exact(Hkv, bs, T):
    from oracle import reshape_and_cache_ref
    from vllm_hymeta import hip_ops

    g_ = gen(11)
    nblk = 9
    key, value = randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    kv = randn_bf16(g_, 2, nblk, bs, Hkv, D)
    slots = torch.randperm(nblk * bs, generator=g_)[:T].to(torch.int64)
    slots[::5] = -1  # PAD_SLOT_ID
    ref = kv.clone()
    reshape_and_cache_ref(key, value, ref[0], ref[1], slots)
    dev = kv.cuda()
    hip_ops.reshape_and_cache(key.cuda(), value.cuda(), dev[0], dev[1], slots.cuda())
    torch.cuda.synchronize()
    assert torch.equal(dev.cpu().view(torch.int16), ref.view(torch.int16))


def test_reshape_and_cache_strided_source_and_short_mapping():
    """K/V as views of a fused qkv buffer; slot_mapping shorter than the token count."""
    from oracle import reshape_and_cache_ref
    from vllm_hymeta import hip_ops

    g_ = gen(12)
    T, H, Hkv, bs, nblk = 19, 28, 4, 32, 4
    qkv = randn_bf16(g_, T, (H + 2 * Hkv) * D)
    key = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
    value = qkv[:, (H + Hkv) * D:].view(T, Hkv, D)
    kv = torch.zeros(2, nblk, bs, Hkv, D, dtype=torch.bfloat16)
    slots = torch.randperm(nblk * bs, generator=g_)[:T - 4].to(torch.int64)
    ref = kv.clone()
    reshape_and_cache_ref(key, value, ref[0], ref[1], slots)
    qkv_d, dev = qkv.cuda(), kv.cuda()
    kd = qkv_d[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
    vd = qkv_d[:, (H + Hkv) * D:].view(T, Hkv, D)
    hip_ops.reshape_and_cache(kd, vd, dev[0], dev[1], slots.cuda())
    torch.cuda.synchronize()
    assert torch.equal(dev.cpu().view(torch.int16), ref.view(torch.int16))


W = 64  # test window (keys visible incl. self), so window_left = W - 1


@pytest.mark.parametrize("H,Hkv,bs,M,window_left,seq_lens,splits", [
    (28, 4, 16, 0, W, [1, W - 1, W, W + 1, 3 * W, 5], 0),        # 7B: no meta, window W+1 keys
    (7, 1, 16, 32, W - 1, [1, W - 1, W, W + 1, 3 * W, 200], 0),  # 70B shard: meta + window
    (7, 1, 32, 128, -1, [1, 33, 500, 1000], 0),                  # 70B full-attention layer
    (28, 4, 256, 128, 4095, [5000, 4096, 4097, 17], 0),          # real geometry, auto splits
    (14, 2, 64, 64, 127, [700, 130, 1], 1),                      # forced single pass
    (14, 2, 64, 64, 127, [700, 130, 1], 5),                      # forced 5 splits
    (32, 1, 16, 0, -1, [77, 300], 3),                            # 32 q-heads per kv-head: two head groups
])
def test_paged_decode(H, Hkv, bs, M, window_left, seq_lens, splits):
    from oracle import meta_swa_paged_decode_ref
    from vllm_hymeta import hip_ops

    B = len(seq_lens)
    kc, vc, bt = _paged_problem(21, seq_lens, Hkv, bs)
    g_ = gen(22)
    q = randn_bf16(g_, B, H, D)
    k2 = randn_bf16(g_, M, Hkv, D) if M else None
    v2 = randn_bf16(g_, M, Hkv, D) if M else None
    lens = torch.tensor(seq_lens, dtype=torch.int32)
    ref = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, k2, v2, window_left)
    out = hip_ops.meta_swa_paged_decode(
        q.cuda(), kc.cuda(), vc.cuda(), lens.cuda(), bt.cuda(),
        None if k2 is None else k2.cuda(), None if v2 is None else v2.cuda(),
        window_left=window_left, num_splits=splits)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"paged decode H={H} Hkv={Hkv} bs={bs} M={M} wl={window_left}")


def test_paged_decode_ignores_garbage_outside_the_window():
    """NaN/Inf in cache rows that are masked (beyond seq_len, before the window, unused blocks)
    must not reach the output."""
    from oracle import meta_swa_paged_decode_ref
    from vllm_hymeta import hip_ops

    H, Hkv, bs, M, wl = 7, 1, 16, 32, 40
    seq_lens = [100, 7, 49]
    kc, vc, bt = _paged_problem(31, seq_lens, Hkv, bs)
    g_ = gen(32)
    q = randn_bf16(g_, len(seq_lens), H, D)
    k2, v2 = randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    ref = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, k2, v2, wl)
    kc_bad, vc_bad = kc.clone(), vc.clone()
    visible = torch.zeros(kc.shape[0], bs, dtype=torch.bool)
    for b, L in enumerate(seq_lens):
        for p in range(max(0, L - 1 - wl), L):
            visible[int(bt[b, p // bs]), p % bs] = True
    kc_bad[~visible] = float("nan")
    vc_bad[~visible] = float("inf")
    out = hip_ops.meta_swa_paged_decode(
        q.cuda(), kc_bad.cuda(), vc_bad.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(),
        bt.cuda(), k2.cuda(), v2.cuda(), window_left=wl)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, "paged decode with poisoned masked rows")


def test_paged_decode_softmax_spike():
    """One key far above the rest late in the sequence forces the online-softmax rescale."""
    from oracle import meta_swa_paged_decode_ref
    from vllm_hymeta import hip_ops

    H, Hkv, bs = 7, 1, 16
    seq_lens = [640]
    kc, vc, bt = _paged_problem(41, seq_lens, Hkv, bs)
    g_ = gen(42)
    q = randn_bf16(g_, 1, H, D)
    # position 601 aligned with head 3's query: score ~ |q|^2 * 3
    blk, off = int(bt[0, 601 // bs]), 601 % bs
    kc[blk, off, 0] = (q[0, 3].float() * 3).to(torch.bfloat16)
    ref = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, None, None, -1)
    for splits in (1, 4):
        out = hip_ops.meta_swa_paged_decode(q.cuda(), kc.cuda(), vc.cuda(),
                                            torch.tensor(seq_lens, dtype=torch.int32).cuda(), bt.cuda(),
                                            window_left=-1, num_splits=splits)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), ref, f"paged decode spike, splits={splits}")


def test_paged_decode_workspace_error():
    from vllm_hymeta import _C, hip_ops
    import ctypes as C

    lib = _C.load()
    assert lib.hymeta_meta_swa_paged_decode_workspace_bytes(4, 28, 128, 1) == 0
    assert lib.hymeta_meta_swa_paged_decode_workspace_bytes(4, 28, 128, 3) == 4 * 28 * 3 * 130 * 4
    a = _C.PagedDecodeArgs()
    assert lib.hymeta_meta_swa_paged_decode(C.byref(a), None) == -1  # null pointers


@pytest.mark.parametrize("M,window_left,splits", [(0, 100, 1), (32, 63, 3), (0, -1, 0)])
def test_paged_decode_fused_kv_write_equals_write_then_decode(M, window_left, splits):
    """The decode call that also stores the decoded token's K/V row: same cache bytes and same
    output bits as hymeta_reshape_and_cache followed by the plain decode (skipped slot included)."""
    from vllm_hymeta import hip_ops

    H, Hkv, bs = 8, 2, 16
    seq_lens = [1, 16, 17, 130, 257, 64]
    kc, vc, bt = _paged_problem(21, seq_lens, Hkv, bs)
    g_ = gen(22)
    B = len(seq_lens)
    q = randn_bf16(g_, B, H, D)
    qkv = randn_bf16(g_, B, 2 * Hkv * D)                 # new K/V as views of one fused buffer
    nk, nv = qkv[:, : Hkv * D].view(B, Hkv, D), qkv[:, Hkv * D:].view(B, Hkv, D)
    k2 = randn_bf16(g_, M, Hkv, D) if M else None
    v2 = randn_bf16(g_, M, Hkv, D) if M else None
    slots = torch.tensor([bt[i, (L - 1) // bs].item() * bs + (L - 1) % bs for i, L in enumerate(seq_lens)],
                         dtype=torch.int64)
    slots[3] = -1                                        # padded entry: nothing written
    lens = torch.tensor(seq_lens, dtype=torch.int32).cuda()
    dev = lambda t: None if t is None else t.cuda()

    k_a, v_a = kc.cuda(), vc.cuda()
    hip_ops.reshape_and_cache(nk.cuda(), nv.cuda(), k_a, v_a, slots.cuda())
    out_a = hip_ops.meta_swa_paged_decode(q.cuda(), k_a, v_a, lens, bt.cuda(), meta_key=dev(k2), meta_value=dev(v2),
                                          window_left=window_left, num_splits=splits)
    k_b, v_b = kc.cuda(), vc.cuda()
    qkv_d = qkv.cuda()
    out_b = hip_ops.meta_swa_paged_decode(q.cuda(), k_b, v_b, lens, bt.cuda(), meta_key=dev(k2), meta_value=dev(v2),
                                          window_left=window_left, num_splits=splits,
                                          new_key=qkv_d[:, : Hkv * D].view(B, Hkv, D),
                                          new_value=qkv_d[:, Hkv * D:].view(B, Hkv, D), slot_mapping=slots.cuda())
    torch.cuda.synchronize()
    assert torch.equal(k_a.cpu().view(torch.int16), k_b.cpu().view(torch.int16))
    assert torch.equal(v_a.cpu().view(torch.int16), v_b.cpu().view(torch.int16))
    assert torch.equal(out_a.cpu().view(torch.int16), out_b.cpu().view(torch.int16))


@pytest.mark.parametrize("H,Hkv,bs,M,window_left,seq_lens,splits", [
    (28, 4, 16, 0, W, [1, W - 1, W, W + 1, 3 * W, 5], 1),        # 7B geometry, one pass
    (28, 4, 256, 128, 4095, [5000, 4096, 4097, 17], 3),          # key splits + merge kernel
    (14, 2, 64, 64, 127, [700, 130, 1], 5),                      # two kv-heads: four waves per kv-head
    (8, 2, 16, 0, -1, [77, 300, 31], 2),
])
def test_paged_decode_all_kv_heads_per_workgroup(H, Hkv, bs, M, window_left, seq_lens, splits, monkeypatch):
    """The workgroup shape used when the sequences alone fill the chip (all kv-heads of a sequence in one
    workgroup), forced here at small batch; two calls in a row."""
    from oracle import meta_swa_paged_decode_ref
    from vllm_hymeta import hip_ops

    monkeypatch.setenv("HYMETA_PAGED_SHAPE", "all")
    B = len(seq_lens)
    kc, vc, bt = _paged_problem(51, seq_lens, Hkv, bs)
    g_ = gen(52)
    k2 = randn_bf16(g_, M, Hkv, D) if M else None
    v2 = randn_bf16(g_, M, Hkv, D) if M else None
    lens = torch.tensor(seq_lens, dtype=torch.int32)
    for call in range(2):
        q = randn_bf16(g_, B, H, D)
        ref = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, k2, v2, window_left)
        out = hip_ops.meta_swa_paged_decode(
            q.cuda(), kc.cuda(), vc.cuda(), lens.cuda(), bt.cuda(),
            None if k2 is None else k2.cuda(), None if v2 is None else v2.cuda(),
            window_left=window_left, num_splits=splits)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), ref, f"paged decode, all kv-heads per workgroup, call {call}")


@pytest.mark.parametrize("splits", [1, 3])
def test_paged_decode_all_kv_heads_fused_kv_write(splits, monkeypatch):
    """Fused KV write in the all-kv-heads shape: same cache bytes and output bits as write-then-decode."""
    from vllm_hymeta import hip_ops

    monkeypatch.setenv("HYMETA_PAGED_SHAPE", "all")
    H, Hkv, bs = 28, 4, 16
    seq_lens = [1, 16, 17, 130, 257, 64]
    kc, vc, bt = _paged_problem(61, seq_lens, Hkv, bs)
    g_ = gen(62)
    B = len(seq_lens)
    q = randn_bf16(g_, B, H, D).cuda()
    nk, nv = randn_bf16(g_, B, Hkv, D).cuda(), randn_bf16(g_, B, Hkv, D).cuda()
    slots = torch.tensor([bt[i, (L - 1) // bs].item() * bs + (L - 1) % bs for i, L in enumerate(seq_lens)],
                         dtype=torch.int64)
    slots[3] = -1
    lens = torch.tensor(seq_lens, dtype=torch.int32).cuda()
    k_a, v_a = kc.cuda(), vc.cuda()
    hip_ops.reshape_and_cache(nk, nv, k_a, v_a, slots.cuda())
    out_a = hip_ops.meta_swa_paged_decode(q, k_a, v_a, lens, bt.cuda(), window_left=100, num_splits=splits)
    k_b, v_b = kc.cuda(), vc.cuda()
    out_b = hip_ops.meta_swa_paged_decode(q, k_b, v_b, lens, bt.cuda(), window_left=100, num_splits=splits,
                                          new_key=nk, new_value=nv, slot_mapping=slots.cuda())
    torch.cuda.synchronize()
    assert torch.equal(k_a.cpu().view(torch.int16), k_b.cpu().view(torch.int16))
    assert torch.equal(v_a.cpu().view(torch.int16), v_b.cpu().view(torch.int16))
    assert torch.equal(out_a.cpu().view(torch.int16), out_b.cpu().view(torch.int16))


def test_paged_decode_through_a_window_trimmed_block_table():
    """What the engine hands the backend for decodes under a sliding window (the V0 runner's lengths clamped to the
    block-aligned window plus the last block's fill, and the builder's table cut to the window's last blocks,
    reference vllm_hymeta/attention/meta_attention.py:479-485; builder here: MetaAttentionMetadataBuilder._add_seq_group):
    the kernel reads the same physical rows through the trimmed table as through the whole one, the NaNs that fill every
    row outside the trimmed view never show, and the result is the oracle's for the untrimmed sequence."""
    import types

    from oracle import meta_swa_paged_decode_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta.attention.meta_attention import MetaAttentionMetadataBuilder

    H, Hkv, bs, M, window = 7, 1, 16, 32, 32
    window_blocks = (window + bs - 1) // bs
    seq_lens = [70, 64, 33, 20, 129]                 # last block part full / full; one sequence shorter than the window
    kc, vc, bt = _paged_problem(41, seq_lens, Hkv, bs)
    g_ = gen(42)
    q = randn_bf16(g_, len(seq_lens), H, D)
    k2, v2 = randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    ref = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, k2, v2, window - 1)

    # the engine's view: lengths clamped, and how many trailing blocks of each table the builder keeps
    inter, clamped = [], []
    for i, L in enumerate(seq_lens):
        suff = L % bs
        cur = min(L, window_blocks * bs + suff)
        nblk = window_blocks + (1 if suff > 0 else 0)
        table = bt[i, : (L + bs - 1) // bs].tolist()
        inter.append(types.SimpleNamespace(
            is_prompt=False, block_tables={i: table}, seq_ids=[i], input_tokens=[[0]], orig_seq_lens=[L], seq_lens=[cur],
            query_lens=[1], context_lens=[L - 1], curr_sliding_window_blocks=[nblk], prefix_cache_hit=False,
            multi_modal_placeholder_maps=None))
        clamped.append(cur)
    runner = types.SimpleNamespace(device=torch.device("cuda"), pin_memory=False)
    ib = types.SimpleNamespace(runner=runner, sliding_window=window, block_size=bs, chunked_prefill_enabled=False,
                               inter_data_list=inter)
    b = MetaAttentionMetadataBuilder(ib)
    b.prepare()
    md = b.build(clamped, [1] * len(seq_lens), -1, len(seq_lens))
    assert md.seq_lens_tensor.tolist() == clamped and md.block_tables.shape[1] == window_blocks + 1
    for i, L in enumerate(seq_lens):                  # the new token's slot still comes from the whole table
        assert int(md.slot_mapping[i]) == int(bt[i, (L - 1) // bs]) * bs + (L - 1) % bs

    # poison every cache row the trimmed view does not reach
    reach = torch.zeros(kc.shape[0], bs, dtype=torch.bool)
    for i, cur in enumerate(clamped):
        for p in range(cur):
            reach[int(md.block_tables[i, p // bs]), p % bs] = True
    kc_bad, vc_bad = kc.clone(), vc.clone()
    kc_bad[~reach] = float("nan")
    vc_bad[~reach] = float("nan")
    out = hip_ops.meta_swa_paged_decode(q.cuda(), kc_bad.cuda(), vc_bad.cuda(), md.seq_lens_tensor, md.block_tables, k2.cuda(), v2.cuda(),
                                        window_left=window - 1)
    whole = hip_ops.meta_swa_paged_decode(q.cuda(), kc.cuda(), vc.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(), bt.cuda(),
                                          k2.cuda(), v2.cuda(), window_left=window - 1)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, "paged decode through the trimmed table")
    assert_bf16_ulp(out.cpu(), whole.cpu(), "trimmed against whole table", ulps=1)
