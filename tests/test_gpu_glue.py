"""RoPE, the 70B two-branch norm combine and the blocks' RMS norm vs the CPU restatement."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128


def test_rope_neox_in_place_with_meta_offset():
    from oracle.layers import rope_cos_sin_cache_ref, rope_neox_ref
    from vllm_hymeta import hip_ops

    T, H, Hkv, M = 37, 7, 1, 128
    g_ = gen(0)
    qkv = randn_bf16(g_, T, (H + 2 * Hkv) * D)
    cache = rope_cos_sin_cache_ref(D, 4096, 1e6)
    pos = torch.randint(0, 3000, (T,), generator=g_)
    q = qkv[:, : H * D].view(T, H, D)
    k = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
    q_ref, k_ref = rope_neox_ref(pos + M, q, k, cache)
    dev = qkv.cuda()
    qd = dev[:, : H * D].view(T, H, D)
    kd = dev[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
    hip_ops.rope_neox_(pos.cuda(), qd, kd, cache.cuda(), position_offset=M)
    torch.cuda.synchronize()
    # same bf16 arithmetic on both sides: exact up to fused-multiply-add contraction of the sum
    assert (qd.cpu().float() - q_ref.float()).abs().max() <= 2.0 ** -7 * q_ref.float().abs().max()
    assert_bf16_close(qd.cpu(), q_ref, "rope q", rel=2e-3)
    assert_bf16_close(kd.cpu(), k_ref, "rope k", rel=2e-3)
    # v (the rest of the fused buffer) untouched
    assert torch.equal(dev.cpu()[:, (H + Hkv) * D:], qkv[:, (H + Hkv) * D:])


def test_rope_last_position_with_meta_offset_stays_inside_the_table():
    """The 70B model ropes real token p at p + num_meta_tokens, so its table holds max_pos + M rows: the last
    position of a full-length sequence reads the last row.  A position past the table is clamped to that row
    (never read out of bounds)."""
    from oracle.layers import rope_cos_sin_cache_ref, rope_neox_ref
    from vllm_hymeta import hip_ops

    H, Hkv, M, max_pos = 7, 1, 128, 512
    g_ = gen(3)
    cache = rope_cos_sin_cache_ref(D, max_pos + M, 1e6)
    pos = torch.tensor([max_pos - 1, 0, max_pos + 5], dtype=torch.int64)       # last row, first row, beyond the table
    q, k = randn_bf16(g_, 3, H, D), randn_bf16(g_, 3, Hkv, D)
    q_ref, k_ref = rope_neox_ref(torch.tensor([max_pos - 1 + M, M, max_pos + M - 1]), q, k, cache)
    # the table sits at the very end of its allocation
    buf = torch.zeros((max_pos + M) * D, dtype=torch.bfloat16, device="cuda")
    buf.copy_(cache.reshape(-1))
    qd, kd = q.cuda(), k.cuda()
    hip_ops.rope_neox_(pos.cuda(), qd, kd, buf.view(max_pos + M, D), position_offset=M)
    torch.cuda.synchronize()
    assert_bf16_close(qd.cpu(), q_ref, "rope q at the table's last row", rel=2e-3)
    assert_bf16_close(kd.cpu(), k_ref, "rope k at the table's last row", rel=2e-3)


def test_70b_model_rope_table_covers_the_meta_offset():
    from types import SimpleNamespace
    from vllm_hymeta.models.modeling_hymeta import HymetaModel

    cfg = SimpleNamespace(num_attention_heads=4, num_key_value_heads=2, head_dim=D, hidden_size=4 * D, num_hidden_layers=1,
                          use_lower_bound=False, num_meta_tokens=128, max_position_embeddings=300, rope_theta=1e6)
    m = HymetaModel(cfg, max_num_seqs=2, device="cuda")
    assert m.cos_sin_cache.shape == (300 + 128, D)


@pytest.mark.parametrize("tp", [1, 4])
def test_hybrid_combine(tp):
    from oracle.layers import hybrid_combine_ref, rms_norm_tp_ref
    from vllm_hymeta import hip_ops

    T, hidden = 19, 3584
    g_ = gen(1)
    o1, o2 = randn_bf16(g_, T, hidden), randn_bf16(g_, T, hidden, scale=3.0)
    w1 = 1 + 0.1 * torch.randn(hidden, generator=g_)
    w2 = 1 + 0.1 * torch.randn(hidden, generator=g_)
    if tp == 1:
        ref = hybrid_combine_ref(o1, o2, w1, w2, 1e-6)
        out = hip_ops.hybrid_combine(o1.cuda(), o2.cuda(), w1.cuda(), w2.cuda(), 1e-6)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), ref, "hybrid combine tp=1", rel=3e-3)
        return
    # tensor parallel: each rank norms its hidden/tp columns with the all-reduced variance
    loc = hidden // tp
    var1 = o1.float().pow(2).mean(-1, keepdim=True)
    var2 = o2.float().pow(2).mean(-1, keepdim=True)
    stats_sum = torch.zeros(T, 2, device="cuda")
    shards = []
    for r in range(tp):
        sl = slice(r * loc, (r + 1) * loc)
        a, b = o1[:, sl].contiguous().cuda(), o2[:, sl].contiguous().cuda()
        shards.append((a, b, w1[sl].contiguous().cuda(), w2[sl].contiguous().cuda()))
        stats_sum += hip_ops.rms_sumsq2(a, b)          # what the all-reduce(sum) would produce
    stats = stats_sum / tp
    torch.cuda.synchronize()
    assert torch.allclose(stats.cpu(), torch.cat([var1, var2], dim=1), rtol=1e-5)
    for r, (a, b, wa, wb) in enumerate(shards):
        sl = slice(r * loc, (r + 1) * loc)
        ref = ((rms_norm_tp_ref(o1[:, sl], w1[sl], 1e-6, var1) + rms_norm_tp_ref(o2[:, sl], w2[sl], 1e-6, var2)) / 2
               ).to(torch.bfloat16)
        out = hip_ops.hybrid_combine(a, b, wa, wb, 1e-6, stats=stats)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), ref, f"hybrid combine shard {r}", rel=3e-3)


@pytest.mark.parametrize("T,hidden", [(1, 3584), (19, 3584), (64, 8192), (5, 136)])
@pytest.mark.parametrize("with_residual", [False, True])
def test_add_rmsnorm(T, hidden, with_residual):
    from oracle.layers import add_rms_norm_ref
    from vllm_hymeta import hip_ops

    g_ = gen(2)
    x = randn_bf16(g_, T, hidden, scale=2.0)
    res = randn_bf16(g_, T, hidden) if with_residual else None
    w = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(torch.bfloat16)
    ref, res_ref = add_rms_norm_ref(x, res, w, 1e-6)
    res_dev = res.cuda() if with_residual else None
    out = hip_ops.add_rmsnorm(x.cuda(), w.cuda(), 1e-6, residual=res_dev)
    torch.cuda.synchronize()
    if with_residual:
        assert torch.equal(res_dev.cpu(), res_ref), "new residual = bf16(x + residual), bit-exact"
    assert_bf16_close(out.cpu(), ref, "add_rmsnorm", rel=3e-3)
    # in place over x, from a strided view of a wider buffer
    wide = torch.zeros(T, hidden + 64, dtype=torch.bfloat16, device="cuda")
    wide[:, :hidden] = x.cuda()
    res_dev2 = res.cuda() if with_residual else None
    hip_ops.add_rmsnorm(wide[:, :hidden], w.cuda(), 1e-6, residual=res_dev2, out=wide[:, :hidden])
    torch.cuda.synchronize()
    assert torch.equal(wide[:, :hidden].cpu(), out.cpu())
    assert not wide[:, hidden:].any()


def test_add_rmsnorm_rejects_bad_layout():
    from vllm_hymeta import hip_ops

    x = torch.zeros(4, 130, dtype=torch.bfloat16, device="cuda")
    w = torch.ones(130, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(Exception):
        hip_ops.add_rmsnorm(x, w)


@pytest.mark.parametrize("T,hidden,N,with_bias", [(1, 3584, 512, True), (64, 3584, 512, True), (5, 512, 256, False), (37, 1024, 128, True)])
def test_lowrank_gate_equals_the_two_linears(T, hidden, N, with_bias):
    """The rank-16 gate projection in one launch against its two linears (fp32 sums of bf16 products,
    the intermediate and the result rounded to bf16, as ``model_for_7B/gla_attention.py:157-176`` computes them)."""
    from vllm_hymeta import hip_ops

    g_ = gen(71)
    x = randn_bf16(g_, T, hidden)
    w0 = randn_bf16(g_, 16, hidden, scale=hidden ** -0.5)
    w1 = randn_bf16(g_, N, 16, scale=0.25)
    b = randn_bf16(g_, N, scale=0.1) if with_bias else None
    t = (x.double() @ w0.double().t()).to(torch.bfloat16)
    ref = t.double() @ w1.double().t() + (b.double() if b is not None else 0.0)
    out = hip_ops.lowrank_gate(x.cuda(), w0.cuda(), w1.cuda(), None if b is None else b.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"lowrank gate T={T}")
    # rows of a wider buffer as input and output
    buf = randn_bf16(g_, T, hidden + 64).cuda()
    wide = torch.zeros(T, N + 32, dtype=torch.bfloat16, device="cuda")
    hip_ops.lowrank_gate(buf[:, :hidden], w0.cuda(), w1.cuda(), None if b is None else b.cuda(), out=wide[:, :N])
    again = hip_ops.lowrank_gate(buf[:, :hidden].contiguous(), w0.cuda(), w1.cuda(), None if b is None else b.cuda())
    torch.cuda.synchronize()
    assert torch.equal(wide[:, :N], again) and (wide[:, N:] == 0).all()


def test_advance_step_on_the_device_equals_the_host_restatement():
    """MetaAttentionMetadata.advance_step (multi-step decoding): the one-launch device update gives the integers of the
    tensor-op restatement that CPU tensors take -- tokens, positions, lengths and cache slots bit for bit, padding
    rows get slot -1 (reference meta_attention.py:338-409 -> vLLM's advance_step_flashattn)."""
    from types import SimpleNamespace

    from test_gpu_layers import _md

    g_ = gen(81)
    B, q, bs, nblk = 9, 7, 16, 6
    lens = [int(v) for v in torch.randint(1, bs * nblk - 2, (B,), generator=g_)]
    bt = torch.randperm(B * nblk, generator=g_).to(torch.int32).view(B, nblk)
    sampled = torch.randint(0, 1000, (q, 1), generator=g_)

    def make(dev):
        slots = torch.full((B,), 5, dtype=torch.int64, device=dev)
        md = _md(dev, decode_seq_lens=list(lens), block_tables=bt.to(dev), slot_mapping=slots)
        md.use_cuda_graph = True
        mi = SimpleNamespace(input_tokens=torch.zeros(B, dtype=torch.int64, device=dev),
                             input_positions=torch.zeros(B, dtype=torch.int64, device=dev))
        return md, mi

    md_c, mi_c = make(torch.device("cpu"))
    md_g, mi_g = make(torch.device("cuda"))
    for _ in range(3):                                    # three steps in a row
        md_c.advance_step(mi_c, sampled, bs, B, q)
        md_g.advance_step(mi_g, sampled.cuda(), bs, B, q)
    torch.cuda.synchronize()
    assert md_g.seq_lens == md_c.seq_lens and md_g.max_decode_seq_len == md_c.max_decode_seq_len
    assert torch.equal(md_g.seq_lens_tensor.cpu(), md_c.seq_lens_tensor)
    assert torch.equal(mi_g.input_tokens.cpu()[:q], mi_c.input_tokens[:q]) and torch.equal(mi_g.input_positions.cpu()[:q], mi_c.input_positions[:q])
    assert torch.equal(md_g.slot_mapping.cpu()[:q], md_c.slot_mapping[:q])
    assert (md_g.slot_mapping.cpu()[q:] == -1).all()
    pos = torch.tensor(lens[:q]) + 2                      # the position written by the third step
    assert torch.equal(md_c.slot_mapping[:q], torch.gather(bt[:q].long(), 1, (pos // bs).unsqueeze(1)).squeeze(1) * bs + pos % bs)
