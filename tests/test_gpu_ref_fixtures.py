"""HIP kernels against outputs of the reference's own eager code (tests/golden/ref_*.npz): the block RMS norm,
the fused per-head norm of the GLA kernels, the 70B branch norms and NeoX RoPE.  Tolerance: these ops repeat the
reference's bf16 roundings, so results are equal except where the device's rsqrt / fused multiply-add differs in the
last fp32 bit before a bf16 rounding: at most 1 bf16 ulp, on a small fraction of the elements."""
import pytest
import torch

from test_ref_fixtures import load

pytestmark = pytest.mark.gpu


def _ulp_close(out, ref, what, max_frac=0.02):
    out, ref = out.float().cpu(), ref.float().cpu()
    tol = 2.0 ** -7 * torch.maximum(out.abs(), ref.abs()) + 1e-30
    assert ((out - ref).abs() <= tol).all(), f"{what}: more than 1 bf16 ulp apart"
    frac = float((out != ref).float().mean())
    assert frac <= max_frac, f"{what}: {frac:.3%} of the elements differ from the reference's bits"


@pytest.mark.parametrize("tag", ["head", "hidden"])
def test_add_rmsnorm_vs_reference_rmsnorm(tag):
    from vllm_hymeta import hip_ops

    f = load("ref_rmsnorm.npz")
    x, w = f[f"{tag}_x"].cuda(), f[f"{tag}_w"].to(torch.bfloat16).cuda()
    y = hip_ops.add_rmsnorm(x, w, float(f["eps"]))
    torch.cuda.synchronize()
    _ulp_close(y, f[f"{tag}_y"], f"add_rmsnorm {tag}")


@pytest.mark.parametrize("state_heads", ["query", "kv"])
def test_gla_decode_g_norm_epilogue_vs_reference_rmsnorm(state_heads):
    """The fused per-head norm (g_norm) of the GLA decode kernel.  The step is set up so that the un-normed output
    equals the fixture's input exactly: zero state, no decay, scale 1, q = k = e_0, v = x  =>  o = v."""
    from vllm_hymeta import hip_ops

    f = load("ref_rmsnorm.npz")
    x, w = f["head_x"], f["head_w"].to(torch.bfloat16)
    B, H, Hkv, D = 3, 4, 4, 128
    v = x.view(B, Hkv, D).cuda()
    e0 = torch.zeros(D, dtype=torch.bfloat16)
    e0[0] = 1
    q = e0.expand(B, H, D).contiguous().cuda()
    k = e0.expand(B, Hkv, D).contiguous().cuda()
    g = torch.zeros(B, Hkv, D, dtype=torch.float32, device="cuda")
    state = torch.zeros(B, H if state_heads == "query" else Hkv, D, D, dtype=torch.bfloat16, device="cuda")
    slots = torch.arange(B, dtype=torch.int32, device="cuda")
    raw = hip_ops.gla_decode(q, k, v, g, state.clone(), slots, scale=1.0)
    out = hip_ops.gla_decode(q, k, v, g, state, slots, scale=1.0, out_norm_weight=w.cuda(), out_norm_eps=float(f["eps"]))
    torch.cuda.synchronize()
    assert torch.equal(raw.cpu().view(B * H, D), x), "set-up: un-normed output must equal the fixture input"
    _ulp_close(out.view(B * H, D), f["head_y"], "gla decode g_norm epilogue")


def test_gla_prefill_g_norm_epilogue_vs_reference_rmsnorm():
    """Same for the prefill kernel: one-token sequences from a zero state."""
    from vllm_hymeta import hip_ops

    f = load("ref_rmsnorm.npz")
    x, w = f["head_x"], f["head_w"].to(torch.bfloat16)
    T, H, D = 3, 4, 128
    v = x.view(T, H, D).cuda()
    e0 = torch.zeros(D, dtype=torch.bfloat16)
    e0[0] = 1
    q = e0.expand(T, H, D).contiguous().cuda()
    g = torch.zeros(T, H, D, dtype=torch.float32, device="cuda")
    state = torch.zeros(T, H, D, D, dtype=torch.bfloat16, device="cuda")
    slots = torch.arange(T, dtype=torch.int32, device="cuda")
    cu = torch.arange(T + 1, dtype=torch.int32, device="cuda")
    ctx = torch.zeros(T, dtype=torch.int32, device="cuda")
    out = hip_ops.gla_prefill_varlen(q, q.clone(), v, g, state, slots, cu, 1, context_lens=ctx, scale=1.0,
                                     out_norm_weight=w.cuda(), out_norm_eps=float(f["eps"]))
    torch.cuda.synchronize()
    _ulp_close(out.view(T * H, D), f["head_y"], "gla prefill g_norm epilogue")


def test_hybrid_combine_vs_reference_rmsnorm():
    """Both branches = x and both weights = w: (norm(x) w + norm(x) w) / 2 = norm(x) w, the reference RMSNorm with
    its fp32 weight (the dtype mix of HymetaRMSNormTP), rounded to bf16."""
    from vllm_hymeta import hip_ops

    f = load("ref_rmsnorm.npz")
    x, w = f["hidden_x"].cuda(), f["hidden_w"].cuda()
    y = hip_ops.hybrid_combine(x, x.clone(), w, w.clone(), float(f["eps"]))
    torch.cuda.synchronize()
    _ulp_close(y, f["hidden_y_w32"].to(torch.bfloat16), "hybrid combine")


def test_rope_kernel_vs_reference_rotary():
    from oracle.layers import rope_cos_sin_cache_ref
    from vllm_hymeta import hip_ops

    f = load("ref_rope.npz")
    pos = f["positions"].long()
    q = f["bf16_q"][0].transpose(0, 1).contiguous().cuda()
    k = f["bf16_k"][0].transpose(0, 1).contiguous().cuda()
    table = rope_cos_sin_cache_ref(q.shape[-1], 131072, float(f["theta"])).cuda()   # rows == the reference's cos/sin (CPU test)
    hip_ops.rope_neox_(pos.cuda(), q, k, table)
    torch.cuda.synchronize()
    # x*cos - y*sin: the reference rounds both products to bf16 before the sum; a contracted multiply-add on the
    # device keeps one product exact, so single elements move by an ulp
    _ulp_close(q, f["bf16_q_out"][0].transpose(0, 1), "rope q", max_frac=0.25)
    _ulp_close(k, f["bf16_k_out"][0].transpose(0, 1), "rope k", max_frac=0.25)


def test_attention_kernels_vs_the_reference_dense_definition():
    """The prefill and paged-decode kernels against outputs of ``naive_metatoken_flash_attn``
    (vllm_hymeta/attention/fa_kernel_metatoken.py:377-410) itself, at the kernels' geometry (head size 128, 32 meta
    tokens, bf16 inputs, window given as ``sliding_window[0] = w`` there = ``window_left = w - 1`` here):
    a whole sequence with the meta queries (their rows come first here, last there); a 20-token chunk against 90 keys
    (bottom-right aligned), un-paged and through a block table; and the chunk's last row as a paged single-token decode."""
    from _util import assert_bf16_close
    from vllm_hymeta import hip_ops

    f = load("ref_meta_attention.npz")
    M, scale = int(f["kernel_num_meta"]), float(f["kernel_scale"])
    dev = torch.device("cuda")
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)

    t = "k_eq_q2"
    q1, k1, v1, k2, v2, q2 = (f[f"{t}_{n}"].to(dev) for n in ("q1", "k1", "v1", "k2", "v2", "q2"))
    L, wl = q1.shape[0], int(f[f"{t}_window"]) - 1
    out = hip_ops.meta_swa_prefill(q1, i32([0, L]), L, key=k1, value=v1, cu_seqlens_k=i32([0, L]), meta_query=q2, meta_key=k2,
                                   meta_value=v2, window_left=wl, scale=scale)
    torch.cuda.synchronize()
    want = f[f"{t}_out"]
    assert_bf16_close(out[M:].cpu(), want[:L], "prefill, real rows vs the reference's dense definition")
    assert_bf16_close(out[:M].cpu(), want[L:], "prefill, meta-query rows vs the reference's dense definition")

    t = "k_chunk"
    q1, k1, v1, k2, v2 = (f[f"{t}_{n}"].to(dev) for n in ("q1", "k1", "v1", "k2", "v2"))
    Lq, Lk, wl = q1.shape[0], k1.shape[0], int(f[f"{t}_window"]) - 1
    want = f[f"{t}_out"]
    out = hip_ops.meta_swa_prefill(q1, i32([0, Lq]), Lq, key=k1, value=v1, cu_seqlens_k=i32([0, Lk]), meta_key=k2, meta_value=v2,
                                   window_left=wl, scale=scale)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), want, "chunk against a longer key range, un-paged")
    # the same keys in a paged cache with a scattered block table
    bs, Hkv = 16, k1.shape[1]
    nb = (Lk + bs - 1) // bs
    perm = torch.randperm(nb + 3, generator=torch.Generator().manual_seed(7))[:nb].to(torch.int32)
    kc = torch.full((nb + 3, bs, Hkv, 128), float("nan"), dtype=torch.bfloat16, device=dev)
    vc = torch.full_like(kc, float("nan"))
    for p in range(Lk):
        kc[int(perm[p // bs]), p % bs], vc[int(perm[p // bs]), p % bs] = k1[p], v1[p]
    bt = perm.view(1, nb).to(dev)
    out = hip_ops.meta_swa_prefill(q1, i32([0, Lq]), Lq, key_cache=kc, value_cache=vc, block_tables=bt, seq_lens_k=i32([Lk]),
                                   meta_key=k2, meta_value=v2, window_left=wl, scale=scale)
    dec = hip_ops.meta_swa_paged_decode(q1[-1:].contiguous(), kc, vc, i32([Lk]), bt, k2, v2, window_left=wl, scale=scale)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), want, "chunk through a block table")
    assert_bf16_close(dec.cpu(), want[-1:], "last row as a paged single-token decode")
