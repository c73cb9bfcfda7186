"""The oracle against outputs of the reference's OWN code (tests/golden/ref_*.npz, made by
tests/golden/make_ref_fixtures.py in the build container from the reference files that import without the
GPU-only wheels): RMSNorm, RoPE, repeat_kv, the eager rolling-window cache, and the single-token GLA kernel
(the reference's Triton source run under Triton's CPU interpreter)."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in z.files:
        if k.endswith("_bf16"):
            out[k[:-5]] = torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16)
        elif z[k].dtype.kind in "US":
            out[k] = [str(x) for x in np.atleast_1d(z[k])]
        else:
            out[k] = torch.from_numpy(np.asarray(z[k]).copy()) if z[k].shape else z[k].item()
    return out


@pytest.mark.parametrize("tag", ["head", "hidden"])
def test_rms_norm_ref_equals_reference_rmsnorm(tag):
    """hf_7B_model/norms.py:13-18 -- bit for bit in all three dtype mixes."""
    from oracle.layers import rms_norm_ref, rms_norm_tp_ref

    f = load("ref_rmsnorm.npz")
    x, w, eps = f[f"{tag}_x"], f[f"{tag}_w"], float(f["eps"])
    assert torch.equal(rms_norm_ref(x, w.to(torch.bfloat16), eps), f[f"{tag}_y"])             # bf16 module
    assert torch.equal(rms_norm_tp_ref(x, w, eps), f[f"{tag}_y_w32"])                            # fp32 weight on bf16 x
    assert torch.equal(rms_norm_ref(x.float(), w, eps), f[f"{tag}_y_f32"])                       # fp32 activations


def test_rope_neox_ref_equals_reference_rotary():
    """hf_7B_model/window_attention.py:23-79: fp32 activations against an fp32 table, bf16 activations against
    the table rounded to bf16 (what the eager model does with bf16 weights, and what vLLM's table holds)."""
    from oracle.layers import rope_cos_sin_cache_ref, rope_neox_ref

    f = load("ref_rope.npz")
    pos = f["positions"].long()
    D = f["f32_q"].shape[-1]
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        table = rope_cos_sin_cache_ref(D, 131072, float(f["theta"]), dtype=dt)
        # the reference's cos/sin rows ([cos, cos] / [sin, sin] over the two halves) are this table's rows
        assert torch.equal(table[pos][:, : D // 2], f[f"{tag}_cos"][0][:, : D // 2])
        assert torch.equal(table[pos][:, D // 2:], f[f"{tag}_sin"][0][:, : D // 2])
        q = f[f"{tag}_q"][0].transpose(0, 1)           # [T, H, D]
        k = f[f"{tag}_k"][0].transpose(0, 1)
        qo, ko = rope_neox_ref(pos, q, k, table)
        assert torch.equal(qo, f[f"{tag}_q_out"][0].transpose(0, 1)), tag
        assert torch.equal(ko, f[f"{tag}_k_out"][0].transpose(0, 1)), tag


def test_gqa_head_mapping_equals_reference_repeat_kv():
    """repeat_kv (hf_7B_model/window_attention.py:82-91): q-head h reads kv-head h // n_rep -- the mapping the
    kernels apply instead of materialising the copies."""
    f = load("ref_rope.npz")
    x, y = f["repeat_kv_in"], f["repeat_kv_out"]
    n_rep = y.shape[1] // x.shape[1]
    for h in range(y.shape[1]):
        assert torch.equal(y[:, h], x[:, h // n_rep])


@pytest.mark.parametrize("tag", ["7b", "70b"])
def test_rolling_cache_restatement_equals_reference_cache(tag):
    from oracle.cache import RollingCacheRef, window_positions_ref

    f = load("ref_cache.npz")
    W, chunks = int(f["window"]), [int(c) for c in f["chunks"]]
    keys, vals = f["keys"], f["values"]
    c, t = RollingCacheRef(), 0
    held = window_positions_ref(chunks, W)
    for i, n in enumerate(chunks):
        meta = (torch.full((1, 2, 3, 8), 7.0), torch.full((1, 2, 3, 8), -7.0)) if (tag == "70b" and i == 0) else None
        st = c.update(0, n, attn_state=(keys[:, :, t:t + n].clone(), vals[:, :, t:t + n].clone()), meta_state=meta, window_size=W)
        t += n
        assert torch.equal(st["attn_state"][0], f[f"{tag}_k_after_{i}"]) and torch.equal(st["attn_state"][1], f[f"{tag}_v_after_{i}"])
        assert c.seen_tokens == int(f[f"{tag}_seen_after_{i}"]) == t
        # the window holds these positions, in order: the paged cache reads the same rows through its block table
        assert torch.equal(st["attn_state"][0], keys[:, :, held[i]])
    if tag == "70b":
        assert torch.equal(c.states[0]["meta_state"][0], f["70b_meta_k_kept"])
    c2 = RollingCacheRef()
    st = c2.update(0, 10, attn_state=(keys[:, :, :10].clone(), vals[:, :, :10].clone()), window_size=W)
    assert torch.equal(st["attn_state"][0], f[f"{tag}_k_long_first"])
    c3 = RollingCacheRef()
    c3.update(0, 5, recurrent_state=torch.ones(1, 2, 3))
    st = c3.update(0, 1, recurrent_state=torch.full((1, 2, 3), 2.0))
    assert torch.equal(st["recurrent_state"], f[f"{tag}_recurrent_after_2"]) and c3.seen_tokens == int(f[f"{tag}_recurrent_seen"])


def test_rolling_window_is_the_paged_window():
    """Once full, the eager window holds the last W positions [L-W, L) in order; the paged decode of the 70B
    path reads exactly those through the block table (window_left = W - 1 keys before the query + the query)."""
    from oracle.cache import window_positions_ref

    W = 6
    held = window_positions_ref([W] + [1] * 9, W)
    for step, h in enumerate(held):
        L = W + step
        assert h == list(range(L - W, L))


@pytest.mark.parametrize("tag,dtype", [("f16", torch.float16), ("f32", torch.float32)])
def test_gla_decode_step_ref_equals_reference_triton_kernel(tag, dtype):
    """vllm_hymeta/models/my_fused_recurrent.py:8-119 and model_for_7B/my_fused_recurrent.py:9-119 (the two files gave
    identical bits on these inputs): two successive steps, one padded sequence, outputs and state bit for bit."""
    from oracle import gla_decode_step_ref

    f = load("ref_gla_decode.npz")
    assert bool(f["variants_agree"])
    slot = f["slot_idx"]
    state = f[f"{tag}_state_in"].clone()
    assert state.dtype == dtype
    for s in range(int(f["steps"])):
        o = gla_decode_step_ref(f[f"{tag}_q{s}"], f[f"{tag}_k{s}"], f[f"{tag}_v{s}"], f[f"{tag}_g{s}"], state, slot,
                                scale=float(f["scale"]))
        assert o.dtype == dtype and torch.equal(o, f[f"{tag}_out{s}"]), (tag, s)
    assert torch.equal(state, f[f"{tag}_state_out"])


def test_gla_decode_scale_is_rounded_to_the_activation_dtype():
    """my_fused_recurrent.py:59: `load(q) * scale` is tensor x scalar, computed in the tensor's dtype.  With bf16
    activations the scale 128 ** -0.5 therefore enters as 0.08837890625 and the product is rounded to bf16."""
    from oracle import gla_decode_step_ref

    D = 128
    q = torch.zeros(1, 1, D, dtype=torch.bfloat16)
    q[0, 0, 0] = 1.0
    q[0, 0, 1] = 1.0078125                     # 1 + 2^-7: the product with the rounded scale needs a rounding
    k = torch.zeros(1, 1, D, dtype=torch.bfloat16)
    v = torch.zeros(1, 1, D, dtype=torch.bfloat16)
    g = torch.zeros(1, 1, D)
    state = torch.zeros(1, 1, D, D, dtype=torch.float32)
    state[0, 0, 0, 0] = 1.0
    state[0, 0, 1, 1] = 1.0
    o = gla_decode_step_ref(q, k, v, g, state, torch.tensor([0], dtype=torch.int32))
    sc = torch.tensor(D ** -0.5).to(torch.bfloat16)
    assert float(sc) == 0.08837890625
    assert float(o[0, 0, 0]) == float(sc)
    assert float(o[0, 0, 1]) == float((q[0, 0, 1] * sc))


def test_meta_attention_restatement_equals_the_reference_dense_definition():
    """``naive_metatoken_flash_attn`` (vllm_hymeta/attention/fa_kernel_metatoken.py:377-410), the reference's own
    executable definition of window + meta-token attention, run on fp32 inputs: ``meta_swa_dense_ref`` gives the same
    rows (its meta-query rows come first, the vLLM-path order; the naive function appends them) for equal lengths with
    meta queries, a chunk against a longer key range (bottom-right alignment) and windows shorter / longer than the
    sequence.  ``sliding_window[0] = w`` there shows keys with q - k < w: ``window_left = w - 1`` here."""
    from oracle.meta_swa import meta_swa_dense_ref, meta_swa_two_pass_ref

    f = load("ref_meta_attention.npz")
    for tag in f["cases"] + f["kernel_cases"]:
        M = int(f["kernel_num_meta" if tag.startswith("k_") else "num_meta"])
        q1, k1, v1, k2, v2 = (f[f"{tag}_{n}"] for n in ("q1", "k1", "v1", "k2", "v2"))
        q2 = f.get(f"{tag}_q2")
        want = f[f"{tag}_out"]
        Lq = q1.shape[0]
        scale = float(f["kernel_scale" if tag.startswith("k_") else "scale"])
        got, _ = meta_swa_dense_ref(q1, k1, v1, k2, v2, q2, window_left=int(f[f"{tag}_window"]) - 1, scale=scale)
        if q2 is not None:
            assert got.shape[0] == M + Lq
            got = torch.cat([got[M:], got[:M]], dim=0)          # the naive function's order: real rows, then meta rows
        assert got.shape == want.shape, tag
        assert torch.allclose(got.float(), want, rtol=2e-5, atol=2e-6), (tag, float((got.float() - want).abs().max()))
        # the two-pass form the kernels' structure follows (window pass + meta pass + merge) is the same function
        if tag.startswith("k_"):
            continue          # bf16 inputs: the two-pass form rounds its partial outputs to bf16, as flash-attn does
        two = meta_swa_two_pass_ref(q1, k1, v1, k2, v2, q2, window_left=int(f[f"{tag}_window"]) - 1, scale=scale)
        if q2 is not None:
            two = torch.cat([two[M:], two[:M]], dim=0)
        assert torch.allclose(two, want, rtol=2e-5, atol=2e-6), tag


def test_lse_merge_ref_equals_reference_update_out_and_lse():
    """``_update_out_and_lse`` (fa_kernel_metatoken.py:110-122) on bf16 partial outputs: bit for bit, and equal to the
    plain log-sum-exp form the kernels' online softmax computes."""
    from oracle.meta_swa import lse_merge_ref

    f = load("ref_meta_attention.npz")
    out, lse = lse_merge_ref(f["merge_out1"], f["merge_lse1"], f["merge_out2"], f["merge_lse2"])
    assert out.dtype == torch.bfloat16 and torch.equal(out, f["merge_out"]) and torch.equal(lse, f["merge_lse"])
    l1, l2 = f["merge_lse1"].double(), f["merge_lse2"].double()
    both = torch.logaddexp(l1, l2)
    plain = (f["merge_out1"].double() * torch.exp(l1 - both).t()[..., None] + f["merge_out2"].double() * torch.exp(l2 - both).t()[..., None])
    assert torch.allclose(lse.double(), both, rtol=1e-6, atol=1e-6)
    assert torch.allclose(out.double(), plain, rtol=0, atol=2.0 ** -7 * float(plain.abs().max()))
