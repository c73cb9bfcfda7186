"""Checkpoint routing and tensor-parallel cuts of the linears (CPU).

Routing expectations are read off the reference's ``load_weights``
(``vllm_hymeta/models/modeling_hymeta.py:1487-1691``,
``vllm_hymeta/model_for_7B/modeling_gla_swa.py:601-724``).  The cuts are checked by what they must
satisfy whatever the storage format: column-parallel shards concatenate to, and row-parallel shards
sum to, the unsharded linear -- bit-exactly on the de-quantised weights."""
import pytest
import torch

from oracle.quant import gptq_dequant_ref, gptq_pack_ref
from vllm_hymeta import loader
from vllm_hymeta.loader import Route


def test_route_70b_names():
    L = 4
    r = lambda n: loader.route_70b(n, L)
    assert r("model.layers.1.attn.vanilla_attn.q_proj.weight") == Route("model.layers.1.attn.vanilla_attn.qkv_proj.weight", "q")
    assert r("model.layers.1.attn.linear_attn.k_proj.qweight") == Route("model.layers.1.attn.linear_attn.qkv_proj.qweight", "k")
    assert r("model.layers.1.attn.linear_attn.v_proj.bias") == Route("model.layers.1.attn.linear_attn.qkv_proj.bias", "v")
    assert r("model.layers.2.attn.out_proj.weight") == Route("model.layers.2.attn.out_proj.weight")
    assert r("model.layers.0.attn.vanilla_attn.rotary_emb.inv_freq") == Route("model.layers.0.attn.vanilla_attn.rotary_emb.inv_freq")
    assert r("model.layers.0.rotary_emb.inv_freq") is None
    # norms first, and the checkpoint's "mode.norm" typo
    assert r("mode.norm.weight") == Route("model.norm.weight")
    assert r("model.layers.3.attn.norm1.weight") == Route("model.layers.3.attn.norm1.weight")
    # experts: vLLM's fused parameters and shard ids; the router gate passes through
    base = "model.layers.0.block_sparse_moe."
    assert r(base + "experts.5.gate_proj.weight") == Route(base + "experts.w13_weight", "w1", 5)
    assert r(base + "experts.15.up_proj.weight") == Route(base + "experts.w13_weight", "w3", 15)
    assert r(base + "experts.1.down_proj.weight") == Route(base + "experts.w2_weight", "w2", 1)
    assert r(base + "gate.weight") == Route(base + "gate.weight")
    # shared expert and dense MLP: fused gate_up with shard ids 0 / 1
    assert r("model.layers.0.shared_mlp.gate_proj.weight") == Route("model.layers.0.shared_mlp.gate_up_proj.weight", 0)
    assert r("model.layers.0.mlp.up_proj.weight") == Route("model.layers.0.mlp.gate_up_proj.weight", 1)
    assert r("model.layers.0.mlp.down_proj.weight") == Route("model.layers.0.mlp.down_proj.weight")
    # layers the configuration does not have are dropped
    assert r("model.layers.4.mlp.down_proj.weight") is None
    assert r("model.embed_tokens.weight") == Route("model.embed_tokens.weight")
    # a parameter list narrows what counts as a norm weight (the reference consults params_dict)
    assert loader.route_70b("model.layers.0.attn.linear_attn.q_norm_proj.weight", L, known_params=[]) == \
        Route("model.layers.0.attn.linear_attn.q_norm_proj.weight")


def test_route_7b_names():
    L = 2
    r = lambda n: loader.route_7b(n, L)
    assert r("model.layers.0.attn.q_proj.weight") == Route("model.layers.0.attn.qkv_proj.weight", "q")
    assert r("model.layers.0.attn.gk_proj.0.weight") == Route("model.layers.0.attn.gk_proj.linear0.weight")
    assert r("model.layers.0.attn.gk_proj.1.bias") == Route("model.layers.0.attn.gk_proj.linear1.bias")
    assert r("model.layers.0.attn.g_norm.weight") == Route("model.layers.0.attn.g_norm.weight")
    assert r("model.layers.1.attn.o_proj.weight") == Route("model.layers.1.attn.o_proj.weight")
    assert r("model.layers.1.mlp.gate_proj.qweight") == Route("model.layers.1.mlp.gate_up_proj.qweight", 0)
    assert r("model.layers.2.attn.q_proj.weight") is None
    assert r("lm_head.weight") == Route("lm_head.weight")
    with pytest.raises(AssertionError):
        r("model.layers.0.mlp.other_proj.weight")


def _gptq(g, K, N, bits=4, group=32):
    w_int = torch.randint(0, 2 ** bits, (K, N), generator=g)
    zeros = torch.randint(1, 2 ** bits, (K // group, N), generator=g)
    qw, qz = gptq_pack_ref(w_int, zeros, bits)
    scales = (torch.rand(K // group, N, generator=g) * 0.02 + 0.005).to(torch.float16)
    return {"qweight": qw, "qzeros": qz, "scales": scales, "g_idx": torch.arange(K, dtype=torch.int32) // group,
            "bias": torch.randn(N, generator=g).to(torch.bfloat16)}


def _bf16(g, K, N):
    return {"weight": torch.randn(N, K, generator=g).to(torch.bfloat16), "bias": torch.randn(N, generator=g).to(torch.bfloat16)}


def _w8a8(g, K, N):
    return {"weight": torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8),
            "weight_scale": torch.rand(N, 1, generator=g) * 0.01 + 0.001}


def _dense(t, bits=4, group=32):
    """[in, out] fp64 weight of a linear in any of the three formats."""
    kind = loader.linear_kind(t)
    if kind == "gptq":
        return gptq_dequant_ref(t["qweight"], t["qzeros"], t["scales"], bits, group)
    w = t["weight"].double().t()
    return w * t["weight_scale"].reshape(-1).double() if kind == "w8a8" else w


@pytest.mark.parametrize("make", [_bf16, _gptq, _w8a8])
@pytest.mark.parametrize("tp", [1, 2, 4])
def test_qkv_shards_hold_their_heads(make, tp):
    g = torch.Generator().manual_seed(0)
    hidden, H, KVH, D = 128, 4, 2, 32
    q, k, v = make(g, hidden, H * D), make(g, hidden, KVH * D), make(g, hidden, KVH * D)
    wq, wk, wv = _dense(q), _dense(k), _dense(v)
    for rank in range(tp):
        fused, sh = loader.shard_qkv(q, k, v, H, KVH, D, tp, rank, bits=4)
        w = _dense(fused)
        hq, hk = sh.num_heads * D, sh.num_kv_heads * D
        assert w.shape == (hidden, hq + 2 * hk)
        assert torch.equal(w[:, :hq], wq[:, sh.head_start * D:(sh.head_start + sh.num_heads) * D])
        assert torch.equal(w[:, hq:hq + hk], wk[:, sh.kv_head_start * D:(sh.kv_head_start + sh.num_kv_heads) * D])
        assert torch.equal(w[:, hq + hk:], wv[:, sh.kv_head_start * D:(sh.kv_head_start + sh.num_kv_heads) * D])
        # q-head h of this rank reads kv-head h // (H / KVH): the replicated kv-head is the right one
        for hl in range(sh.num_heads):
            assert (sh.head_start + hl) // (H // KVH) - sh.kv_head_start in range(sh.num_kv_heads)
        if "bias" in q:
            assert torch.equal(fused["bias"][:hq], q["bias"][sh.head_start * D:(sh.head_start + sh.num_heads) * D])


@pytest.mark.parametrize("make", [_bf16, _gptq, _w8a8])
@pytest.mark.parametrize("tp", [1, 2, 4])
def test_mlp_shards_reassemble(make, tp):
    g = torch.Generator().manual_seed(1)
    hidden, inter = 128, 256
    gate, up, down = make(g, hidden, inter), make(g, hidden, inter), make(g, inter, hidden)
    x = torch.randn(5, hidden, generator=g).double()
    ref_gu = torch.cat([x @ _dense(gate), x @ _dense(up)], dim=1)
    h = torch.randn(5, inter, generator=g).double()
    ref_down = h @ _dense(down)
    per = inter // tp
    acc = torch.zeros_like(ref_down)
    bias_sum = None
    for rank in range(tp):
        gu = loader.shard_gate_up(gate, up, tp, rank)
        y = x @ _dense(gu)
        sl = slice(rank * per, (rank + 1) * per)
        assert torch.equal(y[:, :per], ref_gu[:, sl]) and torch.equal(y[:, per:], ref_gu[:, inter + rank * per: inter + (rank + 1) * per])
        dn = loader.shard_row_parallel(down, tp, rank, bits=4, group_size=32)
        acc += h[:, sl] @ _dense(dn)
        if "bias" in dn:
            assert bias_sum is None, "exactly one rank carries the row-parallel bias"
            bias_sum = dn["bias"]
    assert torch.allclose(acc, ref_down, rtol=1e-12, atol=1e-9)
    assert ("bias" in down) == (bias_sum is not None)


def test_gptq_cuts_must_respect_packing_and_groups():
    g = torch.Generator().manual_seed(2)
    t = _gptq(g, 128, 64)
    with pytest.raises(ValueError):
        loader.slice_out_features(t, 4, 36, bits=4)          # zeros are packed 8 columns per word
    with pytest.raises(ValueError):
        loader.slice_in_features(t, 16, 80, bits=4, group_size=32)
    t["g_idx"] = torch.randperm(128, generator=g).to(torch.int32) // 32
    with pytest.raises(ValueError):
        loader.slice_in_features(t, 0, 64, bits=4, group_size=32)   # act-order is refused
    with pytest.raises(ValueError):
        loader.make_linear(t, bits=4, group_size=32)


def test_stack_experts_layout():
    g = torch.Generator().manual_seed(3)
    E, H, I = 4, 32, 64
    experts = [{"gate_proj": torch.randn(I, H, generator=g), "up_proj": torch.randn(I, H, generator=g),
                "down_proj": torch.randn(H, I, generator=g)} for _ in range(E)]
    for tp in (1, 2):
        per = I // tp
        for rank in range(tp):
            w13, w2 = loader.stack_experts(experts, tp, rank)
            assert w13.shape == (E, 2 * per, H) and w2.shape == (E, H, per)
            for e in range(E):
                assert torch.equal(w13[e, :per], experts[e]["gate_proj"][rank * per:(rank + 1) * per])
                assert torch.equal(w13[e, per:], experts[e]["up_proj"][rank * per:(rank + 1) * per])
                assert torch.equal(w2[e], experts[e]["down_proj"][:, rank * per:(rank + 1) * per])


def test_make_linear_casts_checkpoint_floats_to_bf16():
    """gptqmodel / llmcompressor checkpoints keep biases, scales and the layers they leave un-quantised in the
    checkpoint's float type (often fp16): the linear must end up with bf16 bias and weight (the kernels' types),
    fp16 scales are taken as they are."""
    g = torch.Generator().manual_seed(5)
    t = _gptq(g, 64, 32)
    t["scales"] = t["scales"].to(torch.float16)
    t["bias"] = torch.randn(32, generator=g).to(torch.float16)
    lin = loader.make_linear(t, 4, 32)
    assert lin.kind == "gptq" and lin.bias.dtype == torch.bfloat16 and lin.scales.dtype == torch.float16
    assert torch.equal(lin.bias, t["bias"].to(torch.bfloat16))
    dense = {"weight": torch.randn(32, 64, generator=g).to(torch.float16), "bias": torch.randn(32, generator=g).to(torch.float16)}
    lin = loader.make_linear(dense)
    assert lin.kind == "bf16" and lin.weight.dtype == torch.bfloat16 and lin.bias.dtype == torch.bfloat16
    w8 = {"weight": torch.randint(-128, 128, (32, 64), generator=g).to(torch.int8),
          "weight_scale": torch.rand(32, 1, generator=g).to(torch.float16), "bias": torch.randn(32, generator=g).to(torch.float16)}
    lin = loader.make_linear(w8)
    assert lin.kind == "w8a8" and lin.qweight.dtype == torch.int8 and lin.w_scale.dtype == torch.float32 and lin.bias.dtype == torch.bfloat16


@pytest.mark.parametrize("make", [_bf16, _gptq, _w8a8])
def test_cutting_tensor_by_tensor_equals_cutting_the_whole_linear(make):
    """The streaming loader cuts every checkpoint tensor on arrival; the pieces equal what the whole-linear cuts give."""
    g = torch.Generator().manual_seed(5)
    K, N, tp, group = 128, 64, 2, 32
    t = make(g, K, N)
    t["bias"] = torch.randn(N, generator=g).to(torch.bfloat16)
    for rank in range(tp):
        whole = loader.shard_row_parallel(t, tp, rank, 4, group)
        got = {leaf: loader.cut_leaf(leaf, x, loader.Cut("in", parts=tp, index=rank), 4, group) for leaf, x in t.items()}
        got = {k: v for k, v in got.items() if v is not None}
        assert set(got) == set(whole) - {"g_idx"} and ("bias" in got) == (rank == 0)
        assert all(torch.equal(got[k], whole[k]) for k in got)
        lo, hi = rank * N // tp, (rank + 1) * N // tp
        whole = loader.slice_out_features(t, lo, hi, 4)
        for cut in (loader.Cut("out", lo, hi), loader.Cut("out", parts=tp, index=rank)):
            got = {leaf: loader.cut_leaf(leaf, x, cut, 4, group) for leaf, x in t.items()}
            assert all(torch.equal(got[k], whole[k]) for k in whole if k != "g_idx")
    # no cut: the tensor itself (a per-channel scale flattened)
    assert loader.cut_leaf("weight", t.get("weight", t.get("qweight")), loader.Cut()) is t.get("weight", t.get("qweight"))


def test_a_kept_piece_does_not_hold_the_whole_tensor():
    full = torch.randn(64, 32)
    piece = loader.own(loader.cut_leaf("weight", full, loader.Cut("out", parts=4, index=1)))
    assert piece.untyped_storage().nbytes() == piece.numel() * 4 and torch.equal(piece, full[16:32])
    piece = loader.own(loader.cut_leaf("weight", full, loader.Cut("in", parts=4, index=3)))
    assert piece.is_contiguous() and piece.untyped_storage().nbytes() == piece.numel() * 4 and torch.equal(piece, full[:, 24:])
    whole = loader.own(full)
    assert whole is full
    with pytest.raises(ValueError):
        loader.cut_leaf("g_idx", torch.arange(64, dtype=torch.int32).flip(0) // 32, loader.Cut(), 4, 32)     # act-order
