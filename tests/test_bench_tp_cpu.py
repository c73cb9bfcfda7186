"""Host side of the tensor-parallel bench mode on CPU, world_size 2 over gloo: the synthetic 70B checkpoint goes through
the model's own load_weights on every rank, the ranks' shards tile the full tensors, the group layout and the byte
model of the step are what the JSON line says.  (The step itself needs the GPU kernels.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_tp_layout():
    from bench_tp import tp_layout
    assert [tp_layout(n) for n in (1, 2, 4, 8, 6)] == [(1, 1), (2, 1), (4, 1), (4, 2), (2, 3)]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rank_bookkeeping_of_the_tensor_parallel_bench(world):
    """What every rank of a job of N processes derives for itself agrees over the job: the groups partition the ranks
    in order, a rank is a member of the group it picks at the place it claims, the ranks of a group share their batch
    (replicas do not), their shares of the published vocabulary are whole multiples of 64 rows that tile it, and the
    byte model of the step is the same on every rank of a group, with and without expert parallelism."""
    import bench_tp
    from bench import VOCAB

    plans = [bench_tp.tp_plan(r, world) for r in range(world)]
    tp, replicas = bench_tp.tp_layout(world)
    assert tp * replicas == world and tp in (1, 2, 4)       # 28 query heads / 4 kv-heads: never 8
    assert all(p["group_ranks"] == plans[0]["group_ranks"] for p in plans)      # every rank creates the same groups in the same order
    assert sorted(sum(plans[0]["group_ranks"], [])) == list(range(world))
    for r, p in enumerate(plans):
        mine = p["group_ranks"][p["replica"]]
        assert mine[p["rank_in_group"]] == r and len(mine) == tp
        assert all(plans[q]["batch_seed"] == p["batch_seed"] for q in mine)
    assert len({p["batch_seed"] for p in plans}) == replicas
    assert VOCAB % (64 * tp) == 0
    rows = [(p["rank_in_group"] * (VOCAB // tp), (p["rank_in_group"] + 1) * (VOCAB // tp)) for p in plans if p["replica"] == 0]
    assert rows[0][0] == 0 and rows[-1][1] == VOCAB and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    cfg = bench_tp.config_70b()
    n_mix = sum(1 for n in cfg.num_layer_experts if n > 1)
    total, parts = bench_tp.step_bytes_70b(cfg, "bf16", tp, 64, 32768, [16] * n_mix)
    total_ep, parts_ep = bench_tp.step_bytes_70b(cfg, "bf16", tp, 64, 32768, [16 // tp] * n_mix, expert_parallel=tp > 1)
    assert parts_ep["routed_experts"] == parts["routed_experts"]      # every expert hit: a 1 / tp slice of each = 16 / tp whole ones
    assert total_ep == total


def test_step_bytes_of_the_published_geometry():
    from bench_tp import config_70b, step_bytes_70b
    cfg = config_70b()
    total, parts = step_bytes_70b(cfg, "bf16", 4, 64, 32768, [16] * 21)
    assert parts["attention_linears"] == 28 * (2 * 4608 * 3584 + 3584 * 3584) * 2 // 4
    assert parts["dense_mlp"] == 7 * 3 * 18944 * 3584 * 2 // 4 and parts["shared_expert"] == 21 * 3 * 18944 * 3584 * 2 // 4
    assert parts["routed_experts"] == 21 * 16 * 3 * 18944 * 3584 * 2 // 4
    assert parts["kv_rows"] == 64 * 2 * 128 * 2 * (24 * (4096 + 128) + 4 * (32769 + 128))
    assert parts["gla_states"] == 28 * 64 * 2 * 7 * 128 * 128 * 2 and total == sum(parts.values())
    assert parts["lm_head_and_logits"] == 152064 // 4 * 3584 * 2 + 2 * 64 * (152064 // 4) * 4
    # the quantised checkpoints hold quantised experts too
    _, p8 = step_bytes_70b(cfg, "w8a8", 4, 64, 32768, [16] * 21)
    assert p8["routed_experts"] * 2 == parts["routed_experts"] and p8["dense_mlp"] * 2 == parts["dense_mlp"]
    _, p4 = step_bytes_70b(cfg, "gptq4", 4, 64, 32768, [16] * 21)
    assert p4["routed_experts"] * p8["dense_mlp"] == p8["routed_experts"] * p4["dense_mlp"] and p4["routed_experts"] < p8["routed_experts"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bench_tp import synthetic_checkpoint_70b
        from vllm_hymeta.models.configuration_hymeta import HymetaConfig
        from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM

        cfg = HymetaConfig(vocab_size=64, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                           intermediate_size=256, shared_intermediate_size=128, num_local_experts=4, num_meta_tokens=8,
                           max_position_embeddings=256)
        cpu = torch.device("cpu")
        full = dict(synthetic_checkpoint_70b(cfg, cpu, "bf16"))
        model = HymetaForCausalLM(cfg, max_num_seqs=2, tp_group=dist.group.WORLD, device="cpu")
        loaded = model.load_weights(synthetic_checkpoint_70b(cfg, cpu, "bf16"))
        assert "model.norm.weight" in loaded          # the checkpoint's "mode.norm" spelling is routed
        m = model.model
        assert (m.tp_rank, m.tp_size) == (rank, world)
        D, Hl, Kl = 128, 2, 1
        lay = m.layers[1].attn
        w = lay.vanilla_attn.qkv_proj.weight
        assert w.shape == ((Hl + 2 * Kl) * D, 512)
        p = "model.layers.1.attn.vanilla_attn."
        assert torch.equal(w[: Hl * D], full[p + "q_proj.weight"][rank * Hl * D:(rank + 1) * Hl * D])
        assert torch.equal(w[Hl * D:(Hl + Kl) * D], full[p + "k_proj.weight"][rank * Kl * D:(rank + 1) * Kl * D])
        assert torch.equal(lay.out_proj.weight, full["model.layers.1.attn.out_proj.weight"][:, rank * Hl * D:(rank + 1) * Hl * D])
        # layer 3 is a mixture layer: experts cut along the intermediate size, the shared expert like a dense MLP
        moe = m.layers[3].mlp
        assert moe.w13.shape == (4, 2 * 128, 512) and moe.w2.shape == (4, 512, 128)
        e2 = "model.layers.3.block_sparse_moe.experts.2."
        assert torch.equal(moe.w13[2, :128], full[e2 + "gate_proj.weight"][rank * 128:(rank + 1) * 128])
        assert torch.equal(moe.w2[2], full[e2 + "down_proj.weight"][:, rank * 128:(rank + 1) * 128])
        assert moe.shared_mlp.gate_up_proj.weight.shape == (2 * 64, 512) and not moe.shared_mlp.reduce_results
        # the ranks' shards of a row-parallel weight tile the full tensor
        parts = [torch.empty_like(lay.out_proj.weight) for _ in range(world)]
        dist.all_gather(parts, lay.out_proj.weight.contiguous())
        assert torch.equal(torch.cat(parts, dim=1), full["model.layers.1.attn.out_proj.weight"])
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.destroy_process_group()


def test_synthetic_70b_checkpoint_shards_over_two_ranks():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def _greedy_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace

        from bench_tp import VocabParallelGreedy

        g = torch.Generator().manual_seed(5)
        B, V, hidden = 6, 256, 64
        w = (torch.randn(V, hidden, generator=g) * 0.1).to(torch.bfloat16)
        w[200] = w[17]                                   # two equal logits in different ranks' shares: the lower index wins
        h = torch.randn(B, hidden, generator=g).to(torch.bfloat16)
        h[0] = w[17].float().sign().to(torch.bfloat16)   # row 0 peaks at tokens 17 and 200 alike
        model = SimpleNamespace(_lm_head=w.clone())
        sampler = VocabParallelGreedy(model, dist.group.WORLD, rank, world, B, torch.device("cpu"))
        assert model._lm_head.shape == (V // world, hidden) and torch.equal(model._lm_head, w[rank * V // world:(rank + 1) * V // world])
        out = torch.zeros(B, dtype=torch.int64)
        sampler(h, out)
        full = torch.nn.functional.linear(h.float(), w.float())
        assert torch.equal(out, full.argmax(-1)) or all(full[i, out[i]] == full[i].max() for i in range(B))
        assert int(out[0]) == 17
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.destroy_process_group()


def test_vocabulary_parallel_greedy_sampling_equals_the_full_argmax():
    """world_size 2 over gloo: every rank multiplies with its half of the lm_head, the ranks exchange one (logit, token)
    pair per sequence, and all of them end up with the argmax of the full logits (the lowest index among equals)."""
    world, port = 2, _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_greedy_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
