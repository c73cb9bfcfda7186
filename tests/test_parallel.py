"""Tensor-parallel plumbing on CPU: head-sharding rules and the two collectives of a layer over a
world_size-2 gloo group (the GPU path uses the same calls on RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vllm_hymeta.parallel import all_reduce_sum, all_reduce_variance, shard_heads, tp_rank_and_size


def test_shard_heads_rules():
    s = shard_heads(28, 4, 1, 0)
    assert (s.num_heads, s.num_kv_heads, s.head_start, s.kv_head_start) == (28, 4, 0, 0)
    s = shard_heads(28, 4, 2, 1)
    assert (s.num_heads, s.num_kv_heads, s.head_start, s.kv_head_start) == (14, 2, 14, 2)
    s = shard_heads(28, 4, 4, 3)
    assert (s.num_heads, s.num_kv_heads, s.head_start, s.kv_head_start) == (7, 1, 21, 3)
    # the q-heads of a rank must map onto the kv-heads it holds (group size 7)
    for tp in (1, 2, 4):
        for r in range(tp):
            s = shard_heads(28, 4, tp, r)
            for h in range(s.head_start, s.head_start + s.num_heads):
                assert s.kv_head_start <= h // 7 < s.kv_head_start + s.num_kv_heads
    with pytest.raises(ValueError):
        shard_heads(28, 4, 8, 0)          # 28 heads do not split 8 ways
    with pytest.raises(ValueError):
        shard_heads(28, 4, 3, 0)
    # kv-head replication when there are more ranks than kv-heads
    s = shard_heads(32, 4, 8, 5)
    assert (s.num_heads, s.num_kv_heads, s.head_start, s.kv_head_start) == (4, 1, 20, 2)
    with pytest.raises(ValueError):
        shard_heads(24, 4, 6, 0)          # 6 ranks are not a multiple of 4 kv-heads


def test_collectives_are_noops_without_group():
    x = torch.ones(3)
    assert all_reduce_sum(x) is x and all_reduce_variance(x) is x
    assert tp_rank_and_size() == (0, 1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.gla import gla_recurrent_ref
        from oracle.layers import hybrid_combine_ref, rms_norm_tp_ref

        g = torch.Generator().manual_seed(7)
        T, H, Hkv, D = 6, 4, 2, 16
        hidden = H * D
        q = torch.randn(T, H, D, generator=g)
        k = torch.randn(T, Hkv, D, generator=g)
        v = torch.randn(T, Hkv, D, generator=g)
        gk = -torch.rand(T, Hkv, D, generator=g)
        w1 = 1 + 0.1 * torch.randn(hidden, generator=g)
        w2 = 1 + 0.1 * torch.randn(hidden, generator=g)
        o_attn = torch.randn(T, hidden, generator=g).to(torch.bfloat16)
        w_out = torch.randn(hidden, hidden, generator=g) / hidden ** 0.5

        rep = H // Hkv
        full, _ = gla_recurrent_ref(q, k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1),
                                    gk.repeat_interleave(rep, 1), None)
        full = full.to(torch.bfloat16).reshape(T, hidden)
        y_full = hybrid_combine_ref(o_attn, full, w1, w2, 1e-6).float() @ w_out.t()

        # this rank's heads: attention needs no communication
        s = shard_heads(H, Hkv, world, rank)
        tp = dist.group.WORLD
        assert tp_rank_and_size(tp) == (rank, world)
        # without a group nothing is reduced, even though a process group exists (replicas, DP, PP)
        assert tp_rank_and_size() == (0, 1)
        probe = torch.full((2,), float(rank + 1))
        assert torch.equal(all_reduce_sum(probe.clone()), probe) and torch.equal(all_reduce_variance(probe.clone()), probe)
        hs = slice(s.head_start, s.head_start + s.num_heads)
        ks = slice(s.kv_head_start, s.kv_head_start + s.num_kv_heads)
        lrep = s.num_heads // s.num_kv_heads
        loc, _ = gla_recurrent_ref(q[:, hs], k[:, ks].repeat_interleave(lrep, 1), v[:, ks].repeat_interleave(lrep, 1),
                                   gk[:, ks].repeat_interleave(lrep, 1), None)
        loc = loc.to(torch.bfloat16).reshape(T, -1)
        cols = slice(s.head_start * D, (s.head_start + s.num_heads) * D)
        assert torch.equal(loc, full[:, cols])
        a_loc = o_attn[:, cols]
        # variance all-reduce: per-rank mean of squares -> mean over the full hidden size
        stats = torch.stack([a_loc.float().pow(2).mean(-1), loc.float().pow(2).mean(-1)], dim=-1)
        stats = all_reduce_variance(stats, tp)
        ref_var = torch.stack([o_attn.float().pow(2).mean(-1), full.float().pow(2).mean(-1)], dim=-1)
        torch.testing.assert_close(stats, ref_var, rtol=1e-6, atol=1e-7)
        n1 = rms_norm_tp_ref(a_loc, w1[cols], stats[:, 0:1], 1e-6)
        n2 = rms_norm_tp_ref(loc, w2[cols], stats[:, 1:2], 1e-6)
        y_loc = ((n1.float() + n2.float()) / 2).to(torch.bfloat16)
        # row-parallel out projection: partial products summed over ranks
        y = all_reduce_sum(y_loc.float() @ w_out[:, cols].t(), tp)
        torch.testing.assert_close(y, y_full, rtol=2e-2, atol=2e-2)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_layer_collectives():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}


def _order_worker(rank, world, port, ret):
    """The sum the one-shot exchange computes (rank order, fp32, one rounding) against the group's own all-reduce, on the
    decode step's messages; with no exchange registered for the group the collectives stay on the backend."""
    from vllm_hymeta import parallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tp = dist.group.WORLD
        assert parallel._ONE_SHOT.get(tp) is None
        g = torch.Generator().manual_seed(100 + rank)
        for rows in (1, 64):
            x = torch.randn(rows, 3584, generator=g).to(torch.bfloat16)
            st = torch.rand(rows, 2, generator=g)
            gx = [torch.empty_like(x) for _ in range(world)]
            gs = [torch.empty_like(st) for _ in range(world)]
            dist.all_gather(gx, x)
            dist.all_gather(gs, st)
            ref_x = parallel.one_shot_reduce_ref(gx, torch.bfloat16)
            ref_s = parallel.one_shot_reduce_ref(gs, torch.float32)
            # two addends: the backend's bf16 sum is the correctly rounded sum too, whatever its order
            assert torch.equal(all_reduce_sum(x.clone(), tp), ref_x)
            assert torch.equal(all_reduce_variance(st.clone(), tp), ref_s / world)
            # every rank computes the same bits from the same gathered messages
            mine = ref_x.float().sum().reshape(1)
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert torch.equal(both[0], both[1])
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_one_shot_sum_order_equals_the_group_all_reduce_on_two_ranks():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_order_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}
