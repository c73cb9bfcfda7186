"""MoE host logic and the expert-parallel exchange on CPU (gloo, world_size 2): each rank's share
of the oracle block, summed by the layer's all-reduce, is the whole block."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import gen, randn_bf16


def test_oracle_routing_rules():
    from oracle.moe import moe_route_ref, silu_and_mul_ref

    logits = torch.tensor([[0.0, 2.0, 2.0, -1.0], [3.0, 1.0, 0.0, 0.0]])
    ids, w = moe_route_ref(logits, 2, False)
    assert ids.tolist() == [[1, 2], [0, 1]]                      # ties go to the lower expert index
    probs = torch.softmax(logits, -1)
    assert torch.allclose(w, torch.stack([probs[0, [1, 2]], probs[1, [0, 1]]]))
    _, wr = moe_route_ref(logits, 2, True)
    assert torch.allclose(wr.sum(-1), torch.ones(2))             # the eager module's renormalisation
    x = torch.tensor([[1.0, -2.0, 3.0, 0.5]], dtype=torch.bfloat16)
    ref = torch.nn.functional.silu(x[:, :2].float()) * x[:, 2:].float()
    assert torch.equal(silu_and_mul_ref(x), ref.to(torch.bfloat16))


def test_expert_range():
    from vllm_hymeta.models.moe import expert_range

    assert [expert_range(16, 4, r) for r in range(4)] == [(0, 4), (4, 8), (8, 12), (12, 16)]
    assert expert_range(16, 1, 0) == (0, 16)
    with pytest.raises(ValueError):
        expert_range(16, 5, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.moe import moe_ref
        from vllm_hymeta.models.moe import expert_range
        from vllm_hymeta.parallel import all_reduce_sum

        g_ = gen(3)
        T, H, inter, E = 24, 64, 32, 8
        x = randn_bf16(g_, T, H)
        w13 = randn_bf16(g_, E, 2 * inter, H, scale=H ** -0.5)
        w2 = randn_bf16(g_, E, H, inter, scale=inter ** -0.5)
        logits = randn_bf16(g_, T, E)
        whole = moe_ref(x, logits, w13, w2)
        share = moe_ref(x, logits, w13, w2, local_experts=expert_range(E, world, rank))
        total = all_reduce_sum(share.float(), dist.group.WORLD)
        assert torch.equal(total.to(torch.bfloat16), whole)      # top-1: the shares have disjoint support
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_expert_parallel_all_reduce_world_size_2():
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}
