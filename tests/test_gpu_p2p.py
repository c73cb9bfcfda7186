"""One-shot all-reduce over peer-mapped exchange buffers (``hymeta_p2p_*``), all ranks in one process on the streams of
one GPU: every rank's result equals the rank-ordered fp32 sum of the ranks' messages bit for bit, for the decode step's
message shapes (bf16 rows, fp32 [T, 2] statistics, both in one exchange), over many calls in a row (the two parities of
the slots, changing sizes and block counts), from HIP graphs, and a wait for a peer that never comes ends at its limit.
Stands in for the tensor_model_parallel_all_reduce calls of the reference
(``vllm_hymeta/models/modeling_hymeta.py:158-160,858-861``)."""
import pytest
import torch

from _util import gen, randn_bf16

pytestmark = pytest.mark.gpu
HIDDEN = 3584


def _ranks(world, max_bytes=2 * 1024 * 1024, wait_s=1.0):
    from vllm_hymeta.parallel import OneShotAllReduce

    dev = torch.device("cuda")
    comms = OneShotAllReduce.in_process(world, dev, max_bytes)
    for c in comms:
        c.set_wait_limit(wait_s)      # a scheduling surprise fails the test within a second instead of waiting
    return comms, [torch.cuda.Stream() for _ in range(world)], dev


def _close(comms):
    torch.cuda.synchronize()
    for c in comms:
        c.close()


def _messages(g_, world, rows, with_data, with_stats, dev):
    xs = [randn_bf16(g_, rows, HIDDEN).to(dev) if with_data else None for _ in range(world)]
    ss = [torch.rand(rows, 2, generator=g_).to(dev) if with_stats else None for _ in range(world)]
    return xs, ss


def _reference(parts, dtype):
    from vllm_hymeta.parallel import one_shot_reduce_ref
    return None if parts[0] is None else one_shot_reduce_ref([p.clone() for p in parts], dtype)


@pytest.mark.parametrize("world", [1, 2, 4])
def test_every_rank_gets_the_rank_ordered_sum(world):
    comms, streams, dev = _ranks(world)
    g_ = gen(40 + world)
    # (rows, data, stats): the step's messages; 1 row = a partial 16-byte unit of statistics; 7 rows of statistics alone;
    # 256 rows = the largest message (several blocks); sizes change from call to call, 9 calls = both parities
    cases = [(64, True, True), (1, True, True), (7, False, True), (256, True, False), (64, True, False), (64, False, True),
             (3, True, True), (256, True, True), (8, True, False)]
    for rows, with_data, with_stats in cases:
        xs, ss = _messages(g_, world, rows, with_data, with_stats, dev)
        ref_x, ref_s = _reference(xs, torch.bfloat16), _reference(ss, torch.float32)
        torch.cuda.synchronize()
        for c, st, x, s in zip(comms, streams, xs, ss):
            with torch.cuda.stream(st):
                c.all_reduce(data=x, stats=s)
        torch.cuda.synchronize()
        for r in range(world):
            if with_data:
                assert torch.equal(xs[r], ref_x), f"rank {r} of {world}: data of {rows} rows"
            if with_stats:
                assert torch.equal(ss[r], ref_s), f"rank {r} of {world}: statistics of {rows} rows"
    # the mean of the ranks' statistics: the scale rides in the exchange
    _, ss = _messages(g_, world, 64, False, True, dev)
    ref_mean = _reference(ss, torch.float32) * (1.0 / world)
    torch.cuda.synchronize()
    for c, st, s in zip(comms, streams, ss):
        with torch.cuda.stream(st):
            c.all_reduce(stats=s, stats_scale=1.0 / world)
    torch.cuda.synchronize()
    assert all(torch.equal(s, ref_mean) for s in ss)
    assert all(c.timeouts() == 0 for c in comms)
    _close(comms)


def test_replays_from_graphs_with_eager_calls_between():
    world = 2
    comms, streams, dev = _ranks(world)
    g_ = gen(50)
    xs = [torch.zeros(64, HIDDEN, dtype=torch.bfloat16, device=dev) for _ in range(world)]
    ss = [torch.zeros(64, 2, device=dev) for _ in range(world)]
    graphs = []
    for c, st, x, s in zip(comms, streams, xs, ss):      # nothing runs during a capture: the epochs stay together
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            c.all_reduce(stats=s)
            c.all_reduce(data=x)
        graphs.append(graph)
    for it in range(4):
        new_x, new_s = _messages(g_, world, 64, True, True, dev)
        for x, s, nx, ns in zip(xs, ss, new_x, new_s):
            x.copy_(nx)
            s.copy_(ns)
        ref_x, ref_s = _reference(new_x, torch.bfloat16), _reference(new_s, torch.float32)
        torch.cuda.synchronize()
        for graph, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                graph.replay()
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(xs[r], ref_x) and torch.equal(ss[r], ref_s), f"replay {it}, rank {r}"
        # an eager exchange of another size between the replays (an odd number of calls: the parity moves on)
        ex, _ = _messages(g_, world, 5, True, False, dev)
        ref_e = _reference(ex, torch.bfloat16)
        for c, st, x in zip(comms, streams, ex):
            with torch.cuda.stream(st):
                c.all_reduce(data=x)
        torch.cuda.synchronize()
        assert all(torch.equal(x, ref_e) for x in ex)
    assert all(c.timeouts() == 0 for c in comms)
    _close(comms)


def test_a_wait_for_a_missing_peer_ends_at_its_limit():
    comms, streams, dev = _ranks(2, wait_s=0.002)
    x = randn_bf16(gen(60), 64, HIDDEN).to(dev)
    comms[0].all_reduce(data=x)          # rank 1 never calls: rank 0's blocks give up after 2 ms and the launch ends
    torch.cuda.synchronize()
    assert comms[0].timeouts() > 0 and comms[1].timeouts() == 0
    _close(comms)


def test_limits_and_unsupported_messages():
    from vllm_hymeta import _C
    comms, streams, dev = _ranks(1, max_bytes=4096)
    c = comms[0]
    ok = torch.zeros(2048, dtype=torch.bfloat16, device=dev)
    assert c.supports(ok) and not c.supports(torch.zeros(2056, dtype=torch.bfloat16, device=dev))     # one unit too many
    assert not c.supports(torch.zeros(12, dtype=torch.bfloat16, device=dev))          # not a multiple of 8 values
    assert not c.supports(torch.zeros(16, dtype=torch.float16, device=dev))
    assert not c.supports(ok, torch.zeros(1, 2, device=dev))                          # statistics no longer fit
    with pytest.raises(ValueError):
        c.all_reduce(data=torch.zeros(2056, dtype=torch.bfloat16, device=dev))
    # the C entry point itself refuses what the wrapper filters
    big = torch.zeros(4096, dtype=torch.bfloat16, device=dev)
    lib = _C.load()
    assert lib.hymeta_p2p_all_reduce(c._comm, big.data_ptr(), big.numel(), None, 0, 1.0, None) == _C.ERR_SCRATCH_LIMIT
    assert lib.hymeta_p2p_all_reduce(c._comm, big.data_ptr(), 12, None, 0, 1.0, None) == -2
    x = torch.ones(2048, dtype=torch.bfloat16, device=dev)
    c.all_reduce(data=x)                 # a world of one: the sum is the message
    torch.cuda.synchronize()
    assert torch.equal(x, torch.ones_like(x)) and c.timeouts() == 0
    _close(comms)


@pytest.mark.parametrize("kind", ["bf16", "w8a8"])
def test_exchange_takes_a_k_sliced_linear_straight_from_its_partial_sums(kind):
    """A row-parallel projection in front of its all-reduce: each rank's K-sliced linear leaves its slices' sum to the
    exchange, which adds them as it reads its message -- the same bits as reduce launch + exchange of the tensor."""
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    world, rows, K, N = 2, 20, 2048, 1024
    comms, streams, dev = _ranks(world)
    g_ = gen(70)
    lins, xs = [], []
    for r in range(world):
        w = randn_bf16(g_, N, K, scale=K ** -0.5)
        if kind == "w8a8":
            s = w.float().abs().amax(1, keepdim=True).clamp_min(1e-8) / 127
            lins.append(HymetaLinear.from_w8a8(torch.clamp(torch.round(w.float() / s), -127, 127).to(torch.int8).to(dev), s.float().to(dev)))
        else:
            lins.append(HymetaLinear.from_weight(w.to(dev)))
        xs.append(randn_bf16(g_, rows, K).to(dev))
    torch.cuda.synchronize()
    plain = []
    for lin, x, st in zip(lins, xs, streams):
        with torch.cuda.stream(st):
            plain.append(lin(x))
    torch.cuda.synchronize()
    ref = _reference(plain, torch.bfloat16)
    outs = []
    for c, lin, x, st in zip(comms, lins, xs, streams):
        with torch.cuda.stream(st):
            p = lin(x, defer=True)
            assert isinstance(p, hip_ops.PartialSums), "the case is meant to be K-sliced"
            outs.append(c.all_reduce_partials(p))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs) and all(c.timeouts() == 0 for c in comms)
    _close(comms)

