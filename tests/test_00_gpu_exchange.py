"""One-shot all-reduce over peer-mapped exchange buffers (``hymeta_p2p_*``), all ranks in one process on the streams of
one GPU: every rank's result equals the rank-ordered fp32 sum of the ranks' messages bit for bit, for the decode step's
message shapes (bf16 rows, fp32 [T, 2] statistics, both in one exchange), over many calls in a row (the two parities of
the slots, changing sizes and block counts), from HIP graphs, and a wait for a peer that never comes ends at its limit.
Stands in for the tensor_model_parallel_all_reduce calls of the reference
(``vllm_hymeta/models/modeling_hymeta.py:158-160,858-861``).

The ranks share one process, each on its own stream, and wait for each other inside kernels.  A process has 4 hardware
queues, a stream gets one at its first use and keeps it, and kernels of streams that share a queue run in order: a rank
behind another rank's waiting exchange kernel never gets to send what that kernel waits for.  The tests therefore run on
streams that were measured to run side by side (``parallel.concurrent_streams``; ``tools/hwqueue_probe.py`` prints the
pairs that serialise), and every wait is bounded: a surprise fails at its limit, it does not hang.  The file is named to
run first among the GPU tests, while few streams have a queue yet."""
import pytest
import torch

from _util import gen, randn_bf16

pytestmark = pytest.mark.gpu
HIDDEN = 3584


def _ranks(world, max_bytes=2 * 1024 * 1024, wait_s=1.0):
    from vllm_hymeta.parallel import OneShotAllReduce, concurrent_streams

    dev = torch.device("cuda")
    streams = concurrent_streams(world, dev)      # measured not to share a hardware queue (see the module docstring)
    comms = OneShotAllReduce.in_process(world, dev, max_bytes)
    for c in comms:
        c.set_wait_limit(wait_s)      # a scheduling surprise fails the test within a second instead of waiting
    return comms, streams, dev


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
    """... and what it could not sum comes out as NaN, never as a sum of whatever the slots held; the serving path's check
    turns the sticky count into an error."""
    from vllm_hymeta.parallel import InProcessGroup, OneShotTimeout, raise_on_exchange_timeout

    comms, streams, dev = _ranks(2, wait_s=0.002)
    x = randn_bf16(gen(60), 64, HIDDEN).to(dev)
    s = torch.rand(64, 2, generator=gen(61)).to(dev)
    comms[0].all_reduce(data=x, stats=s)          # rank 1 never calls: rank 0's blocks give up after 2 ms and the launch ends
    torch.cuda.synchronize()
    assert comms[0].timeouts() > 0 and comms[1].timeouts() == 0
    assert bool(torch.isnan(x).all()) and bool(torch.isnan(s).all())
    raise_on_exchange_timeout(InProcessGroup(1, 2, comms[1], streams[1]))      # the rank that never waited: nothing to report
    with pytest.raises(OneShotTimeout):
        raise_on_exchange_timeout(InProcessGroup(0, 2, comms[0], streams[0]))
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
    # group sizes other than 1, 2, 4, 8 and a rank outside its group are refused; so is an exchange before the
    # peers are connected
    import ctypes as C
    comm = C.c_void_p()
    assert lib.hymeta_p2p_create(0, 3, 4096, C.byref(comm), None) == -2
    assert lib.hymeta_p2p_create(2, 2, 4096, C.byref(comm), None) == -1
    assert lib.hymeta_p2p_create(0, 2, 4096, C.byref(comm), None) == 0
    assert lib.hymeta_p2p_all_reduce(comm, big.data_ptr(), 64, None, 0, 1.0, None) == -1
    assert lib.hymeta_p2p_verify(comm) == -1
    assert lib.hymeta_p2p_destroy(comm) == 0
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
    assert [c.timeouts() for c in comms] == [0] * world
    for r, o in enumerate(outs):
        assert torch.equal(o, ref), f"rank {r}: {int((o != ref).sum())} of {o.numel()} values differ from reduce + exchange"
    _close(comms)


D = 128


@pytest.mark.parametrize("expert_parallel,fmt", [(False, "bf16"), (True, "bf16"), (False, "w8a8"), (True, "w8a8"), (False, "gptq4"),
                                                 (True, "gptq4")])
def test_two_tensor_parallel_ranks_on_one_gpu_equal_the_unsharded_model(expert_parallel, fmt):
    """``fmt``: the checkpoint as bf16, as the W8A8 recipe stores it (int8 linears, experts and router) or as the GPTQ
    recipe does (packed linears and experts): the quantised formats are cut per rank on their own boundaries (scales with
    the output rows, packed words and groups with the input rows) and are compared with the unsharded model of the same
    checkpoint.
    ``expert_parallel``: the routed experts dealt whole to the ranks (2 of the 4 each, only those held) instead of
    every expert cut along its intermediate size -- the reference's ``--enable-expert-parallel``; either way the layer's
    one all-reduce sums the ranks.
    Configs with tensor parallelism, collectives included, on one GPU: two ranks of HymetaForCausalLM live in one
    process, each in its own thread on its own stream, with the layers' all-reduces (branch variances, out_proj, dense
    MLP, expert mixture + shared expert) running through the one-shot exchange between them.  Both ranks cut their
    shards from the same checkpoint, prefill two prompts and decode one token.  The ranks' hidden states are bit-identical
    (the exchange adds in rank order on every rank) and equal the unsharded model's and the CPU restatement's within
    the model tolerance; the vocabulary-parallel greedy pick over the ranks' logits equals the unsharded argmax."""
    import threading

    from test_gpu_layers import _md
    from test_gpu_model_70b import _checkpoint
    from _util import assert_bf16_close

    from oracle.model import hymeta_model_ref
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM
    from vllm_hymeta.parallel import InProcessGroup

    dev = torch.device("cuda")
    g_ = gen(31)
    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    ckpt = _checkpoint(g_, cfg)
    lens = [75, 40]
    ids = [torch.randint(0, cfg.vocab_size, (L + 1,), generator=g_) for L in lens]
    refs = [hymeta_model_ref(x, ckpt, cfg) for x in ids] if fmt == "bf16" else None
    quant = None
    if fmt == "w8a8":
        from test_gpu_model_70b import _w8a8_recipe
        ckpt = _w8a8_recipe(ckpt)
    elif fmt == "gptq4":
        from test_gpu_model_70b import _gptq_recipe
        ckpt, _ = _gptq_recipe(ckpt, g_)
        quant = {"bits": 4, "group_size": 128, "zero_offset": 1}
    bs = 16
    per_seq = (max(lens) + 1 + bs - 1) // bs
    bt_cpu = torch.randperm(2 * per_seq + 1, generator=g_)[: 2 * per_seq].to(torch.int32).view(2, per_seq)
    slot = lambda s, t: int(bt_cpu[s, t // bs]) * bs + t % bs
    pre_slots = [slot(s, t) for s, L in enumerate(lens) for t in range(L)]
    dec_slots = [slot(s, L) for s, L in enumerate(lens)]

    def run_rank(group, out):
        """One rank: build, load, prefill, decode.  ``group`` None = the unsharded model."""
        kvh = cfg.num_key_value_heads // (group.size if group is not None else 1)
        model = HymetaForCausalLM(cfg, max_num_seqs=4, tp_group=group, device=dev, expert_parallel=expert_parallel, quant=quant)
        model.load_weights(ckpt.items())
        if group is not None:
            from vllm_hymeta.models.moe import HymetaMoE
            moes = [m for m in model.modules() if isinstance(m, HymetaMoE)]
            assert moes
            for m in moes:      # a rank holds two whole experts, or a half of each of the four
                assert tuple(m.w13.shape) == ((2, 2 * cfg.intermediate_size, cfg.hidden_size) if expert_parallel
                                              else (4, cfg.intermediate_size, cfg.hidden_size))
                assert m.local_experts == ((2 * group.rank, 2 * group.rank + 2) if expert_parallel else None)
        kv_caches = [torch.zeros(2, 2 * per_seq + 1, bs, kvh, D, dtype=torch.bfloat16, device=dev)
                     for _ in range(cfg.num_hidden_layers)]
        run = dict(request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])
        bt = bt_cpu.to(dev)
        md = _md(dev, prefill_lens=lens, context_lens=[0, 0], slot_mapping=torch.tensor(pre_slots, device=dev))
        pre_ids = torch.cat([x[:L] for x, L in zip(ids, lens)]).to(dev)
        pre_pos = torch.cat([torch.arange(L) for L in lens]).to(dev)
        h_pre = model(pre_ids, pre_pos, attn_metadata=md, kv_caches=kv_caches, **run)
        md = _md(dev, decode_seq_lens=[L + 1 for L in lens], block_tables=bt, slot_mapping=torch.tensor(dec_slots, device=dev))
        dec_ids = torch.stack([x[L] for x, L in zip(ids, lens)]).to(dev)
        h_dec = model(dec_ids, torch.tensor(lens, device=dev), attn_metadata=md, kv_caches=kv_caches, **run)
        # the rank's share of the vocabulary (the unsharded model: all of it)
        rank, size = (group.rank, group.size) if group is not None else (0, 1)
        rows = cfg.vocab_size // size
        logits = torch.nn.functional.linear(h_dec.float(), model.lm_head[rank * rows:(rank + 1) * rows].float())
        torch.cuda.current_stream().synchronize()
        out.update(h_pre=h_pre.cpu(), h_dec=h_dec.cpu(), logits=logits.cpu())

    full = {}
    run_rank(None, full)
    torch.cuda.synchronize()

    groups = InProcessGroup.make(2, dev, max_bytes=4 * 1024 * 1024, wait_seconds=30.0)
    outs, errors = [{}, {}], []

    def thread_main(r):
        try:
            with torch.cuda.stream(groups[r].stream):
                run_rank(groups[r], outs[r])
        except Exception as exc:       # surfaces below: a rank that dies leaves the other to its wait limit
            errors.append((r, exc))

    threads = [threading.Thread(target=thread_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    torch.cuda.synchronize()
    assert not errors, errors
    assert all(g.comm.timeouts() == 0 for g in groups)
    for g in groups:
        g.comm.close()

    for key in ("h_pre", "h_dec"):
        assert torch.equal(outs[0][key], outs[1][key]), f"{key}: the two ranks differ"
    if fmt != "bf16":
        # (W8A8: a rank quantises the slice of an activated row that it holds, with the slice's own scale)
        assert_bf16_close(outs[0]["h_pre"], full["h_pre"].float(), f"TP=2 {fmt} model against the unsharded one, prefill", rel=3e-2, frac=0.15)
        assert_bf16_close(outs[0]["h_dec"], full["h_dec"].float(), f"TP=2 {fmt} model against the unsharded one, decode", rel=3e-2, frac=0.15)
        return
    ref_pre = torch.cat([r[0][:L] for r, L in zip(refs, lens)])
    ref_dec = torch.stack([r[0][L] for r, L in zip(refs, lens)])
    assert_bf16_close(outs[0]["h_pre"], ref_pre, "TP=2 model, prefill hidden states", rel=2e-2, frac=2.0 ** -4)
    assert_bf16_close(outs[0]["h_dec"], ref_dec, "TP=2 model, decode hidden states", rel=2e-2, frac=2.0 ** -4)
    assert_bf16_close(outs[0]["h_dec"], full["h_dec"].float(), "TP=2 against the unsharded model", rel=2e-2, frac=2.0 ** -4)
    # vocabulary-parallel greedy pick: best (value, index) pair over the ranks
    both = torch.cat([outs[0]["logits"], outs[1]["logits"]], dim=1)
    assert_bf16_close(both, full["logits"], "TP=2 logits against the unsharded model", rel=2e-2, frac=2.0 ** -4)
    ref_logits = torch.stack([r[1][L] for r, L in zip(refs, lens)])
    top2 = ref_logits.float().topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05 * top2[:, 0].abs()      # rows whose winner is not a near tie
    assert torch.equal(both.argmax(-1)[clear], ref_logits.argmax(-1)[clear])

