"""Tensor-parallel plumbing around the attention kernels: head sharding rules and the two
collectives of a layer, on torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  The attention kernels themselves need no communication: every (sequence, head)
is independent, so a rank simply runs them on its own heads."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class HeadShard:
    """Heads owned by one tensor-parallel rank."""
    tp_size: int
    tp_rank: int
    num_heads: int        # q-heads on this rank
    num_kv_heads: int     # kv-heads on this rank (>= 1: replicated when tp_size > total kv-heads)
    head_start: int       # first global q-head
    kv_head_start: int    # first global kv-head


def shard_heads(total_heads: int, total_kv_heads: int, tp_size: int, tp_rank: int) -> HeadShard:
    """The reference's sharding rules (``vllm_hymeta/models/modeling_hymeta.py:401-420``): q-heads
    split evenly (``H % tp == 0``); kv-heads split evenly when there are at least ``tp`` of them,
    otherwise each kv-head is replicated on ``tp / KVH`` ranks.  With H = 28, KVH = 4 this admits
    tp in {1, 2, 4, 28}; tp = 8 is rejected, as in the reference."""
    if total_heads % tp_size != 0:
        raise ValueError(f"num_heads {total_heads} must be divisible by tensor-parallel size {tp_size}")
    if total_kv_heads >= tp_size:
        if total_kv_heads % tp_size != 0:
            raise ValueError(f"num_kv_heads {total_kv_heads} must be divisible by tensor-parallel size {tp_size}")
        kv_local = total_kv_heads // tp_size
        kv_start = tp_rank * kv_local
    else:
        if tp_size % total_kv_heads != 0:
            raise ValueError(f"tensor-parallel size {tp_size} must be a multiple of num_kv_heads {total_kv_heads}")
        kv_local = 1
        kv_start = tp_rank // (tp_size // total_kv_heads)
    h_local = total_heads // tp_size
    return HeadShard(tp_size, tp_rank, h_local, kv_local, tp_rank * h_local, kv_start)


# Rehearsal switch: run the collectives over a ONE-rank group too (a multi-GPU step exercised on one GPU, graph
# capture of the collective calls included).  Off in serving: a one-rank all-reduce is pure overhead.
SINGLE_RANK_COLLECTIVES = False


# Rehearsal switch: (rank, size) that ``tp_rank_and_size`` reports for any group, so that ONE process builds and runs
# the shard of one rank of a larger tensor-parallel group (its weight cuts, head counts and expert shares) with the
# collectives running over the real, smaller group.  Results are one rank's partial sums: for timing and profiling only.
EMULATED_TP: Optional[Tuple[int, int]] = None


def _active(group: Optional[dist.ProcessGroup]) -> bool:
    """``group is None`` means "no tensor parallelism" and is a strict no-op, whatever process group
    the program holds for other purposes (replicas, data or pipeline parallelism): a collective only
    runs over a group that was passed explicitly (``dist.group.WORLD`` for the whole job)."""
    if isinstance(group, InProcessGroup):
        return group.size > 1 or SINGLE_RANK_COLLECTIVES
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or SINGLE_RANK_COLLECTIVES


# Messages up to this size take the one-shot exchange when a group has one (``enable_one_shot_all_reduce``):
# [256, 3584] bf16 with its [256, 2] fp32 statistics fits; anything larger stays with RCCL, whose rings win there.
ONE_SHOT_MAX_BYTES = 2 * 1024 * 1024


class OneShotAllReduce:
    """One rank's end of the one-shot all-reduce over peer-mapped device memory (``hymeta_p2p_*``): every rank writes
    its message into a slot of every rank's exchange buffer, raises a flag, waits for all flags in its own buffer and
    adds the slots in rank order in fp32 -- one hop over xGMI where a ring takes 2 (W - 1), and bit-identical results
    on all ranks.  Stands in, for the decode step's small messages, for the ``tensor_model_parallel_all_reduce`` calls
    of the reference (``vllm_hymeta/models/modeling_hymeta.py:158-160,858-861``).

    Build with ``for_group`` (one process per GPU; the IPC handles travel through the process group) or
    ``in_process`` (all ranks in one process, e.g. on the streams of one GPU).  All ranks make the same calls in the
    same order; a rank's calls must be ordered on one stream (or inside one graph)."""

    def __init__(self, rank: int, world: int, device: torch.device, max_bytes: int = ONE_SHOT_MAX_BYTES):
        import ctypes as C

        from . import _C
        self._lib, self._C, self._check = _C.load(), C, _C.check
        self.rank, self.world, self.device, self.max_bytes = rank, world, torch.device(device), int(max_bytes)
        self._comm = C.c_void_p()
        handle = (C.c_ubyte * 64)()
        with torch.cuda.device(self.device):
            self._check(self._lib.hymeta_p2p_create(rank, world, self.max_bytes, C.byref(self._comm), handle), "hymeta_p2p_create")
        self.handle = bytes(handle)
        self.group = None

    @classmethod
    def for_group(cls, group: dist.ProcessGroup, device: torch.device, max_bytes: int = ONE_SHOT_MAX_BYTES) -> "OneShotAllReduce":
        """Collective over the group.  A failure on any rank (no uncached memory, a handle that cannot be mapped) raises
        RuntimeError on EVERY rank, after the same collectives, so that the callers fall back together."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_device = dist.get_backend(group) == "nccl"       # gloo moves host tensors only
        where = torch.device(device) if on_device else torch.device("cpu")
        self, err = None, ""
        try:
            self = cls(rank, world, device, max_bytes)
            self.group = group
        except Exception as exc:
            err = f"rank {rank}: {exc}"
        # per rank: the 64-byte handle, a success byte, the device ordinal (one process per GPU on one node sees all of the
        # node's GPUs under the same ordinals)
        dev_index = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        mine = torch.tensor(list(self.handle if self is not None else bytes(64)) + [0 if err else 1, dev_index],
                            dtype=torch.uint8).to(where)
        gathered = torch.empty(world * 66, dtype=torch.uint8, device=where)
        dist.all_gather_into_tensor(gathered, mine, group=group)
        rows = gathered.cpu().view(world, 66)

        def agree(err_now: str) -> bool:      # the same answer on every rank; also a barrier
            ok = torch.tensor([0.0 if err_now else 1.0], device=where)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            return ok.item() == 1.0

        if bool(rows[:, 64].all()):
            try:
                devices = (self._C.c_int32 * world)(*[int(d) for d in rows[:, 65]])
                with torch.cuda.device(self.device):
                    self._check(self._lib.hymeta_p2p_connect(self._comm, bytes(rows[:, :64].reshape(-1).tolist()), devices),
                                "hymeta_p2p_connect")
            except Exception as exc:
                err = f"rank {rank}: {exc}"
        elif not err:
            err = "another rank could not create its exchange buffer"
        good = agree(err)
        if good:
            try:      # every rank has written its token into every buffer: look for them in the own one
                with torch.cuda.device(self.device):
                    self._check(self._lib.hymeta_p2p_verify(self._comm), "hymeta_p2p_verify")
            except Exception as exc:
                err = f"rank {rank}: {exc}"
            good = agree(err)
        if not good:
            if self is not None:
                self.group = None
                self.close()
            raise RuntimeError(f"one-shot all-reduce unavailable: {err or 'another rank could not map the exchange buffers'}")
        return self

    @classmethod
    def in_process(cls, world: int, device: torch.device, max_bytes: int = ONE_SHOT_MAX_BYTES) -> list:
        ranks = [cls(r, world, device, max_bytes) for r in range(world)]
        C = ranks[0]._C
        buffers = (C.c_void_p * world)(*[r._lib.hymeta_p2p_local_buffer(r._comm) for r in ranks])
        for r in ranks:
            r._check(r._lib.hymeta_p2p_connect_pointers(r._comm, buffers), "hymeta_p2p_connect_pointers")
        return ranks

    def supports(self, data: Optional[torch.Tensor], stats: Optional[torch.Tensor] = None) -> bool:
        nbytes = 0
        if data is not None:
            if not (data.is_cuda and data.dtype == torch.bfloat16 and data.is_contiguous() and data.numel() % 8 == 0
                    and data.data_ptr() % 16 == 0):
                return False
            nbytes += data.numel() * 2
        if stats is not None:
            if not (stats.is_cuda and stats.dtype == torch.float32 and stats.is_contiguous()):
                return False
            nbytes += (stats.numel() * 4 + 15) // 16 * 16
        return 0 < nbytes <= self.max_bytes

    def all_reduce(self, data: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, stats_scale: float = 1.0) -> None:
        """``data`` (bf16) and ``stats`` (fp32) become their sums over the ranks, in place, in one exchange, on the
        current stream of their device; the fp32 sums are multiplied by ``stats_scale``."""
        if not self.supports(data, stats):
            raise ValueError("one-shot all-reduce: contiguous bf16 data (a multiple of 8 values) and / or fp32 statistics "
                             f"on the GPU, {self.max_bytes} bytes at most")
        t = data if data is not None else stats
        stream = torch.cuda.current_stream(t.device).cuda_stream
        self._check(self._lib.hymeta_p2p_all_reduce(self._comm, data.data_ptr() if data is not None else None,
                                                    data.numel() if data is not None else 0,
                                                    stats.data_ptr() if stats is not None else None,
                                                    stats.numel() if stats is not None else 0, float(stats_scale), stream),
                    "hymeta_p2p_all_reduce")

    def supports_partials(self, partials) -> bool:
        M, N = partials.shape
        return N % 8 == 0 and 0 < M * N * 2 <= self.max_bytes

    def all_reduce_partials(self, partials) -> torch.Tensor:
        """The sum over the ranks of a K-sliced linear's result that is still in its partial sums
        (``hip_ops.PartialSums``): the exchange adds the slices as it reads them, so the linear's own reduce launch never
        runs.  Returns bf16 ``[M, N]``."""
        partials.check_live()
        M, N = partials.shape
        out = torch.empty(M, N, dtype=torch.bfloat16, device=partials.device)
        stream = torch.cuda.current_stream(partials.device).cuda_stream
        self._check(self._lib.hymeta_p2p_all_reduce_partials(self._comm, self._C.byref(partials.desc), out.data_ptr(), stream),
                    "hymeta_p2p_all_reduce_partials")
        return out

    def set_wait_limit(self, seconds: float) -> None:
        self._check(self._lib.hymeta_p2p_set_wait_limit(self._comm, max(1, int(seconds * 1e8))), "hymeta_p2p_set_wait_limit")

    def timeouts(self) -> int:
        """Waits for a peer that ran into their time limit so far (synchronises the device)."""
        n = self._C.c_uint32(0)
        with torch.cuda.device(self.device):
            self._check(self._lib.hymeta_p2p_timeouts(self._comm, self._C.byref(n)), "hymeta_p2p_timeouts")
        return int(n.value)

    def close(self) -> None:
        if self._comm:
            if self.group is not None:
                torch.cuda.synchronize(self.device)
                dist.barrier(group=self.group)      # no peer writes into a buffer that is going away
            with torch.cuda.device(self.device):
                self._lib.hymeta_p2p_destroy(self._comm)
            self._comm = None


_CONCURRENT: dict = {}


def _streams_serialise(a: "torch.cuda.Stream", b: "torch.cuda.Stream", device: torch.device) -> bool:
    """Whether a kernel on ``b`` has to wait for the kernel launched before it on ``a`` (the two share a hardware queue):
    a two-rank exchange with a 20 ms limit, rank 0 on ``a``, rank 1 on ``b``; every kernel ends at its limit."""
    comms = OneShotAllReduce.in_process(2, device, 1 << 12)
    try:
        for c in comms:
            c.set_wait_limit(0.02)
        xs = [torch.zeros(64, dtype=torch.bfloat16, device=device) for _ in range(2)]
        torch.cuda.synchronize(device)
        for c, st, x in zip(comms, (a, b), xs):
            with torch.cuda.stream(st):
                c.all_reduce(data=x)
        torch.cuda.synchronize(device)
        return comms[0].timeouts() > 0
    finally:
        for c in comms:
            c.close()


def concurrent_streams(count: int, device: torch.device, candidates: int = 16) -> list:
    """``count`` streams of this process whose kernels run side by side.  A process has few hardware queues (4), a
    stream gets one when it is first used and keeps it, and kernels of streams that share a queue run in order -- fatal
    for ranks that wait for each other inside a kernel.  Which streams share is not documented, so this measures it:
    candidates from torch's pool are tried pairwise (``_streams_serialise``) until ``count`` of them are mutually
    independent.  The answer holds for the life of the process and is remembered."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    chosen = _CONCURRENT.setdefault(key, [])
    tried = 0
    while len(chosen) < count and tried < candidates:
        st = torch.cuda.Stream(device)
        tried += 1
        if all(st.cuda_stream != c.cuda_stream and not _streams_serialise(c, st, device) for c in chosen):
            chosen.append(st)
    if len(chosen) < count:
        raise RuntimeError(f"found {len(chosen)} streams that run side by side, {count} wanted (hardware queues of a process: 4)")
    return chosen[:count]


class InProcessGroup:
    """A tensor-parallel group whose ranks all live in ONE process, each driving its own stream (and, to keep a host
    wait of one rank from stalling the others, its own thread) of one GPU: (rank, size) plus the rank's end of a
    one-shot exchange.  Accepted wherever the layers take a ``tp_group``; every all-reduce goes through the exchange
    (there is no backend behind it, so a message it does not support raises).  For tests and rehearsals of the
    sharded model on one GPU."""

    def __init__(self, rank: int, size: int, comm: OneShotAllReduce, stream: "torch.cuda.Stream"):
        self.rank, self.size, self.comm, self.stream = rank, size, comm, stream

    @classmethod
    def make(cls, size: int, device: torch.device, max_bytes: int = ONE_SHOT_MAX_BYTES, wait_seconds: float = 2.0) -> list:
        """``size`` <= 4 ranks (the hardware queues of a process).  Run rank ``r`` on ``groups[r].stream``: streams that
        were measured not to share a hardware queue (``concurrent_streams``)."""
        if size > 4:
            raise ValueError("ranks that share a process need a hardware queue each: 4 at most")
        streams = concurrent_streams(size, device)
        comms = OneShotAllReduce.in_process(size, device, max_bytes)
        for c in comms:
            c.set_wait_limit(wait_seconds)
        return [cls(r, size, c, st) for r, (c, st) in enumerate(zip(comms, streams))]


_ONE_SHOT: dict = {}


class OneShotTimeout(RuntimeError):
    """A wait of the one-shot exchange for a peer ran into its limit: the step that contains it holds NaN rows."""


def raise_on_exchange_timeout(group=None) -> None:
    """The serving path's check of the one-shot exchange, made where the host waits for the device anyway (before a
    step's graph is replayed, at the start of an eager forward): if a wait of ``group``'s exchange (of every group's
    without one) has run into its limit since the last look, the exchange is switched off for that group -- after a
    timeout the ranks' epochs may have drifted apart and it must not be used again; the group's collectives go back to
    its backend -- and ``OneShotTimeout`` is raised.  The kernel has written NaN over the rows it could not sum, so the
    step that missed a peer cannot have produced plausible tokens in the meantime.  The raise is local: under an engine
    it ends the worker, which ends the job on every rank.  Costs one 4-byte read from the device per exchange."""
    if isinstance(group, InProcessGroup):
        n = group.comm.timeouts()
        if n:
            raise OneShotTimeout(f"one-shot all-reduce: {n} waits for a peer timed out on rank {group.rank} of {group.size}")
        return
    groups = [group] if group is not None else list(_ONE_SHOT)
    for g in groups:
        comm = _ONE_SHOT.get(g)
        if comm is None:
            continue
        n = comm.timeouts()
        if n:
            comm.group = None          # no barrier on the way out: the peers may be the ones that are gone
            _ONE_SHOT.pop(g, None)
            comm.close()
            raise OneShotTimeout(f"one-shot all-reduce: {n} waits for a peer timed out on rank {comm.rank} of {comm.world}; "
                                 "the group falls back to its backend")


def _exchange_of(group) -> Optional[OneShotAllReduce]:
    return group.comm if isinstance(group, InProcessGroup) else _ONE_SHOT.get(group)


def exchange_enabled(group) -> bool:
    """Whether ``group``'s small all-reduces go through a one-shot exchange."""
    return _exchange_of(group) is not None


def enable_one_shot_all_reduce(group: dist.ProcessGroup, device: torch.device, max_bytes: int = ONE_SHOT_MAX_BYTES) -> OneShotAllReduce:
    """From now on ``all_reduce_sum`` / ``all_reduce_variance`` over ``group`` send what fits through the one-shot
    exchange.  Collective over the group (every rank calls it)."""
    if group not in _ONE_SHOT:
        _ONE_SHOT[group] = OneShotAllReduce.for_group(group, device, max_bytes)
    return _ONE_SHOT[group]


def disable_one_shot_all_reduce(group: dist.ProcessGroup) -> None:
    comm = _ONE_SHOT.pop(group, None)
    if comm is not None:
        comm.close()


def one_shot_reduce_ref(parts, dtype):
    """What the exchange computes from the ranks' messages: the sum in rank order in fp32, rounded once."""
    acc = torch.zeros_like(parts[0], dtype=torch.float32)
    for p in parts:
        acc = acc + p.float()
    return acc.to(dtype)


def one_shot_self_check(group: dist.ProcessGroup, device: torch.device, rows=(1, 64, 256), hidden: int = 3584,
                        rounds: int = 3) -> Tuple[bool, str]:
    """Run the group's one-shot exchange on random messages of the decode step's shapes and compare, on every rank, with
    the rank-ordered sum of the all-gathered inputs (RCCL) -- bit for bit -- and with zero timeouts.  Returns the same
    answer on all ranks."""
    comm = _ONE_SHOT.get(group)
    if comm is None:
        return False, "not enabled"
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    why = ""
    on_device = dist.get_backend(group) == "nccl"

    def gather(t):       # gloo moves host tensors only
        mine = t if on_device else t.cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        return [p.to(device) for p in parts]

    try:
        for it in range(rounds):
            for r in rows:
                x = torch.randn(r, hidden, generator=gen).to(torch.bfloat16).to(device)
                st = torch.rand(r, 2, generator=gen).to(device)
                gx, gs = gather(x), gather(st)
                comm.all_reduce(stats=st)
                comm.all_reduce(data=x)
                both_x, both_s = gx[rank].clone(), gs[rank].clone()
                comm.all_reduce(data=both_x, stats=both_s)
                torch.cuda.synchronize(device)
                rx, rs = one_shot_reduce_ref(gx, torch.bfloat16), one_shot_reduce_ref(gs, torch.float32)
                if not (torch.equal(x, rx) and torch.equal(st, rs) and torch.equal(both_x, rx) and torch.equal(both_s, rs)):
                    why = why or f"rank {rank}: [{r}, {hidden}] round {it} differs from the rank-ordered sum"
        if comm.timeouts() != 0:
            why = why or f"rank {rank}: {comm.timeouts()} waits timed out"
    except Exception as exc:      # an error on one rank must not leave the others in a collective
        why = f"rank {rank}: {type(exc).__name__}: {exc}"
    ok = torch.tensor([0.0 if why else 1.0], device=device if on_device else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    good = bool(ok.item() == 1.0)
    return good, ("" if good else (why or "another rank failed the check"))


# A second process group over the same ranks that carries the collectives recorded into HIP graphs.  The backend's
# watchdog thread polls the completion events of EAGER collectives; those events sit on the group's communication
# stream, and a poll of one while that stream is being captured is refused by the runtime and takes the process down.
# With the captured calls on a group of their own, the stream that is being captured never holds an eager event.
_CAPTURE_GROUP: dict = {}


def set_capture_group(group: dist.ProcessGroup, capture_group: Optional[dist.ProcessGroup]) -> None:
    """Collectives over ``group`` that are issued while the current stream is being captured go over ``capture_group``
    instead (same ranks, same order; ``None`` removes the entry).  The caller runs one eager collective on
    ``capture_group`` first, well before any capture, so that its communicator exists."""
    if capture_group is None:
        _CAPTURE_GROUP.pop(group, None)
    else:
        _CAPTURE_GROUP[group] = capture_group


def backend_group(group):
    """The process group a collective over ``group`` is sent on right now (see ``set_capture_group``)."""
    alias = _CAPTURE_GROUP.get(group)
    if alias is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return alias
    return group


def defers_into_all_reduce(group) -> bool:
    """Whether a row-parallel linear in front of ``all_reduce_sum(…, group)`` may hand over its K slices' partial sums
    (``defer=True``): the group's one-shot exchange adds them while it reads its message."""
    return _active(group) and _exchange_of(group) is not None


def all_reduce_sum(x, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Sum over the tensor-parallel group, in place (the reduction behind ``RowParallelLinear``:
    ``out_proj`` / ``o_proj`` / ``down_proj``).  No-op for a single rank or without a group.
    ``x`` may be the ``hip_ops.PartialSums`` of a K-sliced linear (see ``defers_into_all_reduce``): the exchange takes
    them as they are; any other path first runs the linear's reduce launch."""
    if not isinstance(x, torch.Tensor):        # hip_ops.PartialSums
        comm = _exchange_of(group) if _active(group) else None
        if comm is not None and comm.supports_partials(x):
            return comm.all_reduce_partials(x)
        x = x.tensor()
    if _active(group):
        comm = _exchange_of(group)
        if comm is not None and (comm.supports(x) or isinstance(group, InProcessGroup)):
            comm.all_reduce(data=x)
        else:
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=backend_group(group))
    return x


def all_reduce_variance(stats: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """``stats [T, 2]`` = per-rank mean of squares of the two branches over the local columns;
    returns the mean over the full hidden size (sum over ranks / tp), one message for both norms
    where the reference sends two ``[T, 1]`` ones (``modeling_hymeta.py:158-160``)."""
    if _active(group):
        comm = _exchange_of(group)
        size = tp_rank_and_size(group)[1]
        if comm is not None and (comm.supports(None, stats) or isinstance(group, InProcessGroup)):
            comm.all_reduce(stats=stats, stats_scale=1.0 / size)      # the division rides in the exchange: no launch of its own
        else:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=backend_group(group))
            stats /= size
    return stats


def all_gather_rows(mine: torch.Tensor, out: torch.Tensor, group, rank: int) -> torch.Tensor:
    """``out [W, *mine.shape]`` (fp32, contiguous) = every rank's ``mine``, rank-major.  Small messages go through the
    group's one-shot exchange as a sum of the ranks' own rows and zeros elsewhere (exact: x + 0), so that a captured step
    holds no call into the backend; otherwise an all-gather on the group's backend."""
    comm = _exchange_of(group) if _active(group) else None
    if comm is not None and out.dtype == torch.float32 and out.is_contiguous() and comm.supports(None, out):
        out.zero_()
        out[rank].copy_(mine)
        comm.all_reduce(stats=out)
    elif _active(group):
        dist.all_gather_into_tensor(out.view(-1, *mine.shape[1:]) if mine.dim() > 1 else out.view(-1), mine, group=backend_group(group))
    else:
        out[0].copy_(mine)
    return out


def tp_rank_and_size(group: Optional[dist.ProcessGroup] = None) -> Tuple[int, int]:
    """Rank and size of the tensor-parallel group; ``(0, 1)`` without one (see ``_active``)."""
    if isinstance(group, InProcessGroup):
        return group.rank, group.size
    if group is not None and dist.is_available() and dist.is_initialized():
        if EMULATED_TP is not None:
            return EMULATED_TP
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def send_to_stage(t: torch.Tensor, group: dist.ProcessGroup, dst: int) -> None:
    """One tensor to rank ``dst`` of ``group`` (a pipeline hand-off).  A gloo group gets a host copy: its transfers of
    device tensors between processes that share a GPU can hang."""
    if dist.get_backend(group) == "gloo":
        t = t.detach().to("cpu")           # synchronous copy
    dist.send(t.contiguous(), group=group, group_dst=dst)


def recv_from_stage(like: torch.Tensor, group: dist.ProcessGroup, src: int) -> torch.Tensor:
    """The counterpart of ``send_to_stage``: a tensor of ``like``'s shape, dtype and device from rank ``src``."""
    if dist.get_backend(group) == "gloo":
        host = torch.empty(like.shape, dtype=like.dtype, device="cpu")
        dist.recv(host, group=group, group_src=src)
        return host.to(like.device)
    out = torch.empty_like(like)
    dist.recv(out, group=group, group_src=src)
    return out
