#!/usr/bin/env python3
"""The one-shot all-reduce between PROCESSES: N ranks (default 2), each its own process with its own exchange buffer,
mapped into the others through HIP IPC handles that travel over a gloo group -- all on GPU 0 when the box has one GPU,
else one GPU per rank.  Every rank checks its results bit for bit against the rank-ordered sum, replays the exchange
from a graph and times it.  The parent never touches the GPU: it only starts the ranks and collects their exit codes.

    python3 tools/p2p_two_process.py [ranks]
"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main():
    sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
    import torch
    import torch.distributed as dist

    from vllm_hymeta import parallel

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.group.WORLD
    comm = parallel.enable_one_shot_all_reduce(group, dev)
    comm.set_wait_limit(2.0)
    ok, why = parallel.one_shot_self_check(group, dev)
    print(f"[rank {rank}] self-check on {dev}: {'ok' if ok else 'FAILED: ' + why}", flush=True)
    if ok:
        # the layer's three messages from one graph, timed
        x = torch.zeros(64, 3584, dtype=torch.bfloat16, device=dev)      # sums of sums: zeros stay finite over the replays
        st = torch.zeros(64, 2, device=dev)
        graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        torch.cuda.synchronize()
        dist.barrier()
        with torch.cuda.graph(graph, stream=side):
            comm.all_reduce(stats=st)
            comm.all_reduce(data=x)
            comm.all_reduce(data=x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        dist.barrier()
        graph.replay()
        torch.cuda.synchronize()
        dist.barrier()
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ok = comm.timeouts() == 0
        print(f"[rank {rank}] {e0.elapsed_time(e1) / reps / 3 * 1e3:.1f} us per exchange from a graph (3 per replay), "
              f"timeouts {comm.timeouts()}", flush=True)
    parallel.disable_one_shot_all_reduce(group)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), P2P_RANK_PROCESS="1")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)], env=env))
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=150))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(-9)
    print("exit codes", codes, flush=True)
    sys.exit(0 if all(c == 0 for c in codes) else 1)


if __name__ == "__main__":
    rank_main() if os.environ.get("P2P_RANK_PROCESS") == "1" else main()
