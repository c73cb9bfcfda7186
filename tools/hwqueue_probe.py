#!/usr/bin/env python3
"""Which pairs of streams of this process can run kernels side by side?  For every pair (a, b) of a set of streams: a
two-rank one-shot exchange with rank 0 on a and rank 1 on b and a 20 ms wait limit.  Streams that share a hardware queue
run their kernels in order, so rank 0's kernel waits for a peer that cannot start: it times out (every kernel ends at its
limit).  Prints the matrix of pairs that timed out for 8 streams from torch's pool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta.parallel import OneShotAllReduce  # noqa: E402


def probe(streams, label):
    dev = torch.device("cuda", 0)
    n = len(streams)
    print(f"{label}: {n} streams; X = the pair serialises (rank 0 on the row's stream, rank 1 on the column's)")
    for i in range(n):
        row = ""
        for j in range(n):
            if i == j:
                row += " ."
                continue
            comms = OneShotAllReduce.in_process(2, dev, 1 << 16)
            for c in comms:
                c.set_wait_limit(0.02)
            xs = [torch.ones(64, dtype=torch.bfloat16, device=dev) for _ in range(2)]
            torch.cuda.synchronize()
            for c, st, x in zip(comms, (streams[i], streams[j]), xs):
                with torch.cuda.stream(st):
                    c.all_reduce(data=x)
            torch.cuda.synchronize()
            row += " X" if comms[0].timeouts() else " -"
            for c in comms:
                c.close()
        print(row, flush=True)


def main():
    torch.cuda.set_device(0)
    pool = [torch.cuda.Stream() for _ in range(8)]
    probe(pool, "torch pool streams")


if __name__ == "__main__":
    main()
