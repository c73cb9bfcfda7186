#!/usr/bin/env python3
"""Library bf16 GEMM (F.linear) at the 7B shapes for prefill-sized and 129..256-row batches, cold weights, with and
without PyTorch's TunableOp (env PYTORCH_TUNABLEOP_ENABLED=1 set by the caller).  Prints us and TFLOP/s per shape.
NN=1 stores the weights [in, out] and runs ``matmul(x, w)`` (the layout a GPTQ checkpoint expands to) instead of
``linear(x, w[out, in])``."""
import os
import sys
import time

import torch

dev = torch.device("cuda")
SHAPES = [("qkv", 3584, 4608), ("o", 3584, 3584), ("gate_up", 3584, 37888), ("down", 18944, 3584)]
ROWS = [int(r) for r in os.environ.get("ROWS", "256,32768").split(",")]


def bench(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


t_start = time.time()
for M in ROWS:
    total = 0.0
    for name, K, N in SHAPES:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        copies = 1 if M > 4096 else max(2, min(16, int(2.0e9 // (N * K * 2))))
        nn_layout = os.environ.get("NN") == "1"
        ws = [torch.randn(*((K, N) if nn_layout else (N, K)), device=dev).to(torch.bfloat16) for _ in range(copies)]

        def run():
            for w in ws:
                if nn_layout:
                    torch.matmul(x, w)
                else:
                    torch.nn.functional.linear(x, w)

        t = bench(run, 3 if M > 4096 else 5) / copies
        total += t
        print(f"M={M:6d} {name:8s} {t:9.1f} us  {2.0 * M * K * N / t / 1e6:8.1f} TFLOP/s  {N * K * 2 / t / 1e6:6.2f} TB/s of weights", flush=True)
        del ws
    print(f"M={M:6d} layer total {total:9.1f} us   ({time.time() - t_start:.0f} s since start)", flush=True)
