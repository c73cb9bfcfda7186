#!/usr/bin/env python3
"""Time the dequant-GEMM kernels at Hymeta-7B projection shapes (needs a GPU).
Reports weight-stream GB/s (decode regime) and TFLOP/s (prefill regime), with torch bf16 matmul beside."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402


def bench(fn, iters=20):
    """GPU time per call: the calls are captured into one graph so that host launch cost (about
    13 us per ctypes call) does not hide the kernels' own duration."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda")
    shapes = [("qkv", 3584, 4608), ("o", 3584, 3584), ("gate_up", 3584, 37888), ("down", 18944, 3584)]
    Ms = [int(m) for m in os.environ.get("MS", "1,16,64,256,2048,8192").split(",")]
    group = int(os.environ.get("GROUP", "128"))
    sweep = [int(v) for v in os.environ.get("SPLITS", "").split(",") if v]
    only = os.environ.get("ONLY", "")
    for name, K, N in shapes:
        if only and name not in only.split(","):
            continue
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // group, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(K // group, N, device=dev) * 0.01).to(torch.float16)
        w8 = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev)
        ws = torch.rand(N, device=dev) * 0.01
        wb = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        for M in Ms:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            t4 = bench(lambda: hip_ops.gptq_gemm(x, qw, qz, sc, 4, group, out=out))
            for sk in sweep:
                ts = bench(lambda: hip_ops.gptq_gemm(x, qw, qz, sc, 4, group, out=out, split_k=sk))
                print(f"    {name} M={M} split_k={sk}: {ts:8.1f} us", flush=True)
            t8 = bench(lambda: hip_ops.w8a8_gemm(x, w8, ws, out=out))
            tb = bench(lambda: torch.matmul(x, wb.t(), out=out))
            fl = 2.0 * M * K * N
            b4 = K * N / 2 + (M * K + M * N) * 2
            b8 = K * N + (M * K + M * N) * 2
            bb = K * N * 2 + (M * K + M * N) * 2
            print(f"{name:8s} M={M:5d} K={K:5d} N={N:5d} | gptq4 {t4:8.1f} us {b4 / t4 / 1e3:7.1f} GB/s {fl / t4 / 1e6:7.1f} TF"
                  f" | w8a8 {t8:8.1f} us {b8 / t8 / 1e3:7.1f} GB/s {fl / t8 / 1e6:7.1f} TF"
                  f" | bf16 {tb:8.1f} us {bb / tb / 1e3:7.1f} GB/s {fl / tb / 1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
