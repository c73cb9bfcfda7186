#!/usr/bin/env python3
"""GPTQ-4 GEMM time against the number of 128-column tiles (needs a GPU): how a full machine, one
tile per compute unit, compares with a few tiles, and what the tiles beyond one round cost."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from tune_quant import bench  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

dev = torch.device("cuda")
group = 128
M = int(os.environ.get("M", "64"))
K = int(os.environ.get("K", "3584"))
for tiles in (32, 64, 128, 192, 256, 264, 296, 384, 512):
    N = tiles * 128
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // group, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(K // group, N, device=dev) * 0.01).to(torch.float16)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t0 = bench(lambda: hip_ops.gptq_gemm(x, qw, qz, sc, 4, group, out=out))
    t1 = bench(lambda: hip_ops.gptq_gemm(x, qw, qz, sc, 4, group, out=out, split_k=1))
    print(f"K={K} tiles={tiles:4d} N={N:6d}: library's choice {t0:7.1f} us ({K * N / 2 / t0 / 1e3:7.1f} GB/s)   one slice {t1:7.1f} us", flush=True)
