#!/usr/bin/env python3
"""Does the GPTQ GEMM's speed depend on the weight row stride (N * 4 bytes)?  Times M=64 at
neighbouring N (needs a GPU)."""
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
for K, Ns in ((3584, (37888, 38016, 38144, 38400, 4608, 4736, 4864)), (18944, (3584, 3712, 3840))):
    for N in Ns:
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // group, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(K // group, N, device=dev) * 0.01).to(torch.float16)
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for sk in (1, 2, 4, 8):
            t = bench(lambda: hip_ops.gptq_gemm(x, qw, qz, sc, 4, group, out=out, split_k=sk))
            print(f"K={K} N={N} (row stride {N * 4} B = {N * 4 / 4096:.3f} x 4 KiB) split_k={sk}: {t:7.1f} us  {K * N / 2 / t / 1e3:7.1f} GB/s", flush=True)
