#!/usr/bin/env python3
"""Time the two whole-matrix GPTQ expansions ([K, N] and the transposed [N, K]) at the 7B shapes (needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from tune_quant import bench  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

dev = torch.device("cuda")
G = 128
for name, K, N in [("qkv", 3584, 4608), ("o", 3584, 3584), ("gate_up", 3584, 37888), ("down", 18944, 3584)]:
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // G, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(K // G, N, device=dev) * (0.25 * K ** -0.5)).to(torch.float16)
    a = torch.empty(K, N, dtype=torch.bfloat16, device=dev)
    b = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    ta = bench(lambda: hip_ops.gptq_dequant(qw, qz, sc, 4, G, out=a), iters=10)
    tb = bench(lambda: hip_ops.gptq_dequant(qw, qz, sc, 4, G, out=b, transposed=True), iters=10)
    assert torch.equal(a.t(), b)
    nbytes = K * N * 2.5
    print(f"{name:8s} [K,N] {ta:7.1f} us ({nbytes / ta / 1e6:5.2f} TB/s)   [N,K] {tb:7.1f} us ({nbytes / tb / 1e6:5.2f} TB/s)", flush=True)
