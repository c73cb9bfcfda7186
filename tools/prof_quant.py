#!/usr/bin/env python3
"""Run one GPTQ-4 GEMM shape a few times (for rocprofv3 passes).  usage: prof_quant.py M K N"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda")
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
sc = (torch.rand(K // 128, N, device=dev) * 0.01).to(torch.float16)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    hip_ops.gptq_gemm(x, qw, qz, sc, 4, 128, out=out)
torch.cuda.synchronize()
