#!/usr/bin/env python3
"""Run one decode-sized bf16 linear a few times on cold weights (for rocprofv3 trace / counter passes).
usage: prof_gemm.py <qkv|o|gate_up|down> [rows] [lib]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

SHAPES = {"qkv": (3584, 4608), "o": (3584, 3584), "gate_up": (3584, 37888), "down": (18944, 3584)}
K, N = SHAPES[sys.argv[1]]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = len(sys.argv) > 3 and sys.argv[3] == "lib"
dev = torch.device("cuda")
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(6)]
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    for w in ws:
        if lib:
            torch.nn.functional.linear(x, w)
        else:
            hip_ops.bf16_gemm(x, w, out=out)
torch.cuda.synchronize()
