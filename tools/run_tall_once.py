"""A few launches of the tall kernel on one 7B projection (for counter passes): python tools/run_tall_once.py gate_up 256"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vllm-hymeta_amd"))
from vllm_hymeta import hip_ops  # noqa: E402

SHAPES = {"qkv": (4608, 3584), "o": (3584, 3584), "gate_up": (37888, 3584), "down": (3584, 18944)}
name, M = sys.argv[1], int(sys.argv[2])
N, K = SHAPES[name]
g = torch.Generator(device="cpu").manual_seed(5)
ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda() for _ in range(3)]
x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
for i in range(9):
    hip_ops.bf16_gemm(x, ws[i % 3])
torch.cuda.synchronize()
