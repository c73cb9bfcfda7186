#!/usr/bin/env python3
"""One MoE block at the 70B TP=4 shapes for a kernel trace: python3 tools/prof_moe.py [tokens]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda")
E, H, inter = 16, 3584, 18944 // 4
w13 = (torch.randn(E, 2 * inter, H, device=dev) * H ** -0.5).to(torch.bfloat16)
w2 = (torch.randn(E, H, inter, device=dev) * inter ** -0.5).to(torch.bfloat16)
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
gate = (torch.randn(E, H, device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty(T, H, device=dev, dtype=torch.bfloat16)
for _ in range(10):
    hip_ops.moe_forward(x, None, w13, w2, out=out, gate_weight=gate)
torch.cuda.synchronize()
