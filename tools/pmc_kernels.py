#!/usr/bin/env python3
"""Run the kernels whose matrix-core / vector utilisation is reported (for rocprofv3 --pmc passes), a few launches each:
the sliding-window prefill of one 32k prompt (7B geometry), the 4-bit GPTQ gate_up linear at 64 rows, the bf16
weight-streaming gate_up linear at 64 rows, the 4-bit streaming kernel on the qkv shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

dev = torch.device("cuda")
H, HKV, D, L, W = 28, 4, 128, 32768, 4096
q = torch.randn(L, H, D, device=dev).to(torch.bfloat16)
k = torch.randn(L, HKV, D, device=dev).to(torch.bfloat16)
v = torch.randn(L, HKV, D, device=dev).to(torch.bfloat16)
cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
for _ in range(3):
    hip_ops.meta_swa_prefill(q, cu, L, key=k, value=v, cu_seqlens_k=cu, window_left=W)
torch.cuda.synchronize()
del q, k, v

M, G = 64, 128
x = torch.randn(M, 3584, device=dev).to(torch.bfloat16)
for K, N in ((3584, 37888), (3584, 4608)):
    ws = []
    for _ in range(4):
        ws.append((torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, device=dev),
                   torch.randint(-2 ** 31, 2 ** 31 - 1, (K // G, N // 8), dtype=torch.int32, device=dev),
                   (torch.rand(K // G, N, device=dev) * (0.25 * K ** -0.5)).to(torch.float16)))
    for _ in range(2):
        for qw, qz, sc in ws:
            hip_ops.gptq_gemm(x, qw, qz, sc, 4, G)
    torch.cuda.synchronize()
    del ws
wb = [(torch.randn(37888, 3584, device=dev) * 3584 ** -0.5).to(torch.bfloat16) for _ in range(4)]
for _ in range(2):
    for w in wb:
        hip_ops.bf16_gemm(x, w)
torch.cuda.synchronize()
