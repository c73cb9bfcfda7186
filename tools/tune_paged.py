#!/usr/bin/env python3
"""Paged sliding-window decode: time per launch against the number of KV splits (needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from tune_quant import bench  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

H, HKV, D, BLOCK, WINDOW = 28, 4, 128, 256, 4096
dev = torch.device("cuda")
for B in [int(b) for b in os.environ.get("BATCHES", "16,64,256").split(",")]:
    ctx = 8192
    bps = (ctx + BLOCK) // BLOCK
    nblk = B * bps
    kv = torch.randn(2, nblk, BLOCK, HKV, D, device=dev, dtype=torch.bfloat16)
    bt = torch.randperm(nblk).to(torch.int32).view(B, bps).to(dev)
    seq = torch.full((B,), ctx + 1, dtype=torch.int32, device=dev)
    q = torch.randn(B, H, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q)
    alg = B * (2 * HKV * D * 2 * (WINDOW + 1) + H * D * 4)
    auto = hip_ops.paged_decode_splits(B, HKV, ctx + 1, WINDOW, 0)
    for sp in (0, 1, 2, 3, 4, 5, 6, 8, 12, 16):
        us = bench(lambda: hip_ops.meta_swa_paged_decode(q, kv[0], kv[1], seq, bt, window_left=WINDOW, max_seqlen_k=ctx + 1,
                                                         num_splits=sp, out=out), iters=20)
        print(f"B={B:4d} splits={sp if sp else f'auto({auto})':>8}: {us:8.1f} us  {alg / us / 1e3:7.1f} GB/s", flush=True)
