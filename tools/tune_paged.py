#!/usr/bin/env python3
"""Sweep split counts of the paged decode op at the 7B decode geometry (needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

H, HKV, D, BS = 28, 4, 128, 256


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    dev = torch.device("cuda")
    ctx = int(os.environ.get("CTX", 8192))
    wl = int(os.environ.get("WL", 4096))
    for B in [int(x) for x in os.environ.get("BATCHES", "1,8,64,256").split(",")]:
        nb = (ctx + BS - 1) // BS + 1
        nblk = B * nb
        # several caches cycled so the working set exceeds the 256 MiB Infinity Cache
        ncache = max(1, min(8, int(2e9 // (nblk * BS * HKV * D * 2 * 2)) + 1))
        caches = [torch.randn(2, nblk, BS, HKV, D, device=dev, dtype=torch.bfloat16) for _ in range(ncache)]
        bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, nb)
        lens = torch.full((B,), ctx + 1, dtype=torch.int32, device=dev)
        q = torch.randn(B, H, D, device=dev, dtype=torch.bfloat16)
        out = torch.empty_like(q)
        keys = min(ctx + 1, wl + 1) if wl >= 0 else ctx + 1
        nbytes = B * (2 * HKV * D * 2 * keys + H * D * 4)
        auto = hip_ops.paged_decode_splits(B, HKV, ctx + 1, wl, 0)
        for splits in sorted(set([1, 2, 4, 8, 16, 32, auto])):
            i = [0]

            def fn():
                kv = caches[i[0] % ncache]
                i[0] += 1
                hip_ops.meta_swa_paged_decode(q, kv[0], kv[1], lens, bt, window_left=wl, max_seqlen_k=ctx + 1,
                                              num_splits=splits, out=out)
            us = bench(fn)
            print(f"B={B:4d} splits={splits:3d}{'*' if splits == auto else ' '} {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
