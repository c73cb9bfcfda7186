#!/usr/bin/env python3
"""Time the two prefill kernels at BASELINE geometries (needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops, _C  # noqa: E402

D = 128


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def swa(lens, H, Hkv, M, wl):
    dev = torch.device("cuda")
    T = sum(lens)
    q = torch.randn(T, H, D, device=dev).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, device=dev).to(torch.bfloat16)
    k2 = torch.randn(M, Hkv, D, device=dev).to(torch.bfloat16) if M else None
    v2 = torch.randn(M, Hkv, D, device=dev).to(torch.bfloat16) if M else None
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    out = torch.empty(T, H, D, dtype=torch.bfloat16, device=dev)
    us = bench(lambda: hip_ops.meta_swa_prefill(q, cu, max(lens), key=k, value=v, cu_seqlens_k=cu, meta_key=k2,
                                               meta_value=v2, window_left=wl, out=out))
    flops = 0
    for L in lens:
        for p in range(0, L, 1024):   # coarse sum of visible keys
            n = min(1024, L - p)
            for pp in (p, p + n - 1):
                pass
        import numpy as np
        pos = np.arange(L)
        vis = np.minimum(pos + 1, wl + 1) if wl >= 0 else pos + 1
        flops += 4.0 * D * H * float((vis + M).sum())
    print(f"swa_prefill lens={lens} H={H} Hkv={Hkv} M={M} wl={wl}: {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


def swa_paged(chunk, ctx, H, Hkv, M, wl, bs=256):
    """One chunk of `chunk` queries of a sequence that already holds ctx - chunk keys in a paged cache
    (the chunked-prefill call of the 70B configuration)."""
    dev = torch.device("cuda")
    nblk = (ctx + bs - 1) // bs
    kv = torch.randn(2, nblk + 3, bs, Hkv, D, device=dev).to(torch.bfloat16)
    bt = torch.randperm(nblk + 3)[:nblk].to(torch.int32).view(1, nblk).to(dev)
    q = torch.randn(chunk, H, D, device=dev).to(torch.bfloat16)
    k2 = torch.randn(M, Hkv, D, device=dev).to(torch.bfloat16) if M else None
    v2 = torch.randn(M, Hkv, D, device=dev).to(torch.bfloat16) if M else None
    cu = torch.tensor([0, chunk], dtype=torch.int32, device=dev)
    sl = torch.tensor([ctx], dtype=torch.int32, device=dev)
    out = torch.empty(chunk, H, D, dtype=torch.bfloat16, device=dev)
    us = bench(lambda: hip_ops.meta_swa_prefill(q, cu, chunk, key_cache=kv[0], value_cache=kv[1], block_tables=bt,
                                               seq_lens_k=sl, meta_key=k2, meta_value=v2, window_left=wl, out=out,
                                               max_seqlen_k=ctx, num_splits=int(os.environ.get("NSPLIT", "0"))))
    import numpy as np
    pos = np.arange(ctx - chunk, ctx)
    vis = np.minimum(pos + 1, wl + 1) if wl >= 0 else pos + 1
    flops = 4.0 * D * H * float((vis + M).sum())
    print(f"swa_prefill PAGED chunk={chunk} ctx={ctx} H={H} Hkv={Hkv} M={M} wl={wl} bs={bs}: {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


def gla(lens, H, Hkv, nseg=0):
    dev = torch.device("cuda")
    T = sum(lens)
    q = torch.randn(T, H, D, device=dev).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, device=dev).to(torch.bfloat16)
    gk = torch.randn(T, Hkv, D, device=dev).to(torch.bfloat16)
    SH = Hkv if os.environ.get("STATE", "kv_head") == "kv_head" else H      # heads in the state tensor
    st = torch.zeros(len(lens), SH, D, D, dtype=torch.bfloat16, device=dev)
    slot = torch.arange(len(lens), dtype=torch.int32, device=dev)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    ctx = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    out = torch.empty(T, H, D, dtype=torch.bfloat16, device=dev)
    us = bench(lambda: hip_ops.gla_prefill_varlen(q, k, v, gk, st, slot, cu, max(lens), context_lens=ctx, out=out,
                                                  gate_mode=_C.GATE_7B, num_segments=nseg))
    nbytes = T * (H * D * 2 * 2 + 3 * Hkv * D * 2) + len(lens) * H * D * D * 2
    print(f"gla_prefill lens={lens[0]}x{len(lens)} H={H} Hkv={Hkv} state heads {SH} nseg={nseg}: {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s  "
          f"{T / us:7.2f} tok/us", flush=True)


if __name__ == "__main__":
    which = os.environ.get("WHICH", "swa,gla")
    if "swa" in which and os.environ.get("ONLY32K"):
        swa([32768], 28, 4, 0, 4096)
        swa([32768], 7, 1, 128, -1)
    elif "swa" in which:
        swa([8192], 28, 4, 0, 4096)
        swa([32768], 28, 4, 0, 4096)
        swa([32768], 7, 1, 128, 4095)
        swa([32768], 7, 1, 128, -1)
        swa([2048] * 8, 28, 4, 0, 4096)
    if "chunks" in which:
        for chunk in (256, 512, 1024, 2048, 4096):
            swa_paged(chunk, 32768, 7, 1, 128, 4095)
        swa_paged(2048, 32768, 7, 1, 128, -1)
        swa_paged(2048, 32768, 28, 4, 0, 4096)
    if "paged" in which:
        swa_paged(8192, 32768, 7, 1, 128, 4095)
        swa_paged(2048, 32768, 7, 1, 128, 4095)
        swa_paged(8192, 32768, 7, 1, 128, -1)
        swa_paged(8192, 32768, 28, 4, 0, 4096)
        swa_paged(8192, 32768, 28, 4, 0, 4096, bs=16)
    if "gla" in which:
        for nseg in (1, 0, 9, 18, 27, 36, 64, 128):
            gla([32768], 28, 4, nseg)
        for nseg in (1, 0, 9, 18):
            gla([8192], 28, 4, nseg)
        for nseg in (1, 0, 2, 4):
            gla([8192] * 8, 28, 4, nseg)
        gla([2048] * 32, 28, 4, 1)
        gla([512] * 4, 28, 4, 0)
