#!/usr/bin/env python3
"""Time the GLA decode op at the 7B decode geometry (needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
from vllm_hymeta import hip_ops, _C  # noqa: E402

H, HKV, D = 28, 4, 128


def bench(fn, iters=40):
    """Mean time of one call inside a replayed graph of ``iters`` calls (the host's launch cost stays outside)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("EAGER") == "1":       # a counter pass: plain launches, no timing
        return float("nan")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e3


def main():
    dev = torch.device("cuda")
    SH = HKV if os.environ.get("STATE", "kv_head") == "kv_head" else H     # heads in the state tensor
    for B in [int(x) for x in os.environ.get("BATCHES", "1,8,64,256").split(",")]:
        nstate = max(1, min(16, int(2e9 // (B * SH * D * D * 2)) + 1))   # cycle states: defeat the Infinity Cache
        states = [(torch.randn(B, SH, D, D, device=dev) * 0.1).to(torch.bfloat16) for _ in range(nstate)]
        q = torch.randn(B, H, D, device=dev).to(torch.bfloat16)
        k, v, gk = (torch.randn(B, HKV, D, device=dev).to(torch.bfloat16) for _ in range(3))
        w = torch.ones(D, dtype=torch.bfloat16, device=dev)
        slot = torch.arange(B, dtype=torch.int32, device=dev)
        out = torch.empty(B, H, D, dtype=torch.bfloat16, device=dev)
        nbytes = B * (2 * SH * D * D * 2 + H * D * 2 * 2 + 3 * HKV * D * 2)
        for name, kw in (("plain", {}), ("fused gate+norm", dict(gate_mode=_C.GATE_7B, out_norm_weight=w))):
            i = [0]

            def fn():
                st = states[i[0] % nstate]
                i[0] += 1
                hip_ops.gla_decode(q, k, v, gk, st, slot, out=out, **kw)
            us = bench(fn)
            print(f"state heads {SH:2d} B={B:4d} {name:16s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
