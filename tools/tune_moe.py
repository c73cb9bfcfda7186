#!/usr/bin/env python3
"""Time the sparse MoE block at Hymeta-70B TP=4 shapes (needs a GPU): 16 experts, top-1,
hidden 3584, expert intermediate 18944 / 4."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from tune_quant import bench  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402


def main():
    dev = torch.device("cuda")
    E, H, inter = 16, 3584, 18944 // 4
    fmt = os.environ.get("FMT", "bf16")            # bf16 | w8a8 | gptq4
    kw = {}
    if fmt == "bf16":
        w13 = (torch.randn(E, 2 * inter, H, device=dev) * H ** -0.5).to(torch.bfloat16)
        w2 = (torch.randn(E, H, inter, device=dev) * inter ** -0.5).to(torch.bfloat16)
        wbytes = (w13.numel() + w2.numel()) * 2
    elif fmt == "w8a8":
        w13 = torch.randint(-127, 128, (E, 2 * inter, H), dtype=torch.int8, device=dev)
        w2 = torch.randint(-127, 128, (E, H, inter), dtype=torch.int8, device=dev)
        kw = dict(w13_scale=torch.rand(E, 2 * inter, device=dev) * 0.004 + 0.0005, w2_scale=torch.rand(E, H, device=dev) * 0.004 + 0.0005)
        wbytes = w13.numel() + w2.numel()
    else:
        def stack(K, N):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (E, K // 8, N), dtype=torch.int32, device=dev)
            qz = (torch.randint(-2 ** 31, 2 ** 31 - 1, (E, K // 128, N // 8), dtype=torch.int32, device=dev) & 0x11111111) + 0x66666666
            sc = ((torch.rand(E, K // 128, N, device=dev) * 0.5 + 0.75) * 0.217 * K ** -0.5).to(torch.float16)
            return hip_ops.GptqExperts(qw, qz, sc, 4, 128)
        w13, w2 = stack(H, 2 * inter), stack(inter, H)
        wbytes = 3 * inter * H * E // 2
    for T in (1, 16, 64, 256, 512, 1024, 2048, 4096, 8192):
        x = torch.randn(T, H, device=dev).to(torch.bfloat16)
        gate = (torch.randn(E, H, device=dev) * 0.5).to(torch.bfloat16)
        out = torch.empty(T, H, device=dev, dtype=torch.bfloat16)
        # the block as the model runs it: router included (inside the routing launch up to 256 tokens)
        us = bench(lambda: hip_ops.moe_forward(x, None, w13, w2, out=out, gate_weight=gate, **kw), iters=5)
        touched = min(T, E) / E * wbytes if T < 64 else wbytes
        flops = 2.0 * T * 3 * inter * H
        print(f"moe {fmt} T={T:5d}: {us:9.1f} us   ~{touched / us / 1e3:7.1f} GB/s weights   {flops / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
