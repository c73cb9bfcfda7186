"""8-bit, group-64 GPTQ linears (the 7B recipe of the reference) on the 7B shapes: time per launch inside a replayed graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vllm-hymeta_amd"))
sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vllm_hymeta import hip_ops  # noqa: E402
from time_tall_gemm import SHAPES, timed  # noqa: E402

bits = int(os.environ.get("BITS", "8"))
group = int(os.environ.get("GROUP", "64"))
pack = 32 // bits
rows = [int(r) for r in sys.argv[1:]] or [64, 256]
g = torch.Generator(device="cpu").manual_seed(3)
for name, (N, K) in SHAPES.items():
    qws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (K // pack, N), generator=g, dtype=torch.int64).to(torch.int32).cuda() for _ in range(3)]
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // group, N // pack), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    sc = (torch.rand(K // group, N, generator=g) * 0.01 + 0.001).to(torch.float16).cuda()
    for M in rows:
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
        st = {"i": 0}

        def ours():
            st["i"] = (st["i"] + 1) % 3
            hip_ops.gptq_gemm(x, qws[st["i"]], qz, sc, bits, group)

        t = timed(ours)
        print(f"{name} M={M} bits={bits} g={group}: {t:.1f} us = {N * K * bits / 8 / 1e6 / t:.2f} TB/s of weights", flush=True)
