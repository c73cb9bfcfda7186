#!/usr/bin/env python3
"""Check and time the decode-sized bf16 linear against the library GEMM at the 7B shapes (needs a GPU).
Env: HYMETA_STREAM_NF / HYMETA_STREAM_WGS override the streaming kernel's tile width / workgroup count."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from tune_quant import bench  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

dev = torch.device("cuda")
SHAPES = [("qkv", 3584, 4608), ("o", 3584, 3584), ("gate_up", 3584, 37888), ("down", 18944, 3584)]
ROWS = [int(r) for r in os.environ.get("ROWS", "1,16,64").split(",")]
if os.environ.get("SHAPES"):
    SHAPES = [sh for sh in SHAPES if sh[0] in os.environ["SHAPES"].split(",")]
for M in ROWS:
    for name, K, N in SHAPES:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        # cold weights, as inside a model step: rotate over copies that together exceed the 256 MB
        # last-level cache several times
        copies = int(os.environ.get("COPIES", 0)) or max(2, min(40, int(1.2e9 // (N * K * 2))))
        ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(copies)]
        bias = torch.randn(N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ref = (x.float() @ ws[0].float().t() + bias.float())
        got = hip_ops.bf16_gemm(x, ws[0], bias).float()
        got2 = hip_ops.bf16_gemm(x, ws[0], bias).float()
        err = float((got - ref).abs().max() / ref.abs().max())
        same = bool(torch.equal(got, got2))

        def lib():
            for w in ws:
                torch.nn.functional.linear(x, w)

        def mine():
            for w in ws:
                hip_ops.bf16_gemm(x, w, out=out)

        t_lib = bench(lib, iters=2) / copies
        t = bench(mine, iters=2) / copies
        print(f"M={M:3d} {name:8s} library {t_lib:7.1f} us ({N * K * 2 / t_lib / 1e6:5.2f} TB/s) | stream {t:6.1f} us "
              f"({N * K * 2 / t / 1e6:5.2f} TB/s) max rel err {err:.2e} deterministic {same}", flush=True)
        assert err < 1e-2, "wrong result"
        del ws
