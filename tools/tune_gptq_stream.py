#!/usr/bin/env python3
"""Check and time the 4-bit GPTQ linear at the 7B shapes on cold weights (needs a GPU).  Env ROWS, and the library's
HYMETA_GPTQ_STREAM_FROM (rows from which the weight-streaming kernel runs; 1000 = the older kernels); SHAPES (a comma
list of qkv, o, gate_up, down) keeps some of the matrices."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tune_quant import bench  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

dev = torch.device("cuda")
SHAPES = [("qkv", 3584, 4608), ("o", 3584, 3584), ("gate_up", 3584, 37888), ("down", 18944, 3584)]
if os.environ.get("SHAPES"):
    SHAPES = [sh for sh in SHAPES if sh[0] in os.environ["SHAPES"].split(",")]
ROWS = [int(r) for r in os.environ.get("ROWS", "1,16,64").split(",")]
G = 128
for M in ROWS:
    for name, K, N in SHAPES:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        copies = max(2, min(40, int(1.2e9 // (N * K // 2))))
        ws = []
        for _ in range(copies):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // G, N // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand(K // G, N, device=dev) * (0.25 * K ** -0.5)).to(torch.float16)
            ws.append((qw, qz, sc))
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        qw, qz, sc = ws[0]
        w = hip_ops.gptq_dequant(qw, qz, sc, 4, G).float()            # [K, N]
        ref = x.float() @ w
        got = hip_ops.gptq_gemm(x, qw, qz, sc, 4, G).float()
        got2 = hip_ops.gptq_gemm(x, qw, qz, sc, 4, G).float()
        err = float((got - ref).abs().max() / ref.abs().max())

        def mine():
            for qw_, qz_, sc_ in ws:
                hip_ops.gptq_gemm(x, qw_, qz_, sc_, 4, G, out=out)

        t = bench(mine, iters=2) / copies
        nbytes = N * K // 2
        print(f"M={M:3d} {name:8s} gptq4 {t:6.1f} us ({nbytes / t / 1e6:5.2f} TB/s) max rel err {err:.2e} deterministic {bool(torch.equal(got, got2))}",
              flush=True)
        assert err < 2e-2 or os.environ.get("NO_CHECK"), "wrong result"
        del ws
