#!/usr/bin/env python3
"""W8A8 linear: the fused int8-MFMA kernel against quantise -> library int8 GEMM -> scale, per row count (needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from tune_quant import bench  # noqa: E402
from vllm_hymeta import hip_ops  # noqa: E402

dev = torch.device("cuda")
for name, K, N in (("qkv", 3584, 4608), ("o", 3584, 3584), ("gate_up", 3584, 37888), ("down", 18944, 3584)):
    # cold weights, as inside a model step: rotate over copies that together exceed the last-level cache
    copies = max(2, min(40, int(1.2e9 // (N * K))))
    w8s = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(copies)]
    w8 = w8s[0]
    ws = (torch.rand(N, device=dev) * 0.01).float()
    for M in [int(m) for m in os.environ.get("MS", "32,48,64,96,128,256").split(",")]:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)

        def lib(w8=w8):
            xq, xs = hip_ops.w8a8_quantize(x)
            return hip_ops.w8a8_scale_out(torch._int_mm(xq, w8.t()), xs, ws)
        t_f = bench(lambda: [hip_ops.w8a8_gemm(x, w, ws) for w in w8s], iters=2) / copies
        try:
            t_l = bench(lambda: [lib(w) for w in w8s], iters=2) / copies
            same = torch.equal(lib(), hip_ops.w8a8_gemm(x, w8, ws))
        except Exception as e:   # the library GEMM has its own shape rules
            t_l, same = float("nan"), repr(e)[:60]
        print(f"{name:8s} M={M:4d}: fused {t_f:7.1f} us   library path {t_l:7.1f} us   identical={same}", flush=True)
