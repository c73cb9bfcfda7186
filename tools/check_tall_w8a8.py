"""Where the 129..256-row int8 kernel's output differs from the decode-sized kernel run on the rows in two batches
(exact integer sums: any difference is a wrong operand)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vllm-hymeta_amd"))
sys.path.insert(0, os.path.dirname(__file__))
from vllm_hymeta import hip_ops  # noqa: E402
from time_tall_gemm import SHAPES  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator(device="cpu").manual_seed(7)
for name, (N, K) in SHAPES.items():
    qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    scale = (torch.rand(N, generator=g) * 0.01 + 0.001).cuda()
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    ref = torch.cat([hip_ops.w8a8_gemm(x[:128], qw, scale), hip_ops.w8a8_gemm(x[128:], qw, scale)])
    for r in range(reps):
        out = hip_ops.w8a8_gemm(x, qw, scale)
        torch.cuda.synchronize()
        bad = (out != ref)
        if bool(bad.any()):
            rows = bad.any(1).nonzero().flatten().tolist()
            cols = bad.any(0).nonzero().flatten().tolist()
            print(f"{name} rep {r}: {int(bad.sum())} wrong; rows {rows[:8]}..{rows[-3:]} ({len(rows)}), "
                  f"cols {cols[:8]}..{cols[-3:]} ({len(cols)}); 16-col fragments {sorted(set(c // 16 for c in cols))[:12]}", flush=True)
        else:
            print(f"{name} rep {r}: equal", flush=True)
