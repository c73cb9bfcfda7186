"""129..256-row W8A8 linears of the 7B model on whichever kernel the library picks (HYMETA_TALL_MIN_ROWS=1000 keeps the
tall kernel out); prints a checksum of every output so that two runs can be compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vllm-hymeta_amd"))
sys.path.insert(0, os.path.dirname(__file__))
from vllm_hymeta import hip_ops  # noqa: E402
from time_tall_gemm import SHAPES, timed  # noqa: E402

rows = [int(r) for r in sys.argv[1:]] or [256, 192, 130]
g = torch.Generator(device="cpu").manual_seed(7)
for name, (N, K) in SHAPES.items():
    ws = [torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).cuda() for _ in range(3)]
    scale = (torch.rand(N, generator=g) * 0.01 + 0.001).cuda()
    for M in rows:
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
        out = hip_ops.w8a8_gemm(x, ws[0], scale)
        torch.cuda.synchronize()
        st = {"i": 0}

        def ours():
            st["i"] = (st["i"] + 1) % 3
            hip_ops.w8a8_gemm(x, ws[st["i"]], scale)

        line = f"{name} M={M}: checksum {out.view(torch.int16).to(torch.int64).sum().item()} {timed(ours):.1f} us"
        if name == "gate_up":
            o2 = hip_ops.gemm_silu_mul(x, ws[0], scale)
            line += f"; gated equals two kernels {torch.equal(o2, hip_ops.silu_and_mul(out))}, {timed(lambda: hip_ops.gemm_silu_mul(x, ws[0], scale)):.1f} us"
        print(line, flush=True)
