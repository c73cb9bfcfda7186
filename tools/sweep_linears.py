"""Seeded sweep of the decode-sized linears (1..256 rows) over random shapes of every weight format: each result against
the fp32 product of the same (de-quantised) operands, and two calls against each other.

    python tools/sweep_linears.py [cases per format]
"""
import os
import random
import sys

import torch

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from vllm_hymeta import hip_ops  # noqa: E402
from vllm_hymeta.linear import HymetaLinear  # noqa: E402
from test_gpu_quant import _gptq_problem  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rnd = random.Random(7)
    g = torch.Generator(device="cpu").manual_seed(7)
    bad = 0
    for fmt in ("bf16", "w8a8", "gptq4g128", "gptq4g64", "gptq8g64", "gptq8g128"):
        for _ in range(cases):
            M = rnd.choice([1, 7, 16, 17, 32, 33, 48, 64, 65, 100, 128, 129, 160, 200, 255, 256])
            K = {"bf16": 64, "w8a8": 128}.get(fmt, 256) * rnd.randint(4, 40 * (256 // {"bf16": 64, "w8a8": 128}.get(fmt, 256)))
            N = 128 * rnd.randint(1, 64) if rnd.random() < 0.8 else 128 * rnd.randint(200, 320)
            x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
            bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).cuda() if rnd.random() < 0.5 else None
            if fmt == "bf16":
                w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
                lin = HymetaLinear.from_weight(w, bias)
                ref = x.float() @ w.float().t()
            elif fmt == "w8a8":
                qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).cuda()
                ws = (torch.rand(N, generator=g) * (2.0 / 127) * K ** -0.5).float().cuda()
                lin = HymetaLinear.from_w8a8(qw, ws, bias=bias)
                xq, xs = hip_ops.w8a8_quantize(x)
                ref = (xq.float() @ qw.float().t()) * xs.view(-1, 1) * ws.view(1, -1)
            else:
                bits, group = int(fmt[4]), int(fmt.split("g")[2])
                qw, qz, sc = _gptq_problem(M + K + N, K, N, bits, group, torch.float16)
                lin = HymetaLinear.from_gptq(qw.cuda(), qz.cuda(), sc.cuda(), bits, group, bias=bias)
                w = hip_ops.gptq_dequant(qw.cuda(), qz.cuda(), sc.cuda(), bits, group)      # [K, N] bf16
                ref = x.float() @ w.float()
            if bias is not None:
                ref = ref + bias.float()
            y, again = lin(x), lin(x)
            part = lin(x, defer=True)
            y2 = part.tensor() if isinstance(part, hip_ops.PartialSums) else part
            torch.cuda.synchronize()
            scale = ref.abs().max().item() + 1e-9
            err = (y.float() - ref).abs().max().item() / scale
            ok = err < 1.2e-2 and torch.equal(y, again) and torch.equal(y, y2)
            if bias is None and N % 256 == 0 and fmt in ("bf16", "w8a8"):      # the gate / up form: activation in the epilogue where the kernel has one
                gated, two = lin.forward_silu_mul(x), hip_ops.silu_and_mul(y)
                if fmt == "w8a8":       # exact int32 sums: the same bits; bf16: the same bits only when the plain call sums K in one slice
                    ok = ok and torch.equal(gated, two)
                else:
                    ok = ok and (gated.float() - two.float()).abs().max().item() <= 2e-2 * (two.float().abs().max().item() + 1e-9)
            if not ok:
                bad += 1
                print(f"FAIL {fmt} M={M} K={K} N={N} bias={bias is not None}: err {err:.4g} same {torch.equal(y, again)} deferred {torch.equal(y, y2)}", flush=True)
        print(f"{fmt}: done", flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
