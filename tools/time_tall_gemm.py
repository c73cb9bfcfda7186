"""129..256-row bf16 linears of the 7B model: the tall streaming kernel against the library GEMM (values and time).

    python tools/time_tall_gemm.py [rows ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vllm-hymeta_amd"))
from vllm_hymeta import hip_ops  # noqa: E402

SHAPES = {"qkv": (4608, 3584), "o": (3584, 3584), "gate_up": (37888, 3584), "down": (3584, 18944)}


def timed(fn, iters=20):
    """Mean time of one call inside a replayed graph of ``iters`` calls (the host's launch cost stays outside)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * iters) * 1e3


def main():
    rows = [int(r) for r in sys.argv[1:]] or [256, 192, 160, 130]
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    for name, (N, K) in SHAPES.items():
        # several copies of the weight, so that no launch finds it in the caches
        ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(3)]
        for M in rows:
            x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
            ref = (x.float() @ ws[0].float().t())
            print(f"{name} M={M} N={N} K={K}", flush=True)
            out = hip_ops.bf16_gemm(x, ws[0])
            torch.cuda.synchronize()
            err = (out.float() - ref).abs().max().item()
            lib = torch.nn.functional.linear(x, ws[0])
            err_lib = (lib.float() - ref).abs().max().item()
            differ = (out != ref.to(torch.bfloat16)).float().mean().item()
            state = {"i": 0}

            def ours():
                state["i"] = (state["i"] + 1) % 3
                hip_ops.bf16_gemm(x, ws[state["i"]])

            def library():
                state["i"] = (state["i"] + 1) % 3
                torch.nn.functional.linear(x, ws[state["i"]])

            t_ours, t_lib = timed(ours), timed(library)
            mb = N * K * 2 / 1e6
            print(f"    max err {err:.4f} (library {err_lib:.4f}), {differ * 100:.3f}% of outputs differ from the rounded fp32 product; "
                  f"tall {t_ours:.1f} us = {mb / t_ours:.2f} TB/s, library {t_lib:.1f} us", flush=True)
            if name == "gate_up":
                out2 = hip_ops.gemm_silu_mul(x, ws[0])
                if out2 is not None:
                    two = hip_ops.silu_and_mul(out)
                    print(f"    gated epilogue equals linear + silu_and_mul: {torch.equal(out2, two)}; "
                          f"{timed(lambda: hip_ops.gemm_silu_mul(x, ws[0])):.1f} us", flush=True)


if __name__ == "__main__":
    main()
