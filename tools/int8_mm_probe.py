"""Is the library int8 GEMM (torch._int_mm) usable and exact on this GPU, and how fast is it from
decode-sized to prefill-sized row counts?  (needs a GPU)"""
import os
import torch
dev = 'cuda'
shapes = [(M, K, N) for M in [int(m) for m in os.environ.get("MS", "64,128,256,2048,8192").split(",")]
          for K, N in ((3584, 37888), (18944, 3584), (3584, 4608))]
for M, K, N in shapes:
    a = torch.randint(-127, 127, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev)
    try:
        c = torch._int_mm(a, w.t())
        ref = (a[:4].cpu().long() @ w[:64].cpu().long().t())
        ok = torch.equal(c[:4, :64].cpu().long(), ref)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c = torch._int_mm(a, w.t())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(M, K, N, 'ok' if ok else 'MISMATCH', f'{ms * 1e3:.1f} us', f'{2 * M * K * N / ms / 1e9:.0f} TOPS', flush=True)
    except Exception as e:
        print(M, K, N, 'ERR', repr(e)[:200], flush=True)
