"""Regenerates the golden vectors in this directory from the CPU restatement (oracle/).

    python tests/golden/make_golden.py

The reference ships no vectors for this path and its modules cannot be imported
here (missing third-party kernels), so these files pin the RESTATEMENT, not the
reference: they are a regression anchor for the oracle and a fixed target for
the HIP kernels.  bf16 tensors are stored as their int16 bit patterns; expected
outputs as float32 of the fp64 result.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import (gla_decode_step_ref, gla_prefill_varlen_ref, meta_swa_paged_decode_ref,  # noqa: E402
                    meta_swa_varlen_ref, reshape_and_cache_ref)
from oracle.layers import gate_70b_ref, gate_7b_ref  # noqa: E402

D = 128


def bits(t):
    return t.contiguous().view(torch.int16).numpy()


def rb(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(torch.bfloat16)


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def gla_decode():
    g_ = gen(0)
    B, H, Hkv, slots, steps = 3, 2, 2, 4, 4
    state = rb(g_, slots, H, D, D, scale=0.1)
    slot = torch.tensor([2, -1, 0], dtype=torch.int32)
    out = {"state0": bits(state), "slot_idx": slot.numpy()}
    st = state.clone()
    for s in range(steps):
        q, k, v = rb(g_, B, H, D), rb(g_, B, Hkv, D), rb(g_, B, Hkv, D)
        g = (torch.nn.functional.logsigmoid(torch.randn(B, Hkv, D, generator=g_)) / 16).float()
        o = gla_decode_step_ref(q, k, v, g, st, slot)
        out.update({f"q{s}": bits(q), f"k{s}": bits(k), f"v{s}": bits(v), f"g{s}": g.numpy(),
                    f"o{s}": bits(o)})
    out["state_final"] = bits(st)
    np.savez_compressed(os.path.join(HERE, "gla_decode.npz"), **out)


def gla_prefill():
    g_ = gen(1)
    lens = [1, 16, 17, 130]
    H, Hkv, slots = 2, 1, 4
    T = sum(lens)
    cu = np.cumsum([0] + lens).astype(np.int32)
    q, k, v = rb(g_, T, H, D), rb(g_, T, Hkv, D), rb(g_, T, Hkv, D)
    gk = rb(g_, T, Hkv, D, scale=2.0)
    out = {"cu_seqlens": cu, "q": bits(q), "k": bits(k), "v": bits(v), "gk": bits(gk)}
    state0 = rb(g_, slots, H, D, D, scale=0.1)
    meta = rb(g_, H, D, D, scale=0.1)
    slot = torch.tensor([3, 0, -1, 1], dtype=torch.int32)
    ctx = torch.tensor([0, 32, 0, 0], dtype=torch.int32)  # seq 1 resumes a chunked prefill
    out.update({"state0": bits(state0), "meta_state": bits(meta), "slot_idx": slot.numpy(),
                "context_lens": ctx.numpy()})
    # 7B gate, zero init for first chunks
    q7, k7, g7 = gate_7b_ref(q, k, gk)
    st = state0.clone()
    o = gla_prefill_varlen_ref(q7, k7, v, g7, st, slot, cu, ctx, None)
    out.update({"o_7b": bits(o), "state_7b": bits(st)})
    # 70B clamp gate, meta-state init for first chunks
    kr = k.repeat_interleave(H // Hkv, dim=1)
    lb = torch.rand(H, D, generator=g_) * 0.5
    out["lower_bound"] = lb.numpy()
    for name, first in (("clamp", True), ("lb", False)):
        qg, kg, gg = gate_70b_ref(q, kr, lb, first)
        st = state0.clone()
        o = gla_prefill_varlen_ref(qg, kg, v, gg, st, slot, cu, ctx, meta)
        out.update({f"o_70b_{name}": bits(o), f"state_70b_{name}": bits(st)})
    np.savez_compressed(os.path.join(HERE, "gla_prefill.npz"), **out)


def swa_prefill():
    g_ = gen(2)
    lens = [5, 300]
    H, Hkv, M, W = 2, 1, 32, 64
    T = sum(lens)
    cu = np.cumsum([0] + lens).astype(np.int32)
    q, k, v = rb(g_, T, H, D), rb(g_, T, Hkv, D), rb(g_, T, Hkv, D)
    q2, k2, v2 = rb(g_, M, H, D), rb(g_, M, Hkv, D), rb(g_, M, Hkv, D)
    out = {"cu_seqlens": cu, "q": bits(q), "k": bits(k), "v": bits(v), "q2": bits(q2), "k2": bits(k2),
           "v2": bits(v2), "window_left_70b": np.int32(W - 1), "window_left_7b": np.int32(W)}
    out["o_meta_q2"] = meta_swa_varlen_ref(q, k, v, cu, cu, k2, v2, q2, W - 1).float().numpy()
    out["o_meta"] = meta_swa_varlen_ref(q, k, v, cu, cu, k2, v2, None, W - 1).float().numpy()
    out["o_full"] = meta_swa_varlen_ref(q, k, v, cu, cu, k2, v2, None, -1).float().numpy()
    out["o_7b"] = meta_swa_varlen_ref(q, k, v, cu, cu, None, None, None, W).float().numpy()
    np.savez_compressed(os.path.join(HERE, "swa_prefill.npz"), **out)


def paged_decode():
    g_ = gen(3)
    H, Hkv, M, W, bs = 4, 2, 32, 64, 16
    seq_lens = [1, W - 1, W, W + 1, 3 * W]
    nb = [(L + bs - 1) // bs for L in seq_lens]
    nblk = sum(nb) + 2
    perm = torch.randperm(nblk, generator=g_)
    bt = torch.zeros(len(seq_lens), max(nb), dtype=torch.int32)
    off = 0
    for i, n in enumerate(nb):
        bt[i, :n] = perm[off:off + n].to(torch.int32)
        off += n
    kc, vc = rb(g_, nblk, bs, Hkv, D), rb(g_, nblk, bs, Hkv, D)
    q = rb(g_, len(seq_lens), H, D)
    k2, v2 = rb(g_, M, Hkv, D), rb(g_, M, Hkv, D)
    out = {"q": bits(q), "key_cache": bits(kc), "value_cache": bits(vc), "k2": bits(k2), "v2": bits(v2),
           "block_tables": bt.numpy(), "cache_seqlens": np.array(seq_lens, dtype=np.int32),
           "window_left_70b": np.int32(W - 1), "window_left_7b": np.int32(W)}
    out["o_meta"] = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, k2, v2, W - 1).float().numpy()
    out["o_full"] = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, k2, v2, -1).float().numpy()
    out["o_7b"] = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, None, None, W).float().numpy()
    np.savez_compressed(os.path.join(HERE, "paged_decode.npz"), **out)


def reshape_cache():
    g_ = gen(4)
    T, Hkv, bs, nblk = 23, 2, 16, 5
    key, value = rb(g_, T, Hkv, D), rb(g_, T, Hkv, D)
    kv = rb(g_, 2, nblk, bs, Hkv, D)
    slots = torch.randperm(nblk * bs, generator=g_)[:T].to(torch.int64)
    slots[::4] = -1
    after = kv.clone()
    reshape_and_cache_ref(key, value, after[0], after[1], slots)
    np.savez_compressed(os.path.join(HERE, "reshape_and_cache.npz"), key=bits(key), value=bits(value),
                        kv_before=bits(kv), kv_after=bits(after), slot_mapping=slots.numpy())


if __name__ == "__main__":
    torch.set_num_threads(4)
    gla_decode()
    gla_prefill()
    swa_prefill()
    paged_decode()
    reshape_cache()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
