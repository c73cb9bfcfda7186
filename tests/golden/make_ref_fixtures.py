"""Outputs of the reference's own importable eager code on seeded inputs -> tests/golden/ref_*.npz.

Runs in the build container only (needs /root/reference; nothing from it travels: the fixtures hold inputs
and outputs, no source).  What loads by file path without the GPU-only third-party wheels:

* hf_7B_model/norms.py          RMSNorm                                   -> ref_rmsnorm.npz
* hf_7B_model/window_attention.py  RotaryEmbedding, apply_rotary_pos_emb, repeat_kv (module level; the layer's
                                forward needs flash_attn and is not called)  -> ref_rope.npz
* hf_7B_model/cache.py, hf_model/cache.py   Cache.update (rolling window, recurrent / meta state slots)
                                                                           -> ref_cache.npz
* vllm_hymeta/models/my_fused_recurrent.py and vllm_hymeta/model_for_7B/my_fused_recurrent.py: the reference's one
  kernel, the single-token GLA step, run under Triton's CPU interpreter (TRITON_INTERPRET=1, set by this script
  before triton is imported; numpy arithmetic, so fp16 and fp32 tensors: the interpreter has no bf16).  Both files
  import two decorators from ``fla.utils`` (``contiguous``: make tensor arguments contiguous; ``autocast_custom_fwd``:
  autocast bookkeeping); fla is not installed, and the script registers a module of that name holding those two
  decorators (no arithmetic in either) before loading the files           -> ref_gla_decode.npz
* vllm_hymeta/attention/fa_kernel_metatoken.py: ``naive_metatoken_flash_attn`` (:377-410), the reference's own
  dense-mask definition of window + meta-token attention, and ``_update_out_and_lse`` (:110-122), its merge of the two
  partial softmaxes.  Both are plain torch.  The module's import lines name flash_attn (absent here, GPU-only): the
  script registers modules of those names whose every attribute raises when called, so that nothing but the two torch
  functions can run; the naive function moves its mask with ``.cuda()`` (:407), which the script makes a no-op for the
  duration of the call (no GPU here; the mask is the same tensor wherever it lives)      -> ref_meta_attention.npz

bf16 tensors are stored as their uint16 bit patterns (key suffix ``_bf16``)."""
import importlib.util
import os
import sys
import types

os.environ.setdefault("TRITON_INTERPRET", "1")   # before triton is imported anywhere

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load(pkg: str, name: str):
    """Import REF/pkg/name.py as ``pkg.name`` without running the package's __init__ (which imports fla)."""
    if pkg not in sys.modules:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]
        sys.modules[pkg] = m
    full = f"{pkg}.{name}"
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, os.path.join(REF, pkg, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod


def put(out, key, t):
    if t.dtype == torch.bfloat16:
        out[key + "_bf16"] = t.contiguous().view(torch.int16).numpy().view(np.uint16)
    else:
        out[key] = t.contiguous().numpy()


def rmsnorm():
    RMSNorm = load("hf_7B_model", "norms").RMSNorm
    g = torch.Generator().manual_seed(100)
    out = {}
    for tag, rows, hidden in (("head", 12, 128), ("hidden", 3, 3584)):
        x = (torch.randn(rows, hidden, generator=g) * 1.7).to(torch.bfloat16)
        w = 1 + 0.2 * torch.randn(hidden, generator=g)
        m = RMSNorm(hidden, eps=1e-6)
        with torch.no_grad():
            m.weight.copy_(w)
            y_w32 = m(x)                       # fp32 weight: bf16 normalised value promoted to fp32
            y_f32 = m(x.float())               # fp32 activations
            mb = RMSNorm(hidden, eps=1e-6).to(torch.bfloat16)
            mb.weight.copy_(w.to(torch.bfloat16))
            y_bf = mb(x)                       # module cast to bf16, as a bf16 checkpoint runs it
        put(out, f"{tag}_x", x)
        put(out, f"{tag}_w", w)
        put(out, f"{tag}_y_w32", y_w32)
        put(out, f"{tag}_y_f32", y_f32)
        put(out, f"{tag}_y", y_bf)
    np.savez_compressed(os.path.join(HERE, "ref_rmsnorm.npz"), eps=np.float32(1e-6), **out)
    return {k: v.shape for k, v in out.items()}


def rope():
    wa = load("hf_7B_model", "window_attention")
    g = torch.Generator().manual_seed(101)
    H, Hkv, D, theta = 4, 2, 128, 1e6
    pos = torch.tensor([[0, 1, 2, 127, 128, 4095, 4096, 32767, 32895, 131071]])
    T = pos.shape[1]
    rot = wa.RotaryEmbedding(D, max_position_embeddings=131072, base=theta)
    out = {"positions": pos[0].numpy(), "theta": np.float64(theta)}
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        q = torch.randn(1, H, T, D, generator=g).to(dt)
        k = torch.randn(1, Hkv, T, D, generator=g).to(dt)
        cos, sin = rot(k, pos)
        qe, ke = wa.apply_rotary_pos_emb(q, k, cos, sin)
        for name, t in (("q", q), ("k", k), ("cos", cos), ("sin", sin), ("q_out", qe), ("k_out", ke)):
            put(out, f"{tag}_{name}", t)
    x = torch.arange(2 * 3 * 5 * 4, dtype=torch.float32).view(2, 3, 5, 4)
    out["repeat_kv_in"] = x.numpy()
    out["repeat_kv_out"] = wa.repeat_kv(x, 7).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_rope.npz"), **out)
    return {k: getattr(v, "shape", ()) for k, v in out.items()}


def cache():
    g = torch.Generator().manual_seed(102)
    out = {}
    Hkv, D, W = 2, 8, 6
    chunks = [4, 1, 1, 1, 1, 3, 1, 2]       # prefill of 4 (< W), decode steps, a 3-token chunk, ...
    out["window"] = np.int64(W)
    out["chunks"] = np.array(chunks)
    total = sum(chunks)
    keys = torch.randn(1, Hkv, total, D, generator=g)
    vals = torch.randn(1, Hkv, total, D, generator=g)
    out["keys"], out["values"] = keys.numpy(), vals.numpy()
    for tag, pkg in (("7b", "hf_7B_model"), ("70b", "hf_model")):
        Cache = load(pkg, "cache").Cache
        c = Cache()
        t = 0
        for i, n in enumerate(chunks):
            kw = dict(attn_state=(keys[:, :, t:t + n].clone(), vals[:, :, t:t + n].clone()), layer_idx=0, offset=n,
                      cache_kwargs={"window_size": W})
            if tag == "70b" and i == 0:
                kw["meta_state"] = (torch.full((1, Hkv, 3, D), 7.0), torch.full((1, Hkv, 3, D), -7.0))
            st = c.update(**kw)
            t += n
            out[f"{tag}_k_after_{i}"] = st["attn_state"][0].numpy().copy()
            out[f"{tag}_v_after_{i}"] = st["attn_state"][1].numpy().copy()
            out[f"{tag}_seen_after_{i}"] = np.int64(c.get_seq_length(0))
        if tag == "70b":
            out["70b_meta_k_kept"] = c[0]["meta_state"][0].numpy()
        # a first chunk longer than the window is cut to its last W tokens
        c2 = Cache()
        st = c2.update(attn_state=(keys[:, :, :10].clone(), vals[:, :, :10].clone()), layer_idx=0, offset=10,
                       cache_kwargs={"window_size": W})
        out[f"{tag}_k_long_first"] = st["attn_state"][0].numpy().copy()
        # the recurrent-state slot: replaced, not accumulated
        c3 = Cache()
        c3.update(recurrent_state=torch.ones(1, 2, 3), layer_idx=0, offset=5)
        st = c3.update(recurrent_state=torch.full((1, 2, 3), 2.0), layer_idx=0, offset=1)
        out[f"{tag}_recurrent_after_2"] = st["recurrent_state"].numpy()
        out[f"{tag}_recurrent_seen"] = np.int64(c3.get_seq_length(0))
    np.savez_compressed(os.path.join(HERE, "ref_cache.npz"), **out)
    return len(out)


def _fla_utils_decorators():
    """``fla.utils.contiguous`` / ``autocast_custom_fwd`` as far as the kernel launchers use them."""
    if "fla.utils" in sys.modules:
        return
    import functools

    def contiguous(fn):
        @functools.wraps(fn)
        def wrapper(ctx, *args, **kwargs):
            return fn(ctx, *(a.contiguous() if isinstance(a, torch.Tensor) else a for a in args),
                      **{k: (v.contiguous() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()})
        return wrapper

    fla, utils = types.ModuleType("fla"), types.ModuleType("fla.utils")
    utils.contiguous, utils.autocast_custom_fwd = contiguous, (lambda fn: fn)
    fla.utils = utils
    sys.modules["fla"], sys.modules["fla.utils"] = fla, utils


def _load_file(path: str, name: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gla_decode():
    """Two successive steps of ``my_fused_recurrent`` (70B file) and of its 7B twin on the same inputs: B = 4 with one
    padded sequence (slot -1), H = 2, D = 128, a state of 3 slots that is not zero; fp16 and fp32."""
    assert os.environ.get("TRITON_INTERPRET") == "1"
    _fla_utils_decorators()
    k70 = _load_file(os.path.join(REF, "vllm_hymeta/models/my_fused_recurrent.py"), "ref_my_fused_recurrent_70b")
    k7 = _load_file(os.path.join(REF, "vllm_hymeta/model_for_7B/my_fused_recurrent.py"), "ref_my_fused_recurrent_7b")
    gen = torch.Generator().manual_seed(103)
    B, H, D, slots, steps = 4, 2, 128, 3, 2
    slot = torch.tensor([2, -1, 0, 1], dtype=torch.int32)
    live = slot != -1
    out = {"slot_idx": slot.numpy(), "scale": np.float64(D ** -0.5), "steps": np.int64(steps)}
    for tag, dt in (("f16", torch.float16), ("f32", torch.float32)):
        state = (torch.randn(slots, H, D, D, generator=gen) * 0.5).to(dt)
        put(out, f"{tag}_state_in", state)
        st70, st7 = state.clone(), state.clone()
        for s in range(steps):
            q = torch.randn(B, H, 1, D, generator=gen).to(dt)
            k = torch.randn(B, H, 1, D, generator=gen).to(dt)
            v = torch.randn(B, H, 1, D, generator=gen).to(dt)
            # log-decays of both gates' ranges: logsigmoid / 16 (7B) up to log(0.05) (70B clamp)
            g = torch.nn.functional.logsigmoid(torch.randn(B, H, 1, D, generator=gen) * 2) * (0.2 if s == 0 else 1.5)
            g = g.to(dt).float()          # representable in the activation dtype: both variants see the same values
            o70 = k70.my_fused_recurrent(q, k, v, g, st70, slot, D ** -0.5)            # fp32 g, [B, H * D]
            o7 = k7.my_fused_recurrent_gla(q, k, v, g.to(dt), st7, slot)                # g in dt, [B, H, D]
            assert o70.shape == (B, H * D) and o7.shape == (B, H, D)
            assert torch.equal(o70.view(B, H, D)[live], o7[live]) and torch.equal(st70, st7), "the two variants differ"
            for name, t in (("q", q[:, :, 0]), ("k", k[:, :, 0]), ("v", v[:, :, 0]), ("g", g[:, :, 0])):
                put(out, f"{tag}_{name}{s}", t)
            o = o7.clone()
            o[~live] = 0            # rows of padded sequences are never written by the kernel (torch.empty)
            put(out, f"{tag}_out{s}", o)
        put(out, f"{tag}_state_out", st70)
    np.savez_compressed(os.path.join(HERE, "ref_gla_decode.npz"), variants_agree=np.bool_(True), **out)
    return {k: getattr(v, "shape", ()) for k, v in out.items()}


def _flash_attn_placeholders():
    """Modules that satisfy ``from flash_attn.flash_attn_interface import ...`` and ``import flash_attn_2_cuda``; any use
    of what they hold raises."""
    def refuse(name):
        def call(*a, **k):
            raise RuntimeError(f"{name}: flash_attn is not installed here; only the module's plain-torch functions run")
        return call

    class Placeholder(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return refuse(f"{self.__name__}.{name}")

    for name in ("flash_attn", "flash_attn.flash_attn_interface", "flash_attn_2_cuda"):
        if name not in sys.modules:
            sys.modules[name] = Placeholder(name)
    sys.modules["flash_attn"].flash_attn_interface = sys.modules["flash_attn.flash_attn_interface"]


def meta_attention():
    """``naive_metatoken_flash_attn`` on fp32 inputs: equal lengths with meta queries, a chunk against a longer key
    range (bottom-right alignment), a window shorter and one longer than the sequence; ``_update_out_and_lse`` on two
    partial softmaxes."""
    _flash_attn_placeholders()
    mod = _load_file(os.path.join(REF, "vllm_hymeta/attention/fa_kernel_metatoken.py"), "ref_fa_kernel_metatoken")
    gen = torch.Generator().manual_seed(104)
    H, D, M = 3, 32, 8
    out = {"num_meta": np.int64(M), "scale": np.float64(D ** -0.5)}
    cases = [("eq_q2", 11, 11, 5, True), ("chunk", 4, 13, 6, False), ("short_window", 9, 9, 1, False), ("long_window", 7, 7, 50, False)]
    out["cases"] = np.array([c[0] for c in cases])
    # two cases at the kernels' own geometry (head size 128, 32 meta tokens, inputs that are exact in bf16, stored as bf16)
    kernel_cases = [("k_eq_q2", 70, 70, 24, True, 2, 128, 32), ("k_chunk", 20, 90, 33, False, 2, 128, 32)]
    out["kernel_cases"] = np.array([c[0] for c in kernel_cases])
    out["kernel_num_meta"], out["kernel_scale"] = np.int64(32), np.float64(128 ** -0.5)
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, Lq, Lk, win, with_q2, H, D, M in [c + (H, D, M) for c in cases] + kernel_cases:
            as_bf16 = tag.startswith("k_")
            rnd = (lambda *shape: torch.randn(*shape, generator=gen).to(torch.bfloat16)) if as_bf16 else \
                (lambda *shape: torch.randn(*shape, generator=gen))
            q1 = rnd(1, Lq, H, D)
            k1, v1 = rnd(1, Lk, H, D), rnd(1, Lk, H, D)
            k2, v2 = rnd(1, M, H, D), rnd(1, M, H, D)
            q2 = rnd(1, M, H, D) if with_q2 else None
            f32 = lambda t: None if t is None else t.float()           # the function runs in fp32 on the bf16 values
            o = mod.naive_metatoken_flash_attn(f32(q1), f32(q2), f32(k1), f32(k2), f32(v1), f32(v2), softmax_scale=D ** -0.5,
                                               num_meta_tokens=M, sliding_window=(win, 0))
            for name, t in (("q1", q1), ("k1", k1), ("v1", v1), ("k2", k2), ("v2", v2)):
                put(out, f"{tag}_{name}", t[0])
            if q2 is not None:
                put(out, f"{tag}_q2", q2[0])
            out[f"{tag}_window"] = np.int64(win)      # sliding_window[0]: keys with q - k < window are visible
            put(out, f"{tag}_out", o[0])              # real-query rows first, meta-query rows (if any) last
    finally:
        torch.Tensor.cuda = saved_cuda
    # the merge: [batch, seq, heads, dim] outputs and [batch, heads, seq] log-sum-exps of two key sets
    L = 6
    o1, o2 = torch.randn(1, L, H, D, generator=gen).to(torch.bfloat16), torch.randn(1, L, H, D, generator=gen).to(torch.bfloat16)
    l1, l2 = torch.randn(1, H, L, generator=gen) * 3, torch.randn(1, H, L, generator=gen) * 3
    mo, ml = mod._update_out_and_lse(o1, l1, o2, l2)
    for name, t in (("merge_out1", o1[0]), ("merge_lse1", l1[0]), ("merge_out2", o2[0]), ("merge_lse2", l2[0]), ("merge_out", mo[0]),
                    ("merge_lse", ml[0])):
        put(out, name, t)
    np.savez_compressed(os.path.join(HERE, "ref_meta_attention.npz"), **out)
    return {k: getattr(v, "shape", ()) for k, v in out.items()}


if __name__ == "__main__":
    print("rmsnorm", rmsnorm())
    print("rope", rope())
    print("cache", cache())
    print("gla_decode", gla_decode())
    print("meta_attention", meta_attention())
