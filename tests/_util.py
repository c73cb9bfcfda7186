"""Shared helpers for the parity tests: tolerances and seeded inputs."""
import torch

# Stated tolerances (the reference pins none): outputs are bf16, compared with
# an fp64 oracle fed the same bf16-rounded inputs.
REL_RMS_TOL = 5e-3          # ||out - ref||_rms / ||ref||_rms
MAX_ABS_FRAC = 2.0 ** -6    # max|out - ref| <= 2^-6 * max|ref|


def rel_rms(out, ref):
    out, ref = out.double().cpu(), ref.double().cpu()
    denom = ref.pow(2).mean().sqrt().clamp_min(1e-30)
    return float((out - ref).pow(2).mean().sqrt() / denom)


def assert_bf16_close(out, ref, what, rel=REL_RMS_TOL, frac=MAX_ABS_FRAC):
    out, ref = out.double().cpu(), ref.double().cpu()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite values in output"
    r = rel_rms(out, ref)
    mx = float((out - ref).abs().max())
    bound = frac * float(ref.abs().max()) + 1e-6
    assert r <= rel, f"{what}: relative RMS error {r:.3e} > {rel:.1e}"
    assert mx <= bound, f"{what}: max abs error {mx:.3e} > {bound:.3e}"


def assert_bf16_ulp(a, b, what, ulps=1):
    """Two bf16 tensors equal within `ulps` units in the last place (plus a small
    absolute term: where the update's two addends cancel, an fp32 last-bit
    difference in either addend is many ulps of the tiny result)."""
    a, b = a.float().cpu(), b.float().cpu()
    tol = ulps * (2.0 ** -7) * torch.maximum(a.abs(), b.abs()) + 2e-5
    bad = (a - b).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} elements differ by more than {ulps} bf16 ulp"


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def randn_bf16(gen_, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen_) * scale).to(torch.bfloat16)
