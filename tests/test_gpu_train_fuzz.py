"""Seeded random-configuration fuzz of the generator's BACKWARD pass (``pytest -m gpu``): widths 24 .. 256,
strides 2 .. 8, kernel sizes 3 .. 11, dilations up to 5, 1 .. 3 ResBlocks of unequal depth, with / without weight norm, ResBlock bias,
AR branch — every element of every parameter / input gradient against the CPU oracle's autograd.  The backward has its own kernels per
tap count (wgrad_taps_kernel<1,2,3,7,11>, wgrad_gemm_kernel) and split-K forms; this walks them.  HIFICAR_FUZZ_CASES raises the count.

LeakyReLU makes gradients discontinuous where a pre-activation is within rounding distance of zero, and ONE activation
that falls on the other side moves a whole output channel's weight gradient of a short sequence by percents (tests/dev/grad_fuzz_probe.py:
cases 13 / 17 / 62 of the first version of this test — the same configurations are exact to 1e-6 on inputs without such an activation).
So the input is chosen, per case, such that the float64 oracle sees NO LeakyReLU input (generator slope, the PastFCEncoder's 0.1, the
output conv's 0.01) closer to zero than 2e-6 of its tensor's scale — a few times the device's forward deviation — by trying input seeds
(the odd, kinked cases are sized to ~5 x 10^4 activations so that a few tries suffice).  Then EVERY element of EVERY gradient must be within
2e-4 of the float64 oracle.  Even cases force negative_slope = 1.0 and keep the drawn, bigger shapes (many tiles, row splits): only the output
conv and the MLP have kinks there; when no clean input turns up in a few seeds they fall back to flip-robust statistics (median tensor
within 2e-5, none beyond 0.2)."""
import os

import numpy as np
import pytest
import torch

from conftest import E2W_PARAMS, rel_err
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict, uniform
from oracle import hificar_oracle as O

pytestmark = pytest.mark.gpu
N_CASES = max(24, int(os.environ.get("HIFICAR_FUZZ_CASES", "24")))
TOL = 2e-4


def draw(rng):
    n_stages = int(rng.integers(1, 4))
    channels = int(rng.choice([24, 32, 48, 64, 96, 128, 256]))
    while channels >> n_stages < 3:
        n_stages -= 1
    scales = [int(rng.choice([2, 3, 4, 5, 8])) for _ in range(n_stages)]
    n_blocks = int(rng.integers(1, 4))
    ks = [int(rng.choice([3, 5, 7, 9, 11])) for _ in range(n_blocks)]
    dils = [[int(rng.integers(1, 6)) for _ in range(int(rng.integers(1, 4)))] for _ in range(n_blocks)]
    use_ar = bool(rng.integers(0, 3))
    cf = int(rng.integers(1, 40))
    return dict(E2W_PARAMS, channels=channels, kernel_size=int(rng.choice([3, 5, 7])), upsample_scales=scales,
                upsample_kernel_sizes=[2 * s for s in scales], resblock_kernel_sizes=ks, resblock_dilations=dils, use_ar=use_ar,
                in_channels=cf + (128 if use_ar else 0), bias=bool(rng.integers(0, 3)), use_weight_norm=bool(rng.integers(0, 3)),
                use_tanh=bool(rng.integers(0, 4)),
                nonlinear_activation_params={"negative_slope": float(rng.choice([0.1, 0.1, 0.2, 0.01, 1.0]))}), cf


def kink_margin(sd, params, c_np, ar_np):
    """Smallest |x| / max|x| over every LeakyReLU input of the float64 oracle forward (slope-1 calls do not count)."""
    margins = []
    real = O.F.leaky_relu

    def spy(x, negative_slope=0.01, *a, **kw):
        if negative_slope != 1.0 and x.numel():
            margins.append(float(x.abs().min() / x.abs().max().clamp_min(1e-300)))
        return real(x, negative_slope, *a, **kw)

    O.F.leaky_relu = spy
    try:
        with torch.no_grad():
            O.generator_forward(O.fold_weight_norm(sd, dtype=torch.float64), params, torch.from_numpy(c_np).double(),
                                torch.from_numpy(ar_np).double() if ar_np is not None else None)
    finally:
        O.F.leaky_relu = real
    return min(margins) if margins else 1.0


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_configuration_gradients(case):
    assert torch.cuda.is_available()
    rng = np.random.default_rng(31000 + case)
    params, cf = draw(rng)
    if case % 2 == 0:
        params["nonlinear_activation_params"] = {"negative_slope": 1.0}
    hop = int(np.prod(params["upsample_scales"]))
    sd = synth_state_dict(params, seed=700 + case)
    g = HiFiGANGenerator(**params, precision="f32")
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g = g.train().cuda()
    B = int(rng.integers(1, 5)) if rng.integers(0, 4) else int(rng.integers(5, 24))
    T = int(rng.integers(2, 30)) if rng.integers(0, 4) else int(rng.integers(30, 120))
    if case % 2:  # kinked: keep the number of activations near 5 x 10^4 (a clean input is then found within a few seeds)
        per_frame, width, rate = 0, params["channels"], 1
        for i, s_ in enumerate(params["upsample_scales"]):
            width, rate = width // 2, rate * s_
            per_frame += rate * width * (1 + 2 * sum(len(d) for d in params["resblock_dilations"]))
        T = max(2, min(T, int(5e4 / max(per_frame * B, 1))))
        B = max(1, min(B, int(5e4 / max(per_frame * T, 1))))
    clean = False
    for attempt in range(40 if case % 2 else 6):
        c_np = synth_features(B, T, cf, seed=case + 1000 * attempt).transpose(0, 2, 1).copy()
        ar_np = ((synth_features(B, 512, 1, seed=case + 1 + 1000 * attempt)[:, :, 0] * 0.4).reshape(B, 1, 512).astype(np.float32)
                 if params["use_ar"] else None)
        margin = kink_margin(sd, params, c_np, ar_np)
        if margin > 2e-6:
            clean = True
            break
    if not clean and case % 2:
        pytest.fail(f"case {case}: no input without a LeakyReLU input within 2e-6 of zero in 40 seeds (B {B}, T {T})")
    cot = uniform(case, "cot", (B, 1, hop * T), -1.0, 1.0)
    c = torch.from_numpy(c_np).cuda().requires_grad_(True)
    ar = torch.from_numpy(ar_np).cuda().requires_grad_(True) if ar_np is not None else None
    y = g(c, ar=ar)
    (y * torch.from_numpy(cot).cuda()).sum().backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out64, ref64 = O.gradients(sd, params, c_np, ar_np, cot, dtype=torch.float64)
    tag = (case, {k: params[k] for k in ("channels", "kernel_size", "upsample_scales", "resblock_kernel_sizes", "resblock_dilations", "use_ar",
                                         "in_channels", "bias", "use_weight_norm", "nonlinear_activation_params")}, B, T)
    assert rel_err(y.detach().cpu().numpy(), out64.numpy()) < 2e-5, tag
    got = {k: p.grad for k, p in g.named_parameters()}
    got["c"] = c.grad
    if ar is not None:
        got["ar"] = ar.grad
    assert sorted(got) == sorted(ref64), tag
    errs = {}
    for k in ref64:
        assert got[k] is not None and bool(torch.isfinite(got[k]).all()), (tag, k)
        errs[k] = rel_err(got[k].cpu().numpy(), ref64[k].numpy())
        if ref64[k].numel() == 1:
            # a one-element gradient (output_conv's weight_g: <dW, v> / ||v||) has no neighbours to set the scale of its error: when the dot
            # product nearly cancels, fp32 summation noise shows up amplified against its own small value (case 356: 5e-4 on this scalar,
            # 5e-7 on every other tensor) — held to 25 x the bar instead
            errs[k] /= 25.0
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    if clean:
        assert worst[0][1] < TOL, (tag, attempt, worst)
    else:  # (an even case whose output conv / MLP input comes within 2e-6 of a kink for every seed tried)
        # Above ~5e-7 the fp32 pre-activation still falls on the oracle's side of the kink and the gradients agree as in the clean cases; below,
        # device and oracle may sit on different sides of the output conv's 100 : 1 kink for one element, which moves every upstream gradient
        # (measured with tests/dev/grad_T_probe.py: 1e-6 at margins above 1e-7, a median of 4e-3 / a maximum of 2e-2 at a margin of 7e-9).
        v = np.array(list(errs.values()))
        assert np.median(v) < (2e-5 if margin > 5e-7 else 1e-2) and v.max() < 0.2, (tag, margin, float(np.median(v)), worst)
