"""Random discriminator configurations (kernel sizes, strides, groups incl. group widths that are no multiple of 4, pooling, periods,
weight norm on / off, odd lengths) on a MI355X against the CPU oracle: every layer output and every gradient.  ``pytest -m gpu``;
HIFICAR_FUZZ_CASES raises the number of cases (default 16)."""
import os

import numpy as np
import pytest
import torch

from articulatory_amd.models import HiFiGANMultiScaleMultiPeriodDiscriminator
from articulatory_amd.utils.synth import synth_disc_state_dict, uniform
from oracle import disc_oracle as DO

pytestmark = pytest.mark.gpu
N_CASES = max(16, int(os.environ.get("HIFICAR_FUZZ_CASES", "16")))


def random_case(i):
    rng = np.random.default_rng(1000 + i)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    scales = int(rng.integers(0, 4))
    periods = sorted(set(int(p) for p in rng.choice([2, 3, 5, 7], size=int(rng.integers(0 if scales else 1, 4)), replace=False)))
    pool = pick([{"kernel_size": 4, "stride": 2, "padding": 2}, {"kernel_size": 2, "stride": 2, "padding": 0}, {"kernel_size": 3, "stride": 2, "padding": 1}])
    mg = pick([1, 4, 16])
    sp = {"in_channels": 1, "out_channels": 1, "kernel_sizes": [pick([3, 5, 15]), pick([5, 11, 41]), pick([3, 5]), 3],
          "channels": pick([16, 32]) if mg == 16 else pick([8, 16, 24]),  # (channel counts the groups divide, as torch requires)
          "max_downsample_channels": pick([32, 64]), "max_groups": mg, "bias": bool(rng.integers(2)),
          "downsample_scales": [pick([1, 2, 4]) for _ in range(int(rng.integers(1, 4)))], "nonlinear_activation": "LeakyReLU",
          "nonlinear_activation_params": {"negative_slope": pick([0.1, 0.2])}}
    pch, pn = pick([4, 8]), int(rng.integers(1, 4))
    # (the reference's output conv takes min(4 * last_width, max) input channels, hifigan.py:380-383: only stacks whose last layer has
    # reached max_downsample_channels are well-formed there, so only those are drawn)
    pp = {"in_channels": 1, "out_channels": 1, "kernel_sizes": [pick([3, 5]), 3], "channels": pch,
          "downsample_scales": [pick([1, 2, 3]) for _ in range(pn)], "max_downsample_channels": min(32, pch * 4 ** (pn - 1)), "bias": True,
          "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1},
          "use_weight_norm": bool(rng.integers(2)), "use_spectral_norm": False}
    params = dict(scales=scales, scale_downsample_pooling="AvgPool1d", scale_downsample_pooling_params=pool, scale_discriminator_params=sp,
                  follow_official_norm=True, periods=periods, period_discriminator_params=pp)
    return params, int(rng.integers(1, 4)), int(rng.integers(300, 900))


@pytest.mark.parametrize("i", range(N_CASES))
def test_random_discriminator_vs_oracle(i):
    params, B, T = random_case(i)
    sd = synth_disc_state_dict(params, seed=50 + i)
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(**params)
    assert list(d.state_dict()) == list(sd)
    d.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    d = d.to("cuda:0")
    x_np = uniform(i, "x", (B, 1, T), -0.7, 0.7)
    x = torch.from_numpy(x_np).cuda().requires_grad_(True)
    outs = d(x)
    shapes = [[tuple(t.shape) for t in o] for o in outs]
    cots = [[uniform(i, f"cot.{a}.{b}", s, -1.0, 1.0) / np.sqrt(np.prod(s[1:])) for b, s in enumerate(o)] for a, o in enumerate(shapes)]
    ref_outs, ref = DO.disc_gradients(sd, params, x_np, cots)
    assert shapes == [[tuple(t.shape) for t in o] for o in ref_outs], (params, shapes)
    loss = 0.0
    for o, r, c in zip(outs, ref_outs, cots):
        for t, tr, ct in zip(o, r, c):
            scale = float(tr.abs().max().clamp_min(1e-6))
            assert float((t.detach().cpu() - tr).abs().max()) < 2e-5 * scale, params
            loss = loss + (t * torch.from_numpy(ct).cuda()).sum()
    loss.backward()
    got = {k: p.grad for k, p in d.named_parameters()}
    got["x"] = x.grad
    assert sorted(got) == sorted(ref)
    for k in ref:
        a, b = got[k].cpu().double().reshape(-1), ref[k].double().reshape(-1)
        err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        cos = 1.0 - float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))
        # (a pre-activation within rounding distance of the LeakyReLU kink flips a handful of elements: direction then decides)
        assert err < 2e-4 or cos < 1e-5, (k, err, cos, params)
