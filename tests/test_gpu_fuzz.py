"""Seeded random-configuration parity fuzz of the HIP path against the CPU oracle (``pytest -m gpu``).
Hyper-parameters, batch, length, ragged lengths and arithmetic are drawn at random from what hificar_create
accepts; HIFICAR_FUZZ_CASES raises the number of cases (default 64, about a second each on the GPU box)."""

import os

import numpy as np
import pytest
import torch

from conftest import E2W_PARAMS, rel_err, same_across_shapes
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from oracle import hificar_oracle as O

pytestmark = pytest.mark.gpu
TOLS = {"f32": 2e-5, "bf16x3": 2e-4}
N_CASES = int(os.environ.get("HIFICAR_FUZZ_CASES", "64"))


def draw(rng):
    n_stages = int(rng.integers(1, 5))
    channels = int(rng.choice([24, 48, 64, 100, 128, 256, 512]))  # any width: narrow stages are padded to 32 channels internally
    while channels >> n_stages < 1:
        n_stages -= 1
    scales = [int(rng.choice([2, 3, 4, 5, 8])) for _ in range(n_stages)]
    n_blocks = int(rng.integers(1, 4))
    ks = [int(rng.choice([3, 5, 7, 9, 11, 13, 15])) for _ in range(n_blocks)]
    dils = [[int(rng.integers(1, 10)) for _ in range(int(rng.integers(1, 4)))] for _ in range(n_blocks)]
    use_ar = bool(rng.integers(0, 2))
    cf = int(rng.integers(1, 90)) if rng.integers(0, 5) else int(rng.integers(90, 400))
    params = dict(E2W_PARAMS, channels=channels, kernel_size=int(rng.choice([3, 5, 7, 9])), upsample_scales=scales,
                  upsample_kernel_sizes=[2 * s for s in scales], resblock_kernel_sizes=ks, resblock_dilations=dils,
                  use_ar=use_ar, in_channels=cf + (128 if use_ar else 0), bias=bool(rng.integers(0, 4)),
                  use_tanh=bool(rng.integers(0, 4)),
                  nonlinear_activation_params={"negative_slope": float(rng.choice([0.1, 0.1, 0.2, 0.01, 0.5, 0.0, 1.0]))})
    return params, cf


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_configuration(case):
    assert torch.cuda.is_available()
    rng = np.random.default_rng(1000 + case)
    params, cf = draw(rng)
    prec = "f32" if case % 2 else "bf16x3"
    hop = int(np.prod(params["upsample_scales"]))
    sd = synth_state_dict(params, seed=500 + case)
    g = HiFiGANGenerator(**params, precision=prec)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    g = g.eval().cuda()
    w = O.fold_weight_norm(sd)
    B = int(rng.integers(1, 6)) if rng.integers(0, 6) else int(rng.integers(6, 80))
    T = int(rng.integers(1, 70)) if rng.integers(0, 4) else int(rng.integers(70, 500))  # mostly short, sometimes many tiles per sequence
    lens = [int(v) for v in rng.integers(0, T + 1, size=B)]
    lens[int(rng.integers(0, B))] = T
    x = synth_features(B, T, cf, seed=case)
    c = torch.from_numpy(x).permute(0, 2, 1).contiguous()
    ar = torch.from_numpy(synth_features(B, 512, 1, seed=case + 1)[:, :, 0] * 0.3).reshape(B, 1, 512) if params["use_ar"] else None
    with torch.no_grad():
        y = g(c.cuda(), ar=ar.cuda() if ar is not None else None).cpu()
        yr = g(c.cuda(), ar=ar.cuda() if ar is not None else None, lengths=lens).cpu()
        ref = O.generator_forward(w, params, c, ar)
        pre = O.generator_forward(w, dict(params, use_tanh=False), c, ar) if params["use_tanh"] else ref
    # the arithmetic's error is a fixed fraction of the PRE-tanh scale (conditioning): a configuration
    # that drives tanh into saturation is held to that, not to the squashed output's scale
    tol = TOLS[prec] * max(1.0, float(pre.abs().max() / ref.abs().max()))
    tag = (case, prec, {k: params[k] for k in ("channels", "kernel_size", "upsample_scales", "resblock_kernel_sizes",
                                                "resblock_dilations", "use_ar", "in_channels", "bias")}, B, T, lens)
    assert y.shape == ref.shape == (B, 1, hop * T), tag
    assert rel_err(y.numpy(), ref.numpy()) < tol, tag
    for b, n in enumerate(lens):  # ragged: each utterance as if alone (zero padding at its own end), nothing beyond it
        assert float(yr[b, :, hop * n:].abs().sum()) == 0.0, tag
        if n:
            with torch.no_grad():
                alone = O.generator_forward(w, params, c[b:b + 1, :, :n], ar[b:b + 1] if ar is not None else None)
            assert rel_err(yr[b:b + 1, :, :hop * n].numpy(), alone.numpy()) < tol * max(1.0, float(ref.abs().max() / alone.abs().max())), tag


@pytest.mark.parametrize("case", range(max(1, N_CASES // 2)))
def test_random_ar_dataset(case):
    """Random HiFi-CAR configuration, random chunk length, a random list of utterance lengths through the continuously
    batched loop (random number in flight) and through the padded-batch loop: each utterance equals the oracle's batch-1
    ``ar_loop`` (decode.py:54-83) and the two device paths agree bit for bit."""
    assert torch.cuda.is_available()
    rng = np.random.default_rng(7000 + case)
    params, cf = draw(rng)
    params.update(use_ar=True, in_channels=cf + 128)
    prec = "bf16x3" if case % 2 else "f32"
    hop = int(np.prod(params["upsample_scales"]))
    chunk = int(rng.integers(-(-512 // hop), -(-512 // hop) + 20))  # ar_input <= hop * chunk (the reference's well-formed case)
    sd = synth_state_dict(params, seed=900 + case)
    g = HiFiGANGenerator(**params, precision=prec)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    g = g.eval().cuda()
    w = O.fold_weight_norm(sd)
    N = int(rng.integers(1, 7))
    lens = sorted((int(v) for v in rng.integers(0, 3 * chunk + 2, size=N)), reverse=True)
    if lens[0] == 0:
        lens[0] = chunk + 1
    Tm = lens[0]
    x = synth_features(N, Tm, cf, seed=case)
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        yp = g.ar_synthesis_packed(feats, chunk, lens, batch=int(rng.integers(1, N + 1)))
        yr = g.ar_synthesis(feats, chunk, lengths=lens)
    tag = (case, prec, chunk, lens, {k: params[k] for k in ("channels", "kernel_size", "upsample_scales", "resblock_kernel_sizes",
                                                             "resblock_dilations", "in_channels")})
    assert same_across_shapes(yp, yr, 5e-6 if prec == "f32" else 2e-4), tag
    for b, n in enumerate(lens):
        assert float(yp[b, hop * n:].abs().sum()) == 0.0, tag
        if n:
            with torch.no_grad():
                ref = O.ar_loop(w, params, torch.from_numpy(x[b, :n]), hop * chunk, hop)
            # a configuration that drives tanh towards saturation feeds rounding differences back through the AR context
            # (conditioning): those are held to the north-star bar, the others to the tight one
            tol = TOLS[prec] if float(ref.abs().max()) < 0.5 else 1e-3
            assert rel_err(yp[b, :hop * n].cpu().numpy(), ref.numpy()) < tol, tag
