"""Constructor options of the reference's HiFiGANGenerator that no shipped YAML uses, on a MI355X through the C ABI (round 5):
``use_additional_convs=False`` (articulatory/layers/residual_block.py:151, 191-205, 217-221), a FOURTH residual block per stage with unequal
dilation counts (articulatory/models/hifigan.py:134-145, 226-230) and ``nonlinear_activation="ReLU"`` (:40-41, 121-123) — against golden vectors of the REAL reference class
(oracle/make_golden_variants.py) and against the CPU oracle.  ``pytest -m gpu``."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err, same_across_shapes
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from oracle import hificar_oracle as O

pytestmark = pytest.mark.gpu
TAGS = ["noadd", "blocks4"]
FWD_TAGS = TAGS + ["relu"]  # nonlinear_activation="ReLU" (hifigan.py:40-41, 121-123): forward fixtures; its gradients against the float64 oracle below


def load(tag):
    g = np.load(os.path.join(GOLDEN, f"gold_variant_{tag}.npz"))
    return g, dict(ast.literal_eval(str(g["params"])))


def build(params, seed, precision="f32", train=False):
    assert torch.cuda.is_available()
    sd = synth_state_dict(params, seed=seed)
    m = HiFiGANGenerator(**params, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    if train:
        return m.train().to("cuda:0"), sd
    m.remove_weight_norm()
    return m.eval().to("cuda:0"), sd


@pytest.mark.parametrize("tag", FWD_TAGS)
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_forward_every_block_and_ar_loop_vs_reference_golden(tag, precision):
    g, params = load(tag)
    model, _ = build(params, 1234, precision)
    nb = len(params["resblock_kernel_sizes"])
    names = [f"upsamples.{i}" for i in range(4)] + [f"blocks.{b}" for b in range(4 * nb)]
    tol = 2e-5 if precision == "f32" else 2e-4
    c, ar = torch.from_numpy(g["c"]).cuda(), torch.from_numpy(g["ar"]).cuda()
    with torch.no_grad():
        y, taps = model.debug_taps(names, c, ar=ar)
        y2 = model(c, ar=ar)
    assert "libhificar.so" in open("/proc/self/maps").read()
    assert torch.equal(y, y2) and rel_err(y.cpu().numpy(), g["out"]) < tol
    for n in names:
        assert rel_err(taps[n].cpu().numpy(), g["tap::" + n]) < tol, n
    x = torch.from_numpy(g["arloop_x"]).cuda()
    with torch.no_grad():
        w = model.ar_synthesis(x.t().unsqueeze(0).contiguous(), 25)
    assert w.shape == (1, 80 * 60) and rel_err(w[0].cpu().numpy(), g["arloop_out"]) < tol


@pytest.mark.parametrize("tag", TAGS)
def test_gradients_vs_reference_golden(tag):
    """Every parameter's gradient (weight norm in the graph), c and ar, LeakyReLU slope 1 (no kinks), against the real reference under autograd."""
    g, params = load(tag)
    gp = dict(params, nonlinear_activation_params={"negative_slope": 1.0})
    model, _ = build(gp, int(g["gseed"]), train=True)
    c = torch.from_numpy(g["gc"]).cuda().requires_grad_(True)
    ar = torch.from_numpy(g["gar"]).cuda().requires_grad_(True)
    y = model(c, ar=ar)
    assert y.requires_grad and O.check_packed(g, "gout", y, 2e-5) < 2e-5
    (y * torch.from_numpy(g["gcot"]).cuda()).sum().backward()
    worst = {}
    for k, p in list(model.named_parameters()) + [("c", c), ("ar", ar)]:
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
        worst[k] = O.check_packed(g, "grad::" + k, p.grad, 2e-4)
    bad = {k: v for k, v in worst.items() if v >= 2e-4}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


@pytest.mark.parametrize("tag", TAGS)
def test_full_width_ragged_batch_vs_oracle(tag):
    """The variant at the recipe's width (channels 512: the wide-stage kernels, a fourth branch in its own launch), a ragged batch of three
    utterances through the batched AR loop against the oracle's per-utterance loop; batch composition does not change an utterance's waveform."""
    _, small = load(tag)
    params = dict(small, channels=512)
    model, sd = build(params, 77)
    w = O.fold_weight_norm(sd)
    lens = [37, 25, 9]
    x = torch.from_numpy(synth_features(3, 37, 13, seed=78))
    with torch.no_grad():
        y = model.ar_synthesis(x.permute(0, 2, 1).contiguous().cuda(), 25, lengths=lens).cpu()
        y0 = model.ar_synthesis(x[:1].permute(0, 2, 1).contiguous().cuda(), 25).cpu()
    assert same_across_shapes(y[0], y0[0])
    for b, n in enumerate(lens):
        with torch.no_grad():
            ref = O.ar_loop(w, params, x[b, :n], 2000, 80)
        assert rel_err(y[b, :80 * n].numpy(), ref.numpy()) < 2e-5, b
        assert float(y[b, 80 * n:].abs().sum()) == 0.0


@pytest.mark.parametrize("tag", FWD_TAGS)
def test_training_iterations_move_the_variant(tag):
    """A few Adam steps through the autograd node: the loss falls, every parameter has a finite gradient — and, with real LeakyReLU slope, the
    gradients agree with the float64 oracle in L2 (flip-robust)."""
    _, params = load(tag)
    model, sd = build(params, 91, train=True)
    B, T = 2, 6
    c_np = synth_features(B, T, 13, seed=92).transpose(0, 2, 1).copy()
    ar_np = np.zeros((B, 1, 512), np.float32)
    cot = np.ones((B, 1, 80 * T), np.float32)
    c, ar = torch.from_numpy(c_np).cuda(), torch.from_numpy(ar_np).cuda()
    y = model(c, ar=ar)
    y.sum().backward()
    _, ref = O.gradients(sd, params, c_np, ar_np, cot, dtype=torch.float64)
    errs = []
    for k, p in model.named_parameters():
        r = ref[k].numpy()
        errs.append(float(np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - r) / max(np.linalg.norm(r), 1e-30)))
    assert np.median(errs) < 1e-5 and max(errs) < 0.1, (np.median(errs), max(errs))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    target = torch.zeros(B, 1, 80 * T).cuda()
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = (model(c, ar=ar) - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
