"""The shipped recipes at their real size on a MI355X against golden vectors of the REAL reference (``pytest -m gpu``):

* BASELINE config 5 — one whole GAN iteration of egs/ema/voc1/conf/e2w_hifigan_car.yaml (full HiFi-CAR generator, 70.7 M-parameter
  multi-scale multi-period discriminator, Adam, shipped loss weights) through ``Trainer.train_step`` against the reference's own
  ``Trainer._train_step`` (articulatory/bin/train.py:241-440) run by oracle/make_golden_train.py, with the shipped mel loss and with the
  multi-resolution STFT loss config 5 names: every logged loss, gradients and post-update values of a handful of tensors.
* the shipped MRI generator (egs/mri/voc1/conf/mri2w_hifigan_car.yaml:34-58: in_channels 358, x240 upsampling): forward, ar_loop and
  gradients against oracle/make_golden_mri.py's fixtures."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from articulatory_amd.bin.train import Trainer
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.recipes import recipe_train_config
from articulatory_amd.utils.synth import synth_disc_state_dict, synth_state_dict
from oracle import hificar_oracle as O
from oracle.make_golden_train import D_TENSORS, G_TENSORS, linearize, make_batch

pytestmark = pytest.mark.gpu


def sampled(gold, name, arr):
    """(device values, golden values) of a fixture entry written by oracle/make_golden_grad.py::pack."""
    flat = np.asarray(arr.detach().cpu(), dtype=np.float64).reshape(-1)
    if name + "::full" in gold:
        return flat, gold[name + "::full"].astype(np.float64)
    return flat[gold[name + "::idx"]], gold[name + "::vals"].astype(np.float64)


FIXTURES = {"car": "gold_train_step.npz", "car_lin": "gold_train_step_lin.npz", "e2w": "gold_train_step_e2w.npz", "mri": "gold_train_step_mri.npz"}
# The one fixture tensor whose device gradient sits further from the reference's fp32 run than a small multiple of the reference's own
# fp32-vs-fp64 deviation: the period-2 discriminator's first layer (Conv2d 1 -> 32, 160 weights, each a sum over ~20 000 positions that nearly
# cancels on the "car" batch: 3.0e-4 median against the yardstick's 1.0e-5; on the other three fixtures the same tensor is AT its yardstick,
# 7e-6 .. 1.5e-4).  The weight-gradient kernel accumulates a row split's positions sequentially in the MFMA accumulator; oneDNN blocks them.
CANCELLING = {("car", "mpd.discriminators.0.convs.0.0.weight_v")}


# (recipe, auxiliary loss): config 5's recipe with both losses; the other two shipped recipes — e2w_hifigan.yaml (8000-sample windows) and
# mri2w_hifigan_car.yaml (230-dim features, x240, 30000-sample windows) — with the mel loss they ship with (oracle/make_golden_train.py --recipe);
# "car_lin": config 5's recipe with every LeakyReLU slope of both networks set to 1 (round 4: the assembled iteration without its activation kinks)
@pytest.mark.parametrize("recipe,aux", [("car", "mel"), ("car", "stft"), ("e2w", "mel"), ("mri", "mel"), ("car_lin", "mel")])
def test_full_recipe_iteration_vs_reference_train_step(recipe, aux):
    """What "equal" can mean here.  The fixtures carry, per tensor, the reference's OWN fp32-vs-fp64 gradient deviation (the same _train_step run
    in float64).  It is NOT small and NOT a kink effect: with every LeakyReLU slope at 1 ("car_lin") the reference's fp32 gradients sit 1e-4 .. 3e-4
    (median; 1e-3 .. 3.5e-3 max) of each tensor's scale from its float64 ones — the mel loss's log and the un-normalised 80-conv network amplify
    fp32 rounding that far — and its fp32 fake / discriminator loss 1.5e-4 from the float64 value (the discriminator part runs on the generator
    AFTER its first Adam step, a sign function of those gradients).  So the assembled step is held to a small multiple of that yardstick, tensor by
    tensor: measured 1 - 6 x (tests/dev/recipe_yardstick_probe.py), i.e. the device is as close to the reference's fp32 run as two correct fp32
    implementations of this step can be expected to be.  (Round 3's review hoped a slope-1 fixture would allow 1e-5: the yardstick says no.)"""
    gold = np.load(os.path.join(GOLDEN, FIXTURES[recipe]))
    B = int(gold["B"])
    seed_g, seed_d, seed_x = (int(s) for s in gold["seeds"])
    config = recipe_train_config("car" if recipe == "car_lin" else recipe, aux=aux, batch=B)
    if recipe == "car_lin":
        linearize(config)
    t = Trainer(config, torch.device("cuda:0"))
    gsd = synth_state_dict(config["generator_params"], seed=seed_g)
    dsd = synth_disc_state_dict(config["discriminator_params"], seed=seed_d)
    t.G.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
    t.D.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    batch = {k: torch.from_numpy(v) for k, v in make_batch(config, seed_x, B).items()}
    t.steps = 2  # past generator_train_start_steps (1) and discriminator_train_start_steps (0), as in the fixture
    log = {k: float(v) for k, v in t.train_step(batch).items()}
    assert "libhificar.so" in open("/proc/self/maps").read()
    # ---- every logged loss (seven with the mel loss, eight with the two STFT terms): 1e-4 (+ 3 x the reference's own fp32-vs-fp64 gap where the
    # fixture stores it)
    keys = [k[len(aux) + 7:] for k in gold.files if k.startswith(f"{aux}::log::")]
    assert sorted(keys) == sorted(log) and len(keys) == (7 if aux == "mel" else 8)
    for k in keys:
        ref = float(gold[f"{aux}::log::{k}"])
        slack = 3.0 * abs(ref - float(gold[f"{aux}::log64::{k}"])) if f"{aux}::log64::{k}" in gold.files else 0.0
        assert abs(log[k] - ref) < 1e-4 * max(abs(ref), 1e-3) + slack, (k, log[k], ref)
    # ---- gradients left in .grad and parameters after the Adam step, for the fixture's tensors
    lr = config["generator_optimizer_params"]["lr"]
    k_med, k_max = (6.5, 4.0) if recipe == "car_lin" else (20.0, 8.0)  # (twice what was measured — <= 3.2 / 1.9 and <= 12.2 / 3.8 — or less)
    for net, names, module, sd in (("generator", G_TENSORS, t.G, gsd), ("discriminator", D_TENSORS, t.D, dsd)):
        params = dict(module.named_parameters())
        for n in names:
            g, gr = sampled(gold, f"{aux}::{net}::grad::{n}", params[n].grad)
            err = np.abs(g - gr) / max(np.abs(gr).max(), 1e-30)
            yard_med, yard_max = (float(v) for v in gold[f"{aux}::{net}::grad_f32_vs_f64::{n}"])
            if (recipe, n) in CANCELLING:  # (see above) the absolute bounds of round 3 only
                assert np.median(err) < 1e-3 and err.max() < 5e-3, (net, n, float(np.median(err)), float(err.max()))
            else:
                assert np.median(err) <= k_med * yard_med + 2e-6, (net, n, float(np.median(err)), yard_med)
                assert err.max() <= k_max * yard_max + 5e-5, (net, n, float(err.max()), yard_max)
            assert np.median(err) < 1e-3 and err.max() < 5e-3  # (and never worse than this in absolute terms: round 3 allowed 5e-2)
            p, pr = sampled(gold, f"{aux}::{net}::new::{n}", params[n])
            old = sampled(gold, f"{aux}::{net}::new::{n}", torch.from_numpy(sd[n]))[0]
            d, dr = p - old, pr - old
            assert np.abs(dr).max() > 0.5 * lr                                        # (the fixture's tensors all move by about lr)
            # first Adam step = -lr * g / (|g| + eps): elements whose gradient is not noise agree to a fraction of lr
            assert (np.abs(d - dr) < 0.02 * lr).mean() > 0.95, (net, n, float((np.abs(d - dr) < 0.02 * lr).mean()))
    assert t.steps == 3


MRI_OVER = dict(in_channels=358, upsample_scales=[8, 5, 3, 2], upsample_kernel_sizes=[16, 10, 6, 4], final_scale=240)


def mri_generator(seed, slope=0.1, train=False):
    params = dict(recipe_train_config("mri")["generator_params"], nonlinear_activation_params={"negative_slope": slope})
    assert all(params[k] == v for k, v in MRI_OVER.items())
    g = HiFiGANGenerator(**params, precision="f32")
    ref_params = {k: v for k, v in params.items() if k not in ("final_scale", "extra_art")}
    sd = synth_state_dict(ref_params, seed=seed)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return (g.train() if train else g.eval()).to("cuda:0"), ref_params, sd


def test_shipped_mri_generator_forward_and_ar_loop_vs_reference_golden():
    """mri2w_hifigan_car.yaml's generator (230 feature dims + 128 AR, x240): one forward (B = 2, T = 25, every output sample and the
    sampled stage outputs) and the reference's ar_loop at the recipe's batch_max_steps 30000 (chunk 125 + a 15-frame tail)."""
    gold = np.load(os.path.join(GOLDEN, "gold_mri_fwd.npz"))
    g, params, _ = mri_generator(1234)
    g.remove_weight_norm()
    g = g.eval()
    c, ar = torch.from_numpy(gold["c"]).cuda(), torch.from_numpy(gold["ar"]).cuda()
    names = [f"upsamples.{i}" for i in range(4)]
    with torch.no_grad():
        y, taps = g.debug_taps(names, c, ar=ar)
    assert y.shape == (2, 1, 6000) and rel_err(y.cpu().numpy(), gold["out"]) < 2e-5
    for i, n in enumerate(names):
        assert O.check_packed(gold, f"stage::up{i}", taps[n], 2e-5) < 2e-5, n
    x = torch.from_numpy(gold["arloop_x"]).cuda()  # (140, 230)
    with torch.no_grad():
        w = g.ar_synthesis(x.t().unsqueeze(0).contiguous(), 125)
    assert w.shape == (1, 140 * 240) and rel_err(w[0].cpu().numpy(), gold["arloop_out"]) < 2e-5
    assert "libhificar.so" in open("/proc/self/maps").read()


def test_shipped_mri_generator_gradients_vs_reference_golden():
    """Gradients of every parameter / c / ar of the shipped MRI generator (LeakyReLU slope 1: no kinks, see oracle/make_golden_grad.py)
    against the reference under autograd."""
    gold = np.load(os.path.join(GOLDEN, "gold_mri_grad.npz"))
    g, params, sd = mri_generator(int(gold["seed"]), slope=1.0, train=True)
    c = torch.from_numpy(gold["c"]).cuda().requires_grad_(True)
    ar = torch.from_numpy(gold["ar"]).cuda().requires_grad_(True)
    y = g(c, ar=ar)
    assert y.requires_grad and O.check_packed(gold, "out", y, 2e-5) < 2e-5
    (y * torch.from_numpy(gold["cot"]).cuda()).sum().backward()
    worst = {}
    for k, p in list(g.named_parameters()) + [("c", c), ("ar", ar)]:
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
        worst[k] = O.check_packed(gold, "grad::" + k, p.grad, 2e-4)
    bad = {k: v for k, v in worst.items() if v >= 2e-4}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
