"""Dev probe: per fixture tensor of the recipe-size train-step goldens, the device's gradient deviation from the reference's fp32 run (median /
max relative to the tensor's scale) next to the reference's OWN fp32-vs-fp64 deviation stored in the fixture.  python tests/dev/recipe_yardstick_probe.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from articulatory_amd.bin.train import Trainer  # noqa: E402
from articulatory_amd.utils.recipes import recipe_train_config  # noqa: E402
from articulatory_amd.utils.synth import synth_disc_state_dict, synth_state_dict  # noqa: E402
from oracle.make_golden_train import D_TENSORS, G_TENSORS, linearize, make_batch  # noqa: E402
from test_gpu_recipe import sampled  # noqa: E402

FILES = {"car": "gold_train_step.npz", "car_lin": "gold_train_step_lin.npz", "e2w": "gold_train_step_e2w.npz", "mri": "gold_train_step_mri.npz"}
for recipe, aux in (("car_lin", "mel"), ("car", "mel"), ("car", "stft"), ("e2w", "mel"), ("mri", "mel")):
    gold = np.load(os.path.join(REPO, "tests", "golden", FILES[recipe]))
    B = int(gold["B"])
    seed_g, seed_d, seed_x = (int(s) for s in gold["seeds"])
    config = recipe_train_config("car" if recipe == "car_lin" else recipe, aux=aux, batch=B)
    if recipe == "car_lin":
        linearize(config)
    t = Trainer(config, torch.device("cuda:0"))
    t.G.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(config["generator_params"], seed=seed_g).items()})
    t.D.load_state_dict({k: torch.from_numpy(v) for k, v in synth_disc_state_dict(config["discriminator_params"], seed=seed_d).items()})
    batch = {k: torch.from_numpy(v) for k, v in make_batch(config, seed_x, B).items()}
    t.steps = 2
    log = {k: float(v) for k, v in t.train_step(batch).items()}
    print(f"==== {recipe} / {aux}: worst logged-loss deviation",
          max(abs(log[k[len(aux) + 7:]] - float(gold[k])) / max(abs(float(gold[k])), 1e-3) for k in gold.files if k.startswith(f"{aux}::log::")))
    for net, names, module in (("generator", G_TENSORS, t.G), ("discriminator", D_TENSORS, t.D)):
        params = dict(module.named_parameters())
        for n in names:
            g, gr = sampled(gold, f"{aux}::{net}::grad::{n}", params[n].grad)
            err = np.abs(g - gr) / max(np.abs(gr).max(), 1e-30)
            yard = gold[f"{aux}::{net}::grad_f32_vs_f64::{n}"]
            print(f"  {net[:3]} {n:44s} dev median {np.median(err):.1e} max {err.max():.1e} | ref fp32-vs-fp64 median {yard[0]:.1e} max {yard[1]:.1e}")
    del t
