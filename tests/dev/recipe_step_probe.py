"""Per-tensor deviation of Trainer.train_step (full e2w_hifigan_car.yaml, batch 8) from the reference's _train_step fixture, next to the
reference's own fp32-vs-fp64 deviation (tests/golden/gold_train_step.npz).  python tests/dev/recipe_step_probe.py [mel|stft]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_recipe import D_TENSORS, G_TENSORS, GOLDEN, Trainer, make_batch, recipe_train_config, sampled, synth_disc_state_dict, synth_state_dict  # noqa: E402

aux = sys.argv[1] if len(sys.argv) > 1 else "mel"
gold = np.load(os.path.join(GOLDEN, "gold_train_step.npz"))
B = int(gold["B"])
seed_g, seed_d, seed_x = (int(s) for s in gold["seeds"])
config = recipe_train_config("car", aux=aux, batch=B)
t = Trainer(config, torch.device("cuda:0"))
gsd = synth_state_dict(config["generator_params"], seed=seed_g)
dsd = synth_disc_state_dict(config["discriminator_params"], seed=seed_d)
t.G.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
t.D.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
batch = {k: torch.from_numpy(v) for k, v in make_batch(config, seed_x, B).items()}
t.steps = 2
log = {k: float(v) for k, v in t.train_step(batch).items()}
for k in sorted(log):
    ref = float(gold[f"{aux}::log::{k}"])
    print(f"{k:40s} {log[k]:.7f} ref {ref:.7f} rel {abs(log[k] - ref) / abs(ref):.1e}")
lr = 1e-4
for net, names, module, sd in (("generator", G_TENSORS, t.G, gsd), ("discriminator", D_TENSORS, t.D, dsd)):
    params = dict(module.named_parameters())
    for n in names:
        g, gr = sampled(gold, f"{aux}::{net}::grad::{n}", params[n].grad)
        err = np.abs(g - gr) / max(np.abs(gr).max(), 1e-30)
        p, pr = sampled(gold, f"{aux}::{net}::new::{n}", params[n])
        old = sampled(gold, f"{aux}::{net}::new::{n}", torch.from_numpy(sd[n]))[0]
        d, dr = p - old, pr - old
        y = gold[f"{aux}::{net}::grad_f32_vs_f64::{n}"]
        print(f"{net[:3]} {n:45s} n={g.size:5d} grad err median {np.median(err):.1e} max {err.max():.1e} | ref fp32-vs-fp64 {y[0]:.1e} {y[1]:.1e} | "
              f"Adam delta agree {(np.abs(d - dr) < 0.02 * lr).mean():.3f}")
