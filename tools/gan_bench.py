#!/usr/bin/env python3
"""One full GAN training iteration of the HiFi-CAR recipe (BASELINE config 5's step on one GPU: e2w_hifigan_car.yaml — batch 64,
batch_max_steps 2000, generator + multi-scale multi-period discriminator, mel + adversarial + feature-matching losses, Adam):
   python tools/gan_bench.py [--batch 64] [--steps 10]
Every step runs Trainer.train_step of articulatory_amd/bin/train.py on synthetic windows."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from articulatory_amd.bin.train import SyntheticPairs, Trainer, WindowCollater  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--recipe", default="car", choices=["car", "e2w", "mri"],
                help="car: e2w_hifigan_car.yaml (batch 64 x 2000 samples); e2w: e2w_hifigan.yaml (32 x 8000, mel hop 80); "
                     "mri: mri2w_hifigan_car.yaml (16 x 30000, 230-dim features, x240 upsampling, 20 kHz)")
ap.add_argument("--aux", default="mel", choices=["mel", "stft"], help="auxiliary loss: the shipped YAMLs' mel loss, or the multi-resolution "
                "STFT loss BASELINE config 5 names (reference defaults: fft 1024 / 2048 / 512)")
ap.add_argument("--foreach-adam", action="store_true", help="config key fused_optimizers: false — torch's default foreach Adam instead of the fused kernel")
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--torch-profile", action="store_true")
ap.add_argument("--pageable", action="store_true", help="keep the batch in pageable host memory (every .to(device) then drains the stream)")
ap.add_argument("--serial-aux", action="store_true", help="config key overlap_aux_loss: false — the auxiliary loss on the main stream (A/B)")
ap.add_argument("--late-real", action="store_true", help="config key early_real_gradient: false — both backward passes of the discriminator update inside it (A/B)")
a = ap.parse_args()
from articulatory_amd.utils.recipes import recipe_train_config  # noqa: E402

config = recipe_train_config(a.recipe, aux=a.aux, batch=a.batch, fused_optimizers=not a.foreach_adam)
config["overlap_aux_loss"] = not a.serial_aux
config["early_real_gradient"] = not a.late_real
a.batch = config["batch_size"]
r_steps = config["batch_max_steps"]
r_hop = int(np.prod(config["generator_params"]["upsample_scales"]))
r_dims = config["generator_params"]["in_channels"] - config["generator_params"]["ar_output"]
trainer = Trainer(config, torch.device("cuda"))
trainer.steps = 1  # past discriminator_train_start_steps: the full iteration
data = SyntheticPairs(a.batch, 2 * r_steps // r_hop, r_dims, r_hop, seed=0)
batch = WindowCollater(r_steps, r_hop, 512, np.random.default_rng(0))([data[i] for i in range(a.batch)])
if not a.pageable:  # the recipe's DataLoader delivers pinned batches (pin_memory: true): the copies to the device are asynchronous
    batch = {k: v.pin_memory() for k, v in batch.items()}
for _ in range(3):
    log = trainer.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    log = trainer.train_step(batch)
t_host = (time.perf_counter() - t0) / a.steps  # host time to ENQUEUE an iteration (the GPU runs behind it)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"host enqueue time per iteration {t_host * 1e3:.2f} ms (wall {dt * 1e3:.2f} ms)")
print(f"GAN iteration (G step + D step, recipe {a.recipe}, batch {a.batch} x {r_steps} samples): {dt * 1e3:.2f} ms, {a.batch / dt:.0f} windows/s, "
      + ", ".join(f"{k.split('/')[1]} {float(v):.4f}" for k, v in sorted(log.items())))
if a.torch_profile:
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            trainer.train_step(batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
