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
from articulatory_amd.utils.synth import disc_params  # noqa: E402
from bench import CAR_PARAMS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--recipe", default="car", choices=["car", "e2w", "mri"],
                help="car: e2w_hifigan_car.yaml (batch 64 x 2000 samples); e2w: e2w_hifigan.yaml (32 x 8000, mel hop 80); "
                     "mri: mri2w_hifigan_car.yaml (16 x 30000, 230-dim features, x240 upsampling, 20 kHz)")
ap.add_argument("--aux", default="mel", choices=["mel", "stft"], help="auxiliary loss: the shipped YAMLs' mel loss, or the multi-resolution "
                "STFT loss BASELINE config 5 names (reference defaults: fft 1024 / 2048 / 512)")
ap.add_argument("--fused-adam", action="store_true", help="config key fused_optimizers: torch.optim.Adam(fused=True)")
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--torch-profile", action="store_true")
a = ap.parse_args()
RECIPES = {  # (generator overrides, batch_size, batch_max_steps, hop, feature dims, mel fs, mel hop) — restated from the shipped YAMLs
    "car": ({}, 64, 2000, 80, 13, 16000, 256),
    "e2w": ({}, 32, 8000, 80, 13, 16000, 80),
    "mri": ({"in_channels": 358, "upsample_scales": [8, 5, 3, 2], "upsample_kernel_sizes": [16, 10, 6, 4], "final_scale": 240, "extra_art": False},
            16, 30000, 240, 230, 20000, 256),
}
g_over, r_batch, r_steps, r_hop, r_dims, r_fs, r_melhop = RECIPES[a.recipe]
a.batch = a.batch or r_batch
adam = {"lr": 1.0e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}
sched = {"gamma": 0.5, "milestones": [40000, 80000, 120000, 160000]}
config = dict(  # e2w_hifigan_car.yaml
    generator_type="HiFiGANGenerator", generator_params=dict(CAR_PARAMS, **g_over),
    discriminator_type="HiFiGANMultiScaleMultiPeriodDiscriminator",
    discriminator_params=dict(scale_discriminator_params=dict(disc_params()["scale_discriminator_params"], downsample_scales=[4, 4, 4, 4, 1])),
    use_stft_loss=a.aux == "stft", use_mel_loss=a.aux == "mel", stft_loss_params={},
    mel_loss_params=dict(fs=r_fs, fft_size=1024, hop_size=r_melhop, win_length=None, window="hann", num_mels=80, fmin=0, fmax=11025, log_base=None),
    generator_adv_loss_params={"average_by_discriminators": False}, discriminator_adv_loss_params={"average_by_discriminators": False},
    use_feat_match_loss=True, feat_match_loss_params={"average_by_discriminators": False, "average_by_layers": False, "include_final_outputs": False},
    lambda_aux=45.0, lambda_adv=1.0, lambda_feat_match=2.0, batch_size=a.batch, batch_max_steps=r_steps,
    generator_optimizer_type="Adam", generator_optimizer_params=adam, generator_scheduler_type="MultiStepLR", generator_scheduler_params=sched,
    generator_grad_norm=-1, discriminator_optimizer_type="Adam", discriminator_optimizer_params=adam, discriminator_scheduler_type="MultiStepLR",
    discriminator_scheduler_params=sched, discriminator_grad_norm=-1, discriminator_train_start_steps=0, distributed=False, fused_optimizers=a.fused_adam)
trainer = Trainer(config, torch.device("cuda"))
trainer.steps = 1  # past discriminator_train_start_steps: the full iteration
data = SyntheticPairs(a.batch, 2 * r_steps // r_hop, r_dims, r_hop, seed=0)
batch = WindowCollater(r_steps, r_hop, 512, np.random.default_rng(0))([data[i] for i in range(a.batch)])
for _ in range(3):
    log = trainer.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    log = trainer.train_step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"GAN iteration (G step + D step, recipe {a.recipe}, batch {a.batch} x {r_steps} samples): {dt * 1e3:.2f} ms, {a.batch / dt:.0f} windows/s, "
      + ", ".join(f"{k.split('/')[1]} {float(v):.4f}" for k, v in sorted(log.items())))
if a.torch_profile:
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            trainer.train_step(batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
