#!/usr/bin/env python3
"""Golden vectors of the REAL reference's training iteration — ``articulatory.bin.train.Trainer._train_step`` (train.py:241-440) run
unmodified on the FULL shipped recipe egs/ema/voc1/conf/e2w_hifigan_car.yaml (13.5 M-parameter HiFi-CAR generator, 70.7 M-parameter
multi-scale multi-period discriminator, Adam, all loss weights as shipped) at batch 8 x (2000 samples + 512 AR context samples), with

  mel    the shipped auxiliary loss (use_mel_loss, mel_loss_params of the YAML)
  stft   BASELINE config 5's variant: use_stft_loss with the reference's default resolutions instead

Stored: every value the step adds to ``total_train_loss`` (the logged losses), and — for a handful of generator and discriminator
tensors — the gradient left in ``.grad`` and the parameter after the Adam update (sampled, oracle/make_golden_grad.py::pack).
The real Trainer class is constructed with the reference's own loss modules and torch optimizers / schedulers exactly as the
reference's ``main`` does (train.py:1649-1789); import shims as in oracle/make_golden.py and oracle/make_golden_loss.py (torch.stft
``return_complex=False``, restated librosa mel basis, a no-op tensorboardX.SummaryWriter and tqdm handle).  Fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_train.py
"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from make_golden import REF, import_reference  # noqa: E402
from make_golden_grad import pack  # noqa: E402

G_TENSORS = ["input_conv.weight_v", "input_conv.bias", "upsamples.0.1.weight_g", "upsamples.2.1.weight_v", "blocks.0.convs1.0.1.weight_v",
             "blocks.5.convs2.1.1.weight_g", "blocks.11.convs2.2.1.bias", "output_conv.1.weight_v", "ar_model.model.0.weight", "ar_model.model.8.bias"]
D_TENSORS = ["msd.discriminators.0.layers.0.0.weight", "msd.discriminators.1.layers.2.0.weight", "msd.discriminators.2.layers.7.bias", "msd.discriminators.2.layers.4.0.weight",
             "mpd.discriminators.0.convs.0.0.weight_v", "mpd.discriminators.4.convs.3.0.weight_g", "mpd.discriminators.2.output_conv.bias",
             "mpd.discriminators.3.convs.4.0.weight_v"]
STFT_DEFAULTS = {"fft_sizes": [1024, 2048, 512], "hop_sizes": [120, 240, 50], "win_lengths": [600, 1200, 240], "window": "hann_window"}
B, SEED_G, SEED_D, SEED_X = 8, 41, 42, 43
# recipe -> (YAML under /root/reference, batch, fixture file, auxiliary-loss variants).  "car" is the fixture described above; the other two
# shipped recipes (same Trainer, same checks) at a batch the CPU finishes in minutes:  --recipe e2w | mri
#   e2w  egs/ema/voc1/conf/e2w_hifigan.yaml          the same networks on 8000-sample windows (100 frames: several row chunks per sequence)
#   mri  egs/mri/voc1/conf/mri2w_hifigan_car.yaml    230-dim features, x240 upsampling (scales 8, 5, 3, 2), 30000-sample windows at 20 kHz
#   car_lin  the "car" recipe with every LeakyReLU SLOPE SET TO 1 in both networks (generator_params / scale_ / period_discriminator_params
#            nonlinear_activation_params): the assembled iteration without the kinks that make recipe-size gradients a coin flip per activation
#            (what remains: the generator's hard-coded LeakyReLU(0.01) in front of its output conv and LeakyReLU(0.1) in the PastFCEncoder,
#            |.| of the L1 mel / feature-matching terms) — lets the device's assembled step be held element-wise, see tests/test_gpu_recipe.py
RECIPES = {"car": ("egs/ema/voc1/conf/e2w_hifigan_car.yaml", 8, "gold_train_step.npz", ("mel", "stft")),
           "car_lin": ("egs/ema/voc1/conf/e2w_hifigan_car.yaml", 8, "gold_train_step_lin.npz", ("mel",)),
           "e2w": ("egs/ema/voc1/conf/e2w_hifigan.yaml", 4, "gold_train_step_e2w.npz", ("mel",)),
           "mri": ("egs/mri/voc1/conf/mri2w_hifigan_car.yaml", 2, "gold_train_step_mri.npz", ("mel",))}


def recipe_config(recipe="car"):
    import yaml

    with open(os.path.join(REF, RECIPES[recipe][0])) as f:
        cfg = yaml.safe_load(f)
    cfg["generator_params"] = {k: v for k, v in cfg["generator_params"].items() if k not in ("final_scale", "extra_art")}
    if recipe == "car_lin":
        linearize(cfg)
    return cfg


def linearize(cfg):
    """Every configurable LeakyReLU slope of both networks -> 1 (in place; also used by the tests on the package's own config)."""
    cfg["generator_params"] = dict(cfg["generator_params"], nonlinear_activation_params={"negative_slope": 1.0})
    dp = cfg["discriminator_params"] = dict(cfg["discriminator_params"])
    for k in ("scale_discriminator_params", "period_discriminator_params"):
        dp[k] = dict(dp[k], nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 1.0})
    return cfg


def make_batch(cfg, seed=SEED_X, batch=B):
    from articulatory_amd.utils.synth import synth_train_batch

    return synth_train_batch(cfg, seed, batch)


def main():
    import argparse

    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--recipe", default="car", choices=sorted(RECIPES))
    recipe = ap.parse_args().recipe
    _, batch_size, fixture, auxes = RECIPES[recipe]

    from articulatory_amd.utils.synth import synth_disc_state_dict, synth_state_dict
    from disc_oracle import mel_filterbank

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, _, _ = import_reference()
    sys.modules["tensorboardX"].SummaryWriter = lambda *a, **k: None
    lib = sys.modules["librosa"]
    lib.filters = types.ModuleType("librosa.filters")
    lib.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax: mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    real_stft = torch.stft

    def stft_compat(*a, return_complex=None, **kw):
        if return_complex is False:
            return torch.view_as_real(real_stft(*a, return_complex=True, **kw))
        return real_stft(*a, return_complex=return_complex, **kw)

    torch.stft = stft_compat
    try:
        import articulatory.bin.train as ref_train
        from articulatory.losses import (DiscriminatorAdversarialLoss, FeatureMatchLoss, GeneratorAdversarialLoss, MelSpectrogramLoss,
                                         MultiResolutionSTFTLoss)

        out = {}
        for aux in auxes:
            kept = {}
            for dtype in (torch.float32, torch.float64):  # float64: the same step again, a yardstick for the tests' tolerances only
                cfg = recipe_config(recipe)
                cfg["outdir"] = "/tmp"
                if aux == "stft":
                    cfg.update(use_stft_loss=True, use_mel_loss=False, stft_loss_params=dict(STFT_DEFAULTS))
                for flag in ("use_subband_stft_loss", "use_inter_loss", "use_ph_loss"):
                    cfg.setdefault(flag, False)
                gsd = synth_state_dict(cfg["generator_params"], seed=SEED_G)
                dsd = synth_disc_state_dict(cfg["discriminator_params"], seed=SEED_D)
                model = {"generator": getattr(ref_models, cfg["generator_type"])(**cfg["generator_params"]),
                         "discriminator": getattr(ref_models, cfg["discriminator_type"])(**cfg["discriminator_params"])}
                model["generator"].load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
                model["discriminator"].load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
                criterion = {"gen_adv": GeneratorAdversarialLoss(**cfg["generator_adv_loss_params"]),
                             "dis_adv": DiscriminatorAdversarialLoss(**cfg["discriminator_adv_loss_params"]),
                             "feat_match": FeatureMatchLoss(**cfg["feat_match_loss_params"])}
                if cfg["use_mel_loss"]:
                    criterion["mel"] = MelSpectrogramLoss(**cfg["mel_loss_params"])
                if cfg["use_stft_loss"]:
                    criterion["stft"] = MultiResolutionSTFTLoss(**cfg["stft_loss_params"])
                for m in list(model.values()) + list(criterion.values()):
                    m.to(dtype)
                optimizer = {k: getattr(torch.optim, cfg[f"{k}_optimizer_type"])(model[k].parameters(), **cfg[f"{k}_optimizer_params"])
                             for k in ("generator", "discriminator")}
                scheduler = {k: getattr(torch.optim.lr_scheduler, cfg[f"{k}_scheduler_type"])(optimizer=optimizer[k], **cfg[f"{k}_scheduler_params"])
                             for k in ("generator", "discriminator")}
                trainer = ref_train.Trainer(steps=2, epochs=0, data_loader={}, sampler={}, model=model, criterion=criterion, optimizer=optimizer,
                                            scheduler=scheduler, config=cfg, device=torch.device("cpu"))
                trainer.tqdm = types.SimpleNamespace(update=lambda n: None)
                nb = make_batch(cfg, batch=batch_size)
                batch = {"x": (torch.from_numpy(nb["x"]).to(dtype),), "y": torch.from_numpy(nb["y"]).to(dtype), "ar": torch.from_numpy(nb["ar"]).to(dtype)}
                model["generator"].train()
                model["discriminator"].train()
                trainer._train_step(batch)
                assert trainer.steps == 3
                kept[dtype] = (dict(trainer.total_train_loss), {net: {n: (p.grad.double().numpy().copy(), p.detach().double().numpy().copy())
                                                                       for n, p in model[net].named_parameters()} for net in model})
            logs, tensors = kept[torch.float32]
            for k, v in logs.items():
                out[f"{aux}::log::{k}"] = np.array(v)
                if recipe == "car_lin":  # (the yardstick for the logged losses too: the discriminator part runs on the UPDATED generator, whose
                    out[f"{aux}::log64::{k}"] = np.array(kept[torch.float64][0][k])  # first Adam step is a sign function of noisy gradients)
                print(f"{aux}: {k} = {v:.7f}   (float64: {kept[torch.float64][0][k]:.7f})")
            for net, names in (("generator", G_TENSORS), ("discriminator", D_TENSORS)):
                for n in names:
                    g32, p32 = tensors[net][n]
                    g64, _ = kept[torch.float64][1][net][n]
                    pack(f"{aux}::{net}::grad::{n}", g32, out)
                    pack(f"{aux}::{net}::new::{n}", p32, out)
                    e = np.abs(g32 - g64) / max(np.abs(g64).max(), 1e-30)
                    out[f"{aux}::{net}::grad_f32_vs_f64::{n}"] = np.array([np.median(e), e.max()])
                    print(f"    {net} {n}: reference fp32 vs fp64 gradient: median {np.median(e):.1e} max {e.max():.1e}")
    finally:
        torch.stft = real_stft
    out["B"], out["seeds"] = np.array(batch_size), np.array([SEED_G, SEED_D, SEED_X])
    path = os.path.join(REPO, "tests", "golden", fixture)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
