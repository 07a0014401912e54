#!/usr/bin/env python3
"""Golden vectors of the REAL reference discriminator and GAN losses (training path, articulatory/bin/train.py:341-440):
HiFiGANMultiScaleMultiPeriodDiscriminator outputs (every layer of every sub-discriminator), its gradients (weight norm in the graph)
for loss = sum over outputs of sum(out * cot), and the reference loss modules' values on those outputs.  Same rules as
oracle/make_golden.py: imports the reference in THIS container only; fixtures are data.  Big tensors are stored as samples
(oracle/make_golden_grad.py::pack).

  default  the shipped e2w_hifigan_car.yaml discriminator_params (70.7 M parameters), B = 2, T = 2512 (= ar 512 + batch_max_steps 2000,
           train.py:340-346)
  small    a narrow variant (2 scales, periods 2 / 3 / 5, T = 1031: not a multiple of any period -> reflect padding), every tensor in full

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_disc.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from make_golden import REF, import_reference  # noqa: E402
from make_golden_grad import pack  # noqa: E402

SMALL = dict(
    scales=2,
    scale_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3], "channels": 16,
                                "max_downsample_channels": 64, "max_groups": 4, "bias": True, "downsample_scales": [4, 4, 1],
                                "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1}},
    periods=[2, 3, 5],
    period_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3], "channels": 8,
                                 "downsample_scales": [3, 3, 1], "max_downsample_channels": 64, "bias": True,
                                 "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1},
                                 "use_weight_norm": True, "use_spectral_norm": False},
)


def main():
    import torch
    import yaml

    from articulatory_amd.utils.synth import synth_disc_state_dict, uniform

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, _, _ = import_reference()
    sys.path.insert(0, REF)
    import types

    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    from articulatory.losses.adversarial_loss import DiscriminatorAdversarialLoss, GeneratorAdversarialLoss
    from articulatory.losses.feat_match_loss import FeatureMatchLoss

    with open(os.path.join(REF, "egs/ema/voc1/conf/e2w_hifigan_car.yaml")) as f:
        cfg = yaml.safe_load(f)
    cases = {"default": (cfg["discriminator_params"], 2, 2512), "small": (SMALL, 3, 1031)}
    for tag, (params, B, T) in cases.items():
        for seed in range(900, 940):
            sd = synth_disc_state_dict(params, seed=seed)
            x_np = uniform(seed, "x", (B, 1, T), -0.6, 0.6)
            xh_np = uniform(seed, "x_hat", (B, 1, T), -0.6, 0.6)
            res = {}
            for dtype in (torch.float32, torch.float64):
                D = ref_models.HiFiGANMultiScaleMultiPeriodDiscriminator(**params).to(dtype)
                D.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in sd.items()})
                x = torch.from_numpy(x_np).to(dtype).requires_grad_(True)
                outs = D(x)
                loss = 0.0
                cots = []
                for i, o in enumerate(outs):
                    cots.append([])
                    for l, t in enumerate(o):
                        c = uniform(seed, f"cot.{i}.{l}", tuple(t.shape), -1.0, 1.0) / np.sqrt(t[0].numel())
                        cots[-1].append(c)
                        loss = loss + (t * torch.from_numpy(c).to(dtype)).sum()
                loss.backward()
                g = {k: p.grad.detach().double().numpy() for k, p in D.named_parameters()}
                g["x"] = x.grad.detach().double().numpy()
                res[dtype] = (outs, g)
            # a seed whose gradients hinge on a LeakyReLU kink (fp32 and fp64 reference disagree) pins nothing: next seed
            worst = max(np.abs(res[torch.float32][1][k] - res[torch.float64][1][k]).max() / max(np.abs(res[torch.float64][1][k]).max(), 1e-30)
                        for k in res[torch.float64][1])
            print(f"{tag}: seed {seed}: fp32-vs-fp64 reference gradients differ by {worst:.2e}")
            if worst < 1e-4:
                break
        else:
            raise SystemExit("no kink-free seed found")
        outs, grads = res[torch.float32]
        out = {"seed": np.array(seed), "B": np.array(B), "T": np.array(T), "keys": np.array(list(sd.keys())),
               "shapes": np.array([str(tuple(v.shape)) for v in sd.values()]), "n_disc": np.array(len(outs)),
               "n_layers": np.array([len(o) for o in outs])}
        full = tag == "small"
        for i, o in enumerate(outs):
            for l, t in enumerate(o):
                out[f"shape::{i}.{l}"] = np.array(t.shape)
                if full:
                    out[f"out::{i}.{l}::full"] = t.detach().numpy().reshape(-1)
                else:
                    pack(f"out::{i}.{l}", t.detach().numpy(), out)
        for k, v in grads.items():
            if full and v.size <= 1 << 16:
                out["grad::" + k + "::full"] = v.astype(np.float32).reshape(-1)
            else:
                pack("grad::" + k, v, out)
        # the reference's loss modules on (fake = D(x_hat), real = D(x)), both flag settings
        D = ref_models.HiFiGANMultiScaleMultiPeriodDiscriminator(**params)
        D.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        with torch.no_grad():
            p_real = D(torch.from_numpy(x_np))
            p_fake = D(torch.from_numpy(xh_np))
        for avg in (False, True):
            for lt in ("mse", "hinge"):
                out[f"loss::gen_adv::{lt}::{int(avg)}"] = np.array(float(GeneratorAdversarialLoss(avg, lt)(p_fake)))
                r, f_ = DiscriminatorAdversarialLoss(avg, lt)(p_fake, p_real)
                out[f"loss::dis_real::{lt}::{int(avg)}"] = np.array(float(r))
                out[f"loss::dis_fake::{lt}::{int(avg)}"] = np.array(float(f_))
            for inc in (False, True):
                out[f"loss::feat_match::{int(avg)}::{int(inc)}"] = np.array(float(FeatureMatchLoss(avg, avg, inc)(p_fake, p_real)))
        path = os.path.join(REPO, "tests", "golden", f"gold_disc_{tag}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
