#!/usr/bin/env python3
"""Golden vectors of the REAL reference generator for two constructor options no shipped YAML uses (round 5):
  noadd    use_additional_convs=False (articulatory/layers/residual_block.py:151, 191-205, 217-221: a ResBlock layer is x = x + convs1[d](x))
  relu     nonlinear_activation="ReLU" (hifigan.py:40-41, 121-123: any torch.nn activation module by name); forward fixtures only
  blocks4  FOUR residual blocks per stage (articulatory/models/hifigan.py:134-145 builds one per resblock_kernel_sizes entry; the MRF mean
           is cs / 4, :226-230) with unequal numbers of dilations per block
Per variant: a forward of the width-64 model with every ResBlock output and every upsampler output tapped (eval, weight norm removed, as
predict_wav.py:114-124 runs it), the reference's own ar_loop on a short utterance, and gradients of every parameter / c / ar with weight norm in
the graph at LeakyReLU slope 1 (no kinks: see oracle/make_golden_grad.py's header; the output conv's LeakyReLU(0.01) inputs must stay 5e-6 of their
scale away from zero and the fp32 gradients within 1e-4 of a float64 run, else the next seed is taken).  Same rules as oracle/make_golden.py:
only data is written.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_variants.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from make_golden import import_reference, yaml_generator_params  # noqa: E402
from make_golden_grad import pack  # noqa: E402

VARIANTS = {
    "noadd": dict(channels=64, use_additional_convs=False),
    "blocks4": dict(channels=64, resblock_kernel_sizes=[3, 5, 7, 11], resblock_dilations=[[1, 3], [1, 3, 5], [1], [1, 3, 5]]),
    # another activation module by name (hifigan.py:40-41, 121-123): torch.nn.ReLU — forward fixtures only (its kinks make element-wise gradient
    # fixtures a coin flip; the GPU test compares gradients with the float64 oracle flip-robustly)
    "relu": dict(channels=64, nonlinear_activation="ReLU", nonlinear_activation_params={}),
}


def main():
    import torch

    from articulatory_amd.utils.synth import synth_features, synth_state_dict, uniform

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, ref_ar_loop, _ = import_reference()
    outdir = os.path.join(REPO, "tests", "golden")
    cfg = yaml_generator_params("e2w_hifigan.yaml")
    full = cfg["generator_params"]
    for tag, over in VARIANTS.items():
        params = dict(full, **over)
        nb = len(params["resblock_kernel_sizes"])
        g = ref_models.HiFiGANGenerator(**params)
        sd = synth_state_dict(params, seed=1234)
        assert list(g.state_dict().keys()) == list(sd.keys()), "param spec disagrees with the reference's state_dict keys"
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g.remove_weight_norm()
        g = g.eval()
        out = {"params": np.array(repr(params)), "keys": np.array("\n".join(f"{k} {' '.join(str(int(s)) for s in v.shape)}" for k, v in sd.items()))}
        taps = {}
        for i in range(4):
            g.upsamples[i].register_forward_hook(lambda m, inp, o, i=i: taps.__setitem__(f"upsamples.{i}", o.detach().numpy().copy()))
        for b in range(4 * nb):
            g.blocks[b].register_forward_hook(lambda m, inp, o, b=b: taps.__setitem__(f"blocks.{b}", o.detach().numpy().copy()))
        B, T = 2, 8
        c = synth_features(B, T, 13, seed=141).transpose(0, 2, 1).copy()
        ar = (synth_features(B, 512, 1, seed=142)[:, :, 0] * 0.5 - 0.25).reshape(B, 1, 512).astype(np.float32)
        with torch.no_grad():
            y = g(torch.from_numpy(c), ar=torch.from_numpy(ar))
        out.update(c=c, ar=ar, out=y.numpy(), **{"tap::" + k: v for k, v in taps.items()})
        x = synth_features(1, 60, 13, seed=143)[0]
        with torch.no_grad():
            out["arloop_x"] = x
            out["arloop_out"] = ref_ar_loop(g, torch.from_numpy(x), dict(cfg, generator_params=params, batch_max_steps=2000)).numpy()  # chunks of 25 frames + a 10-frame tail
        if params["nonlinear_activation"] != "LeakyReLU":
            np.savez_compressed(os.path.join(outdir, f"gold_variant_{tag}.npz"), **out)
            print(f"gold_variant_{tag}.npz", os.path.getsize(os.path.join(outdir, f"gold_variant_{tag}.npz")), "(forward only)")
            continue
        # gradients: weight norm in the graph, slope 1
        gparams = dict(params, nonlinear_activation_params={"negative_slope": 1.0})
        for seed in range(880, 920):
            gt = ref_models.HiFiGANGenerator(**gparams)
            gsd = synth_state_dict(gparams, seed=seed)
            gt.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
            gt.train()
            Bg, Tg = 2, 6
            cg = torch.from_numpy(synth_features(Bg, Tg, 13, seed=seed + 10).transpose(0, 2, 1).copy()).requires_grad_(True)
            arg = torch.from_numpy((synth_features(Bg, 512, 1, seed=seed + 11)[:, :, 0] * 0.5 - 0.25).reshape(Bg, 1, 512).astype(np.float32)).requires_grad_(True)
            cot = torch.from_numpy(uniform(seed + 12, "cotangent", (Bg, 1, 80 * Tg), -1.0, 1.0))
            margins = []
            gt.output_conv[0].register_forward_hook(lambda m, i, o: margins.append(float(i[0].abs().min() / i[0].abs().max())))
            yg = gt(cg, ar=arg)
            (yg * cot).sum().backward()
            if margins[0] < 5e-6:
                continue
            g64 = ref_models.HiFiGANGenerator(**gparams)
            g64.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
            g64 = g64.double().train()
            c64, ar64 = cg.detach().double().requires_grad_(True), arg.detach().double().requires_grad_(True)
            (g64(c64, ar=ar64) * cot.double()).sum().backward()
            worst = max(float((p.grad.double() - q.grad).abs().max() / q.grad.abs().max()) for (_, p), (_, q) in zip(gt.named_parameters(), g64.named_parameters()))
            if worst > 1e-4:
                continue
            out.update(gseed=np.array(seed), gc=cg.detach().numpy(), gar=arg.detach().numpy(), gcot=cot.numpy())
            pack("gout", yg.detach().numpy(), out)
            pack("grad::c", cg.grad.numpy(), out)
            pack("grad::ar", arg.grad.numpy(), out)
            for k, p in gt.named_parameters():
                pack("grad::" + k, p.grad.numpy(), out)
            break
        else:
            raise SystemExit(f"{tag}: no kink-free seed found")
        np.savez_compressed(os.path.join(outdir, f"gold_variant_{tag}.npz"), **out)
        print(f"gold_variant_{tag}.npz", os.path.getsize(os.path.join(outdir, f"gold_variant_{tag}.npz")), "grad seed", int(out["gseed"]))


if __name__ == "__main__":
    main()
