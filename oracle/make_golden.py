#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (read-only at /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

What is pinned:  the reference's own ``articulatory.models.HiFiGANGenerator``,
``articulatory.bin.decode.ar_loop`` and ``articulatory.utils.load_model`` are imported unmodified
(two import shims for packages absent from this image: ``scipy.signal.kaiser`` alias, stub modules
for h5py/soundfile/librosa/resampy/kaldiio/tensorboardX/tkinter — none of them is on the hot path),
the synthetic checkpoint of ``articulatory_amd.utils.synth`` is loaded with ``load_state_dict``,
``remove_weight_norm()`` + ``eval()`` + ``no_grad`` as predict_wav.py:114-124 does, and inputs/outputs
are written as small fixtures.  Only data is written: no reference source text goes into the repo.
"""

import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HIFICAR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def import_reference():
    import scipy.signal
    import scipy.signal.windows

    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser  # layers/pqmf.py:12 uses the removed alias
    for m in ["h5py", "soundfile", "librosa", "resampy", "kaldiio", "tensorboardX"]:
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    if "tkinter" not in sys.modules:
        tk = types.ModuleType("tkinter")
        tk.X = None
        sys.modules["tkinter"] = tk
    sys.path.insert(0, REF)
    import articulatory.models as ref_models
    from articulatory.bin.decode import ar_loop as ref_ar_loop
    from articulatory.utils import load_model as ref_load_model

    return ref_models, ref_ar_loop, ref_load_model


def yaml_generator_params(name):
    import yaml

    with open(os.path.join(REF, "egs/ema/voc1/conf", name)) as f:
        return yaml.safe_load(f)


def main():
    import torch

    from articulatory_amd.utils.synth import synth_features, synth_state_dict

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, ref_ar_loop, ref_load_model = import_reference()
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    cfg = yaml_generator_params("e2w_hifigan.yaml")
    full_params = cfg["generator_params"]

    def build(params, seed=1234):
        g = ref_models.HiFiGANGenerator(**params)
        sd = synth_state_dict(params, seed=seed)
        ref_keys = list(g.state_dict().keys())
        assert ref_keys == list(sd.keys()), "param spec disagrees with the reference's state_dict keys"
        for k, v in g.state_dict().items():
            assert tuple(v.shape) == sd[k].shape, k
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return g, sd

    # ---- (5) weight-norm fold: one Conv1d, one ConvTranspose1d --------------------------------
    small_params = dict(full_params, channels=64)
    g, sd = build(small_params)
    g.remove_weight_norm()
    fold = {}
    for name in ["upsamples.0.1", "blocks.4.convs2.1.1", "output_conv.1"]:
        fold[name + ".weight_g"] = sd[name + ".weight_g"]
        fold[name + ".weight_v"] = sd[name + ".weight_v"]
        fold[name + ".weight"] = g.state_dict()[name + ".weight"].numpy()
    np.savez_compressed(os.path.join(outdir, "gold_wnfold.npz"), **fold)
    # state_dict key order + shapes of the full model (pins generator_param_spec)
    gfull, sdfull = build(full_params)
    with open(os.path.join(outdir, "gold_state_dict_keys.txt"), "w") as f:
        for k, v in gfull.state_dict().items():
            f.write(f"{k} {' '.join(str(int(s)) for s in v.shape)}\n")

    # ---- (1) small model, every layer output ---------------------------------------------------
    g = g.eval()
    taps = {}

    def hook(name):
        def _h(mod, inp, out):
            taps[name] = out.detach().numpy().copy()
        return _h

    g.ar_model.register_forward_hook(hook("ar_feats"))
    g.input_conv.register_forward_hook(hook("input_conv"))
    for i in range(4):
        g.upsamples[i].register_forward_hook(hook(f"upsample{i}"))
    for b in range(12):
        g.blocks[b].register_forward_hook(hook(f"block{b}"))
        for d in (range(3) if b in (0, 7) else ()):
            g.blocks[b].convs1[d].register_forward_hook(hook(f"block{b}.convs1.{d}"))
            g.blocks[b].convs2[d].register_forward_hook(hook(f"block{b}.convs2.{d}"))
    B, T = 2, 8
    c = synth_features(B, T, 13, seed=101).transpose(0, 2, 1).copy()  # (B, 13, T)
    ar = (synth_features(B, 512, 1, seed=102)[:, :, 0] * 0.5 - 0.25).reshape(B, 1, 512).astype(np.float32)
    with torch.no_grad():
        y = g(torch.from_numpy(c), ar=torch.from_numpy(ar))
    np.savez_compressed(os.path.join(outdir, "gold_fwd_small.npz"), c=c, ar=ar, out=y.numpy(),
                        **{"tap." + k: v for k, v in taps.items()})

    # ---- (1b) small MRI-shaped model (odd scales 5 and 3, kernels 2s) --------------------------
    mri_params = dict(full_params, channels=64, in_channels=20 + 128, upsample_scales=[8, 5, 3, 2],
                      upsample_kernel_sizes=[16, 10, 6, 4])
    gm, _ = build(mri_params)
    gm.remove_weight_norm()
    gm = gm.eval()
    cm = synth_features(1, 9, 20, seed=111).transpose(0, 2, 1).copy()
    arm = (synth_features(1, 512, 1, seed=112)[:, :, 0] * 0.5 - 0.25).reshape(1, 1, 512).astype(np.float32)
    with torch.no_grad():
        ym = gm(torch.from_numpy(cm), ar=torch.from_numpy(arm))
    np.savez_compressed(os.path.join(outdir, "gold_fwd_small_mri.npz"), c=cm, ar=arm, out=ym.numpy())

    # ---- (2) full model, B=2, T=25 --------------------------------------------------------------
    gfull.remove_weight_norm()
    gfull = gfull.eval()
    stage = {}
    for i in range(4):
        def _h(mod, inp, out, i=i):
            stage[f"up{i}"] = out.detach().numpy().copy()
        gfull.upsamples[i].register_forward_hook(_h)
    B, T = 2, 25
    c = synth_features(B, T, 13, seed=201).transpose(0, 2, 1).copy()
    ar = (synth_features(B, 512, 1, seed=202)[:, :, 0] * 0.5 - 0.25).reshape(B, 1, 512).astype(np.float32)
    with torch.no_grad():
        y = gfull(torch.from_numpy(c), ar=torch.from_numpy(ar))
    stats = {}
    for k, v in stage.items():
        flat = v.reshape(-1).astype(np.float64)
        idx = (np.arange(16) * (flat.size // 16 + 1) * 7919) % flat.size
        stats[k + ".sum"] = np.array(flat.sum())
        stats[k + ".abssum"] = np.array(np.abs(flat).sum())
        stats[k + ".idx"] = idx
        stats[k + ".vals"] = flat[idx].astype(np.float32)
    np.savez_compressed(os.path.join(outdir, "gold_fwd_full.npz"), c=c, ar=ar, out=y.numpy(), **stats)

    # ---- (4) ar_loop on a (260, 13) utterance, chunk 25 and chunk 100 (ragged tails) -----------
    x = synth_features(1, 260, 13, seed=301)[0]
    res = {"x": x}
    for bms in (2000, 8000):
        conf = dict(cfg, batch_max_steps=bms)
        with torch.no_grad():
            yy = ref_ar_loop(gfull, torch.from_numpy(x), conf)
        res[f"out_bms{bms}"] = yy.numpy()
    np.savez_compressed(os.path.join(outdir, "gold_arloop.npz"), **res)

    # ---- (4b) the WSOLA driver variant (decode.py:84-100): half-overlapping chunks of 100 frames -------------
    conf = dict(cfg, batch_max_steps=8000)
    conf["generator_params"] = dict(conf["generator_params"], extra_art=False)
    with torch.no_grad():
        outs, ins = ref_ar_loop(gfull, torch.from_numpy(x), conf, do_wsola=True)
    np.savez_compressed(os.path.join(outdir, "gold_arloop_wsola.npz"), x=x,
                        n=np.array(len(outs)), **{f"out{i}": o.numpy() for i, o in enumerate(outs)},
                        **{f"in_len{i}": np.array(len(a)) for i, a in enumerate(ins)})

    # ---- (3) non-AR 12-dim model through .inference() -------------------------------------------
    nonar_params = dict(full_params, in_channels=12, use_ar=False)
    gn, _ = build(nonar_params)
    gn.remove_weight_norm()
    gn = gn.eval()
    xn = synth_features(1, 300, 12, seed=401)[0]
    with torch.no_grad():
        yn = gn.inference(xn)
    np.savez_compressed(os.path.join(outdir, "gold_nonar.npz"), x=xn, out=yn.numpy())

    # ---- (6) plumbing: checkpoint file in the reference layout -> load_model -> ar_loop ---------
    with tempfile.TemporaryDirectory() as td:
        ckpt = os.path.join(td, "checkpoint-1steps.pkl")
        torch.save({"model": {"generator": {k: torch.from_numpy(v) for k, v in sdfull.items()}},
                    "steps": 1, "epochs": 0}, ckpt)
        conf = dict(cfg)
        model = ref_load_model(ckpt, conf)
        model.remove_weight_norm()
        model = model.eval()
        xu = synth_features(1, 700, 13, seed=501)[0].astype(np.float64)  # .npy files are float64 on disk
        with torch.no_grad():
            cu = torch.tensor(xu, dtype=torch.float)
            yu = ref_ar_loop(model, cu, conf)
    np.savez_compressed(os.path.join(outdir, "gold_predict_wav.npz"), x=xu.astype(np.float32), out=yu.numpy())

    for fn in sorted(os.listdir(outdir)):
        print(fn, os.path.getsize(os.path.join(outdir, fn)))


if __name__ == "__main__":
    main()
