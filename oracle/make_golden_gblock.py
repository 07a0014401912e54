#!/usr/bin/env python3
"""Golden vectors of the REAL reference's ``GBlockGenerator`` (articulatory/models/gblock_gen.py:14-132, GBlock at
articulatory/layers/pytorch_layers.py:32-91) — SURVEY.md §8 f4's last component.  Same rules as oracle/make_golden.py: the reference is
imported unmodified in THIS container only; what is written is data.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_gblock.py

Which configurations run.  The class's DEFAULT arguments do not: ``g_kernel_sizes=(16, 16, 4, 4)`` are even (a GBlock's main path then loses
samples against its residual path: "size of tensor a (78) must match the size of tensor b (80)"), and four scales stop the hard-coded
ten-entry channel lists (gblock_gen.py:63-64) at ``channels // 2`` while the output conv expects ``channels // 8``.  The class is written
for TEN GBlocks with ODD kernel sizes, and that is what is pinned here:

  gold_gblock_small.npz   channels 64, kernel 3, x80 (g_scales 5,1,4,1,1,2,1,2,1,1), use_ar: B = 2, T = 8 — inputs, output and EVERY
                          GBlock's conv1 / res1 / block output (forward hooks), the PastFCEncoder output, the input conv
  gold_gblock_full.npz    channels 512 (11.99 M parameters), kernel 3, use_ar: B = 2, T = 25 — output in full, per-block statistics
  gold_gblock_k5spk.npz   channels 64, kernel 5, use_ar + use_spk_id: B = 3, T = 10 — output, per-block statistics
  gold_gblock_arloop.npz  the reference's own ar_loop (articulatory/bin/decode.py:31-83) on a (60, 13) utterance at chunk 25 (ragged tail of
                          10 frames) and on a (230, 13) utterance at chunk 100, channels 64 / kernel 3
  gold_gblock_nonar.npz   use_ar False, in_channels 1: ``.inference()`` of a (T,) input — the only input rank gblock_gen.py:172-190 accepts
                          (it unsqueezes BEFORE transposing: a (T, C) input becomes 4-D)
  gold_gblock_grad.npz    gradients of every state_dict parameter / c / ar under the reference's autograd with weight norm in the graph
                          (gblock_gen.py:161-170), two cases: "k3" (channels 64, use_ar, B = 2, T = 6) and "k5spk" (kernel 5, use_ar +
                          use_spk_id, B = 2, T = 5).  ReLU makes gradients discontinuous where a pre-activation is within rounding distance of
                          zero, and the class has no slope to set to 1: the script takes the first seed whose fp32 and fp64 reference
                          gradients agree to 1e-4 of every tensor's scale AND whose every ReLU / LeakyReLU input stays 2e-6 of its tensor's
                          scale away from zero (float64 run), and stores that margin next to the values.
  gold_gblock_keys.txt    state_dict keys + shapes of the channels-512 use_ar + use_spk_id model
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from make_golden import import_reference  # noqa: E402
from make_golden_grad import pack  # noqa: E402

SCALES = (5, 1, 4, 1, 1, 2, 1, 2, 1, 1)


def gparams(channels=64, k=3, **kw):
    return dict(dict(in_channels=141, out_channels=1, channels=channels, kernel_size=7, g_scales=list(SCALES), g_kernel_sizes=[k] * 10,
                     use_weight_norm=True, use_ar=True, ar_input=512, ar_hidden=256, ar_output=128, use_tanh=True), **kw)


def main():
    import torch

    from articulatory_amd.utils.synth import synth_features, synth_gblock_state_dict, uniform

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, ref_ar_loop, _ = import_reference()
    outdir = os.path.join(REPO, "tests", "golden")

    def build(params, seed=1234, train=False, dtype=None):
        g = ref_models.GBlockGenerator(**params)
        sd = synth_gblock_state_dict(params, seed=seed)
        assert list(g.state_dict().keys()) == list(sd.keys()), "gblock_param_spec disagrees with the reference's state_dict keys"
        for k, v in g.state_dict().items():
            assert tuple(v.shape) == sd[k].shape, k
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        if dtype is not None:
            g = g.to(dtype)
        if train:
            return g.train(), sd
        g.remove_weight_norm()
        return g.eval(), sd

    def ar_ctx(B, seed):
        return (synth_features(B, 512, 1, seed=seed)[:, :, 0] * 0.5 - 0.25).reshape(B, 1, 512).astype(np.float32)

    def hook_blocks(g, taps, inner):
        for i in range(10):
            g.resamples[i].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"resamples.{i}", o.detach().numpy().copy()))
            if inner:
                g.resamples[i].conv1.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"resamples.{i}.conv1", o.detach().numpy().copy()))
                g.resamples[i].res1.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"resamples.{i}.res1", o.detach().numpy().copy()))

    # ---- key list of the largest variant
    pk = gparams(512, 3, use_spk_id=True, num_spk=4)
    gk, _ = build(pk, train=True)
    with open(os.path.join(outdir, "gold_gblock_keys.txt"), "w") as f:
        for k, v in gk.state_dict().items():
            f.write(f"{k} {' '.join(str(int(s)) for s in v.shape)}\n")
    del gk

    # ---- small model, every block's tensors
    p = gparams(64, 3)
    g, _ = build(p)
    taps = {}
    g.ar_model.register_forward_hook(lambda m, a, o: taps.__setitem__("ar_feats", o.detach().numpy().copy()))
    g.input_conv.register_forward_hook(lambda m, a, o: taps.__setitem__("input_conv", o.detach().numpy().copy()))
    hook_blocks(g, taps, inner=True)
    B, T = 2, 8
    c = synth_features(B, T, 13, seed=901).transpose(0, 2, 1).copy()
    ar = ar_ctx(B, 902)
    with torch.no_grad():
        y = g(torch.from_numpy(c), ar=torch.from_numpy(ar))
    assert y.shape == (B, 1, 80 * T)
    np.savez_compressed(os.path.join(outdir, "gold_gblock_small.npz"), c=c, ar=ar, out=y.numpy(), params=np.array(repr(sorted(p.items()))),
                        **{"tap::" + k: v for k, v in taps.items()})

    # ---- ar_loop of the same model (decode.py:31-83): chunk 25 with a ragged tail, chunk 100
    out = {}
    for tag, frames, bms, seed in (("c25", 60, 2000, 903), ("c100", 230, 8000, 904)):
        x = synth_features(1, frames, 13, seed=seed)[0]
        conf = {"generator_params": dict(p, extra_art=False), "batch_max_steps": bms, "hop_size": 80, "dataset_mode": "a2w"}  # (the shipped YAMLs' mode)
        with torch.no_grad():
            yy = ref_ar_loop(g, torch.from_numpy(x), conf)
        assert yy.shape == (80 * frames,)
        out[f"{tag}_x"], out[f"{tag}_out"], out[f"{tag}_batch_max_steps"] = x, yy.numpy(), np.array(bms)
    np.savez_compressed(os.path.join(outdir, "gold_gblock_arloop.npz"), **out)

    # ---- full width
    p = gparams(512, 3)
    g, _ = build(p)
    print("channels 512:", sum(v.numel() for v in g.parameters()), "parameters")
    taps = {}
    hook_blocks(g, taps, inner=False)
    B, T = 2, 25
    c = synth_features(B, T, 13, seed=905).transpose(0, 2, 1).copy()
    ar = ar_ctx(B, 906)
    with torch.no_grad():
        y = g(torch.from_numpy(c), ar=torch.from_numpy(ar))
    out = {"c": c, "ar": ar, "out": y.numpy(), "params": np.array(repr(sorted(p.items())))}
    for k, v in taps.items():
        pack("tap::" + k, v, out)
    np.savez_compressed(os.path.join(outdir, "gold_gblock_full.npz"), **out)

    # ---- kernel 5 + speaker conditioning
    p = gparams(64, 5, use_spk_id=True, num_spk=4)
    g, _ = build(p)
    taps = {}
    hook_blocks(g, taps, inner=False)
    B, T = 3, 10
    c = synth_features(B, T, 13, seed=907).transpose(0, 2, 1).copy()
    ar = ar_ctx(B, 908)
    spk = np.array([2, 0, 3], dtype=np.int64)
    with torch.no_grad():
        y = g(torch.from_numpy(c), spk_id=torch.from_numpy(spk), ar=torch.from_numpy(ar))
    out = {"c": c, "ar": ar, "spk_id": spk, "out": y.numpy(), "params": np.array(repr(sorted(p.items())))}
    for k, v in taps.items():
        pack("tap::" + k, v, out)
    np.savez_compressed(os.path.join(outdir, "gold_gblock_k5spk.npz"), **out)

    # ---- non-AR .inference() on a 1-D input (the only rank the reference's method accepts)
    p = gparams(64, 3, in_channels=1, use_ar=False)
    g, _ = build(p)
    x = synth_features(1, 37, 1, seed=909)[0, :, 0].copy()
    with torch.no_grad():
        yy = g.inference(torch.from_numpy(x))
    assert yy.shape == (37 * 80, 1)
    np.savez_compressed(os.path.join(outdir, "gold_gblock_nonar.npz"), x=x, out=yy.numpy(), params=np.array(repr(sorted(p.items()))))

    # ---- gradients
    gout = {}
    for tag, p, B, T, seed0 in (("k3", gparams(64, 3), 2, 6, 920), ("k5spk", gparams(64, 5, use_spk_id=True, num_spk=4), 2, 5, 960)):
        for seed in range(seed0, seed0 + 40):
            g, sd = build(p, seed=seed, train=True)
            c = torch.from_numpy(synth_features(B, T, 13, seed=seed + 100).transpose(0, 2, 1).copy()).requires_grad_(True)
            ar = torch.from_numpy(ar_ctx(B, seed + 101)).requires_grad_(True)
            spk = torch.from_numpy(np.array([(seed + b) % 4 for b in range(B)], dtype=np.int64)) if p.get("use_spk_id") else None
            cot = torch.from_numpy(uniform(seed + 102, "cotangent", (B, 1, 80 * T), -1.0, 1.0))
            y = g(c, spk_id=spk, ar=ar)
            (y * cot).sum().backward()
            g64, _ = build(p, seed=seed, train=True, dtype=torch.float64)
            c64, ar64 = c.detach().double().requires_grad_(True), ar.detach().double().requires_grad_(True)
            margins = []
            for m in g64.modules():
                if isinstance(m, (torch.nn.ReLU, torch.nn.LeakyReLU)):
                    m.register_forward_hook(lambda mod, a, o: margins.append(float(a[0].abs().min() / a[0].abs().max())))
            (g64(c64, spk_id=spk, ar=ar64) * cot.double()).sum().backward()
            worst = max(float((q.grad.double() - r.grad).abs().max() / r.grad.abs().max())
                        for (_, q), (_, r) in zip(g.named_parameters(), g64.named_parameters()))
            worst = max(worst, float((c.grad.double() - c64.grad).abs().max() / c64.grad.abs().max()),
                        float((ar.grad.double() - ar64.grad).abs().max() / ar64.grad.abs().max()))
            if worst > 1e-4:
                print(f"{tag} seed {seed}: fp32 and fp64 reference gradients differ by {worst:.1e} (a ReLU kink): next seed")
                continue
            if min(margins) < 2e-6:  # (the margin DESIGN.md §2 uses everywhere: another correct fp32 implementation may round across it)
                print(f"{tag} seed {seed}: a ReLU input is {min(margins):.1e} of its tensor's scale from zero: next seed")
                continue
            o = {"c": c.detach().numpy(), "ar": ar.detach().numpy(), "cot": cot.numpy(), "seed": np.array(seed),
                 "margin": np.array(min(margins)), "fp32_vs_fp64": np.array(worst), "params": np.array(repr(sorted(p.items())))}
            if spk is not None:
                o["spk_id"] = spk.numpy()
            pack("out", y.detach().numpy(), o)
            pack("grad::c", c.grad.numpy(), o)
            pack("grad::ar", ar.grad.numpy(), o)
            for k, q in g.named_parameters():
                pack("grad::" + k, q.grad.numpy(), o)
            gout.update({f"{tag}/{k}": v for k, v in o.items()})
            print(f"{tag}: seed {seed}, fp32-vs-fp64 {worst:.1e}, smallest ReLU margin {min(margins):.1e}")
            break
        else:
            raise SystemExit(f"{tag}: no kink-free seed found")
    np.savez_compressed(os.path.join(outdir, "gold_gblock_grad.npz"), **gout)
    for f in sorted(os.listdir(outdir)):
        if f.startswith("gold_gblock"):
            print(f, os.path.getsize(os.path.join(outdir, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
