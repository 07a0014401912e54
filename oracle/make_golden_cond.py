#!/usr/bin/env python3
"""Golden vectors for the speaker / phoneme conditioning branches of the REAL reference generator (hifigan.py:176-189, 212-220,
232-237).  Same rules as oracle/make_golden.py (the reference is imported unmodified from /root/reference in the build container;
only data is written).  Kept separate so that the fixtures of make_golden.py regenerate byte for byte.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cond.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from make_golden import import_reference, yaml_generator_params  # noqa: E402


def main():
    import torch

    from articulatory_amd.utils.synth import synth_features, synth_state_dict

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, _, _ = import_reference()
    outdir = os.path.join(REPO, "tests", "golden")
    full = yaml_generator_params("e2w_hifigan.yaml")["generator_params"]

    def build(params, seed):
        g = ref_models.HiFiGANGenerator(**params)
        sd = synth_state_dict(params, seed=seed)
        assert list(g.state_dict().keys()) == list(sd.keys()), "param spec disagrees with the reference's state_dict keys"
        for k, v in g.state_dict().items():
            assert tuple(v.shape) == sd[k].shape, k
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g.remove_weight_norm()
        return g.eval()

    # (a) speaker conditioning on the HiFi-CAR generator, width 128 (5 speakers)
    spk_params = dict(full, channels=128, use_spk_id=True, num_spk=5, spk_emb_size=32)
    g = build(spk_params, seed=4321)
    B, T = 3, 21
    c = synth_features(B, T, 13, seed=601).transpose(0, 2, 1).copy()
    ar = (synth_features(B, 512, 1, seed=602)[:, :, 0] * 0.5 - 0.25).reshape(B, 1, 512).astype(np.float32)
    spk = np.array([4, 0, 2], dtype=np.int64)
    with torch.no_grad():
        y = g(torch.from_numpy(c), spk_id=torch.from_numpy(spk), ar=torch.from_numpy(ar))
    np.savez_compressed(os.path.join(outdir, "gold_fwd_spk.npz"), c=c, ar=ar, spk_id=spk, out=y.numpy())

    # (b) phoneme conditioning + phoneme-loss head, non-AR 12-dim features + 8-dim phoneme embedding (in_channels 20), 11 phonemes
    ph_params = dict(full, channels=128, in_channels=12 + 8, use_ar=False, use_ph=True, num_ph=11, ph_emb_size=8, use_ph_loss=True)
    g = build(ph_params, seed=4322)
    B, T = 2, 19
    c = synth_features(B, T, 12, seed=611).transpose(0, 2, 1).copy()
    ph = np.random.Generator(np.random.PCG64(612)).integers(0, 11, size=(B, T)).astype(np.int64)
    with torch.no_grad():
        y, ph_out = g(torch.from_numpy(c), ph=torch.from_numpy(ph))
    np.savez_compressed(os.path.join(outdir, "gold_fwd_ph.npz"), c=c, ph=ph, out=y.numpy(), ph_out=ph_out.numpy())
    for fn in ("gold_fwd_spk.npz", "gold_fwd_ph.npz"):
        print(fn, os.path.getsize(os.path.join(outdir, fn)))

    # (c) GRADIENTS of the conditioned variants under the reference's autograd, weight norm in the graph (train.py:276 passes spk_id= / ph=):
    # loss = sum(out * cot) (+ sum(ph_out * cot_ph)); 2-stage width-128 generators (few activations: the script refuses a seed whose fp32 and
    # fp64 reference gradients differ by more than 1e-4, i.e. one with an activation on a LeakyReLU kink — see oracle/make_golden_grad.py)
    from articulatory_amd.utils.synth import uniform
    from make_golden_grad import pack

    small = dict(full, channels=128, upsample_scales=[5, 4], upsample_kernel_sizes=[10, 8])
    cases = {
        "spk": (dict(small, use_spk_id=True, num_spk=5, spk_emb_size=32), 3, 9),
        "ph": (dict(small, in_channels=12 + 8, use_ar=False, use_ph=True, num_ph=11, ph_emb_size=8, use_ph_loss=True), 2, 11),
        "ph_ar": (dict(small, in_channels=13 + 128 + 8, use_ph=True, num_ph=7, ph_emb_size=8, use_ph_loss=True), 2, 10),
    }
    for tag, (params, B, T) in cases.items():
        hop = int(np.prod(params["upsample_scales"]))
        dims = params["in_channels"] - (128 if params["use_ar"] else 0) - (params["ph_emb_size"] if params.get("use_ph") else 0)
        for seed in range(870, 910):
            sd = synth_state_dict(params, seed=seed)
            c_np = synth_features(B, T, dims, seed=seed + 10).transpose(0, 2, 1).copy()
            ar_np = (synth_features(B, 512, 1, seed=seed + 11)[:, :, 0] * 0.5 - 0.25).reshape(B, 1, 512).astype(np.float32) if params["use_ar"] else None
            cot = uniform(seed + 12, "cotangent", (B, 1, hop * T), -1.0, 1.0)
            kw_np = {}
            if params.get("use_spk_id"):
                kw_np["spk_id"] = np.random.Generator(np.random.PCG64(seed)).integers(0, params["num_spk"], size=(B,)).astype(np.int64)
            if params.get("use_ph"):
                kw_np["ph"] = np.random.Generator(np.random.PCG64(seed + 1)).integers(0, params["num_ph"], size=(B, T)).astype(np.int64)
            cot_ph = uniform(seed + 13, "cot_ph", (B, params["num_ph"], T), -1.0, 1.0) if params.get("use_ph_loss") else None
            res = {}
            for dtype in (torch.float32, torch.float64):
                g = ref_models.HiFiGANGenerator(**params)
                g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
                g = g.to(dtype).train()
                c = torch.from_numpy(c_np).to(dtype).requires_grad_(True)
                ar = torch.from_numpy(ar_np).to(dtype).requires_grad_(True) if ar_np is not None else None
                y = g(c, ar=ar, **{k: torch.from_numpy(v) for k, v in kw_np.items()})
                if cot_ph is not None:
                    y, ph_out = y
                    loss = (y * torch.from_numpy(cot).to(dtype)).sum() + (ph_out * torch.from_numpy(cot_ph).to(dtype)).sum()
                else:
                    ph_out, loss = None, (y * torch.from_numpy(cot).to(dtype)).sum()
                loss.backward()
                gr = {k: p.grad.double().numpy() for k, p in g.named_parameters()}
                gr["c"] = c.grad.double().numpy()
                if ar is not None:
                    gr["ar"] = ar.grad.double().numpy()
                res[dtype] = (y.detach(), ph_out.detach() if ph_out is not None else None, gr)
            worst = max(np.abs(res[torch.float32][2][k] - res[torch.float64][2][k]).max() / max(np.abs(res[torch.float64][2][k]).max(), 1e-30)
                        for k in res[torch.float64][2])
            if worst > 1e-4:
                print(f"{tag}: seed {seed}: fp32 and fp64 reference gradients differ by {worst:.1e} (a LeakyReLU kink): next seed")
                continue
            y, ph_out, gr = res[torch.float32]
            out = {"c": c_np, "cot": cot, "seed": np.array(seed), **kw_np}
            if ar_np is not None:
                out["ar"] = ar_np
            if cot_ph is not None:
                out["cot_ph"] = cot_ph
                out["ph_out"] = ph_out.numpy()
            pack("out", y.numpy(), out)
            for k, v in gr.items():
                pack("grad::" + k, v, out)
            path = os.path.join(outdir, f"gold_grad_{tag}.npz")
            np.savez_compressed(path, **out)
            print(f"gold_grad_{tag}.npz", os.path.getsize(path), "seed", seed, f"fp32-vs-fp64 {worst:.1e}")
            break
        else:
            raise SystemExit(f"{tag}: no kink-free seed found")


if __name__ == "__main__":
    main()
