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


if __name__ == "__main__":
    main()
