#!/usr/bin/env python3
"""Golden vectors of the REAL reference's auxiliary losses — ``articulatory.losses.stft_loss.MultiResolutionSTFTLoss`` (stft_loss.py:128-170,
the loss BASELINE config 5 names) and ``articulatory.losses.mel_loss.MelSpectrogramLoss`` (mel_loss.py:114-166, the loss the shipped YAMLs
use) — values AND gradients with respect to the predicted waveform.  Same rules as oracle/make_golden.py: the reference is imported in
THIS container only; fixtures are data (inputs are regenerated from seeds, outputs are stored).

Two third-party-API shims, of the same kind as the ``scipy.signal.kaiser`` alias of oracle/make_golden.py:

  torch.stft      the reference passes ``return_complex=False`` (stft_loss.py:28-31, mel_loss.py:70-71), which torch >= 2.0 rejects.  The
                  shim calls the same torch.stft with ``return_complex=True`` and returns ``torch.view_as_real`` of it — exactly the
                  (..., 2) real view torch 1.9 (the reference's pin) returned.  Nothing of the reference's own code is changed: framing,
                  window, clamp (1e-7 / eps), sqrt, log base, Frobenius / L1 reductions and the averaging all run from the reference files.
  librosa         absent from this image.  ``librosa.filters.mel`` is stubbed by the restated Slaney filterbank of oracle/disc_oracle.py
                  (``mel_filterbank``); the BASIS therefore stays parity-unpinned (a restatement of librosa's published algorithm), while
                  everything the reference does around it (STFT, |.|, clamp, matmul, clamp, log, L1) is pinned.  The basis' checksum is
                  stored so a later change of the restatement shows.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_loss.py
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


def main():
    import torch
    import yaml

    torch.manual_seed(0)
    torch.set_num_threads(8)
    import_reference()
    sys.path.insert(0, REF)
    from disc_oracle import loss_test_signals as signals, mel_filterbank

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
        from articulatory.losses.mel_loss import MelSpectrogram, MelSpectrogramLoss
        from articulatory.losses.stft_loss import MultiResolutionSTFTLoss

        with open(os.path.join(REF, "egs/ema/voc1/conf/e2w_hifigan_car.yaml")) as f:
            cfg = yaml.safe_load(f)
        mel_recipe = cfg["mel_loss_params"]
        out = {"mel_recipe_params": np.array(yaml.safe_dump(mel_recipe))}
        # (B, T): the recipe's window (batch_max_steps 2000), a length that is no multiple of any hop, and a long one with silence
        cases = {"recipe": (3, 2000, 501), "odd": (2, 2511, 502), "silence": (2, 4096, 503)}
        mel_sets = {"recipe": mel_recipe, "default": {}}
        stft_sets = {"default": {}, "alt": {"fft_sizes": [512, 256], "hop_sizes": [128, 64], "win_lengths": [512, 200]}}
        for tag, (B, T, seed) in cases.items():
            y_hat_np, y_np = signals(seed, B, T)
            out[f"{tag}::B"], out[f"{tag}::T"], out[f"{tag}::seed"] = np.array(B), np.array(T), np.array(seed)
            for dtype, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
                y = torch.from_numpy(y_np).to(dtype)
                for sname, sp in stft_sets.items():
                    crit = MultiResolutionSTFTLoss(**sp).to(dtype)
                    for which in ("sc", "mag"):
                        yh = torch.from_numpy(y_hat_np).to(dtype).requires_grad_(True)
                        sc, mag = crit(yh, y)
                        (sc if which == "sc" else mag).backward()
                        out[f"{tag}::stft::{sname}::{which}::{dname}"] = np.array(float(sc if which == "sc" else mag))
                        out[f"{tag}::stft::{sname}::d{which}::{dname}"] = yh.grad.numpy().astype(np.float32 if dtype == torch.float32 else np.float64)
                for mname, mp in mel_sets.items():
                    crit = MelSpectrogramLoss(**mp).to(dtype)
                    yh = torch.from_numpy(y_hat_np).to(dtype).requires_grad_(True)
                    loss = crit(yh, y)
                    loss.backward()
                    out[f"{tag}::mel::{mname}::loss::{dname}"] = np.array(float(loss))
                    out[f"{tag}::mel::{mname}::dloss::{dname}"] = yh.grad.numpy().astype(np.float32 if dtype == torch.float32 else np.float64)
                    if dtype == torch.float32:
                        spec = MelSpectrogram(**mp)(torch.from_numpy(y_hat_np))
                        out[f"{tag}::mel::{mname}::spec"] = spec.numpy()
        for mname, mp in mel_sets.items():
            m = MelSpectrogram(**mp)
            out[f"melmat::{mname}::sum"] = np.array(float(m.melmat.double().sum()))
            out[f"melmat::{mname}::abssum_rows"] = m.melmat.double().abs().sum(0).numpy()
    finally:
        torch.stft = real_stft
    # float64 gradients are only a yardstick for the test tolerances: keep the values, drop the big arrays
    for k in [k for k in out if k.endswith("::f64") and out[k].ndim > 0]:
        g64 = out.pop(k)
        g32 = out[k[:-5] + "::f32"]
        out[k[:-5] + "::f32_vs_f64"] = np.array(float(np.abs(g32 - g64).max() / max(np.abs(g64).max(), 1e-30)))
    path = os.path.join(REPO, "tests", "golden", "gold_loss_aux.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    for k in sorted(out):
        if out[k].ndim == 0 and "::" in k and not k.endswith(("::B", "::T", "::seed")):
            print(f"  {k} = {out[k]}")


if __name__ == "__main__":
    main()
