"""The discriminator / GAN-loss oracle (oracle/disc_oracle.py) against golden vectors of the REAL reference
(tests/golden/gold_disc_*.npz, written by oracle/make_golden_disc.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from articulatory_amd.utils.synth import disc_param_spec, synth_disc_state_dict, uniform
from oracle import disc_oracle as DO
from oracle import hificar_oracle as O
from oracle.make_golden_disc import SMALL

TOL = 2e-5


def case_params(tag):
    if tag == "small":
        return SMALL
    import yaml

    # the shipped e2w_hifigan_car.yaml discriminator_params (restated: /root/reference is not available where tests run)
    return yaml.safe_load("""
scales: 3
scale_downsample_pooling: "AvgPool1d"
scale_downsample_pooling_params: {kernel_size: 4, stride: 2, padding: 2}
scale_discriminator_params:
    in_channels: 1
    out_channels: 1
    kernel_sizes: [15, 41, 5, 3]
    channels: 128
    max_downsample_channels: 1024
    max_groups: 16
    bias: true
    downsample_scales: [4, 4, 4, 4, 1]
    nonlinear_activation: "LeakyReLU"
    nonlinear_activation_params: {negative_slope: 0.1}
follow_official_norm: true
periods: [2, 3, 5, 7, 11]
period_discriminator_params:
    in_channels: 1
    out_channels: 1
    kernel_sizes: [5, 3]
    channels: 32
    downsample_scales: [3, 3, 3, 3, 1]
    max_downsample_channels: 1024
    bias: true
    nonlinear_activation: "LeakyReLU"
    nonlinear_activation_params: {negative_slope: 0.1}
    use_weight_norm: true
""")


def load(tag):
    gold = np.load(os.path.join(GOLDEN, f"gold_disc_{tag}.npz"))
    params = case_params(tag)
    seed, B, T = int(gold["seed"]), int(gold["B"]), int(gold["T"])
    sd = synth_disc_state_dict(params, seed=seed)
    x = uniform(seed, "x", (B, 1, T), -0.6, 0.6)
    xh = uniform(seed, "x_hat", (B, 1, T), -0.6, 0.6)
    return gold, params, sd, x, xh, seed


@pytest.mark.parametrize("tag", ["small", "default"])
def test_state_dict_layout_matches_reference(tag):
    gold, params, sd, *_ = load(tag)
    spec = disc_param_spec(**params)
    assert list(spec) == [str(k) for k in gold["keys"]]
    assert [str(tuple(v)) for v in spec.values()] == [str(s) for s in gold["shapes"]]


@pytest.mark.parametrize("tag", ["small", "default"])
def test_forward_and_gradients_vs_reference(tag):
    gold, params, sd, x, _, seed = load(tag)
    n_layers = [int(n) for n in gold["n_layers"]]
    cots = [[uniform(seed, f"cot.{i}.{l}", tuple(int(v) for v in gold[f"shape::{i}.{l}"]), -1.0, 1.0)
             / np.sqrt(np.prod(gold[f"shape::{i}.{l}"][1:])) for l in range(n)] for i, n in enumerate(n_layers)]
    outs, grads = DO.disc_gradients(sd, params, x, cots)
    assert [len(o) for o in outs] == n_layers
    for i, o in enumerate(outs):
        for l, t in enumerate(o):
            assert tuple(t.shape) == tuple(gold[f"shape::{i}.{l}"])
            assert O.check_packed(gold, f"out::{i}.{l}", t, TOL) < TOL, (i, l)
    bad = {k: O.check_packed(gold, "grad::" + k, g, 1e-4) for k, g in grads.items()}
    bad = {k: v for k, v in bad.items() if v >= 1e-4}
    assert not bad, bad


@pytest.mark.parametrize("tag", ["small", "default"])
def test_losses_vs_reference(tag):
    gold, params, sd, x, xh, _ = load(tag)
    w = DO.fold_disc_weight_norm(sd)
    with torch.no_grad():
        real = DO.disc_forward(w, params, torch.from_numpy(x))
        fake = DO.disc_forward(w, params, torch.from_numpy(xh))

    def close(a, key):
        ref = float(gold[key])
        assert abs(float(a) - ref) <= 2e-5 * max(abs(ref), 1e-3), (key, float(a), ref)

    for avg in (False, True):
        for lt in ("mse", "hinge"):
            close(DO.gen_adv_loss(fake, avg, lt), f"loss::gen_adv::{lt}::{int(avg)}")
            r, f = DO.dis_adv_loss(fake, real, avg, lt)
            close(r, f"loss::dis_real::{lt}::{int(avg)}")
            close(f, f"loss::dis_fake::{lt}::{int(avg)}")
        for inc in (False, True):
            close(DO.feat_match_loss(fake, real, avg, avg, inc), f"loss::feat_match::{int(avg)}::{int(inc)}")


def test_mel_filterbank_and_spectrogram_against_float64_dft():
    """librosa is not in this image (the reference's MelSpectrogram cannot be constructed): the restated Slaney filterbank is checked
    for its defining properties, and the STFT path against a direct float64 DFT of the same frames."""
    fb = DO.mel_filterbank(16000, 1024, 80, 0, 11025)  # e2w_hifigan_car.yaml:103-111 (fmax above Nyquist, as shipped)
    assert fb.shape == (80, 513) and (fb >= 0).all()
    peaks = fb.argmax(axis=1)
    live = fb.sum(axis=1) > 0
    assert live[:60].all() and (np.diff(peaks[live]) >= 0).all()
    # slaney norm: every complete triangle has area 2 / width * width / 2 = 1 in Hz
    df = 8000.0 / 512
    areas = fb.sum(axis=1) * df
    assert np.allclose(areas[5:55], 1.0, atol=0.08)
    rng = np.random.default_rng(3)
    y = (rng.standard_normal((2, 2000)) * 0.1).astype(np.float32)
    kw = dict(fs=16000, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=0, fmax=11025, log_base=None)
    got = DO.mel_spectrogram(torch.from_numpy(y), **kw).numpy()
    # direct restatement: reflect-pad n_fft // 2, periodic hann, frames at hop, |DFT|, filterbank, log
    pad = np.pad(y.astype(np.float64), ((0, 0), (512, 512)), mode="reflect")
    n = np.arange(1024)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / 1024)
    frames = 1 + 2000 // 256
    ref = np.zeros((2, 80, frames))
    for t in range(frames):
        seg = pad[:, t * 256 : t * 256 + 1024] * win
        amp = np.sqrt(np.maximum(np.abs(np.fft.rfft(seg, axis=1)) ** 2, 1e-10))
        ref[:, :, t] = np.log(np.maximum(amp @ fb.T.astype(np.float64), 1e-10))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-3  # log-mel of fp32 STFT vs float64


def test_stft_loss_oracle_against_float64_dft():
    """The reference's stft() calls torch.stft(..., return_complex=False), which torch 2.x rejects, so its MultiResolutionSTFTLoss cannot
    be run here: the restatement (same torch.stft parameters, complex output) is checked against a direct float64 DFT of the frames —
    window shorter than the frame (centred, zero-padded), reflect padding — and the two loss formulas are recomputed by hand."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((2, 1500)) * 0.1).astype(np.float32)
    y = (x + rng.standard_normal((2, 1500)) * 0.02).astype(np.float32)
    fs, ss, wl = 512, 50, 240

    def mags(sig):
        pad = np.pad(sig.astype(np.float64), ((0, 0), (fs // 2, fs // 2)), mode="reflect")
        n = np.arange(wl)
        win = np.zeros(fs)
        win[(fs - wl) // 2:(fs - wl) // 2 + wl] = 0.5 - 0.5 * np.cos(2 * np.pi * n / wl)
        frames = 1 + sig.shape[1] // ss
        out = np.zeros((sig.shape[0], frames, fs // 2 + 1))
        for t in range(frames):
            out[:, t] = np.sqrt(np.maximum(np.abs(np.fft.rfft(pad[:, t * ss:t * ss + fs] * win, axis=1)) ** 2, 1e-7))
        return out

    got = DO.stft_magnitude(torch.from_numpy(x), fs, ss, wl).numpy()
    ref = mags(x)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-5 * ref.max()
    sc, mag = DO.multi_resolution_stft_loss(torch.from_numpy(x), torch.from_numpy(y), [fs], [ss], [wl])
    xm, ym = mags(x), mags(y)
    assert abs(float(sc) - np.linalg.norm(ym - xm) / np.linalg.norm(ym)) < 1e-4 * float(sc)
    assert abs(float(mag) - np.abs(np.log(ym) - np.log(xm)).mean()) < 1e-4 * float(mag)


def test_stand_alone_discriminator_classes_state_dict_layout():
    """HiFiGANMultiScaleDiscriminator / HiFiGANMultiPeriodDiscriminator hold their sub-discriminators at the top level
    ("discriminators.<i>. ...", hifigan.py:451-500,666-738): the combined class's keys without the msd. / mpd. prefix (checked against the
    real reference classes when this was written)."""
    from articulatory_amd.models import HiFiGANMultiPeriodDiscriminator, HiFiGANMultiScaleDiscriminator

    spec = disc_param_spec()
    assert list(HiFiGANMultiScaleDiscriminator().state_dict()) == [k[4:] for k in spec if k.startswith("msd.")]
    assert list(HiFiGANMultiPeriodDiscriminator().state_dict()) == [k[4:] for k in spec if k.startswith("mpd.")]
    assert len(HiFiGANMultiScaleDiscriminator(scales=2).state_dict()) == 32 and len(HiFiGANMultiPeriodDiscriminator(periods=[2, 5]).state_dict()) == 36


# ---------------------------------------------------------------- auxiliary losses vs the REAL reference modules (gold_loss_aux.npz)
LOSS_CASES = [(tag, sname) for tag in ("recipe", "odd", "silence") for sname in ("default", "alt")]
STFT_SETS = {"default": {}, "alt": {"fft_sizes": [512, 256], "hop_sizes": [128, 64], "win_lengths": [512, 200]}}
MEL_SETS = {"recipe": dict(fs=16000, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=0, fmax=11025, log_base=None),
            "default": {}}


def kinked_gradient_close(g, ref, frac=0.98):
    """|.| terms flip sign where two values agree to rounding (the reference in fp32 and fp64 differs by 1e-4 there, see the
    ``f32_vs_f64`` entries of the fixture): direction + nearly all elements."""
    g, ref = np.asarray(g, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    cos = 1.0 - float(g @ ref) / float(np.linalg.norm(g) * np.linalg.norm(ref))
    return cos < 1e-5 and (np.abs(g - ref) < 1e-3 * np.abs(ref).max()).mean() > frac


@pytest.mark.parametrize("tag,sname", LOSS_CASES)
def test_stft_loss_oracle_vs_reference_golden(tag, sname):
    """multi_resolution_stft_loss vs articulatory.losses.stft_loss.MultiResolutionSTFTLoss run by oracle/make_golden_loss.py."""
    gold = np.load(os.path.join(GOLDEN, "gold_loss_aux.npz"))
    yh_np, y_np = DO.loss_test_signals(int(gold[f"{tag}::seed"]), int(gold[f"{tag}::B"]), int(gold[f"{tag}::T"]))
    for which in ("sc", "mag"):
        yh = torch.from_numpy(yh_np).requires_grad_(True)
        sc, mag = DO.multi_resolution_stft_loss(yh, torch.from_numpy(y_np), **STFT_SETS[sname])
        val = sc if which == "sc" else mag
        val.backward()
        ref = float(gold[f"{tag}::stft::{sname}::{which}::f32"])
        assert abs(float(val.detach()) - ref) < 2e-6 * abs(ref)
        g, gr = yh.grad.numpy(), gold[f"{tag}::stft::{sname}::d{which}::f32"]
        if which == "sc":
            assert np.abs(g - gr).max() < 2e-5 * np.abs(gr).max()
        else:
            assert kinked_gradient_close(g, gr)


@pytest.mark.parametrize("tag", ["recipe", "odd", "silence"])
@pytest.mark.parametrize("mname", ["recipe", "default"])
def test_mel_loss_oracle_vs_reference_golden(tag, mname):
    """mel_loss / mel_spectrogram vs articulatory.losses.mel_loss.MelSpectrogramLoss / MelSpectrogram (restated filterbank on both sides)."""
    gold = np.load(os.path.join(GOLDEN, "gold_loss_aux.npz"))
    yh_np, y_np = DO.loss_test_signals(int(gold[f"{tag}::seed"]), int(gold[f"{tag}::B"]), int(gold[f"{tag}::T"]))
    kw = MEL_SETS[mname]
    fb = DO.mel_filterbank(kw.get("fs", 22050), kw.get("fft_size", 1024), kw.get("num_mels", 80), kw.get("fmin", 80), kw.get("fmax", 7600))
    assert abs(float(fb.astype(np.float64).sum()) - float(gold[f"melmat::{mname}::sum"])) < 1e-9  # the basis the fixture was made with
    spec = DO.mel_spectrogram(torch.from_numpy(yh_np), **kw).numpy()
    ref_spec = gold[f"{tag}::mel::{mname}::spec"]
    assert spec.shape == ref_spec.shape and np.abs(spec - ref_spec).max() < 1e-4
    yh = torch.from_numpy(yh_np).requires_grad_(True)
    loss = DO.mel_loss(yh, torch.from_numpy(y_np), **kw)
    loss.backward()
    ref = float(gold[f"{tag}::mel::{mname}::loss::f32"])
    assert abs(float(loss.detach()) - ref) < 2e-6 * abs(ref)
    assert kinked_gradient_close(yh.grad.numpy(), gold[f"{tag}::mel::{mname}::dloss::f32"])


# ---------------------------------------------------------------- spectral norm on the period discriminators (gold_disc_sn.npz)
def test_spectral_norm_period_discriminators_oracle_vs_reference_golden():
    """use_spectral_norm (hifigan.py:390-399, 440-448; torch.nn.utils.spectral_norm, one power iteration per training-mode forward): two
    consecutive forwards — the power iteration advances between them —, the second one's gradients and the u / v buffers after each,
    against the real reference (oracle/make_golden_disc_sn.py)."""
    from oracle.make_golden_disc_sn import PERIODS, SN_PERIOD

    gold = np.load(os.path.join(GOLDEN, "gold_disc_sn.npz"))
    params = dict(scales=0, periods=PERIODS, period_discriminator_params=SN_PERIOD)
    seed, B, T = int(gold["seed"]), int(gold["B"]), int(gold["T"])
    sd = synth_disc_state_dict(params, seed=seed)
    assert [k[4:] for k in sd] == [str(k) for k in gold["keys"]]
    x = torch.from_numpy(uniform(seed, "x", (B, 1, T), -0.6, 0.6)).requires_grad_(True)
    leaves = {k: torch.from_numpy(v).clone().requires_grad_(k.endswith(("weight_orig", "bias"))) for k, v in sd.items()}
    for tag in ("1", "2"):
        w, state = DO.fold_disc_spectral_norm(leaves, training=True)
        for k, v in state.items():  # the advanced vectors are what the next forward starts from
            leaves[k] = v
            assert np.abs(v.numpy() - gold[f"state{tag}::{k[4:]}"]).max() < 2e-6, k
        outs = DO.disc_forward(w, params, x)
        for i, o in enumerate(outs):
            for l, t in enumerate(o):
                ref = gold[f"out{tag}::{i}.{l}"]
                assert np.abs(t.detach().numpy() - ref).max() < 2e-6 * max(np.abs(ref).max(), 1e-3), (tag, i, l)
    loss = 0.0
    for i, o in enumerate(outs):
        for l, t in enumerate(o):
            loss = loss + (t * torch.from_numpy(uniform(seed, f"cot.{i}.{l}", tuple(t.shape), -1.0, 1.0) / np.sqrt(t[0].numel()))).sum()
    loss.backward()
    got = {k[4:]: v.grad for k, v in leaves.items() if v.requires_grad}
    got["x"] = x.grad
    for k, g in got.items():
        ref = gold["grad::" + k]
        assert np.abs(g.numpy() - ref).max() < 2e-5 * np.abs(ref).max(), k
