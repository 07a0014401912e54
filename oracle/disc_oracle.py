"""CPU restatement of the reference's discriminators and GAN losses — TEST INFRASTRUCTURE ONLY (see oracle/hificar_oracle.py's
header for the rules: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path never does).

Follows, function by function:
  articulatory/models/hifigan.py:317-389,391-417  HiFiGANPeriodDiscriminator: reflect pad to a multiple of the period, view
                                                  (B, C, T/P, P), Conv2d (k, 1) stride (s, 1) + LeakyReLU x 5, output Conv2d
                                                  (kernel_sizes[1] - 1, 1) padding ((kernel_sizes[1] - 1) // 2, 0), flatten
  articulatory/models/hifigan.py:451-500          HiFiGANMultiPeriodDiscriminator
  articulatory/models/hifigan.py:503-643          HiFiGANScaleDiscriminator: Conv1d k0 -> grouped strided Conv1d x n -> Conv1d k2 -> Conv1d k3
  articulatory/models/hifigan.py:645-663          its apply_weight_norm / apply_spectral_norm test isinstance(m, Conv2d): NO norm is applied
  articulatory/models/hifigan.py:666-738          HiFiGANMultiScaleDiscriminator: AvgPool1d(4, 2, 2) between scales
  articulatory/models/hifigan.py:741-825          HiFiGANMultiScaleMultiPeriodDiscriminator: msd outputs + mpd outputs
  articulatory/losses/adversarial_loss.py:12-123  Generator / Discriminator adversarial losses (mse, hinge)
  articulatory/losses/feat_match_loss.py:12-54    FeatureMatchLoss
  articulatory/losses/mel_loss.py:16-166          MelSpectrogram / MelSpectrogramLoss (torch.stft, center, hann; librosa.filters.mel)
  articulatory/losses/stft_loss.py:16-170         stft magnitude, SpectralConvergenceLoss, LogSTFTMagnitudeLoss, MultiResolutionSTFTLoss
Pinned by tests/golden/gold_disc_*.npz (oracle/make_golden_disc.py, real reference) and, for the mel / multi-resolution STFT losses,
tests/golden/gold_loss_aux.npz (oracle/make_golden_loss.py: the reference's own MelSpectrogramLoss / MultiResolutionSTFTLoss modules run
behind a torch.stft ``return_complex=False`` compatibility shim) — except the mel FILTERBANK MATRIX: librosa is not in this image, so
``mel_filterbank`` restates librosa.filters.mel 0.9 (Slaney scale, slaney norm) from its published algorithm; the reference's mel loss
was run with that restated basis, so the basis alone stays PARITY UNPINNED (checked for its defining properties instead).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from articulatory_amd.utils.synth import disc_params, period_disc_layers, scale_disc_layers


def fold_disc_weight_norm(sd, dtype=torch.float32):
    """weight_g / weight_v -> weight (torch.nn.utils.weight_norm, dim 0), everything as torch tensors."""
    out = OrderedDict()
    for k, v in sd.items():
        v = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(dtype)
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            g = torch.as_tensor(np.asarray(sd[k[:-1] + "g"]) if not isinstance(sd[k[:-1] + "g"], torch.Tensor) else sd[k[:-1] + "g"]).to(dtype)
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
            out[k[: -len("_v")]] = v * (g / norm)
        else:
            out[k] = v
    return out


def fold_disc_spectral_norm(sd, training=True, dtype=torch.float32, eps=1e-12):
    """torch.nn.utils.spectral_norm (hifigan.py:440-448; one power iteration per forward in training mode): every "<conv>.weight_orig" with
    its "<conv>.weight_u" / "<conv>.weight_v" becomes "<conv>.weight" = weight_orig / sigma.  Returns (folded tensors, {key: advanced u / v}).
    weight_orig tensors that require grad stay in the graph (u, v are constants, as in torch)."""
    out, state = OrderedDict(), {}
    for k, v in sd.items():
        t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
        t = t.to(dtype) if not t.requires_grad else t
        if k.endswith(".weight_u") or (k.endswith(".weight_v") and k[:-1] + "u" in sd):
            continue
        if k.endswith(".weight_orig"):
            base = k[: -len("weight_orig")]
            as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).detach().to(t.dtype)  # noqa: E731
            u, vv = as_t(sd[base + "weight_u"]), as_t(sd[base + "weight_v"])
            mat = t.reshape(t.shape[0], -1)
            if training:
                with torch.no_grad():
                    vv = F.normalize(torch.mv(mat.t(), u), dim=0, eps=eps)
                    u = F.normalize(torch.mv(mat, vv), dim=0, eps=eps)
            state[base + "weight_u"], state[base + "weight_v"] = u, vv
            out[base + "weight"] = t / torch.dot(u, torch.mv(mat, vv))
        else:
            out[k] = t
    return out, state


def scale_disc_forward(w, prefix, layers, x, slope):
    outs = []
    for l, L in enumerate(layers):
        base = f"{prefix}.layers.{l}" + (".0" if L["act"] else "")
        x = F.conv1d(x, w[base + ".weight"], w.get(base + ".bias"), stride=L["stride"], padding=L["pad"], groups=L["groups"])
        if L["act"]:
            x = F.leaky_relu(x, slope)
        outs.append(x)
    return outs


def period_disc_forward(w, prefix, layers, period, x, slope):
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t += n_pad
    x = x.view(b, c, t // period, period)
    outs = []
    for l, L in enumerate(layers):
        base = prefix + (f".convs.{l}.0" if L["act"] else ".output_conv")
        x = F.conv2d(x, w[base + ".weight"], w[base + ".bias"], stride=(L["stride"], 1), padding=(L["pad"], 0))
        if L["act"]:
            x = F.leaky_relu(x, slope)
            outs.append(x)
    outs.append(torch.flatten(x, 1, -1))
    return outs


def disc_forward(w, params, x):
    """w: FOLDED weights (fold_disc_weight_norm).  x: (B, 1, T).  Returns the reference's list (msd scales, then mpd periods) of lists
    of layer outputs."""
    p = disc_params(**params)
    outs = []
    sp = p["scale_discriminator_params"]
    pool = p["scale_downsample_pooling_params"]
    assert p["scale_downsample_pooling"] == "AvgPool1d"
    xs = x
    for i in range(p["scales"]):
        outs.append(scale_disc_forward(w, f"msd.discriminators.{i}", scale_disc_layers(**sp), xs,
                                       sp.get("nonlinear_activation_params", {"negative_slope": 0.1}).get("negative_slope", 0.01)))
        xs = F.avg_pool1d(xs, pool["kernel_size"], pool["stride"], pool["padding"])
    pp = p["period_discriminator_params"]
    for i, period in enumerate(p["periods"]):
        outs.append(period_disc_forward(w, f"mpd.discriminators.{i}", period_disc_layers(**pp), period, x,
                                        pp.get("nonlinear_activation_params", {"negative_slope": 0.1}).get("negative_slope", 0.01)))
    return outs


# ---------------------------------------------------------------- losses
def gen_adv_loss(outs, average_by_discriminators=True, loss_type="mse"):
    loss = 0.0
    for i, o in enumerate(outs):
        o = o[-1]
        loss = loss + (F.mse_loss(o, torch.ones_like(o)) if loss_type == "mse" else -o.mean())
    return loss / (i + 1) if average_by_discriminators else loss


def dis_adv_loss(outs_hat, outs, average_by_discriminators=True, loss_type="mse"):
    real, fake = 0.0, 0.0
    for i, (oh, o) in enumerate(zip(outs_hat, outs)):
        oh, o = oh[-1], o[-1]
        if loss_type == "mse":
            real = real + F.mse_loss(o, torch.ones_like(o))
            fake = fake + F.mse_loss(oh, torch.zeros_like(oh))
        else:
            real = real - torch.mean(torch.min(o - 1, torch.zeros_like(o)))
            fake = fake - torch.mean(torch.min(-oh - 1, torch.zeros_like(oh)))
    if average_by_discriminators:
        real, fake = real / (i + 1), fake / (i + 1)
    return real, fake


def feat_match_loss(feats_hat, feats, average_by_layers=True, average_by_discriminators=True, include_final_outputs=False):
    total = 0.0
    for i, (fh, f) in enumerate(zip(feats_hat, feats)):
        if not include_final_outputs:
            fh, f = fh[:-1], f[:-1]
        part = 0.0
        for j, (a, b) in enumerate(zip(fh, f)):
            part = part + F.l1_loss(a, b.detach())
        if average_by_layers:
            part = part / (j + 1)
        total = total + part
    return total / (i + 1) if average_by_discriminators else total


# ---------------------------------------------------------------- mel loss
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults htk=False, norm='slaney' (librosa 0.8-0.10): triangular
    filters on the Slaney mel scale, each scaled by 2 / (its band width in Hz).  (n_mels, 1 + n_fft // 2) float32."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def mel_spectrogram(x, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80, fmax=7600,
                    center=True, eps=1e-10, log_base=10.0):
    """mel_loss.py:16-112 (normalized=False, onesided=True).  x: (B, T) or (B, 1, T) -> (B, num_mels, frames)."""
    if x.dim() == 3:
        x = x.reshape(-1, x.size(2))
    win_length = fft_size if win_length is None else win_length
    win = getattr(torch, f"{window}_window")(win_length, dtype=x.dtype)
    spec = torch.stft(x, n_fft=fft_size, hop_length=hop_size, win_length=win_length, window=win, center=center, normalized=False,
                      onesided=True, return_complex=True)
    spec = torch.view_as_real(spec).transpose(1, 2)  # (B, frames, freqs, 2)
    power = spec[..., 0] ** 2 + spec[..., 1] ** 2
    amp = torch.sqrt(torch.clamp(power, min=eps))
    fmin = 0 if fmin is None else fmin
    fmax = fs / 2 if fmax is None else fmax
    melmat = torch.from_numpy(mel_filterbank(fs, fft_size, num_mels, fmin, fmax).T.copy()).to(x.dtype)
    mel = torch.clamp(torch.matmul(amp, melmat), min=eps)
    log = torch.log if log_base is None else {2.0: torch.log2, 10.0: torch.log10}[log_base]
    return log(mel).transpose(1, 2)


def mel_loss(y_hat, y, **kw):
    return F.l1_loss(mel_spectrogram(y_hat, **kw), mel_spectrogram(y, **kw))


def stft_magnitude(x, fft_size, hop_size, win_length):
    """stft_loss.py:16-40 (hann window, center, reflect padding; clamp 1e-7)."""
    win = torch.hann_window(win_length, dtype=x.dtype)
    spec = torch.view_as_real(torch.stft(x, fft_size, hop_size, win_length, win, return_complex=True))
    return torch.sqrt(torch.clamp(spec[..., 0] ** 2 + spec[..., 1] ** 2, min=1e-7)).transpose(2, 1)


def multi_resolution_stft_loss(x, y, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240)):
    """stft_loss.py:43-170: x predicted, y ground truth, (B, T) or (B, C, T) -> (sc_loss, mag_loss)."""
    if x.dim() == 3:
        x, y = x.reshape(-1, x.size(2)), y.reshape(-1, y.size(2))
    sc, mag = 0.0, 0.0
    for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths):
        xm, ym = stft_magnitude(x, fs, ss, wl), stft_magnitude(y, fs, ss, wl)
        sc = sc + torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
        mag = mag + F.l1_loss(torch.log(ym), torch.log(xm))
    return sc / len(fft_sizes), mag / len(fft_sizes)


def loss_test_signals(seed, B, T):
    """Inputs of tests/golden/gold_loss_aux.npz (oracle/make_golden_loss.py): articulatory_amd.utils.synth.synth_waveforms."""
    from articulatory_amd.utils.synth import synth_waveforms

    return synth_waveforms(seed, B, T)


def disc_gradients(sd, params, x, cots, dtype=torch.float32):
    """d(sum_i sum_l sum(out[i][l] * cots[i][l])) / d(every raw state_dict parameter, x) with weight norm in the graph."""
    leaves = {k: torch.as_tensor(np.asarray(v)).to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    w = OrderedDict()
    for k, v in leaves.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            g = leaves[k[:-1] + "g"]
            w[k[: -len("_v")]] = v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))
        else:
            w[k] = v
    x = torch.as_tensor(np.asarray(x)).to(dtype).clone().requires_grad_(True)
    outs = disc_forward(w, params, x)
    loss = 0.0
    for o, c in zip(outs, cots):
        for a, b in zip(o, c):
            loss = loss + (a * torch.as_tensor(np.asarray(b)).to(dtype)).sum()
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    grads["x"] = x.grad
    return [[t.detach() for t in o] for o in outs], grads
