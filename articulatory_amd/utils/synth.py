"""Deterministic synthetic checkpoints for the HiFi-GAN / HiFi-CAR generator.

Trained checkpoints of the reference are Google-Drive links (reference README.md:36)
and unreachable offline, so every parity test, golden fixture and benchmark in this
repo runs on *synthesised* weights.  The generator here is numpy-only and keyed by
tensor name, so that

  * the golden-vector script (oracle/make_golden.py, which imports the real reference),
  * the CPU oracle (oracle/hificar_oracle.py),
  * the HIP path (articulatory_amd.models.HiFiGANGenerator) and
  * bench.py

all regenerate bit-identical fp32 tensors from (seed, name, shape) without shipping
the 54 MB state_dict.  The tensors are produced in the reference's *checkpoint*
layout: ``*.weight_g`` / ``*.weight_v`` / ``*.bias`` for every Conv1d/ConvTranspose1d
(old-style ``torch.nn.utils.weight_norm``; reference articulatory/models/hifigan.py:268-278)
and plain ``weight`` / ``bias`` for the PastFCEncoder Linear layers
(reference articulatory/layers/pytorch_layers.py:438-449).
"""

from collections import OrderedDict

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """n float64 samples in [0, 1), a pure function of (seed, name, index)."""
    base = (_fnv1a64(name) ^ (int(seed) * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(base)) & _MASK
    bits = _splitmix64(_splitmix64(ctr))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed: int, name: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def default_paddings(upsample_scales):
    """Reference padding rule (articulatory/models/hifigan.py:82-103)."""
    pads = [s // 2 + s % 2 for s in upsample_scales]
    opads = [s % 2 for s in upsample_scales]
    return pads, opads


def generator_param_spec(
    in_channels=80,
    out_channels=1,
    channels=512,
    kernel_size=7,
    upsample_scales=(8, 8, 2, 2),
    upsample_kernel_sizes=(16, 16, 4, 4),
    resblock_kernel_sizes=(3, 7, 11),
    resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
    use_additional_convs=True,
    bias=True,
    use_weight_norm=True,
    use_ar=False,
    ar_input=512,
    ar_hidden=256,
    ar_output=128,
    use_spk_id=False,
    num_spk=None,
    spk_emb_size=32,
    use_ph=False,
    num_ph=None,
    ph_emb_size=8,
    use_ph_loss=False,
    **_ignored,
):
    """Ordered {state_dict key: shape} of the reference generator for these kwargs.

    Key names and shapes follow what ``HiFiGANGenerator(**kwargs).state_dict()`` yields in the
    reference (articulatory/models/hifigan.py:108-175; verified key-for-key by
    tests/test_oracle_golden.py against the fixture written by oracle/make_golden.py).
    ConvTranspose1d weights are (Cin, Cout, K) and their weight_g is per-*Cin* (dim 0).
    """
    spec = OrderedDict()

    def conv(prefix, w_shape, n_bias, has_bias=True):
        if has_bias:
            spec[prefix + ".bias"] = (n_bias,)
        if use_weight_norm:
            spec[prefix + ".weight_g"] = (w_shape[0], 1, 1)
            spec[prefix + ".weight_v"] = tuple(w_shape)
        else:
            spec[prefix + ".weight"] = tuple(w_shape)

    # torch orders a weight-normed module's params as bias, weight_g, weight_v
    conv("input_conv", (channels, in_channels, kernel_size), channels)
    n_up = len(upsample_kernel_sizes)
    n_blocks = len(resblock_kernel_sizes)
    ups = []
    blocks = []
    for i in range(n_up):
        cin = channels // (2 ** i)
        cout = channels // (2 ** (i + 1))
        ups.append((f"upsamples.{i}.1", (cin, cout, upsample_kernel_sizes[i]), cout))
        for j in range(n_blocks):
            k = resblock_kernel_sizes[j]
            b = i * n_blocks + j
            c1 = [(f"blocks.{b}.convs1.{d}.1", (cout, cout, k), cout) for d in range(len(resblock_dilations[j]))]
            c2 = [(f"blocks.{b}.convs2.{d}.1", (cout, cout, k), cout) for d in range(len(resblock_dilations[j]))]
            blocks.append((c1, c2 if use_additional_convs else []))
    for p, s, nb in ups:
        conv(p, s, nb)
    # only the ResidualBlock convs honour the `bias` flag (residual_block.py:172-205); the input,
    # upsample and output convs are built without one and so always carry a bias (hifigan.py:108-159)
    for c1, c2 in blocks:
        for p, s, nb in c1:
            conv(p, s, nb, bias)
        for p, s, nb in c2:
            conv(p, s, nb, bias)
    c_last = channels // (2 ** n_up)
    conv("output_conv.1", (out_channels, c_last, kernel_size), out_channels)
    if use_ar:
        dims = [ar_input] + [ar_hidden] * 4 + [ar_output]
        for li in range(5):
            spec[f"ar_model.model.{2 * li}.weight"] = (dims[li + 1], dims[li])
            spec[f"ar_model.model.{2 * li}.bias"] = (dims[li + 1],)
    # speaker / phoneme conditioning (hifigan.py:176-189), in the reference's module registration order; Embedding and Linear carry
    # no weight norm (apply_weight_norm only touches Conv1d / ConvTranspose1d, hifigan.py:268-278)
    if use_spk_id:
        spec["spk_emb_mat.weight"] = (num_spk, spk_emb_size)
        spec["spk_fc.weight"] = (in_channels, spk_emb_size)
        spec["spk_fc.bias"] = (in_channels,)
    if use_ph:
        spec["ph_emb_mat.weight"] = (num_ph, ph_emb_size)
    if use_ph_loss:
        spec["ph_fc.weight"] = (num_ph, c_last)
        spec["ph_fc.bias"] = (num_ph,)
    return spec


def synth_state_dict(generator_params: dict, seed: int = 1234, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Synthesise a reference-layout generator state_dict as float32 numpy arrays.

    ``weight_v ~ U(-b, b)`` with ``b = gain * sqrt(3 / fan_in)`` (unit-variance-preserving),
    ``weight_g = ||v|| * U(0.8, 1.2)`` so that the weight-norm fold is *not* the identity,
    ``bias ~ U(-0.05, 0.05)``.
    """
    spec = generator_param_spec(**generator_params)
    out = OrderedDict()
    for name, shape in spec.items():
        if name.endswith(".weight_v") or name.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            b = gain * np.sqrt(3.0 / fan_in)
            out[name] = uniform(seed, name, shape, -b, b)
        elif name.endswith(".bias"):
            out[name] = uniform(seed, name, shape, -0.05, 0.05)
    for name, shape in spec.items():
        if name.endswith(".weight_g"):
            v = out[name[: -len("weight_g")] + "weight_v"].astype(np.float64)
            norm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1)).reshape(shape)
            jitter = uniform(seed, name, shape, 0.8, 1.2).astype(np.float64)
            out[name] = (norm * jitter).astype(np.float32)
    # restore spec order
    return OrderedDict((k, out[k]) for k in spec)


GBLOCK_IN = (1, 1, 1, 2, 2, 2, 2, 4, 4, 8)    # divisors of ``channels`` for the ten GBlocks' inputs (gblock_gen.py:63)
GBLOCK_OUT = (1, 1, 2, 2, 2, 2, 4, 4, 8, 8)   # ... and outputs (gblock_gen.py:64)


def gblock_param_spec(
    in_channels=80,
    out_channels=1,
    channels=512,
    kernel_size=7,
    g_scales=(8, 8, 2, 2),
    g_kernel_sizes=(16, 16, 4, 4),
    use_weight_norm=True,
    use_ar=False,
    ar_input=512,
    ar_hidden=256,
    ar_output=128,
    use_spk_id=False,
    num_spk=None,
    spk_emb_size=32,
    **_ignored,
):
    """Ordered {state_dict key: shape} of the reference's ``GBlockGenerator(**kwargs)`` (articulatory/models/gblock_gen.py:17-109; the
    GBlock's layers articulatory/layers/pytorch_layers.py:32-83).  A GBlock's Sequentials hold [ReLU, (Upsample,) conv, ReLU, conv]
    (conv1), [(Upsample,) conv] (res1) and [ReLU, conv, ReLU, conv] (conv2): the conv indices move by one when ``upsample > 1``.
    The GBlocks are built with ``norm=False`` but the generator's apply_weight_norm (gblock_gen.py:161-170) wraps every Conv1d, theirs
    included.  Checked key-for-key against the real class by oracle/make_golden_gblock.py."""
    spec = OrderedDict()

    def conv(prefix, w_shape):
        spec[prefix + ".bias"] = (w_shape[0],)
        if use_weight_norm:
            spec[prefix + ".weight_g"] = (w_shape[0], 1, 1)
            spec[prefix + ".weight_v"] = tuple(w_shape)
        else:  # a plain torch module registers weight before bias
            del spec[prefix + ".bias"]
            spec[prefix + ".weight"] = tuple(w_shape)
            spec[prefix + ".bias"] = (w_shape[0],)

    conv("input_conv", (channels, in_channels, kernel_size))
    for i, (s, k) in enumerate(zip(g_scales, g_kernel_sizes)):
        cin, cout = channels // GBLOCK_IN[i], channels // GBLOCK_OUT[i]
        u = 1 if s > 1 else 0
        base = f"resamples.{i}"
        conv(f"{base}.conv1.{1 + u}", (cout, cin, k))
        conv(f"{base}.conv1.{3 + u}", (cout, cout, k))
        conv(f"{base}.res1.{u}", (cout, cin, 1))
        conv(f"{base}.conv2.1", (cout, cout, k))
        conv(f"{base}.conv2.3", (cout, cout, k))
    conv("output_conv.1", (out_channels, channels // 8, kernel_size))
    if use_ar:
        dims = [ar_input] + [ar_hidden] * 4 + [ar_output]
        for li in range(5):
            spec[f"ar_model.model.{2 * li}.weight"] = (dims[li + 1], dims[li])
            spec[f"ar_model.model.{2 * li}.bias"] = (dims[li + 1],)
    if use_spk_id:
        spec["spk_emb_mat.weight"] = (num_spk, spk_emb_size)
        spec["spk_fc.weight"] = (in_channels, spk_emb_size)
        spec["spk_fc.bias"] = (in_channels,)
    return spec


def _synth_from_spec(spec, seed, gain):
    out = OrderedDict()
    for name, shape in spec.items():
        if name.endswith(".weight_v") or name.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            b = gain * np.sqrt(3.0 / fan_in)
            out[name] = uniform(seed, name, shape, -b, b)
        elif name.endswith(".bias"):
            out[name] = uniform(seed, name, shape, -0.05, 0.05)
    for name, shape in spec.items():
        if name.endswith(".weight_g"):
            v = out[name[: -len("weight_g")] + "weight_v"].astype(np.float64)
            norm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1)).reshape(shape)
            jitter = uniform(seed, name, shape, 0.8, 1.2).astype(np.float64)
            out[name] = (norm * jitter).astype(np.float32)
    return OrderedDict((k, out[k]) for k in spec)


def synth_gblock_state_dict(generator_params: dict, seed: int = 1234, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Synthetic reference-layout state_dict of a GBlockGenerator (same recipe as synth_state_dict)."""
    return _synth_from_spec(gblock_param_spec(**generator_params), seed, gain)


def synth_features(batch: int, frames: int, dims: int, seed: int) -> np.ndarray:
    """Synthetic EMA(+pitch) features, (B, T, dims) fp32: N(0,1) with channel 0 ~ U(0,1).

    Mirrors the min-max-normalised pitch channel of the reference's feature files
    (egs/ema/voc1/local/combine_feats.py:42-62).  numpy PCG64, as SURVEY.md §8(d) specifies.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((batch, frames, dims)).astype(np.float32)
    x[:, :, 0] = rng.random((batch, frames)).astype(np.float32)
    return x


# ------------------------------------------------------------------------------------------------
# HiFiGANMultiScaleMultiPeriodDiscriminator (reference articulatory/models/hifigan.py:317-825)
# ------------------------------------------------------------------------------------------------
def synth_waveforms(seed, B, T):
    """Waveform pairs for the loss / training fixtures (tests/golden/gold_loss_aux.npz, gold_train_step.npz).  (y_hat, y): band-limited pseudo-speech (a few decaying harmonics + noise) so that the spectra have structure, plus — for the
    LAST sequence — a stretch of exact silence in y_hat longer than the largest frame, which puts whole frames on the magnitude clamps
    (stft_loss.py:40, mel_loss.py:97,100)."""
    t = np.arange(T, dtype=np.float64) / 16000.0
    out = []
    for name in ("y_hat", "y"):
        sig = np.zeros((B, T))
        for b in range(B):
            f0 = 90.0 + 35.0 * b + (7.0 if name == "y" else 0.0)
            amp = uniform(seed, f"{name}.amp.{b}", (8,), 0.02, 0.12).astype(np.float64)
            for h in range(8):
                sig[b] += amp[h] * np.sin(2 * np.pi * f0 * (h + 1) * t + 0.3 * h + b)
        sig += uniform(seed, f"{name}.noise", (B, T), -0.05, 0.05)
        out.append(sig.astype(np.float32))
    y_hat, y = out
    if T >= 2600:
        y_hat[-1, 200:2500] = 0.0
    return y_hat[:, None, :], y[:, None, :]


def synth_train_batch(config, seed, batch):
    """One training batch of a recipe as numpy: {"x": (B, dims, frames), "y": (B, 1, batch_max_steps), "ar": (B, 1, ar_input)} — what the
    reference's collater hands _train_step for the a2w + AR recipes (train.py:1071-1097); the AR context and the window are one
    continuous waveform."""
    gp = config["generator_params"]
    hop = int(np.prod(gp["upsample_scales"]))
    frames = config["batch_max_steps"] // hop
    dims = gp["in_channels"] - gp["ar_output"]
    x = synth_features(batch, frames, dims, seed=seed).transpose(0, 2, 1).copy()
    _, wav = synth_waveforms(seed + 1, batch, gp["ar_input"] + config["batch_max_steps"])
    return {"x": x.astype(np.float32), "y": wav[:, :, gp["ar_input"]:].copy(), "ar": wav[:, :, : gp["ar_input"]].copy()}


DISC_DEFAULTS = dict(
    scales=3,
    scale_downsample_pooling="AvgPool1d",
    scale_downsample_pooling_params={"kernel_size": 4, "stride": 2, "padding": 2},
    scale_discriminator_params={
        "in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3], "channels": 128, "max_downsample_channels": 1024,
        "max_groups": 16, "bias": True, "downsample_scales": [2, 2, 4, 4, 1], "nonlinear_activation": "LeakyReLU",
        "nonlinear_activation_params": {"negative_slope": 0.1},
    },
    follow_official_norm=True,
    periods=[2, 3, 5, 7, 11],
    period_discriminator_params={
        "in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3], "channels": 32, "downsample_scales": [3, 3, 3, 3, 1],
        "max_downsample_channels": 1024, "bias": True, "nonlinear_activation": "LeakyReLU",
        "nonlinear_activation_params": {"negative_slope": 0.1}, "use_weight_norm": True, "use_spectral_norm": False,
    },
)


def disc_params(**kw):
    """discriminator_params of a YAML config merged over the reference's defaults (hifigan.py:744-781)."""
    p = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DISC_DEFAULTS.items()}
    for k, v in kw.items():
        if k not in p:
            raise TypeError(f"unexpected discriminator parameter {k!r}")
        p[k] = dict(v) if isinstance(v, dict) else v
    return p


def scale_disc_layers(in_channels=1, out_channels=1, kernel_sizes=(15, 41, 5, 3), channels=128, max_downsample_channels=1024,
                      max_groups=16, bias=True, downsample_scales=(2, 2, 4, 4, 1), **_ignored):
    """Conv1d layers of one HiFiGANScaleDiscriminator (hifigan.py:549-617): dicts of cin, cout, k, stride, groups, act."""
    assert len(kernel_sizes) == 4 and all(k % 2 == 1 for k in kernel_sizes)
    layers = [dict(cin=in_channels, cout=channels, k=kernel_sizes[0], stride=1, groups=1, act=True, bias=bias)]
    in_chs, out_chs, groups = channels, channels, 4
    for s in downsample_scales:
        layers.append(dict(cin=in_chs, cout=out_chs, k=kernel_sizes[1], stride=s, groups=groups, act=True, bias=bias))
        in_chs = out_chs
        out_chs = min(in_chs * 2, max_downsample_channels)
        groups = min(groups * 4, max_groups)
    out_chs = min(in_chs * 2, max_downsample_channels)
    layers.append(dict(cin=in_chs, cout=out_chs, k=kernel_sizes[2], stride=1, groups=1, act=True, bias=bias))
    layers.append(dict(cin=out_chs, cout=out_channels, k=kernel_sizes[3], stride=1, groups=1, act=False, bias=bias))
    for L in layers:
        L["pad"] = (L["k"] - 1) // 2
    return layers


def period_disc_layers(in_channels=1, out_channels=1, kernel_sizes=(5, 3), channels=32, downsample_scales=(3, 3, 3, 3, 1),
                       max_downsample_channels=1024, bias=True, **_ignored):
    """Conv2d (k, 1) layers of one HiFiGANPeriodDiscriminator (hifigan.py:357-389).  NB the output conv's kernel is
    kernel_sizes[1] - 1 with padding (kernel_sizes[1] - 1) // 2 (:383-389), and every Conv2d has a bias whatever ``bias`` says."""
    assert len(kernel_sizes) == 2 and kernel_sizes[0] % 2 == 1 and kernel_sizes[1] % 2 == 1
    layers = []
    in_chs, out_chs = in_channels, channels
    for s in downsample_scales:
        layers.append(dict(cin=in_chs, cout=out_chs, k=kernel_sizes[0], stride=s, groups=1, act=True, bias=True,
                           pad=(kernel_sizes[0] - 1) // 2))
        in_chs = out_chs
        out_chs = min(out_chs * 4, max_downsample_channels)
    layers.append(dict(cin=out_chs, cout=out_channels, k=kernel_sizes[1] - 1, stride=1, groups=1, act=False, bias=True,
                       pad=(kernel_sizes[1] - 1) // 2))
    return layers


def disc_param_spec(**kw):
    """Ordered {state_dict key: shape} of HiFiGANMultiScaleMultiPeriodDiscriminator(**kw) (checked key-for-key against the real
    reference by tests/test_oracle_golden.py through the fixture of oracle/make_golden_disc.py).  The scale discriminators never
    get a norm (their apply_weight_norm / apply_spectral_norm test for Conv2d on a Conv1d stack, hifigan.py:645-663): plain
    weights.  The period discriminators are weight-normed Conv2d (bias, weight_g, weight_v)."""
    p = disc_params(**kw)
    spec = OrderedDict()
    for i in range(p["scales"]):
        layers = scale_disc_layers(**p["scale_discriminator_params"])
        for l, L in enumerate(layers):
            base = f"msd.discriminators.{i}.layers.{l}" + (".0" if L["act"] else "")
            spec[base + ".weight"] = (L["cout"], L["cin"] // L["groups"], L["k"])
            if L["bias"]:
                spec[base + ".bias"] = (L["cout"],)
    pp = p["period_discriminator_params"]
    sn = pp.get("use_spectral_norm", False)
    wn = pp.get("use_weight_norm", True)
    if sn and wn:
        raise ValueError("Either use use_weight_norm or use_spectral_norm.")  # hifigan.py:390-391
    for i, _period in enumerate(p["periods"]):
        layers = period_disc_layers(**pp)
        for l, L in enumerate(layers):
            base = f"mpd.discriminators.{i}." + (f"convs.{l}.0" if L["act"] else "output_conv")
            shape = (L["cout"], L["cin"], L["k"], 1)
            if sn:  # torch.nn.utils.spectral_norm: bias, weight_orig (parameters), weight_u, weight_v (buffers: power-iteration vectors)
                spec[base + ".bias"] = (L["cout"],)
                spec[base + ".weight_orig"] = shape
                spec[base + ".weight_u"] = (L["cout"],)
                spec[base + ".weight_v"] = (L["cin"] * L["k"],)
            elif wn:  # a weight-normed module lists bias first (bias, weight_g, weight_v), a plain one weight, bias
                spec[base + ".bias"] = (L["cout"],)
                spec[base + ".weight_g"] = (L["cout"], 1, 1, 1)
                spec[base + ".weight_v"] = shape
            else:
                spec[base + ".weight"] = shape
                spec[base + ".bias"] = (L["cout"],)
    return spec


def synth_disc_state_dict(discriminator_params: dict, seed: int = 4321, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Synthetic discriminator state_dict (same recipe as synth_state_dict)."""
    spec = disc_param_spec(**discriminator_params)
    out = OrderedDict()
    for name, shape in spec.items():
        if len(shape) == 1 and (name.endswith(".weight_u") or name.endswith(".weight_v")):  # spectral norm's unit vectors
            v = uniform(seed, name, shape, -1.0, 1.0).astype(np.float64)
            out[name] = (v / np.sqrt((v * v).sum())).astype(np.float32)
        elif name.endswith(".weight_v") or name.endswith(".weight") or name.endswith(".weight_orig"):
            b = gain * np.sqrt(3.0 / int(np.prod(shape[1:])))
            out[name] = uniform(seed, name, shape, -b, b)
        elif name.endswith(".bias"):
            out[name] = uniform(seed, name, shape, -0.05, 0.05)
    for name, shape in spec.items():
        if name.endswith(".weight_g"):
            v = out[name[: -len("weight_g")] + "weight_v"].astype(np.float64)
            norm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1)).reshape(shape)
            out[name] = (norm * uniform(seed, name, shape, 0.8, 1.2).astype(np.float64)).astype(np.float32)
    return OrderedDict((k, out[k]) for k in spec)
