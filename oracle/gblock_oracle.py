"""CPU oracle for the GAN-TTS style ``GBlockGenerator``.  TEST INFRASTRUCTURE ONLY.

A CPU *restatement* of the reference's algorithm; the checker the HIP path is compared with.  Only ``tests/`` may import it — nothing
under ``articulatory_amd/`` does (the product path fails loudly without ``libhificar.so`` / a GPU).

Reference lines followed:
  articulatory/models/gblock_gen.py:111-132      generator forward (AR concat, speaker vector, input conv, ten GBlocks, output conv)
  articulatory/models/gblock_gen.py:63-69        channel plan of the ten GBlocks
  articulatory/layers/pytorch_layers.py:32-91    GBlock: conv1 = ReLU, nearest Upsample, conv(k), ReLU, conv(k, dilation 3);
                                                 res1 = nearest Upsample, conv(1) on the RAW input; conv2 = ReLU, conv(k, d 9), ReLU,
                                                 conv(k, d 27);  x = conv1(x) + res1(x);  x = x + conv2(x)
  articulatory/layers/pytorch_layers.py:24-29    get_padding = dilation * (kernel_size - 1) // 2
  articulatory/layers/pytorch_layers.py:438-460  PastFCEncoder (shared with the HiFi-CAR oracle)
  articulatory/bin/decode.py:45-83               ar_loop (non-WSOLA branch; model(c, ar=prev) is all it calls)

PARITY PIN: ``oracle/make_golden_gblock.py`` imports the REAL reference class at runnable configurations (ten GBlocks, odd kernel sizes —
the class's defaults do not run, see that script's header), loads the synthetic checkpoint of ``articulatory_amd.utils.synth`` and writes
``tests/golden/gold_gblock_*.npz``; ``tests/test_gblock_oracle.py`` holds this file to every one of them.

The arithmetic lives in PyTorch (third party; ``torch==1.9.1`` pinned by the reference's requirements.txt:82, 2.10.0 here), so the oracle
calls the same ATen operators on folded weights; ``naive_forward`` is an independent float64 numpy restatement from the defining sums.
"""

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .hificar_oracle import _np_conv1d, _np_lrelu, fold_weight_norm, past_fc_encoder  # noqa: F401  (fold_weight_norm re-exported)

GBLOCK_IN = (1, 1, 1, 2, 2, 2, 2, 4, 4, 8)   # gblock_gen.py:63
GBLOCK_OUT = (1, 1, 2, 2, 2, 2, 4, 4, 8, 8)  # gblock_gen.py:64


def _cfg(params):
    p = dict(in_channels=80, out_channels=1, channels=512, kernel_size=7, g_scales=(8, 8, 2, 2), g_kernel_sizes=(16, 16, 4, 4),
             use_ar=False, ar_input=512, ar_hidden=256, ar_output=128, use_tanh=True, use_spk_id=False, num_spk=None, spk_emb_size=32)
    p.update({k: v for k, v in params.items() if k in p})
    return p


def _pad(k, d=1):
    return d * (k - 1) // 2  # pytorch_layers.py:24-29


def layer_names(i, scale):
    """state_dict prefixes of GBlock i's five convs: (conv1 first, conv1 second, res1, conv2 first, conv2 second)."""
    u = 1 if scale > 1 else 0
    b = f"resamples.{i}"
    return f"{b}.conv1.{1 + u}", f"{b}.conv1.{3 + u}", f"{b}.res1.{u}", f"{b}.conv2.1", f"{b}.conv2.3"


def gblock(w, i, x, scale, k, taps=None):
    """pytorch_layers.py:85-91."""
    n1, n2, nr, n3, n4 = layer_names(i, scale)
    a = F.relu(x)
    xr = x
    if scale > 1:  # torch.nn.Upsample(scale_factor=s), mode "nearest": out[t] = in[t // s]
        a = a.repeat_interleave(scale, dim=2)
        xr = x.repeat_interleave(scale, dim=2)
    h = F.conv1d(a, w[n1 + ".weight"], w[n1 + ".bias"], padding=_pad(k))
    if taps is not None:
        taps[f"resamples.{i}.conv1a"] = h
    h = F.conv1d(F.relu(h), w[n2 + ".weight"], w[n2 + ".bias"], dilation=3, padding=_pad(k, 3))
    r = F.conv1d(xr, w[nr + ".weight"], w[nr + ".bias"])
    if taps is not None:
        taps[f"resamples.{i}.res1"] = r
    x = h + r
    if taps is not None:
        taps[f"resamples.{i}.mid"] = x
    h = F.conv1d(F.relu(x), w[n3 + ".weight"], w[n3 + ".bias"], dilation=9, padding=_pad(k, 9))
    if taps is not None:
        taps[f"resamples.{i}.conv2a"] = h
    h = F.conv1d(F.relu(h), w[n4 + ".weight"], w[n4 + ".bias"], dilation=27, padding=_pad(k, 27))
    return x + h


def generator_forward(w, params, c, ar=None, spk_id=None, taps=None):
    """gblock_gen.py:111-132 on folded weights.  c: (B, C, T); ar: (B, 1, ar_input); spk_id: (B,) long."""
    p = _cfg(params)
    if p["use_ar"]:
        ar_feats = past_fc_encoder(w, ar)
        if taps is not None:
            taps["ar_feats"] = ar_feats
        c = torch.cat((c, ar_feats.unsqueeze(2).repeat(1, 1, c.shape[2])), dim=1)
    if p["use_spk_id"]:
        spk = F.linear(F.embedding(spk_id, w["spk_emb_mat.weight"]), w["spk_fc.weight"], w["spk_fc.bias"])
        c = c + spk.unsqueeze(2)
    ks = p["kernel_size"]
    c = F.conv1d(c, w["input_conv.weight"], w["input_conv.bias"], padding=(ks - 1) // 2)
    if taps is not None:
        taps["input_conv"] = c
    for i, (s, k) in enumerate(zip(p["g_scales"], p["g_kernel_sizes"])):
        c = gblock(w, i, c, s, k, taps)
        if taps is not None:
            taps[f"resamples.{i}"] = c
    out = F.conv1d(F.leaky_relu(c, 0.01), w["output_conv.1.weight"], w["output_conv.1.bias"], padding=(ks - 1) // 2)
    return torch.tanh(out) if p["use_tanh"] else out


def ar_loop(w, params, x, batch_max_steps, hop_size):
    """decode.py:54-83 for one utterance.  x: (T, C) -> (hop * T,)."""
    p = _cfg(params)
    in_chunk = int(batch_max_steps / hop_size)
    past = p["ar_input"]
    assert past <= batch_max_steps
    prev = torch.zeros((1, p["out_channels"], past), dtype=x.dtype)
    outs = []
    for i in range(0, len(x), in_chunk):
        cout = generator_forward(w, p, x[i:i + in_chunk].unsqueeze(0).permute(0, 2, 1), ar=prev)
        outs.append(cout[0][0])
        prev = cout[:, :, -past:]
    return torch.cat(outs, dim=0)


def ar_loop_batched(w, params, x, batch_max_steps, hop_size):
    """ar_loop for B equal-length utterances at once (they never interact).  x: (B, T, C) -> (B, hop * T)."""
    p = _cfg(params)
    in_chunk = int(batch_max_steps / hop_size)
    past = p["ar_input"]
    prev = torch.zeros((x.shape[0], p["out_channels"], past), dtype=x.dtype)
    outs = []
    for i in range(0, x.shape[1], in_chunk):
        cout = generator_forward(w, p, x[:, i:i + in_chunk].permute(0, 2, 1), ar=prev)
        outs.append(cout[:, 0])
        prev = cout[:, :, -past:] if cout.shape[2] >= past else torch.cat((prev, cout), dim=2)[:, :, -past:]
    return torch.cat(outs, dim=1)


def gradients(state_dict, params, c, ar, cot, dtype=torch.float32, spk_id=None):
    """d sum(out * cot) / d(every state_dict parameter, c, ar), weight norm inside the graph (gblock_gen.py:161-170)."""
    leaves = {k: torch.as_tensor(np.asarray(v)).to(dtype).clone().requires_grad_(True) for k, v in state_dict.items()}
    w = OrderedDict()
    for k, v in leaves.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len("weight_v")]
            g = leaves[base + "weight_g"]
            w[base + "weight"] = v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))
        else:
            w[k] = v
    c = torch.as_tensor(np.asarray(c)).to(dtype).clone().requires_grad_(True)
    ar_t = torch.as_tensor(np.asarray(ar)).to(dtype).clone().requires_grad_(True) if ar is not None else None
    spk_t = torch.as_tensor(np.asarray(spk_id)).long() if spk_id is not None else None
    out = generator_forward(w, params, c, ar_t, spk_id=spk_t)
    (out * torch.as_tensor(np.asarray(cot)).to(dtype)).sum().backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    grads["c"] = c.grad
    if ar_t is not None:
        grads["ar"] = ar_t.grad
    return out.detach(), grads


def relu_margin(w, params, c, ar=None, spk_id=None):
    """Smallest |pre-activation| / max|pre-activation| over every ReLU / LeakyReLU input of one forward (float64 weights and inputs
    recommended): a gradient comparison is only meaningful element-wise when no activation input sits within rounding distance of
    its kink (DESIGN.md §2 "Kinks")."""
    taps = {}
    generator_forward(w, params, c, ar, spk_id=spk_id, taps=taps)
    p = _cfg(params)
    worst = 1.0
    for i in range(len(p["g_scales"])):
        # the four ReLU inputs of a GBlock: its input, conv1's first conv, conv1 + res1, conv2's first conv
        for name in (f"resamples.{i - 1}" if i else "input_conv", f"resamples.{i}.conv1a", f"resamples.{i}.mid", f"resamples.{i}.conv2a"):
            t = taps[name]
            worst = min(worst, float(t.abs().min() / t.abs().max()))
    t = taps[f"resamples.{len(p['g_scales']) - 1}"]
    return min(worst, float(t.abs().min() / t.abs().max()))


def naive_forward(w, params, c, ar=None):
    """The same network from the defining sums in float64 numpy (no conv library; no speaker branch).  Small sizes only."""
    p = _cfg(params)
    W = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    c = np.asarray(c, dtype=np.float64)
    if p["use_ar"]:
        h = np.asarray(ar, dtype=np.float64).reshape(c.shape[0], -1)
        for li in range(5):
            h = h @ W[f"ar_model.model.{2 * li}.weight"].T + W[f"ar_model.model.{2 * li}.bias"]
            if li < 4:
                h = _np_lrelu(h, 0.1)
        c = np.concatenate([c, np.repeat(h[:, :, None], c.shape[2], axis=2)], axis=1)
    ks = p["kernel_size"]
    c = _np_conv1d(c, W["input_conv.weight"], W["input_conv.bias"], 1, (ks - 1) // 2)
    for i, (s, k) in enumerate(zip(p["g_scales"], p["g_kernel_sizes"])):
        n1, n2, nr, n3, n4 = layer_names(i, s)
        up = lambda v: v[:, :, np.arange(v.shape[2] * s) // s]  # noqa: E731  nearest: out[t] = in[t // s]
        h = _np_conv1d(up(np.maximum(c, 0)), W[n1 + ".weight"], W[n1 + ".bias"], 1, _pad(k))
        h = _np_conv1d(np.maximum(h, 0), W[n2 + ".weight"], W[n2 + ".bias"], 3, _pad(k, 3))
        c = h + _np_conv1d(up(c), W[nr + ".weight"], W[nr + ".bias"], 1, 0)
        h = _np_conv1d(np.maximum(c, 0), W[n3 + ".weight"], W[n3 + ".bias"], 9, _pad(k, 9))
        c = c + _np_conv1d(np.maximum(h, 0), W[n4 + ".weight"], W[n4 + ".bias"], 27, _pad(k, 27))
    c = _np_conv1d(_np_lrelu(c, 0.01), W["output_conv.1.weight"], W["output_conv.1.bias"], 1, (ks - 1) // 2)
    return np.tanh(c) if p["use_tanh"] else c
