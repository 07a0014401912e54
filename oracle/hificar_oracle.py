"""CPU oracle for the HiFi-GAN / HiFi-CAR generator forward pass.  TEST INFRASTRUCTURE ONLY.

This file is a CPU *restatement* of the reference's algorithm for the hot path; it is the
checker the HIP kernels are compared with.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under ``articulatory_amd/`` does:
the product path fails loudly when ``libhificar.so`` or a GPU is missing.

Two restatements live here:

* ``generator_forward`` / ``ar_loop`` / ``ar_loop_batched`` — fp32 (or fp64) ``torch.nn.functional``
  on *folded* weights.  The reference computes the same thing through ``torch.nn`` modules;
  the arithmetic itself lives in PyTorch (third-party, pinned ``torch==1.9.1`` by the reference's
  requirements.txt:82, 2.10.0 in this image), so the oracle calls the same ATen operators.
* ``naive_forward`` — an independent numpy float64 implementation written from the defining
  sums (no convolution library call), used at small sizes to check that the operator-level
  restatement means what we think it means (padding, dilation, transposed-conv indexing).

PARITY PIN: the reference ships no tests and no golden vectors, so the pin is
constructed: ``oracle/make_golden.py`` imports the *real* reference from /root/reference in the
build container, loads the synthetic checkpoint of ``articulatory_amd.utils.synth`` into it, and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle against every
one of them.  The fixtures travel; the reference does not.

Reference lines followed:
  articulatory/models/hifigan.py:198-239     generator forward (AR concat, input conv, stage loop, output conv)
  articulatory/models/hifigan.py:82-103      ConvTranspose1d padding / output_padding rule
  articulatory/models/hifigan.py:256-278     weight-norm fold (torch.nn.utils.weight_norm, dim=0)
  articulatory/layers/residual_block.py:207-222  ResBlock forward
  articulatory/layers/pytorch_layers.py:438-460  PastFCEncoder
  articulatory/bin/decode.py:45-83           ar_loop (non-WSOLA branch)
"""

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# weight-norm fold
# --------------------------------------------------------------------------------------
def fold_weight_norm(state_dict, dtype=torch.float32):
    """{*.weight_g, *.weight_v} -> {*.weight}: w = v * g / ||v||, norm over all dims but 0.

    Follows torch.nn.utils.weight_norm(dim=0) as applied at hifigan.py:268-278 and baked by
    remove_weight_norm (hifigan.py:256-266).  For ConvTranspose1d dim 0 is Cin.
    Accepts numpy arrays or tensors; returns an OrderedDict of tensors of ``dtype``.
    """
    out = OrderedDict()
    sd = {k: torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v for k, v in state_dict.items()}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len("weight_v")]
            g = sd[base + "weight_g"].to(torch.float32)
            vv = v.to(torch.float32)
            # same expression order as torch._weight_norm: v * (g / norm(v))
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(g.shape)
            out[base + "weight"] = (vv * (g / norm)).to(dtype)
        else:
            out[k] = v.to(dtype)
    return out


# --------------------------------------------------------------------------------------
# operator-level restatement
# --------------------------------------------------------------------------------------
def _cfg(params):
    p = dict(
        in_channels=80, out_channels=1, channels=512, kernel_size=7,
        upsample_scales=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
        resblock_kernel_sizes=(3, 7, 11), resblock_dilations=((1, 3, 5),) * 3,
        use_additional_convs=True, bias=True,
        nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
        use_ar=False, ar_input=512, ar_hidden=256, ar_output=128, use_tanh=True,
        use_spk_id=False, num_spk=None, spk_emb_size=32, use_ph=False, num_ph=None, ph_emb_size=8, use_ph_loss=False,
    )
    p.update({k: v for k, v in params.items() if k in p})
    # hifigan.py:121-123: getattr(torch.nn, nonlinear_activation)(**params); ReLU and Identity are LeakyReLU at slope 0 / 1
    if p["nonlinear_activation"] == "ReLU":
        p["nonlinear_activation_params"] = {"negative_slope": 0.0}
    elif p["nonlinear_activation"] == "Identity":
        p["nonlinear_activation_params"] = {"negative_slope": 1.0}
    else:
        assert p["nonlinear_activation"] == "LeakyReLU"
    return p


def past_fc_encoder(w, ar):
    """pytorch_layers.py:451-460: reshape(B,-1) -> 4x[Linear, LeakyReLU(0.1)] -> Linear."""
    x = ar.reshape(ar.shape[0], -1)
    for li in range(5):
        x = F.linear(x, w[f"ar_model.model.{2 * li}.weight"], w[f"ar_model.model.{2 * li}.bias"])
        if li < 4:
            x = F.leaky_relu(x, 0.1)
    return x


def residual_block(w, prefix, x, kernel_size, dilations, slope, use_additional_convs=True):
    """residual_block.py:217-221."""
    for idx, d in enumerate(dilations):
        p1 = f"{prefix}.convs1.{idx}.1"
        xt = F.conv1d(F.leaky_relu(x, slope), w[p1 + ".weight"], w.get(p1 + ".bias"),
                      dilation=d, padding=(kernel_size - 1) // 2 * d)
        if use_additional_convs:
            p2 = f"{prefix}.convs2.{idx}.1"
            xt = F.conv1d(F.leaky_relu(xt, slope), w[p2 + ".weight"], w.get(p2 + ".bias"),
                          dilation=1, padding=(kernel_size - 1) // 2)
        x = xt + x
    return x


def generator_forward(w, params, c, ar=None, taps=None, spk_id=None, ph=None):
    """hifigan.py:198-239 on folded weights ``w`` (dict of tensors).  c: (B, C, T); ar: (B,1,ar_input); spk_id: (B,) long
    (use_spk_id); ph: (B, T) long (use_ph).  Returns the waveform, or (waveform, ph_out) when use_ph_loss.

    ``taps``: optional dict that receives every intermediate (for per-layer golden checks).
    """
    p = _cfg(params)
    slope = p["nonlinear_activation_params"]["negative_slope"]
    if p["use_ar"]:
        ar_feats = past_fc_encoder(w, ar)
        if taps is not None:
            taps["ar_feats"] = ar_feats
        c = torch.cat((c, ar_feats.unsqueeze(2).repeat(1, 1, c.shape[2])), dim=1)
    if p["use_spk_id"]:  # hifigan.py:212-216: a per-utterance vector added to every frame of every input channel
        spk = F.linear(F.embedding(spk_id, w["spk_emb_mat.weight"]), w["spk_fc.weight"], w["spk_fc.bias"])
        c = c + spk.unsqueeze(2)
    if p["use_ph"]:  # hifigan.py:217-220: per-frame phoneme embeddings appended as channels
        c = torch.cat((c, F.embedding(ph, w["ph_emb_mat.weight"]).transpose(1, 2)), dim=1)
    ks = p["kernel_size"]
    c = F.conv1d(c, w["input_conv.weight"], w["input_conv.bias"], padding=(ks - 1) // 2)
    if taps is not None:
        taps["input_conv"] = c
    nb = len(p["resblock_kernel_sizes"])
    for i, (s, k) in enumerate(zip(p["upsample_scales"], p["upsample_kernel_sizes"])):
        c = F.conv_transpose1d(F.leaky_relu(c, slope), w[f"upsamples.{i}.1.weight"], w[f"upsamples.{i}.1.bias"],
                               stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
        if taps is not None:
            taps[f"upsample{i}"] = c
        cs = 0.0
        for j in range(nb):
            bj = residual_block(w, f"blocks.{i * nb + j}", c, p["resblock_kernel_sizes"][j],
                                p["resblock_dilations"][j], slope, p["use_additional_convs"])
            if taps is not None:
                taps[f"blocks.{i * nb + j}"] = bj
            cs = cs + bj
        c = cs / nb
        if taps is not None:
            taps[f"stage{i}"] = c
    # output conv uses LeakyReLU() default slope 0.01 (hifigan.py:150)
    out = F.conv1d(F.leaky_relu(c, 0.01), w["output_conv.1.weight"], w["output_conv.1.bias"], padding=(ks - 1) // 2)
    if p["use_tanh"]:
        out = torch.tanh(out)
    if p["use_ph_loss"]:  # hifigan.py:232-237 (+ :183-189): per-sample phoneme logits, average-pooled back to the frame rate
        hop = int(np.prod(p["upsample_scales"]))
        ph_out = F.linear(c.transpose(1, 2), w["ph_fc.weight"], w["ph_fc.bias"]).transpose(1, 2)
        ph_out = F.avg_pool1d(ph_out, kernel_size=hop * 2, stride=hop, padding=hop // 2)
        return out, ph_out
    return out


def inference(w, params, c, mean=None, scale=None):
    """hifigan.py:298-314: (T, C) -> (T*prod(scales), out_channels)."""
    c = torch.as_tensor(c, dtype=next(iter(w.values())).dtype)
    if mean is not None:
        c = (c - mean) / scale
    return generator_forward(w, params, c.transpose(1, 0).unsqueeze(0)).squeeze(0).transpose(1, 0)


def ar_loop(w, params, x, batch_max_steps, hop_size):
    """decode.py:54-83 for one utterance.  x: (T, C) tensor -> (hop*T,) tensor."""
    p = _cfg(params)
    in_chunk = int(batch_max_steps / hop_size)
    past = p["ar_input"]
    prev = torch.zeros((1, p["out_channels"], past), dtype=x.dtype)
    outs = []
    for i in range(0, len(x), in_chunk):
        cin = x[i:i + in_chunk].unsqueeze(0).permute(0, 2, 1)
        cout = generator_forward(w, p, cin, ar=prev)
        outs.append(cout[0][0])
        if past <= batch_max_steps:
            prev = cout[:, :, -past:]
        else:
            prev = _shift_prev(prev, cout, in_chunk)
    return torch.cat(outs, dim=0)


def ar_loop_wsola(w, params, x, batch_max_steps, hop_size, extra_art=0):
    """decode.py:84-100 (do_wsola branch): half-overlapping chunks, each conditioned on the ar_input samples that end
    at the middle of the previous chunk.  Returns (list of chunk waveforms, list of chunk inputs)."""
    p = _cfg(params)
    in_chunk = int(batch_max_steps / hop_size)
    past = p["ar_input"]
    assert in_chunk % 2 == 0
    ins = [x[i:i + in_chunk + int(extra_art)] for i in range(0, len(x), int(in_chunk / 2))]
    prev = torch.zeros((1, 1, past), dtype=x.dtype)
    outs = []
    for art_i, art in enumerate(ins):
        signal = generator_forward(w, p, art.unsqueeze(0).permute(0, 2, 1), ar=prev)
        outs.append(signal[0][0])
        if art_i < len(ins) - 1:
            prev = signal[:, :, int(batch_max_steps / 2) - past:int(batch_max_steps / 2)]
            assert prev.shape[2] == past
    return outs, ins


def _shift_prev(prev, cout, in_chunk):
    # decode.py:79-81 (only reached when the chunk's audio is shorter than ar_input)
    prev = prev.clone()
    prev[:, :, :-in_chunk] = prev[:, :, in_chunk:].clone()
    prev[:, :, -in_chunk:] = cout
    return prev


def ar_loop_batched(w, params, x, batch_max_steps, hop_size):
    """Batched form of ar_loop for equal-length utterances.  x: (B, T, C) -> (B, hop*T).

    The reference's driver is batch-1 (decode.py:59,65); utterances never interact, so the batched
    loop equals B independent ar_loop calls (checked in tests/test_oracle_golden.py).
    """
    p = _cfg(params)
    in_chunk = int(batch_max_steps / hop_size)
    past = p["ar_input"]
    assert past <= batch_max_steps
    B = x.shape[0]
    prev = torch.zeros((B, p["out_channels"], past), dtype=x.dtype)
    outs = []
    for i in range(0, x.shape[1], in_chunk):
        cin = x[:, i:i + in_chunk].permute(0, 2, 1)
        cout = generator_forward(w, p, cin, ar=prev)
        outs.append(cout[:, 0])
        prev = cout[:, :, -past:] if cout.shape[2] >= past else torch.cat((prev, cout), dim=2)[:, :, -past:]
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------------------
# training path: gradients (the generator half of articulatory/bin/train.py:276-440)
# --------------------------------------------------------------------------------------
def gradients(state_dict, params, c, ar, cot, dtype=torch.float32, spk_id=None, ph=None, cot_ph=None):
    """d(sum(out * cot) [+ sum(ph_out * cot_ph)]) / d(every state_dict parameter, c, ar): the reference runs the generator under PyTorch
    autograd with weight norm in the graph (w = v * g / ||v||, hifigan.py:268-278); the restatement differentiates ``generator_forward``
    on weights folded INSIDE the graph.  state_dict: reference-layout arrays (weight_g / weight_v / bias / Linear / Embedding).
    spk_id / ph: the conditioning indices (use_spk_id / use_ph); cot_ph: cotangent of the phoneme-loss head's output (use_ph_loss).
    Returns (out, {"c": .., "ar": .., state_dict key: ..}); out is the pair (out, ph_out) for a use_ph_loss model."""
    leaves = {k: torch.as_tensor(np.asarray(v)).to(dtype).clone().requires_grad_(True) for k, v in state_dict.items()}
    w = OrderedDict()
    for k, v in leaves.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len("weight_v")]
            g = leaves[base + "weight_g"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
            w[base + "weight"] = v * (g / norm)
        else:
            w[k] = v
    c = torch.as_tensor(np.asarray(c)).to(dtype).clone().requires_grad_(True)
    ar_t = torch.as_tensor(np.asarray(ar)).to(dtype).clone().requires_grad_(True) if ar is not None else None
    spk_t = torch.as_tensor(np.asarray(spk_id)).long() if spk_id is not None else None
    ph_t = torch.as_tensor(np.asarray(ph)).long() if ph is not None else None
    out = generator_forward(w, params, c, ar_t, spk_id=spk_t, ph=ph_t)
    if isinstance(out, tuple):
        loss = (out[0] * torch.as_tensor(np.asarray(cot)).to(dtype)).sum()
        if cot_ph is not None:
            loss = loss + (out[1] * torch.as_tensor(np.asarray(cot_ph)).to(dtype)).sum()
        out = (out[0].detach(), out[1].detach())
    else:
        loss = (out * torch.as_tensor(np.asarray(cot)).to(dtype)).sum()
        out = out.detach()
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    grads["c"] = c.grad
    if ar_t is not None:
        grads["ar"] = ar_t.grad
    return out, grads


def check_packed(gold, name, arr, tol):
    """Compare ``arr`` with a fixture entry written by oracle/make_golden_grad.py (full tensor, or sum / |.|-sum / 64 samples).
    Returns the worst relative deviation (each relative to the entry's own scale)."""
    flat = np.asarray(arr.detach().cpu() if hasattr(arr, "detach") else arr, dtype=np.float64).reshape(-1)
    if name + "::full" in gold:
        ref = gold[name + "::full"].astype(np.float64)
        assert ref.shape == flat.shape, (name, ref.shape, flat.shape)
        return float(np.abs(flat - ref).max() / max(np.abs(ref).max(), 1e-30))
    idx = gold[name + "::idx"]
    vals = gold[name + "::vals"].astype(np.float64)
    scale = max(np.abs(vals).max(), float(gold[name + "::abssum"]) / flat.size, 1e-30)
    e1 = float(np.abs(flat[idx] - vals).max() / scale)
    e2 = abs(float(np.abs(flat).sum()) - float(gold[name + "::abssum"])) / max(float(gold[name + "::abssum"]), 1e-30)
    e3 = abs(float(flat.sum()) - float(gold[name + "::sum"])) / max(float(gold[name + "::abssum"]), 1e-30)
    return max(e1, e2, e3)


# --------------------------------------------------------------------------------------
# independent definition-level restatement (numpy float64, small sizes only)
# --------------------------------------------------------------------------------------
def _np_lrelu(x, slope):
    return np.where(x >= 0, x, x * slope)


def _np_conv1d(x, wt, b, dilation, padding):
    """y[b,co,t] = bias[co] + sum_ci sum_k W[co,ci,k] * x[b,ci,t + k*d - p]  (zero outside)."""
    B, Ci, L = x.shape
    Co, _, K = wt.shape
    xp = np.zeros((B, Ci, L + 2 * padding))
    xp[:, :, padding:padding + L] = x
    y = np.zeros((B, Co, L))
    for k in range(K):
        seg = xp[:, :, k * dilation:k * dilation + L]  # x[t + k*d - p]
        y += np.einsum("oc,bct->bot", wt[:, :, k], seg)
    if b is not None:
        y += b[None, :, None]
    return y


def _np_conv_transpose1d(x, wt, b, stride, padding, output_padding):
    """y[b,co,i*s - p + k] += x[b,ci,i] * W[ci,co,k];  L_out = (L-1)s - 2p + K + op."""
    B, Ci, L = x.shape
    _, Co, K = wt.shape
    Lout = (L - 1) * stride - 2 * padding + K + output_padding
    y = np.zeros((B, Co, Lout))
    for i in range(L):
        for k in range(K):
            t = i * stride - padding + k
            if 0 <= t < Lout:
                y[:, :, t] += x[:, :, i] @ wt[:, :, k]
    return y + b[None, :, None]


def naive_forward(w, params, c, ar=None):
    """Same network from the defining sums, float64 numpy.  Slow: use T <= ~16 and narrow models."""
    p = _cfg(params)
    W = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    c = np.asarray(c, dtype=np.float64)
    slope = p["nonlinear_activation_params"]["negative_slope"]
    if p["use_ar"]:
        h = np.asarray(ar, dtype=np.float64).reshape(c.shape[0], -1)
        for li in range(5):
            h = h @ W[f"ar_model.model.{2 * li}.weight"].T + W[f"ar_model.model.{2 * li}.bias"]
            if li < 4:
                h = _np_lrelu(h, 0.1)
        c = np.concatenate([c, np.repeat(h[:, :, None], c.shape[2], axis=2)], axis=1)
    ks = p["kernel_size"]
    c = _np_conv1d(c, W["input_conv.weight"], W["input_conv.bias"], 1, (ks - 1) // 2)
    nb = len(p["resblock_kernel_sizes"])
    for i, (s, k) in enumerate(zip(p["upsample_scales"], p["upsample_kernel_sizes"])):
        c = _np_conv_transpose1d(_np_lrelu(c, slope), W[f"upsamples.{i}.1.weight"], W[f"upsamples.{i}.1.bias"],
                                 s, s // 2 + s % 2, s % 2)
        cs = 0.0
        for j in range(nb):
            x = c
            kk = p["resblock_kernel_sizes"][j]
            for idx, d in enumerate(p["resblock_dilations"][j]):
                p1 = f"blocks.{i * nb + j}.convs1.{idx}.1"
                xt = _np_conv1d(_np_lrelu(x, slope), W[p1 + ".weight"], W.get(p1 + ".bias"), d, (kk - 1) // 2 * d)
                if p["use_additional_convs"]:
                    p2 = f"blocks.{i * nb + j}.convs2.{idx}.1"
                    xt = _np_conv1d(_np_lrelu(xt, slope), W[p2 + ".weight"], W.get(p2 + ".bias"), 1, (kk - 1) // 2)
                x = xt + x
            cs = cs + x
        c = cs / nb
    c = _np_conv1d(_np_lrelu(c, 0.01), W["output_conv.1.weight"], W["output_conv.1.bias"], 1, (ks - 1) // 2)
    return np.tanh(c) if p["use_tanh"] else c
