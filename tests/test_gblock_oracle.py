"""The GBlockGenerator oracle (oracle/gblock_oracle.py) against the golden vectors of the REAL reference class
(oracle/make_golden_gblock.py; articulatory/models/gblock_gen.py:14-132, articulatory/layers/pytorch_layers.py:32-91).

The oracle runs the same ATen operators on the same fp32 inputs: every GBlock output to 2e-6 of its scale; the waveform to 1e-5 (42 convs
deep without any normalisation and a pre-tanh scale of ~8, the REFERENCE's own fp32 output is 2.8e-6 away from its float64 run — measured,
the oracle in float64 is as far from the fixture as the oracle in fp32); gradients (kink-free seeds chosen by the generator
script) to 2e-5 of each tensor's scale.
"""

import ast
import os

import numpy as np
import torch

from conftest import GOLDEN, rel_err
from articulatory_amd.utils.synth import gblock_param_spec, synth_gblock_state_dict
from oracle import gblock_oracle as G
from oracle.hificar_oracle import check_packed

TOL = 2e-6
TOL_OUT = 1e-5


def _params(g, key="params"):
    return dict(ast.literal_eval(str(g[key])))


def _folded(params, seed=1234, dtype=torch.float32):
    return G.fold_weight_norm(synth_gblock_state_dict(params, seed=seed), dtype=dtype)


def test_state_dict_keys_match_reference():
    lines = open(os.path.join(GOLDEN, "gold_gblock_keys.txt")).read().strip().splitlines()
    ref = [(l.split()[0], tuple(int(s) for s in l.split()[1:])) for l in lines]
    p = dict(in_channels=141, channels=512, g_scales=[5, 1, 4, 1, 1, 2, 1, 2, 1, 1], g_kernel_sizes=[3] * 10, use_ar=True,
             use_spk_id=True, num_spk=4)
    assert ref == [(k, tuple(v)) for k, v in gblock_param_spec(**p).items()]
    # ten GBlocks x five convs + input / output conv = 52 convs x (bias, weight_g, weight_v) + 10 PastFCEncoder + 3 speaker tensors
    assert len(ref) == 52 * 3 + 10 + 3
    # nearest-upsample moves the conv indices of a GBlock's Sequentials by one (pytorch_layers.py:46-57)
    names = [k for k, _ in ref]
    assert "resamples.0.conv1.2.weight_v" in names and "resamples.0.res1.1.bias" in names      # upsample 5
    assert "resamples.1.conv1.1.weight_v" in names and "resamples.1.res1.0.bias" in names      # upsample 1


def test_small_model_every_block():
    g = np.load(os.path.join(GOLDEN, "gold_gblock_small.npz"))
    p = _params(g)
    w = _folded(p)
    taps = {}
    with torch.no_grad():
        y = G.generator_forward(w, p, torch.from_numpy(g["c"]), torch.from_numpy(g["ar"]), taps=taps)
    assert rel_err(y.numpy(), g["out"]) < TOL_OUT
    assert rel_err(taps["ar_feats"].numpy(), g["tap::ar_feats"]) < TOL
    assert rel_err(taps["input_conv"].numpy(), g["tap::input_conv"]) < TOL
    for i in range(10):
        assert rel_err(taps[f"resamples.{i}"].numpy(), g[f"tap::resamples.{i}"]) < TOL, i
        assert rel_err(taps[f"resamples.{i}.res1"].numpy(), g[f"tap::resamples.{i}.res1"]) < TOL, i
        # conv1(x) + res1(x) is the stream conv2 is added to (pytorch_layers.py:89)
        assert rel_err(taps[f"resamples.{i}.mid"].numpy(), g[f"tap::resamples.{i}.conv1"] + g[f"tap::resamples.{i}.res1"]) < TOL, i
    assert y.shape == (2, 1, 80 * 8)


def test_naive_definition_agrees():
    """The float64 restatement from the defining sums (nearest upsample out[t] = in[t // s], zero padding, dilations) on the same fixture."""
    g = np.load(os.path.join(GOLDEN, "gold_gblock_small.npz"))
    p = _params(g)
    w = _folded(p)
    y = G.naive_forward({k: v.numpy() for k, v in w.items()}, p, g["c"][:1, :, :4], g["ar"][:1])
    with torch.no_grad():
        y_ref = G.generator_forward(_folded(p, dtype=torch.float64), p, torch.from_numpy(g["c"][:1, :, :4]).double(),
                                    torch.from_numpy(g["ar"][:1]).double())
    assert rel_err(y, y_ref.numpy()) < 1e-12


def test_full_width_model():
    g = np.load(os.path.join(GOLDEN, "gold_gblock_full.npz"))
    p = _params(g)
    assert p["channels"] == 512
    w = _folded(p)
    taps = {}
    with torch.no_grad():
        y = G.generator_forward(w, p, torch.from_numpy(g["c"]), torch.from_numpy(g["ar"]), taps=taps)
    assert rel_err(y.numpy(), g["out"]) < TOL_OUT
    for i in range(10):
        assert check_packed(g, f"tap::resamples.{i}", taps[f"resamples.{i}"], TOL) < TOL * 5, i


def test_kernel5_speaker_model():
    g = np.load(os.path.join(GOLDEN, "gold_gblock_k5spk.npz"))
    p = _params(g)
    assert p["g_kernel_sizes"] == [5] * 10 and p["use_spk_id"]
    w = _folded(p)
    taps = {}
    with torch.no_grad():
        y = G.generator_forward(w, p, torch.from_numpy(g["c"]), torch.from_numpy(g["ar"]), spk_id=torch.from_numpy(g["spk_id"]), taps=taps)
    assert rel_err(y.numpy(), g["out"]) < TOL_OUT
    for i in range(10):
        assert check_packed(g, f"tap::resamples.{i}", taps[f"resamples.{i}"], TOL) < TOL * 5, i


def test_ar_loop_vs_reference_ar_loop():
    g = np.load(os.path.join(GOLDEN, "gold_gblock_arloop.npz"))
    p = _params(np.load(os.path.join(GOLDEN, "gold_gblock_small.npz")))
    w = _folded(p)
    for tag in ("c25", "c100"):
        x = torch.from_numpy(g[f"{tag}_x"])
        bms = int(g[f"{tag}_batch_max_steps"])
        with torch.no_grad():
            y = G.ar_loop(w, p, x, bms, 80)
            yb = G.ar_loop_batched(w, p, x[None], bms, 80)[0]
        assert y.shape == (80 * len(x),)
        assert rel_err(y.numpy(), g[f"{tag}_out"]) < 2e-5, tag  # (chained chunks on top of the single forward's 3e-6)
        assert rel_err(yb.numpy(), g[f"{tag}_out"]) < 2e-5, tag


def test_nonar_inference():
    g = np.load(os.path.join(GOLDEN, "gold_gblock_nonar.npz"))
    p = _params(g)
    w = _folded(p)
    with torch.no_grad():  # gblock_gen.py:185-190 on a (T,) input: (T,) -> (1, 1, T) -> forward -> (80 T, 1)
        y = G.generator_forward(w, p, torch.from_numpy(g["x"])[None, None, :]).squeeze(0).transpose(1, 0)
    assert y.shape == g["out"].shape
    assert rel_err(y.numpy(), g["out"]) < TOL_OUT


def test_gradients_vs_reference_autograd():
    g = np.load(os.path.join(GOLDEN, "gold_gblock_grad.npz"))
    for tag in ("k3", "k5spk"):
        sub = {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "/")}
        p = _params(sub)
        assert float(sub["margin"]) >= 2e-6  # every ReLU input of the fixture stays clear of its kink
        sd = synth_gblock_state_dict(p, seed=int(sub["seed"]))
        out, grads = G.gradients(sd, p, sub["c"], sub["ar"], sub["cot"], spk_id=sub.get("spk_id"))
        assert check_packed(sub, "out", out, TOL_OUT) < TOL_OUT
        worst = max(check_packed(sub, "grad::" + k, v, 2e-5) for k, v in grads.items())
        assert worst < 2e-5, (tag, worst)
        # the float64 margin helper agrees with what the generator script measured on the reference's own modules
        w64 = G.fold_weight_norm(sd, dtype=torch.float64)
        m = G.relu_margin(w64, p, torch.from_numpy(sub["c"]).double(), torch.from_numpy(sub["ar"]).double(),
                          spk_id=torch.from_numpy(sub["spk_id"]) if "spk_id" in sub else None)
        assert abs(m - float(sub["margin"])) <= 0.05 * float(sub["margin"])
