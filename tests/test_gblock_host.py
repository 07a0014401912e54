"""Host side of ``GBlockGenerator`` without a GPU: the plugin lookup, the reference's state_dict keys (gold_gblock_keys.txt, written from the real
class by oracle/make_golden_gblock.py), weight-norm fold, construction-time errors, and that nothing runs on the CPU."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from articulatory_amd import models
from articulatory_amd.models import GBlockGenerator
from articulatory_amd.utils.synth import synth_gblock_state_dict
from oracle import gblock_oracle as G

P = dict(in_channels=141, out_channels=1, channels=512, kernel_size=7, g_scales=[5, 1, 4, 1, 1, 2, 1, 2, 1, 1], g_kernel_sizes=[3] * 10,
         use_weight_norm=True, use_ar=True, ar_input=512, ar_hidden=256, ar_output=128, use_tanh=True, use_spk_id=True, num_spk=4)


def test_plugin_lookup_and_reference_state_dict_keys():
    cls = getattr(models, "GBlockGenerator")  # train.py:1649-1653 / utils.py:325-328 do exactly this on articulatory.models
    g = cls(**P)
    lines = open(os.path.join(GOLDEN, "gold_gblock_keys.txt")).read().strip().splitlines()
    ref = [(l.split()[0], tuple(int(s) for s in l.split()[1:])) for l in lines]
    assert [(k, tuple(v.shape)) for k, v in g.state_dict().items()] == ref
    assert sum(v.numel() for v in g.parameters()) == sum(int(np.prod(s)) for _, s in ref)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in synth_gblock_state_dict(P).items()})  # a reference-layout checkpoint loads


def test_remove_weight_norm_bakes_the_fold():
    p = dict(P, channels=64, use_spk_id=False)
    sd = synth_gblock_state_dict(p, seed=5)
    g = GBlockGenerator(**p)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    w = G.fold_weight_norm(sd)
    got = g.state_dict()
    assert sorted(got) == sorted(w)
    for k in w:
        assert torch.allclose(got[k], w[k], rtol=1e-6, atol=1e-8), k
    g.apply_weight_norm()
    assert "resamples.3.conv2.3.weight_g" in g.state_dict()


def test_configurations_the_reference_cannot_run_are_rejected_at_construction():
    with pytest.raises(ValueError, match="9 or 10 GBlocks"):
        GBlockGenerator()  # the reference's own defaults: forward fails there with a size mismatch
    with pytest.raises(ValueError, match="odd"):
        GBlockGenerator(**dict(P, g_kernel_sizes=[4] * 10))
    with pytest.raises(ValueError, match="channel plan"):
        GBlockGenerator(**dict(P, g_scales=[1] * 11, g_kernel_sizes=[3] * 11))
    with pytest.raises(AssertionError):
        GBlockGenerator(**dict(P, kernel_size=6))
    GBlockGenerator(**dict(P, g_scales=[5, 1, 4, 1, 1, 2, 1, 2, 1], g_kernel_sizes=[3] * 9))  # nine GBlocks end at channels // 8 too


def test_no_cpu_path():
    g = GBlockGenerator(**dict(P, channels=64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 13, 4), ar=torch.zeros(1, 1, 512), spk_id=torch.zeros(1, dtype=torch.long))
    with pytest.raises(TypeError, match="phoneme"):
        g(torch.zeros(1, 13, 4), ar=torch.zeros(1, 1, 512), ph=torch.zeros(1, 4, dtype=torch.long))
