"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the golden vectors captured from
the real reference.  Run on a MI355X: ``pytest -m gpu``.

Tolerance: BASELINE.json's north_star states 1e-3 relative fp32 (max|y_gpu - y_ref| / max|y_ref|).
The fp32-MFMA path is exact-fp32 arithmetic in a different summation order, so we hold it to 2e-5;
the split-bf16 path (when selected) to 2e-4.  Both are far inside the stated 1e-3.
"""

import os

import numpy as np
import pytest
import torch

from conftest import E2W_PARAMS, GOLDEN, rel_err, same_across_shapes
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from oracle import hificar_oracle as O

pytestmark = pytest.mark.gpu

NORTH_STAR_TOL = 1e-3
# every test runs for both conv arithmetics unless HIFICAR_PRECISION narrows it
PRECISIONS = [os.environ["HIFICAR_PRECISION"]] if os.environ.get("HIFICAR_PRECISION") else ["f32", "bf16x3"]
TOLS = {"f32": 2e-5, "bf16x3": 2e-4}
# same utterance, different launch shapes (split-K vs dense accumulation order): fp32 rounding for the exact arithmetic; the bf16x3
# arithmetic re-splits every activation into hi + lo, so a last-bit change moves its own 2^-16-level rounding pattern
XSHAPE_TOL = {"f32": 5e-6, "bf16x3": 2e-4}


def _require_gpu():
    assert torch.cuda.is_available(), "these tests need a GPU; run with -m 'not gpu' on CPU boxes"


@pytest.fixture(params=PRECISIONS, scope="module")
def prec(request):
    return request.param


def make(params, prec, seed=1234, remove_wn=True):
    _require_gpu()
    sd = synth_state_dict(params, seed=seed)
    g = HiFiGANGenerator(**params, precision=prec)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    if remove_wn:
        g.remove_weight_norm()
    return g.eval().to("cuda:0"), O.fold_weight_norm(sd)


@pytest.fixture(scope="module")
def car(prec):
    g, w = make(dict(E2W_PARAMS), prec)
    g.tol = TOLS[prec]
    return g, w


def test_native_library_is_the_path_that_runs(car):
    g, _ = car
    with torch.no_grad():
        g(torch.zeros(1, 13, 4, device="cuda:0"), ar=torch.zeros(1, 1, 512, device="cuda:0"))
    assert g._handle is not None
    maps = open("/proc/self/maps").read()
    assert "libhificar.so" in maps


def test_golden_full_forward(car):
    g, _ = car
    gold = np.load(os.path.join(GOLDEN, "gold_fwd_full.npz"))
    with torch.no_grad():
        y = g(torch.from_numpy(gold["c"]).cuda(), ar=torch.from_numpy(gold["ar"]).cuda())
    assert y.shape == (2, 1, 2000)
    assert rel_err(y.cpu().numpy(), gold["out"]) < g.tol


def test_golden_ar_loop_ragged_tail(car):
    from articulatory_amd.bin.decode import ar_loop
    g, _ = car
    gold = np.load(os.path.join(GOLDEN, "gold_arloop.npz"))
    x = torch.from_numpy(gold["x"]).cuda()
    for bms in (2000, 8000):
        config = dict(batch_max_steps=bms, hop_size=80, generator_params=E2W_PARAMS, dataset_mode="a2w")
        with torch.no_grad():
            y = ar_loop(g, x, config)
        assert y.shape == (20800,)
        assert rel_err(y.cpu().numpy(), gold[f"out_bms{bms}"]) < 2 * g.tol, bms


def test_golden_predict_wav_utterance(car):
    g, _ = car
    gold = np.load(os.path.join(GOLDEN, "gold_predict_wav.npz"))
    with torch.no_grad():
        y = g.ar_synthesis(torch.from_numpy(gold["x"]).cuda().t().unsqueeze(0), 100)
    assert rel_err(y[0].cpu().numpy(), gold["out"]) < 2 * g.tol


def test_golden_nonar_inference(prec):
    params = dict(E2W_PARAMS, in_channels=12, use_ar=False)
    g, _ = make(params, prec)
    g.tol = TOLS[prec]
    gold = np.load(os.path.join(GOLDEN, "gold_nonar.npz"))
    with torch.no_grad():
        y = g.inference(gold["x"])  # ndarray in, as predict_wav.py:136 may pass
    assert y.shape == (24000, 1)
    assert rel_err(y.cpu().numpy(), gold["out"]) < g.tol


def test_golden_small_model_every_layer(prec):
    """The width-64 model of the reference's per-layer fixture (stage widths 32 / 16 / 8 / 4: narrower than an MFMA tile,
    padded to 32 channels internally) — EVERY tapped layer output of the real reference (forward hooks on its modules,
    oracle/make_golden.py) against the device's intermediates (hificar_debug_tap), not just the final waveform."""
    params = dict(E2W_PARAMS, channels=64)
    g, _ = make(params, prec)
    gold = np.load(os.path.join(GOLDEN, "gold_fwd_small.npz"))
    names = ["ar_feats", "input_conv"] + [f"upsamples.{i}" for i in range(4)] + [f"blocks.{b}" for b in range(12)]
    for b in (0, 7):
        names += [f"blocks.{b}.convs1.{d}" for d in range(3)] + [f"blocks.{b}.x.{d}" for d in range(3)]
    with torch.no_grad():
        y, taps = g.debug_taps(names, torch.from_numpy(gold["c"]).cuda(), ar=torch.from_numpy(gold["ar"]).cuda())
        y_plain = g(torch.from_numpy(gold["c"]).cuda(), ar=torch.from_numpy(gold["ar"]).cuda())
    tol = TOLS[prec]
    assert rel_err(y.cpu().numpy(), gold["out"]) < tol and rel_err(y_plain.cpu().numpy(), gold["out"]) < tol
    want = {"ar_feats": gold["tap.ar_feats"], "input_conv": gold["tap.input_conv"]}
    for i in range(4):
        want[f"upsamples.{i}"] = gold[f"tap.upsample{i}"]
    for b in range(12):
        want[f"blocks.{b}"] = gold[f"tap.block{b}"]
    for b in (0, 7):
        x = gold[f"tap.upsample{b // 3}"]
        for d in range(3):
            want[f"blocks.{b}.convs1.{d}"] = gold[f"tap.block{b}.convs1.{d}"]
            x = gold[f"tap.block{b}.convs2.{d}"] + x  # residual_block.py:221
            want[f"blocks.{b}.x.{d}"] = x
    assert sorted(want) == sorted(names)
    for name in names:
        got = taps[name].cpu().numpy()
        assert got.shape == want[name].shape, name
        assert np.isfinite(got).all(), name
        assert rel_err(got, want[name]) < tol, name


@pytest.mark.gpu
def test_debug_taps_at_a_length_that_is_not_a_whole_bucket():
    """A forward of more than 32 frames covers a bucket of 32-frame multiples (the tail is masked): the debug taps hold the bucket's rows and
    debug_taps returns the utterance's own.  The tapped last block reproduces the waveform through the output conv's oracle arithmetic."""
    params = dict(E2W_PARAMS, channels=64)
    g, w = make(params, "f32")
    B, T = 2, 40
    c = torch.from_numpy(synth_features(B, T, 13, seed=41)).permute(0, 2, 1).contiguous().cuda()
    ar = torch.from_numpy(synth_features(B, 512, 1, seed=42)[:, :, 0] * 0.3).reshape(B, 1, 512).cuda()
    names = ["input_conv", "upsamples.0", "blocks.11"]
    with torch.no_grad():
        y, taps = g.debug_taps(names, c, ar=ar)
        ref = O.generator_forward(w, params, c.cpu(), ar.cpu())
    assert rel_err(y.cpu().numpy(), ref.numpy()) < TOLS["f32"]
    assert tuple(taps["input_conv"].shape) == (B, 64, T)
    assert tuple(taps["upsamples.0"].shape) == (B, 32, T * 5)
    assert tuple(taps["blocks.11"].shape) == (B, 4, T * 80)
    for n in names:
        assert bool(torch.isfinite(taps[n]).all()), n


def test_golden_small_mri_model(prec):
    """MRI-shaped narrow model (scales 8/5/3/2, stage widths 32/16/8/4) against the reference's own output."""
    params = dict(E2W_PARAMS, channels=64, in_channels=20 + 128, upsample_scales=[8, 5, 3, 2], upsample_kernel_sizes=[16, 10, 6, 4])
    g, _ = make(params, prec)
    gold = np.load(os.path.join(GOLDEN, "gold_fwd_small_mri.npz"))
    with torch.no_grad():
        y = g(torch.from_numpy(gold["c"]).cuda(), ar=torch.from_numpy(gold["ar"]).cuda())
    assert rel_err(y.cpu().numpy(), gold["out"]) < TOLS[prec]


def test_golden_speaker_conditioning(prec):
    """use_spk_id (hifigan.py:176-178, 212-216) against the real reference's output (oracle/make_golden_cond.py)."""
    params = dict(E2W_PARAMS, channels=128, use_spk_id=True, num_spk=5, spk_emb_size=32)
    g, w = make(params, prec, seed=4321)
    gold = np.load(os.path.join(GOLDEN, "gold_fwd_spk.npz"))
    c, ar, spk = (torch.from_numpy(gold[k]).cuda() for k in ("c", "ar", "spk_id"))
    with torch.no_grad():
        y = g(c, spk_id=spk, ar=ar)
        y_other = g(c, spk_id=(spk + 1) % 5, ar=ar)
    assert rel_err(y.cpu().numpy(), gold["out"]) < TOLS[prec]
    assert rel_err(y_other.cpu().numpy(), gold["out"]) > 1e-2  # the speaker vector matters
    with pytest.raises(RuntimeError, match="spk_id"):
        g(c, ar=ar)
    with pytest.raises(IndexError):
        g(c, spk_id=spk + 5, ar=ar)
    with pytest.raises(ValueError, match="hificar_forward_cond"):
        g.ar_synthesis(c, 25)  # the reference's ar_loop passes no spk_id either (decode.py:72)


def test_golden_phoneme_conditioning_and_loss_head(prec):
    """use_ph + use_ph_loss (hifigan.py:179-189, 217-220, 232-237): waveform and frame-rate phoneme logits against the real
    reference's pair (out, ph_out); ragged lengths leave the frames past an utterance's end untouched."""
    params = dict(E2W_PARAMS, channels=128, in_channels=20, use_ar=False, use_ph=True, num_ph=11, ph_emb_size=8, use_ph_loss=True)
    g, w = make(params, prec, seed=4322)
    gold = np.load(os.path.join(GOLDEN, "gold_fwd_ph.npz"))
    c, ph = torch.from_numpy(gold["c"]).cuda(), torch.from_numpy(gold["ph"]).cuda()
    with torch.no_grad():
        y, ph_out = g(c, ph=ph)
    assert rel_err(y.cpu().numpy(), gold["out"]) < TOLS[prec]
    assert ph_out.shape == (2, 11, 19)
    assert rel_err(ph_out.cpu().numpy(), gold["ph_out"]) < TOLS[prec]
    # a longer, ragged batch against the oracle (bucketed launch geometry: 45 frames run as 64)
    B, T, lens = 3, 45, [45, 20, 33]
    x = synth_features(B, T, 12, seed=71)
    phs = torch.from_numpy(np.random.default_rng(72).integers(0, 11, (B, T)))
    cc = torch.from_numpy(x).permute(0, 2, 1).contiguous()
    with torch.no_grad():
        y, ph_out = g(cc.cuda(), ph=phs.cuda(), lengths=lens)
        for b, n in enumerate(lens):
            ry, rp = O.generator_forward(w, params, cc[b:b + 1, :, :n], ph=phs[b:b + 1, :n])
            assert rel_err(y[b:b + 1, :, :80 * n].cpu().numpy(), ry.numpy()) < TOLS[prec], b
            assert rel_err(ph_out[b:b + 1, :, :n].cpu().numpy(), rp.numpy()) < TOLS[prec], b
            assert float(ph_out[b, :, n:].abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="ph="):
        g(c)


@pytest.mark.parametrize("B,T", [(1, 1), (1, 7), (3, 33), (8, 25), (2, 129), (5, 64)])
def test_forward_vs_oracle_shapes(car, B, T):
    g, w = car
    c = torch.from_numpy(synth_features(B, T, 13, seed=1000 + 7 * B + T)).permute(0, 2, 1).contiguous()
    ar = torch.from_numpy(synth_features(B, 512, 1, seed=2000 + T)[:, :, 0] * 0.5 - 0.25).reshape(B, 1, 512)
    with torch.no_grad():
        y = g(c.cuda(), ar=ar.cuda()).cpu()
        y_ref = O.generator_forward(w, E2W_PARAMS, c, ar)
    assert y.shape == y_ref.shape == (B, 1, 80 * T)
    assert rel_err(y.numpy(), y_ref.numpy()) < g.tol


def test_forward_without_remove_weight_norm_equals_folded(prec):
    """Evaluating with weight norm still applied (as the reference trainer's eval does) folds on the fly."""
    g1, _ = make(dict(E2W_PARAMS), prec, remove_wn=False)
    g2, _ = make(dict(E2W_PARAMS), prec, remove_wn=True)
    c = torch.from_numpy(synth_features(2, 10, 13, seed=5)).permute(0, 2, 1).contiguous().cuda()
    ar = torch.zeros(2, 1, 512, device="cuda:0")
    with torch.no_grad():
        # the fold runs on the device here and on the host there: ulp-level differences in ||v|| only
        # (which the bf16 split may round differently)
        assert rel_err(g1(c, ar=ar).cpu().numpy(), g2(c, ar=ar).cpu().numpy()) < (2e-6 if prec == "f32" else TOLS[prec])


UNIT_CONFIGS = {
    # name: overrides of E2W_PARAMS exercising one kernel shape each (SURVEY.md §4 unit level)
    "up5_k3_d1": dict(channels=64, upsample_scales=[5], upsample_kernel_sizes=[10], resblock_kernel_sizes=[3],
                      resblock_dilations=[[1]]),
    "up4_k7_d135": dict(channels=256, upsample_scales=[4], upsample_kernel_sizes=[8], resblock_kernel_sizes=[7],
                        resblock_dilations=[[1, 3, 5]]),
    "up2_k11_d5_twoblocks": dict(channels=128, upsample_scales=[2], upsample_kernel_sizes=[4],
                                 resblock_kernel_sizes=[11, 3], resblock_dilations=[[5], [1, 3]]),
    "two_stages_c256_c128": dict(channels=512, upsample_scales=[2, 2], upsample_kernel_sizes=[4, 4]),
    "mri_scales_8532": dict(in_channels=148, upsample_scales=[8, 5, 3, 2], upsample_kernel_sizes=[16, 10, 6, 4]),
    "no_resblock_bias_no_tanh": dict(channels=128, upsample_scales=[2, 2], upsample_kernel_sizes=[4, 4], bias=False,
                                     use_tanh=False),
    "nonar_in80": dict(in_channels=80, use_ar=False, channels=128, upsample_scales=[4, 2], upsample_kernel_sizes=[8, 4]),
    "k9_input_kernel": dict(channels=128, kernel_size=9, upsample_scales=[3, 2], upsample_kernel_sizes=[6, 4]),
    # widths that are not MFMA-tile multiples (the reference accepts any, hifigan.py:108-145: channels // 2**i)
    "narrow_c48_24_12": dict(channels=48, upsample_scales=[4, 2], upsample_kernel_sizes=[8, 4]),
    "odd_c100_50_25_12": dict(channels=100, upsample_scales=[2, 3, 2], upsample_kernel_sizes=[4, 6, 4], resblock_kernel_sizes=[3, 7],
                              resblock_dilations=[[1, 3], [1, 3, 5]]),
    "tiny_c8_nonar": dict(channels=8, in_channels=5, use_ar=False, upsample_scales=[2, 2], upsample_kernel_sizes=[4, 4]),
}


@pytest.mark.parametrize("name", sorted(UNIT_CONFIGS))
def test_unit_kernel_shapes(name, prec):
    params = dict(E2W_PARAMS, use_tanh=True)
    params.update(UNIT_CONFIGS[name])
    g, w = make(params, prec, seed=99)
    g.tol = TOLS[prec]
    cf = params["in_channels"] - (128 if params["use_ar"] else 0)
    for B, T in ((2, 37), (1, 200)):
        c = torch.from_numpy(synth_features(B, T, cf, seed=31 + T)).permute(0, 2, 1).contiguous()
        ar = torch.from_numpy(synth_features(B, 512, 1, seed=32)[:, :, 0] * 0.3).reshape(B, 1, 512) if params["use_ar"] else None
        with torch.no_grad():
            y = g(c.cuda(), ar=ar.cuda() if ar is not None else None).cpu()
            y_ref = O.generator_forward(w, params, c, ar)
        assert y.shape == y_ref.shape
        assert rel_err(y.numpy(), y_ref.numpy()) < g.tol, (name, B, T)


def test_baseline_size_properties(car):
    """BASELINE config 3 at full size (batch 64, 10 s, chunk 25): properties that need no oracle run.
    (a) utterances are independent: a batch-64 run reproduces a batch-1 run of the same utterance (bit for bit with HIFICAR_KSPLIT=0, to fp32 rounding otherwise)
        and duplicated utterances give duplicated waveforms;  (b) the first chunk equals forward(ar = 0);
    (c) an oracle check on a window: chunk k of utterance u equals the oracle's forward of that chunk fed
        with the GPU's own previous 512 samples."""
    g, w = car
    B, T, chunk = 64, 2000, 25
    x = synth_features(B, T, 13, seed=20260929 + 3)
    x[63] = x[0]
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        y = g.ar_synthesis(feats, chunk)
        y1 = g.ar_synthesis(feats[17:18].contiguous(), chunk)
        first = g(feats[:, :, :chunk].contiguous(), ar=torch.zeros(B, 1, 512, device="cuda:0"))
    assert y.shape == (64, 160000) and bool(torch.isfinite(y).all())
    assert torch.equal(y[0], y[63])
    assert same_across_shapes(y[17], y1[0], XSHAPE_TOL[g.precision])  # batch 64 vs batch 1: different launch shapes
    assert torch.equal(y[:, :2000], first[:, 0])
    assert float(y.abs().max()) <= 1.0  # tanh range
    yc = y.cpu()
    for u, k in ((3, 1), (40, 57), (63, 79)):
        cin = torch.from_numpy(x[u:u + 1, k * chunk:(k + 1) * chunk]).permute(0, 2, 1)
        prev = yc[u:u + 1, k * 2000 - 512:k * 2000].reshape(1, 1, 512)
        with torch.no_grad():
            ref = O.generator_forward(w, E2W_PARAMS, cin, prev)
        assert rel_err(yc[u, k * 2000:(k + 1) * 2000].numpy(), ref[0, 0].numpy()) < g.tol


_FULL_ORACLE = {}


def _full_oracle(chunk_frames, n_utt):
    """The CPU oracle on whole 10-s utterances of BASELINE config 3, computed once per chunk size."""
    if chunk_frames not in _FULL_ORACLE:
        x = synth_features(64, 2000, 13, seed=20260929 + 3)[:n_utt]
        w = O.fold_weight_norm(synth_state_dict(dict(E2W_PARAMS), seed=1234))
        torch.set_num_threads(min(16, os.cpu_count() or 1))  # the oracle's best thread count on the GPU box's host
        with torch.no_grad():
            _FULL_ORACLE[chunk_frames] = (x, O.ar_loop_batched(w, E2W_PARAMS, torch.from_numpy(x), 80 * chunk_frames, 80))
    return _FULL_ORACLE[chunk_frames]


@pytest.mark.parametrize("chunk_frames,n_utt", [(25, 64), (100, 16)])
def test_baseline_size_every_sample_vs_oracle(car, chunk_frames, n_utt):
    """BASELINE config 3 at full size against the oracle on every one of the 10.24 M samples (80 chained AR steps:
    the whole feedback path is inside the comparison); 16 utterances at chunk 100 (the oracle is slow there)."""
    g, _ = car
    x, ref = _full_oracle(chunk_frames, n_utt)
    with torch.no_grad():
        y = g.ar_synthesis(torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda(), chunk_frames).cpu()
    assert y.shape == ref.shape == (n_utt, 160000)
    assert rel_err(y.numpy(), ref.numpy()) < g.tol
    # per utterance too: no single utterance may hide behind the loudest one
    per_utt = (y - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
    assert float(per_utt.max()) < NORTH_STAR_TOL


# Trained HiFi-GAN checkpoints drive the waveform near full scale (ours are unreachable offline), the plain
# synthetic weights only to max|y| ~ 0.2.  Two rescalings of the same synthetic checkpoint put the output where trained
# models live: (gain, out_gain) = every conv's weight_v scale, and the output conv's weight_g scale (tests/dev/
# trained_scale_probe.py: peaks 0.91 and 0.96, 15-27 % of the samples above 0.5).  Scaling EVERY layer by 1.3 instead makes
# the synthetic network chaotic through the AR feedback — two CPU computations of it (fp32 vs fp64) then differ by 0.66 of
# full scale (tests/test_oracle_golden.py::test_saturated_synthetic_network_is_chaotic_on_cpu_too), so no implementation can
# be compared there.
TRAINED_SCALE = [(1.0, 6.0), (1.15, 3.0)]
_TRAINED_ORACLE = {}


def _trained_scale_sd(gain, out_gain):
    sd = synth_state_dict(dict(E2W_PARAMS), seed=1234, gain=gain)
    sd["output_conv.1.weight_g"] = sd["output_conv.1.weight_g"] * np.float32(out_gain)
    return sd


@pytest.mark.parametrize("gain,out_gain", TRAINED_SCALE)
def test_trained_checkpoint_scale_full_ar_loop(prec, gain, out_gain):
    """BASELINE config 3 at full size (batch 64, 10 s, 80 chained AR steps) with the waveform near full scale, every sample of
    every utterance against the CPU oracle, both arithmetics: the 1e-3 bar in the regime trained checkpoints occupy."""
    _require_gpu()
    sd = _trained_scale_sd(gain, out_gain)
    key = (gain, out_gain)
    if key not in _TRAINED_ORACLE:
        x = synth_features(64, 2000, 13, seed=20260929 + 3)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        with torch.no_grad():
            _TRAINED_ORACLE[key] = (x, O.ar_loop_batched(O.fold_weight_norm(sd), E2W_PARAMS, torch.from_numpy(x), 2000, 80))
    x, ref = _TRAINED_ORACLE[key]
    g = HiFiGANGenerator(**E2W_PARAMS, precision=prec)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    g = g.eval().to("cuda:0")
    with torch.no_grad():
        y = g.ar_synthesis(torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda(), 25).cpu()
    peak = float(ref.abs().max())
    assert 0.85 < peak <= 1.0, peak  # near full scale, as a trained vocoder's output
    assert float((ref.abs() > 0.5).float().mean()) > 0.1
    per_utt = (y - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
    assert float(per_utt.max()) < {"f32": 2e-5, "bf16x3": 2e-4}[prec] < NORTH_STAR_TOL, (prec, float(per_utt.max()))
    # the feedback path does not amplify the difference: the last chunk is as close as the first
    e = (y - ref).abs()
    assert float(e[:, -2000:].max()) < 4 * max(float(e[:, :2000].max()), 1e-7)


@pytest.mark.parametrize("B", [1, 8])
def test_small_batch_ar_loop_vs_oracle(car, B):
    """BASELINE's batch 1 / 8 points (north_star: "throughput ... at batch 1/8/64"): the full 10-s, 80-step AR loop against the
    oracle on every sample, through the split-K conv form the small launches take."""
    g, w = car
    x = synth_features(B, 2000, 13, seed=20260929 + 30 + B)
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        y = g.ar_synthesis(feats, 25).cpu()
        g.profile_begin()
        g.ar_synthesis(feats[:, :, :50].contiguous(), 25)
        names = {s["name"].split("<")[0] for s in g.profile_end()}
        ref = O.ar_loop_batched(w, E2W_PARAMS, torch.from_numpy(x), 2000, 80)
    assert any(n.startswith("conv_sk_") for n in names), names
    per_utt = (y - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
    assert float(per_utt.max()) < g.tol, float(per_utt.max())


def test_nonar_baseline_size_vs_oracle_window(prec):
    """BASELINE config 2 (non-AR 12-dim, batch 8, 10 s): full-size run; oracle comparison on one utterance's
    interior window computed from a halo'd excerpt (receptive field of the whole generator < 60 frames)."""
    params = dict(E2W_PARAMS, in_channels=12, use_ar=False)
    g, w = make(params, prec)
    g.tol = TOLS[prec]
    B, T = 8, 2000
    x = synth_features(B, T, 12, seed=20260929 + 2)
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous()
    with torch.no_grad():
        y = g(feats.cuda()).cpu()
        lo, hi, halo = 900, 1000, 64
        ref = O.generator_forward(w, params, feats[5:6, :, lo - halo:hi + halo])
    assert y.shape == (8, 1, 160000)
    got = y[5, 0, lo * 80:hi * 80].numpy()
    want = ref[0, 0, halo * 80:(halo + hi - lo) * 80].numpy()
    assert rel_err(got, want) < g.tol
    # the sequence edges (zero padding at t = 0 and t = T) against an oracle run of the edge excerpts
    with torch.no_grad():
        ref0 = O.generator_forward(w, params, feats[0:1, :, :100])
        ref1 = O.generator_forward(w, params, feats[7:8, :, -100:])
    assert rel_err(y[0, 0, :30 * 80].numpy(), ref0[0, 0, :30 * 80].numpy()) < g.tol
    assert rel_err(y[7, 0, -30 * 80:].numpy(), ref1[0, 0, -30 * 80:].numpy()) < g.tol


def test_error_behaviour(car):
    g, _ = car
    dev = "cuda:0"
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="Expected input of shape"):
            g(torch.zeros(1, 12, 8, device=dev), ar=torch.zeros(1, 1, 512, device=dev))
        with pytest.raises(RuntimeError, match="needs ar"):
            g(torch.zeros(1, 13, 8, device=dev))
        with pytest.raises(RuntimeError, match="elements"):
            g(torch.zeros(2, 13, 8, device=dev), ar=torch.zeros(1, 1, 512, device=dev))
        with pytest.raises(ValueError, match="ar_input"):
            g.ar_synthesis(torch.zeros(1, 13, 50, device=dev), 4)  # 4 frames = 320 samples < ar_input
        # a single short chunk is fine (no feedback needed)
        assert g.ar_synthesis(torch.zeros(1, 13, 3, device=dev), 4).shape == (1, 240)
    nonar, _ = make(dict(E2W_PARAMS, in_channels=12, use_ar=False, channels=64, upsample_scales=[2],
                         upsample_kernel_sizes=[4], resblock_kernel_sizes=[3], resblock_dilations=[[1]]), "f32")
    with pytest.raises(RuntimeError, match="use_ar=True"):
        nonar.ar_synthesis(torch.zeros(1, 12, 50, device=dev), 25)


def test_profile_hooks_account_for_all_flops(car):
    g, _ = car
    feats = torch.from_numpy(synth_features(4, 50, 13, seed=3)).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        g.profile_begin()
        g.ar_synthesis(feats, 25)
        stats = g.profile_end()
    macs = 2 * g.macs(4, 25)
    assert abs(sum(s["flops"] for s in stats) / (2 * macs) - 1) < 1e-9
    assert all(s["total_ms"] > 0 for s in stats)
    assert stats[0]["name"].startswith("conv_")


def test_precisions_agree_and_switch_in_place(car):
    """set_precision flips the arithmetic of a live handle; both stay inside the north-star tolerance of each other."""
    g, _ = car
    c = torch.from_numpy(synth_features(2, 25, 13, seed=77)).permute(0, 2, 1).contiguous().cuda()
    ar = torch.zeros(2, 1, 512, device="cuda:0")
    orig = g.precision
    with torch.no_grad():
        g.set_precision("f32")
        y32 = g(c, ar=ar)
        g.set_precision("bf16x3")
        y16 = g(c, ar=ar)
        g.set_precision(orig)
    assert not torch.equal(y32, y16)
    assert rel_err(y16.cpu().numpy(), y32.cpu().numpy()) < TOLS["bf16x3"] < NORTH_STAR_TOL


def test_fused_pair_kernel_matches_layer_by_layer(monkeypatch, prec):
    """Narrow stages (C = 32 / 64) run conv1 -> conv2 as one fused kernel (both arithmetics) when the launch has enough tiles and
    the tile quantisation is small (24 x 49 frames: both widths fuse in both arithmetics); HIFICAR_PAIR=0 runs them layer by layer.
    Same arithmetic in a different tiling: agreement to rounding noise; a ragged batch covers the masking inside fused tiles."""
    params = dict(E2W_PARAMS)
    B, T = 24, 49
    c = torch.from_numpy(synth_features(B, T, 13, seed=123)).permute(0, 2, 1).contiguous().cuda()
    ar = torch.from_numpy(synth_features(B, 512, 1, seed=124)[:, :, 0] * 0.3).reshape(B, 1, 512).cuda()
    lens = [T, 17, 1] + [int(v) for v in np.random.default_rng(5).integers(0, T + 1, B - 3)]
    outs, ragged = {}, {}
    kernel = "conv_pair_f32_kernel" if prec == "f32" else "conv_pair_bf16x3_kernel"
    for flag in ("1", "0"):
        monkeypatch.setenv("HIFICAR_PAIR", flag)  # read by hificar_create
        g, w = make(params, prec)
        with torch.no_grad():
            outs[flag] = g(c, ar=ar).cpu()
            ragged[flag] = g(c, ar=ar, lengths=lens).cpu()
            g.profile_begin()
            g(c, ar=ar)
            names = {s["name"] for s in g.profile_end()}
        fused = {n for n in names if n.startswith(kernel)}
        assert (len(fused) == 2) == (flag == "1"), names  # both the C = 64 and the C = 32 instantiation
    with torch.no_grad():
        ref = O.generator_forward(w, params, c.cpu(), ar.cpu())
    assert rel_err(outs["1"].numpy(), outs["0"].numpy()) < TOLS[prec]
    assert rel_err(outs["1"].numpy(), ref.numpy()) < TOLS[prec]
    assert rel_err(ragged["1"].numpy(), ragged["0"].numpy()) < TOLS[prec]
    assert torch.equal(ragged["1"][0], outs["1"][0])
    # a small launch (few tall tiles would leave most CUs idle) runs layer by layer even with fusion enabled
    monkeypatch.setenv("HIFICAR_PAIR", "1")
    g, _ = make(params, prec)
    with torch.no_grad():
        g.profile_begin()
        g(c[:1], ar=ar[:1])
        assert not any(s["name"].startswith("conv_pair") for s in g.profile_end())


def test_small_tile_fused_pair_at_mid_size_launches(monkeypatch):
    """Exact fp32, C = 32, a launch too small for 512-row fused tiles but above ~8000 rows (batch 8 x 25 frames = 16000): the 128-row
    fused pair (conv_pair_f32_kernel<1,4,1,2>) runs; HIFICAR_PAIR_SMALL=0 runs the same launches layer by layer.  Both against the
    oracle and each other; batch 1 (2000 rows) stays layer by layer (test above)."""
    params = dict(E2W_PARAMS)
    B, T = 8, 25
    c = torch.from_numpy(synth_features(B, T, 13, seed=321)).permute(0, 2, 1).contiguous().cuda()
    ar = torch.from_numpy(synth_features(B, 512, 1, seed=322)[:, :, 0] * 0.3).reshape(B, 1, 512).cuda()
    lens = [T, 9, 1, 0, 25, 13, 24, 2]
    outs, ragged = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("HIFICAR_PAIR_SMALL", flag)  # read by hificar_create
        g, w = make(params, "f32")
        with torch.no_grad():
            outs[flag] = g(c, ar=ar).cpu()
            ragged[flag] = g(c, ar=ar, lengths=lens).cpu()
            g.profile_begin()
            g(c, ar=ar)
            names = {s["name"] for s in g.profile_end()}
        assert ("conv_pair_f32_kernel<1,4,1,2>" in names) == (flag == "1"), names
    with torch.no_grad():
        ref = O.generator_forward(w, params, c.cpu(), ar.cpu())
    assert rel_err(outs["1"].numpy(), ref.numpy()) < TOLS["f32"]
    assert rel_err(outs["1"].numpy(), outs["0"].numpy()) < TOLS["f32"]
    assert rel_err(ragged["1"].numpy(), ragged["0"].numpy()) < TOLS["f32"]
    for b, n in enumerate(lens):  # every utterance of the ragged batch as if it were alone in it (rounding: other tile shapes)
        if n:
            with torch.no_grad():
                alone = O.generator_forward(w, params, c[b:b + 1, :, :n].cpu(), ar[b:b + 1].cpu())
            assert rel_err(ragged["1"][b, :, :80 * n].numpy(), alone[0].numpy()) < TOLS["f32"], (b, n)


def test_repeated_runs_are_bit_identical(car):
    """Race screen for the LDS ring / out-buffer hand-offs of the persistent kernels: 12 back-to-back syntheses of the
    same batch (different tile timing every time) must give bit-identical waveforms."""
    g, _ = car
    feats = torch.from_numpy(synth_features(24, 100, 13, seed=99)).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        first = g.ar_synthesis(feats, 25).clone()
        for _ in range(11):
            assert torch.equal(g.ar_synthesis(feats, 25), first)


def test_golden_ar_loop_wsola(car):
    """The WSOLA driver variant (decode.py:84-100) through model.forward, against the reference's own chunks."""
    from articulatory_amd.bin.decode import ar_loop
    g, _ = car
    gold = np.load(os.path.join(GOLDEN, "gold_arloop_wsola.npz"))
    config = dict(batch_max_steps=8000, hop_size=80, generator_params=dict(E2W_PARAMS, extra_art=False), dataset_mode="a2w")
    with torch.no_grad():
        outs, ins = ar_loop(g, torch.from_numpy(gold["x"]).cuda(), config, do_wsola=True)
    assert len(outs) == int(gold["n"])
    for i, o in enumerate(outs):
        assert rel_err(o.cpu().numpy(), gold[f"out{i}"]) < 2 * g.tol, i


def test_inference_normalize_before(prec, tmp_path):
    """register_stats + inference(normalize_before=True) (hifigan.py:280-314) on a non-AR generator."""
    params = dict(E2W_PARAMS, in_channels=12, use_ar=False, channels=128, upsample_scales=[4, 2], upsample_kernel_sizes=[8, 4])
    g, w = make(params, prec, seed=7)
    mean = np.linspace(-0.5, 0.5, 12).astype(np.float32)
    scale = np.linspace(0.5, 2.0, 12).astype(np.float32)
    np.save(tmp_path / "stats.npy", np.stack([mean, scale]))
    g.register_stats(str(tmp_path / "stats.npy"))
    g = g.to("cuda:0")
    x = synth_features(1, 90, 12, seed=8)[0]
    with torch.no_grad():
        y = g.inference(x, normalize_before=True).cpu()
        y_ref = O.inference(w, params, x, torch.from_numpy(mean), torch.from_numpy(scale))
    assert y.shape == y_ref.shape == (720, 1)
    assert rel_err(y.numpy(), y_ref.numpy()) < TOLS[prec]


def test_pcm16_on_device_matches_host_writer(tmp_path):
    from articulatory_amd.bin.predict_wav import write_wav
    from articulatory_amd.utils import pcm16
    import wave
    y = torch.from_numpy(np.concatenate([np.linspace(-1.2, 1.2, 4001), [0.5 / 32767, 1.5 / 32767, -2.5 / 32767]]).astype(np.float32))
    got = pcm16(y.cuda()).cpu().numpy()
    # libsndfile's arithmetic for float32 input: lrintf(src * 32767.f) (float32 product), clipped
    want = np.clip(np.rint(y.numpy() * np.float32(32767.0)), -32768, 32767).astype(np.int16)
    rng = np.random.default_rng(0)
    y = torch.cat([y, torch.from_numpy(rng.uniform(-1, 1, 200000).astype(np.float32))])
    got = pcm16(y.cuda()).cpu().numpy()
    want = np.clip(np.rint(y.numpy() * np.float32(32767.0)), -32768, 32767).astype(np.int16)
    assert got.dtype == np.int16
    assert np.array_equal(got, want)  # integer output: bit-exact with the host writer
    write_wav(str(tmp_path / "a.wav"), got, 16000)
    with wave.open(str(tmp_path / "a.wav")) as f:
        assert f.getnframes() == y.numel() and f.getsampwidth() == 2
        assert np.array_equal(np.frombuffer(f.readframes(y.numel()), dtype="<i2"), got)


def test_ragged_batch_equals_one_at_a_time(car):
    """hificar_ar_loop_ragged: a padded batch of utterances of different lengths gives, per utterance, the waveform of
    that utterance synthesised alone (same_across_shapes: bit for bit in the dense conv form, zero padding at its own end),
    which in turn matches the oracle's batch-1 ``ar_loop`` (decode.py:54-83) incl. each utterance's own short tail chunk."""
    g, w = car
    lens = [260, 25, 131, 7, 200, 0, 99]
    Tm = max(lens)
    x = synth_features(len(lens), Tm, 13, seed=4242)
    for b, n in enumerate(lens):
        x[b, n:] = 7.5  # padding frames must not matter, whatever they hold
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        y = g.ar_synthesis(feats, 25, lengths=lens)
    assert y.shape == (len(lens), 80 * Tm)
    for b, n in enumerate(lens):
        assert float(y[b, 80 * n:].abs().max()) == 0.0 if n < Tm else True
        if n == 0:
            continue
        with torch.no_grad():
            alone = g.ar_synthesis(feats[b:b + 1, :, :n].contiguous(), 25)
            ref = O.ar_loop(w, E2W_PARAMS, torch.from_numpy(x[b, :n]), 2000, 80)
        assert same_across_shapes(y[b, :80 * n], alone[0], XSHAPE_TOL[g.precision]), (b, n)
        assert rel_err(y[b, :80 * n].cpu().numpy(), ref.numpy()) < g.tol, (b, n)
    # one chunk through hificar_forward_ragged with an AR context per utterance
    ar = torch.from_numpy(synth_features(len(lens), 512, 1, seed=8)[:, :, 0] * 0.3).reshape(len(lens), 1, 512).cuda()
    flens = [min(n, 25) for n in lens]
    with torch.no_grad():
        yf = g(feats[:, :, :25].contiguous(), ar=ar, lengths=flens)
        for b, n in enumerate(flens):
            if n:
                alone = g(feats[b:b + 1, :, :n].contiguous(), ar=ar[b:b + 1])
                assert same_across_shapes(yf[b, :, :80 * n], alone[0], XSHAPE_TOL[g.precision]), (b, n)
            assert float(yf[b, :, 80 * n:].abs().sum()) == 0.0


def test_batch_invariance_is_bitwise_without_split_k(monkeypatch, prec):
    """HIFICAR_KSPLIT=0 (dense conv form for every launch shape): an utterance's waveform does not depend on what else is in the
    batch — ragged batch, continuous batching and alone are bit-identical; with the default (split-K on small launches) they
    agree to fp32 rounding (same_across_shapes)."""
    monkeypatch.setenv("HIFICAR_KSPLIT", "0")
    g, _ = make(dict(E2W_PARAMS), prec)
    lens = [130, 25, 77, 7, 100, 0]
    Tm = max(lens)
    x = synth_features(len(lens), Tm, 13, seed=4243)
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        y = g.ar_synthesis(feats, 25, lengths=lens)
        yp = g.ar_synthesis_packed(feats, 25, lens, batch=2)
        for b, n in enumerate(lens):
            if n:
                alone = g.ar_synthesis(feats[b:b + 1, :, :n].contiguous(), 25)
                assert torch.equal(y[b, :80 * n], alone[0]) and torch.equal(yp[b, :80 * n], alone[0]), (b, n)
    monkeypatch.delenv("HIFICAR_KSPLIT")
    g2, _ = make(dict(E2W_PARAMS), prec)
    with torch.no_grad():
        y2 = g2.ar_synthesis(feats, 25, lengths=lens)
        g2.profile_begin()
        g2.ar_synthesis(feats[:1, :, :50].contiguous(), 25)
        names = {s["name"].split("<")[0] for s in g2.profile_end()}
    assert any(n.startswith("conv_sk_") for n in names), names  # the small launch really took the split-K form
    assert same_across_shapes(y2, y, XSHAPE_TOL[prec])


def test_ar_loop_on_two_streams_equals_one_stream(monkeypatch, prec):
    """hificar_ar_loop splits mid-size batches into two halves on two streams (csrc/hificar.hip, hificar_ar_loop_ragged): every utterance's
    waveform equals the single-stream loop's — bit for bit in the batch-invariant mode, to fp32 rounding otherwise (the halves' launches may
    pick other tile forms) — for an odd batch, a tail chunk and repeated calls (the side stream re-uses the handle's schedules)."""
    x = synth_features(5, 60, 13, seed=77)
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    monkeypatch.setenv("HIFICAR_AR_DUAL_MAX", "0")
    g1, _ = make(dict(E2W_PARAMS), prec)
    monkeypatch.setenv("HIFICAR_AR_DUAL_MIN", "2")
    monkeypatch.setenv("HIFICAR_AR_DUAL_MAX", "64")
    g2, _ = make(dict(E2W_PARAMS), prec)
    with torch.no_grad():
        y1 = g1.ar_synthesis(feats, 25)
        y2 = g2.ar_synthesis(feats, 25)
        y2b = g2.ar_synthesis(feats, 25)
        y2c = g2.ar_synthesis(feats[:2].contiguous(), 25)
    assert torch.equal(y2, y2b)
    assert same_across_shapes(y2, y1, XSHAPE_TOL[prec])
    assert same_across_shapes(y2c, y1[:2], XSHAPE_TOL[prec])
    monkeypatch.setenv("HIFICAR_KSPLIT", "0")
    g3, _ = make(dict(E2W_PARAMS), prec)
    monkeypatch.setenv("HIFICAR_AR_DUAL_MAX", "0")
    g4, _ = make(dict(E2W_PARAMS), prec)
    with torch.no_grad():
        assert torch.equal(g3.ar_synthesis(feats, 25), g4.ar_synthesis(feats, 25))


@pytest.mark.parametrize("kernels", [[3, 7, 11], [3, 7], [3, 5, 7, 11]])
def test_mrf_mean_folded_into_the_upsampler(kernels):
    """Exact fp32 inference stages the MRF mean (hifigan.py:226-230: cs += block_j(c) in order; c = cs / n) inside the next upsampler's loader waves
    — no mrf_split_kernel launch, no buffer for the mean.  The forward under autograd still runs mrf_split_kernel (its tape keeps the activated
    mean) with the same sums, the same division and the same LeakyReLU; the two forwards differ only where the taped one differs anyway (weight norm
    folded on the device, narrow stages layer by layer instead of fused pairs): fp32 rounding.  3 blocks per stage (the shipped recipe), 2 and 4;
    a ragged batch; a batch wide enough for the dense multi-tile launches; and the oracle on the folded path."""
    params = dict(E2W_PARAMS, resblock_kernel_sizes=kernels, resblock_dilations=[[1, 3, 5]] * len(kernels))
    g, _ = make(params, "f32", remove_wn=False)
    x = torch.from_numpy(synth_features(24, 50, 13, seed=783)).permute(0, 2, 1).contiguous().cuda()
    ar = torch.from_numpy(np.random.default_rng(783).uniform(-0.5, 0.5, (24, 1, 512)).astype(np.float32)).cuda()
    with torch.no_grad():
        g.profile_begin()
        y_inf = g(x, ar=ar)
        names = {s["name"].split("<")[0] for s in g.profile_end()}
    assert "mrf_split_kernel" not in names and "conv_f32do_kernel" in names
    g.train()
    g.profile_begin()
    y_tape = g(x, ar=ar)  # (gradients enabled: the taped forward)
    names_t = {s["name"].split("<")[0] for s in g.profile_end()}
    g.eval()
    assert "mrf_split_kernel" in names_t
    assert rel_err(y_inf.cpu().numpy(), y_tape.detach().cpu().numpy()) < 5e-6
    with torch.no_grad():
        ref = O.generator_forward(O.fold_weight_norm({k: v.detach().cpu().numpy() for k, v in g.state_dict().items()}), params, x[:4].cpu(), ar[:4].cpu())
    assert rel_err(y_inf[:4].cpu().numpy(), ref.numpy()) < 2e-5
    lens = torch.tensor([50 - 2 * (i % 13) for i in range(24)])
    with torch.no_grad():
        y_r = g(x, ar=ar, lengths=lens)
        for i in (0, 5, 12):  # a ragged batch's utterance = that utterance alone at its own length
            n = int(lens[i])
            y_1 = g(x[i:i + 1, :, :n].contiguous(), ar=ar[i:i + 1])
            assert rel_err(y_r[i:i + 1, :, :n * 80].cpu().numpy(), y_1.cpu().numpy()) < 5e-6


def test_two_handles_on_one_device_run_concurrently_on_two_streams(prec):
    """One handle per module: two generators on one device own separate native state (schedule arenas, tile-pick caches, workspaces), so their
    calls may be in flight at the same time on different streams.  Interleaved AR syntheses and forwards of two models (different widths, so every
    launch shape and schedule differs) on two streams equal each model's own serial results bit for bit, over repeated rounds."""
    pa = dict(E2W_PARAMS)
    pb = dict(E2W_PARAMS, channels=256)
    ga, _ = make(pa, prec, seed=11)
    gb, _ = make(pb, prec, seed=12)
    xa = torch.from_numpy(synth_features(9, 60, 13, seed=91)).permute(0, 2, 1).contiguous().cuda()
    xb = torch.from_numpy(synth_features(5, 85, 13, seed=92)).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        ya0, yb0 = ga.ar_synthesis(xa, 25), gb.ar_synthesis(xb, 25)
        torch.cuda.synchronize()
        assert ga._native_handle().value != gb._native_handle().value
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(3):
            outs = []
            for _ in range(2):  # enqueue alternately: both handles' launches are in flight together
                with torch.cuda.stream(sa):
                    outs.append(("a", ga.ar_synthesis(xa, 25)))
                with torch.cuda.stream(sb):
                    outs.append(("b", gb.ar_synthesis(xb, 25)))
            torch.cuda.synchronize()
            for tag, y in outs:
                assert torch.equal(y, ya0 if tag == "a" else yb0)


def test_ragged_forward_non_ar(prec):
    """hificar_forward_ragged on the non-AR generator: per utterance identical to a forward of that utterance alone."""
    params = dict(E2W_PARAMS, in_channels=12, use_ar=False)
    g, w = make(params, prec)
    lens = [300, 41, 128, 1]
    Tm = max(lens)
    x = synth_features(len(lens), Tm, 12, seed=777)
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        y = g(feats, lengths=torch.tensor(lens))
        for b, n in enumerate(lens):
            alone = g(feats[b:b + 1, :, :n].contiguous())
            assert same_across_shapes(y[b, :, :80 * n], alone[0], XSHAPE_TOL[prec]), (b, n)
            assert float(y[b, :, 80 * n:].abs().sum()) == 0.0
        ref = O.generator_forward(w, params, torch.from_numpy(x[1:2, :41]).permute(0, 2, 1))
    assert rel_err(y[1, :, :80 * 41].cpu().numpy(), ref[0].numpy()) < TOLS[prec]
    with pytest.raises(RuntimeError):
        g(feats, lengths=[1, 2, 3])
    with pytest.raises(RuntimeError):
        g(feats, lengths=[1, 2, 3, Tm + 1])


def test_decode_dataset_ragged_batches_on_device(car, tmp_path):
    """articulatory-decode counterpart over a small dataset of mixed lengths: --batch-size 4 writes the very waveforms
    that one-utterance-at-a-time decoding writes."""
    from articulatory_amd.bin import decode as D
    g, _ = car
    config = dict(generator_params=dict(E2W_PARAMS, extra_art=False), sampling_rate=16000, hop_size=80, batch_max_steps=2000,
                  dataset_mode="a2w")
    items = [(f"u{i}", synth_features(1, T, 13, seed=60 + i)[0]) for i, T in enumerate([310, 64, 255, 129, 26])]
    one, four = {}, {}
    n1, _ = D.decode_dataset(g, iter(items), config, "cuda:0", str(tmp_path), writer=lambda p, y, sr: one.__setitem__(os.path.basename(p), y))
    n4, rtf = D.decode_dataset(g, iter(items), config, "cuda:0", str(tmp_path), batch_size=4,
                               writer=lambda p, y, sr: four.__setitem__(os.path.basename(p), y))
    assert n1 == n4 == 5 and rtf > 0 and sorted(one) == sorted(four)
    for k in one:
        assert one[k].shape == four[k].shape and same_across_shapes(one[k], four[k], XSHAPE_TOL[g.precision]), k


def test_packed_ar_loop_equals_one_at_a_time(car):
    """hificar_ar_loop_packed (continuous batching: 3 utterances in flight out of 9, a finished one replaced by the next):
    every utterance equals synthesising it alone (same_across_shapes); zero-length and single-chunk utterances included."""
    g, w = car
    lens = [260, 131, 130, 99, 64, 26, 25, 7, 0]
    Tm = max(lens)
    x = synth_features(len(lens), Tm, 13, seed=515)
    for b, n in enumerate(lens):
        x[b, n:] = -3.0  # padding frames must not matter
    feats = torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        y = g.ar_synthesis_packed(feats, 25, lens, batch=3)
        y_all = g.ar_synthesis_packed(feats, 25, lens, batch=64)
    assert y.shape == (len(lens), 80 * Tm) and same_across_shapes(y, y_all, XSHAPE_TOL[g.precision])
    for b, n in enumerate(lens):
        assert float(y[b, 80 * n:].abs().sum()) == 0.0
        if n:
            with torch.no_grad():
                alone = g.ar_synthesis(feats[b:b + 1, :, :n].contiguous(), 25)
            assert same_across_shapes(y[b, :80 * n], alone[0], XSHAPE_TOL[g.precision]), (b, n)
    ref = O.ar_loop(w, E2W_PARAMS, torch.from_numpy(x[1, :131]), 2000, 80)
    assert rel_err(y[1, :80 * 131].cpu().numpy(), ref.numpy()) < g.tol
    with pytest.raises(RuntimeError):
        g.ar_synthesis_packed(feats, 25, lens[:-1])


def test_cli_mains_end_to_end_on_device(tmp_path, monkeypatch):
    """The two console-script counterparts, argv to wav files, on the device: reference-layout checkpoint + config.yml +
    scp of .npy features; --batch-size 4 writes the same samples as the default one-at-a-time run; under a torchrun
    environment (WORLD_SIZE=2) the two ranks' shares are disjoint and together complete."""
    import wave
    import yaml
    from articulatory_amd.bin import decode as D, predict_wav as PW
    params = dict(E2W_PARAMS)
    sd = synth_state_dict(params, seed=1234)
    ckpt = tmp_path / "checkpoint-1steps.pkl"
    torch.save({"model": {"generator": {k: torch.from_numpy(v) for k, v in sd.items()}}}, ckpt)
    with open(tmp_path / "config.yml", "w") as f:
        yaml.dump(dict(generator_type="HiFiGANGenerator", generator_params=params, format="npy", sampling_rate=16000,
                       hop_size=80, batch_max_steps=2000, dataset_mode="a2w"), f)
    lens = [300, 260, 411, 275, 333]
    with open(tmp_path / "feats.scp", "w") as f:
        for i, T in enumerate(lens):
            np.save(tmp_path / f"u{i}.npy", synth_features(1, T, 13, seed=90 + i)[0].astype(np.float64))
            f.write(f"u{i} {tmp_path / f'u{i}.npy'}\n")

    def read(path):
        with wave.open(str(path)) as f:
            return np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")

    def run(mod, outdir, *extra):
        mod.main(["--feats-scp", str(tmp_path / "feats.scp"), "--outdir", str(outdir), "--checkpoint", str(ckpt), "--verbose", "0", *extra])

    run(D, tmp_path / "d1")
    run(D, tmp_path / "d4", "--batch-size", "4")
    run(PW, tmp_path / "p4", "--batch-size", "4")
    for i, T in enumerate(lens):
        a = read(tmp_path / "d1" / f"u{i}_gen.wav")
        assert len(a) == 80 * T
        # batch 1 vs batch 4 launches may differ in the last fp32 bit (split-K form on small launches): at most one PCM step
        for other in (read(tmp_path / "d4" / f"u{i}_gen.wav"), read(tmp_path / "p4" / f"u{i}.wav")):
            assert len(other) == len(a) and int(np.abs(a.astype(np.int32) - other.astype(np.int32)).max()) <= 1
    monkeypatch.setenv("WORLD_SIZE", "2")
    for r in (0, 1):
        monkeypatch.setenv("RANK", str(r))
        run(D, tmp_path / f"r{r}", "--batch-size", "4")
    got = [sorted(os.listdir(tmp_path / f"r{r}")) for r in (0, 1)]
    assert not set(got[0]) & set(got[1]) and sorted(got[0] + got[1]) == sorted(os.listdir(tmp_path / "d1"))
