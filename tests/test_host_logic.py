"""Host-side mirror of the reference's plugin surface: registry lookup, constructor keywords, state_dict
layout, weight-norm fold, load_model, predict_wav plumbing, and the no-CPU-fallback rule.  CPU only."""

import os

import numpy as np
import pytest
import torch
import yaml

from conftest import E2W_PARAMS, GOLDEN, rel_err
import articulatory_amd
from articulatory_amd.bin import decode as D
from articulatory_amd.bin import predict_wav as PW
from articulatory_amd.utils import load_model
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from oracle import hificar_oracle as O


def _ref_keys():
    return [(l.split()[0], tuple(int(s) for s in l.split()[1:]))
            for l in open(os.path.join(GOLDEN, "gold_state_dict_keys.txt")).read().strip().splitlines()]


def test_registry_lookup_and_yaml_kwargs():
    cls = getattr(articulatory_amd.models, "HiFiGANGenerator")
    # e2w_hifigan_car.yaml carries final_scale / extra_art, which the reference class rejects
    g = cls(**dict(E2W_PARAMS, final_scale=80, extra_art=False))
    assert [(k, tuple(v.shape)) for k, v in g.state_dict().items()] == _ref_keys()
    assert sum(p.numel() for p in g.parameters() if p.requires_grad) == 13467778
    with pytest.raises(TypeError):
        cls(**dict(E2W_PARAMS, not_a_kwarg=1))
    with pytest.raises(AssertionError, match="odd"):
        cls(**dict(E2W_PARAMS, kernel_size=6))


def test_load_state_dict_and_fold_match_reference():
    params = dict(E2W_PARAMS, channels=64)
    sd = synth_state_dict(params, seed=1234)
    g = articulatory_amd.models.HiFiGANGenerator(**params)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    gold = np.load(os.path.join(GOLDEN, "gold_wnfold.npz"))
    after = g.state_dict()
    for name in ["upsamples.0.1", "blocks.4.convs2.1.1", "output_conv.1"]:
        assert name + ".weight_g" not in after
        assert rel_err(after[name + ".weight"].numpy(), gold[name + ".weight"]) < 1e-6
    # key order after remove_weight_norm is (bias, weight) per layer, as torch leaves it
    assert list(after.keys())[:2] == ["input_conv.bias", "input_conv.weight"]
    # folded_state() of a still-normalised model equals the baked weights
    g2 = articulatory_amd.models.HiFiGANGenerator(**params)
    g2.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    f2 = g2.folded_state()
    for k, v in g.folded_state().items():
        assert torch.equal(v, f2[k]), k
    # apply -> remove round trip leaves weights unchanged
    g.apply_weight_norm()
    assert "input_conv.weight_g" in g.state_dict()
    g.remove_weight_norm()
    assert rel_err(g.state_dict()["upsamples.0.1.weight"].numpy(), gold["upsamples.0.1.weight"]) < 1e-6


def test_strict_load_rejects_wrong_checkpoint():
    g = articulatory_amd.models.HiFiGANGenerator(**E2W_PARAMS)
    sd = synth_state_dict(dict(E2W_PARAMS, in_channels=140), seed=1)
    with pytest.raises(RuntimeError, match="size mismatch"):
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})


def test_no_cpu_fallback():
    g = articulatory_amd.models.HiFiGANGenerator(**E2W_PARAMS).eval()
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            g(torch.zeros(1, 13, 25), ar=torch.zeros(1, 1, 512))
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            g.ar_synthesis(torch.zeros(1, 13, 50), 25)
    g.train()  # under autograd the forward is a native autograd node: still no CPU path
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 13, 25), ar=torch.zeros(1, 1, 512))
    # conditioned variants construct with the reference's parameter names (hifigan.py:176-189); the ill-formed combination is refused
    g = articulatory_amd.models.HiFiGANGenerator(**dict(E2W_PARAMS, use_spk_id=True, num_spk=4))
    assert tuple(g.state_dict()["spk_fc.weight"].shape) == (141, 32) and tuple(g.state_dict()["spk_emb_mat.weight"].shape) == (4, 32)
    g = articulatory_amd.models.HiFiGANGenerator(**dict(E2W_PARAMS, in_channels=149, use_ph=True, num_ph=9, use_ph_loss=True))
    assert tuple(g.state_dict()["ph_emb_mat.weight"].shape) == (9, 8) and tuple(g.state_dict()["ph_fc.weight"].shape) == (9, 32)
    with pytest.raises(ValueError, match="ill-formed"):
        articulatory_amd.models.HiFiGANGenerator(**dict(E2W_PARAMS, use_spk_id=True, num_spk=4, use_ph=True, num_ph=9))


def test_product_code_never_imports_the_oracle():
    import re
    root = os.path.join(os.path.dirname(GOLDEN), "..", "articulatory_amd")
    for dp, _, fns in os.walk(root):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def _write_checkpoint(tmp_path, params, cfg_extra=None):
    sd = synth_state_dict(params, seed=1234)
    ckpt = tmp_path / "checkpoint-1steps.pkl"
    # the reference trainer's layout (train.py:147-176)
    torch.save({"model": {"generator": {k: torch.from_numpy(v) for k, v in sd.items()}, "discriminator": {}},
                "optimizer": {}, "scheduler": {}, "steps": 1, "epochs": 0}, ckpt)
    config = dict(generator_type="HiFiGANGenerator", generator_params=dict(params), format="npy",
                  sampling_rate=16000, hop_size=80, batch_max_steps=8000, dataset_mode="a2w")
    config.update(cfg_extra or {})
    with open(tmp_path / "config.yml", "w") as f:
        yaml.dump(config, f)
    return str(ckpt), config, sd


def test_load_model_reads_reference_checkpoint_layout(tmp_path):
    params = dict(E2W_PARAMS, channels=64)
    ckpt, config, sd = _write_checkpoint(tmp_path, params)
    np.save(tmp_path / "stats.npy", np.stack([np.arange(13.0), np.arange(13.0) + 1]))
    m = load_model(ckpt)  # config.yml found beside the checkpoint, stats.npy registered
    assert isinstance(m, articulatory_amd.models.HiFiGANGenerator)
    assert torch.equal(m.state_dict()["input_conv.weight_v"], torch.from_numpy(sd["input_conv.weight_v"]))
    assert hasattr(m, "mean") and m.scale.shape == (13,)
    with pytest.raises(AttributeError, match="ParallelWaveGANGenerator"):
        load_model(ckpt, {k: v for k, v in config.items() if k != "generator_type"})
    # typo workaround (utils.py:330-333)
    gp = dict(config["generator_params"])
    gp["upsample_kernal_sizes"] = gp.pop("upsample_kernel_sizes")
    assert load_model(ckpt, dict(config, generator_params=gp)) is not None


class _OracleBackedModel:
    """Stands in for the HIP generator on the CPU box so that the *plumbing* (scp parsing, chunk length,
    batching by length, skipping short utterances, wav writing) can be tested here.  Test-only."""

    def __init__(self, params, sd):
        self.params, self.w = params, O.fold_weight_norm(sd)
        self.calls = []

    def ar_synthesis(self, c, chunk_frames, lengths=None):
        self.calls.append((tuple(c.shape), chunk_frames) if lengths is None else (tuple(c.shape), chunk_frames, list(lengths)))
        if lengths is None:
            return O.ar_loop_batched(self.w, self.params, c.permute(0, 2, 1), chunk_frames * 80, 80)
        out = torch.zeros(c.shape[0], 80 * c.shape[2])  # ragged: each utterance alone, as the HIP path guarantees
        for b, n in enumerate(lengths):
            out[b, :80 * n] = O.ar_loop(self.w, self.params, c[b, :, :n].permute(1, 0), chunk_frames * 80, 80)
        return out

    def ar_synthesis_packed(self, c, chunk_frames, lengths, batch=64):
        self.calls.append((tuple(c.shape), chunk_frames, list(lengths), batch))
        out = torch.zeros(c.shape[0], 80 * c.shape[2])  # each utterance alone, as the HIP path guarantees
        for b, n in enumerate(lengths):
            out[b, :80 * n] = O.ar_loop(self.w, self.params, c[b, :, :n].permute(1, 0), chunk_frames * 80, 80)
        return out

    def __call__(self, c, ar=None):
        return O.generator_forward(self.w, self.params, c, ar)


def test_predict_wav_plumbing_matches_reference_pin(tmp_path):
    """configs[0]: one (700, 13) utterance through the predict_wav counterpart; the waveform handed to the
    wav writer must equal what the real predict_wav.py hands to sf.write (gold_predict_wav.npz)."""
    import wave
    gold = np.load(os.path.join(GOLDEN, "gold_predict_wav.npz"))
    sd = synth_state_dict(E2W_PARAMS, seed=1234)
    model = _OracleBackedModel(E2W_PARAMS, sd)
    feat = tmp_path / "utt1.npy"
    np.save(feat, gold["x"].astype(np.float64))  # .npy features are float64 on disk (mk_ema_feats.py:67-72)
    short = tmp_path / "short.npy"
    np.save(short, synth_features(1, 250, 13, seed=5)[0].astype(np.float64))
    scp = tmp_path / "feats.scp"
    scp.write_text(f"utt1 {feat}\nshort {short}\n")
    fids, featps = PW.read_scp(str(scp))
    assert fids == ["utt1", "short"]
    config = dict(generator_params=dict(E2W_PARAMS), sampling_rate=16000, hop_size=80, batch_max_steps=8000,
                  dataset_mode="a2w")
    captured = {}

    def writer(path, y, sr):
        captured[os.path.basename(path)] = np.asarray(y)
        PW.write_wav(path, y, sr)

    written = PW.synthesize_file_list(model, fids, featps, config, "cpu", str(tmp_path), writer=writer)
    assert written == ["utt1"]  # T <= 250 is skipped, as in the reference
    assert model.calls == [((1, 13, 700), 100)]
    assert rel_err(captured["utt1.wav"], gold["out"]) < 2e-5
    with wave.open(str(tmp_path / "utt1.wav")) as f:
        assert (f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()) == (16000, 1, 2, 56000)


def test_predict_wav_ragged_batches(tmp_path):
    params = dict(E2W_PARAMS)
    model = _OracleBackedModel(params, synth_state_dict(params, seed=1234))
    paths = []
    for i, T in enumerate([260, 300, 260]):
        p = tmp_path / f"u{i}.npy"
        np.save(p, synth_features(1, T, 13, seed=10 + i)[0])
        paths.append(str(p))
    config = dict(generator_params=params, sampling_rate=16000, hop_size=80, batch_max_steps=8000)
    outs = {}
    PW.synthesize_file_list(model, ["u0", "u1", "u2"], paths, config, "cpu", str(tmp_path), batch_size=2,
                            writer=lambda p, y, sr: outs.__setitem__(os.path.basename(p), y))
    assert model.calls == [((3, 13, 300), 100, [300, 260, 260], 2)]  # one call: longest first, 2 utterances in flight
    outs3 = {}
    model3 = _OracleBackedModel(params, synth_state_dict(params, seed=1234))
    PW.synthesize_file_list(model3, ["u0", "u1", "u2"], paths, config, "cpu", str(tmp_path), batch_size=3,
                            writer=lambda p, y, sr: outs3.__setitem__(os.path.basename(p), y))
    assert model3.calls == [((3, 13, 300), 100, [300, 260, 260], 3)]
    assert {k: len(v) for k, v in outs3.items()} == {"u0.wav": 20800, "u1.wav": 24000, "u2.wav": 20800}
    for k in outs:
        assert rel_err(outs3[k], outs[k]) < 1e-6
    single = _OracleBackedModel(params, synth_state_dict(params, seed=1234))
    y0 = D.ar_loop(single, torch.from_numpy(np.load(paths[0])).float(), config)
    assert rel_err(outs["u0.wav"], y0.numpy()) < 2e-5


def test_ar_loop_rejects_unbuilt_variants():
    with pytest.raises(NotImplementedError):
        D.ar_loop(None, torch.zeros(10, 13), dict(batch_max_steps=2000, hop_size=80, generator_params=E2W_PARAMS), modality=0)
    with pytest.raises(NotImplementedError):
        D.ar_loop(None, torch.zeros(10, 13), dict(batch_max_steps=2000, hop_size=80, generator_params=E2W_PARAMS,
                                                   dataset_mode="w2a"))


def test_decode_cli_plumbing(tmp_path):
    """articulatory-decode counterpart: scp and dump-dir inputs, <utt>_gen.wav naming, RTF; WSOLA chunk files."""
    import wave
    gold = np.load(os.path.join(GOLDEN, "gold_arloop.npz"))
    sd = synth_state_dict(E2W_PARAMS, seed=1234)
    model = _OracleBackedModel(E2W_PARAMS, sd)
    dump = tmp_path / "dump"
    dump.mkdir()
    np.save(dump / "uttA-feats.npy", gold["x"])
    scp = tmp_path / "feats.scp"
    scp.write_text(f"uttA {dump / 'uttA-feats.npy'}\n")
    assert [u for u, _ in D.iter_features(dumpdir=str(dump))] == ["uttA"]
    assert [u for u, _ in D.iter_features(feats_scp=str(scp))] == ["uttA"]
    with pytest.raises(ValueError, match="either"):
        list(D.iter_features())
    config = dict(generator_params=dict(E2W_PARAMS, extra_art=False), sampling_rate=16000, hop_size=80,
                  batch_max_steps=2000, dataset_mode="a2w")
    out = tmp_path / "out"
    out.mkdir()
    got = {}
    n, rtf = D.decode_dataset(model, D.iter_features(feats_scp=str(scp)), config, "cpu", str(out),
                              writer=lambda p, y, sr: (got.__setitem__(os.path.basename(p), y), PW.write_wav(p, y, sr)))
    assert n == 1 and rtf > 0
    assert rel_err(got["uttA_gen.wav"], gold["out_bms2000"]) < 2e-5
    with wave.open(str(out / "uttA_gen.wav")) as f:
        assert f.getnframes() == 20800 and f.getsampwidth() == 2
    # ragged batches over a dataset: same waveforms as one at a time
    extra = synth_features(1, 77, 13, seed=3)[0]
    np.save(dump / "uttB-feats.npy", extra)
    got_b = {}
    model_b = _OracleBackedModel(E2W_PARAMS, sd)
    n, rtf = D.decode_dataset(model_b, D.iter_features(dumpdir=str(dump)), config, "cpu", str(out), batch_size=4,
                              writer=lambda p, y, sr: got_b.__setitem__(os.path.basename(p), y))
    assert n == 2 and rtf > 0 and model_b.calls == [((2, 13, 260), 25, [260, 77], 4)]
    assert rel_err(got_b["uttA_gen.wav"], gold["out_bms2000"]) < 2e-5 and len(got_b["uttB_gen.wav"]) == 77 * 80
    (dump / "uttB-feats.npy").unlink()
    # WSOLA variant: half-overlapping 100-frame chunks, one wav + one input .npy per chunk
    gw = np.load(os.path.join(GOLDEN, "gold_arloop_wsola.npz"))
    config_w = dict(config, batch_max_steps=8000, wsola=True)
    got.clear()
    D.decode_dataset(model, D.iter_features(feats_scp=str(scp)), config_w, "cpu", str(out),
                     writer=lambda p, y, sr: got.__setitem__(os.path.basename(p), y))
    assert len(got) == int(gw["n"])
    for i in range(int(gw["n"])):
        assert rel_err(got[f"uttA_{i}_gen.wav"], gw[f"out{i}"]) < 2e-5
        assert np.load(out / f"uttA_{i}.npy").shape[0] == int(gw[f"in_len{i}"])
    with pytest.raises(AssertionError):  # odd chunk length: the reference asserts too (decode.py:87)
        D.ar_loop(model, torch.from_numpy(gold["x"]), dict(config, wsola=True), do_wsola=True)


def test_decode_lists_then_loads_lazily(tmp_path):
    """articulatory-decode counterpart: the utterance list is (utt_id, path) pairs, lengths come from the .npy headers, and
    feature files are read one at a time while decoding (a rank never loads utterances of another rank's share)."""
    from articulatory_amd.bin import decode as D
    from articulatory_amd.bin.shard import shard_items
    lens = [30, 300, 31, 290, 32, 280]
    with open(tmp_path / "feats.scp", "w") as f:
        for i, T in enumerate(lens):
            np.save(tmp_path / f"u{i}.npy", np.zeros((T, 13)))
            f.write(f"u{i} {tmp_path / f'u{i}.npy'}\n")
    pairs = D.list_features(str(tmp_path / "feats.scp"))
    assert [u for u, _ in pairs] == [f"u{i}" for i in range(6)]
    assert [D.npy_frames(p) for _, p in pairs] == lens
    shares = [shard_items(pairs, 2, r, length_of=lambda kv: D.npy_frames(kv[1])) for r in range(2)]
    assert sorted(shares[0] + shares[1]) == sorted(pairs) and not set(shares[0]) & set(shares[1])
    loaded = []
    real_load = np.load

    def spy(path, *a, **kw):
        if not kw.get("mmap_mode"):
            loaded.append(os.path.basename(str(path)))
        return real_load(path, *a, **kw)

    np.load = spy
    try:
        it = D.load_features(shares[0])
        assert loaded == []  # nothing read before the loop asks for it
        first = next(it)
        assert loaded == [os.path.basename(shares[0][0][1])] and first[1].shape[1] == 13
    finally:
        np.load = real_load
    with pytest.raises(ValueError):
        D.list_features(None, None)


def test_optimizer_steps_invalidate_the_native_hand_over():
    """torch's fused optimizers update in place without bumping Parameter._version, the modules' only own signal: every module registers with the
    process-wide optimizer post-step hook (articulatory_amd/utils/optim_hook.py), so ANY optimizer.step() over its parameters invalidates the
    hand-over — with or without the Trainer."""
    from articulatory_amd.models import GBlockGenerator, HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator
    from articulatory_amd.utils.synth import disc_params

    g = HiFiGANGenerator(**dict(E2W_PARAMS, channels=32))
    b = GBlockGenerator(in_channels=13, channels=32, g_scales=[5, 1, 4, 1, 1, 2, 1, 2, 1, 1], g_kernel_sizes=[3] * 10)
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(**disc_params())
    calls = {"g": 0, "b": 0, "d": 0}
    for key, m in (("g", g), ("b", b), ("d", d)):
        orig = m.invalidate_parameters
        m.invalidate_parameters = (lambda orig=orig, key=key: (calls.__setitem__(key, calls[key] + 1), orig())[1])
    for p in list(g.parameters()) + list(d.parameters()):
        p.grad = torch.zeros_like(p)
    og = torch.optim.Adam(g.parameters(), lr=1e-3)
    od = torch.optim.SGD(d.parameters(), lr=1e-3)
    og.step()
    assert calls == {"g": 1, "b": 0, "d": 0}
    od.step()
    od.step()
    assert calls == {"g": 1, "b": 0, "d": 2}
    other = torch.nn.Linear(2, 2)
    other.weight.grad = torch.zeros_like(other.weight)
    torch.optim.SGD([other.weight], lr=0.1).step()  # an unrelated optimizer touches nobody
    assert calls == {"g": 1, "b": 0, "d": 2}


def test_reassigned_parameter_objects_refresh_the_cached_lists():
    from articulatory_amd.models import HiFiGANGenerator

    g = HiFiGANGenerator(**dict(E2W_PARAMS, channels=32))
    before = g._plist()
    names0, tensors0 = g._raw_parameters()
    new = torch.nn.Parameter(torch.zeros_like(g.input_conv.weight_v))
    g.input_conv.weight_v = new  # direct surgery on a conv holder: no load_state_dict / .to() / weight-norm call in between
    after = g._plist()
    assert any(p is new for p in after) and not any(p is new for p in before)
    names1, tensors1 = g._raw_parameters()
    assert names1 == names0 and any(t is new for t in tensors1)


def test_copies_and_pickles_of_native_modules_register_themselves_and_share_no_native_state():
    """copy.deepcopy / pickle skip __init__ (round-4 advisor finding): the copy carries no native handle, no cached parameter list of the original,
    and is watched by the optimizer post-step hook — a fused optimizer's step over ITS parameters invalidates ITS hand-over."""
    import copy
    import pickle

    from articulatory_amd.models import GBlockGenerator, HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator
    from articulatory_amd.utils import optim_hook

    g = HiFiGANGenerator(in_channels=13, channels=64, use_ar=False)
    gb = GBlockGenerator(in_channels=13, channels=64, g_scales=[5, 1, 4, 1, 1, 2, 1, 2, 1, 1], g_kernel_sizes=[3] * 10, use_ar=False)
    d = HiFiGANMultiScaleMultiPeriodDiscriminator()
    for m in (g, gb, d):
        list(m.parameters())
        for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
            assert clone in optim_hook._watched
            assert clone._handle is None and clone._lib is None
            assert all(torch.equal(a, b) and a is not b for a, b in zip(m.state_dict().values(), clone.state_dict().values()))
            if hasattr(clone, "_plist"):
                assert all(p is q for p, q in zip(clone._plist(), clone.parameters()))  # its own tensors, not the original's
            flag = []
            clone.invalidate_parameters = lambda flag=flag: flag.append(1)
            opt = torch.optim.SGD(list(clone.parameters())[:1], lr=0.1)
            next(iter(clone.parameters())).grad = torch.zeros_like(next(iter(clone.parameters())))
            opt.step()
            assert flag == [1]
