"""Host logic of the training counterpart (articulatory_amd/bin/train.py) that needs no GPU: the random-window collater (reference
articulatory/bin/train.py:1013-1035, 1071-1097), the .npy pair dataset, loud failure without a GPU."""
import os

import numpy as np
import pytest
import torch

from articulatory_amd.bin import train as T


def test_window_collater_cuts_matching_windows_and_ar_context():
    hop, frames_w, ar_len = 20, 5, 64
    rng = np.random.default_rng(0)
    items = []
    for n in (9, 30, 6, 5):
        feats = np.arange(n * 3, dtype=np.float32).reshape(n, 3) + 1000 * n
        audio = np.arange(n * hop, dtype=np.float32) + 1
        items.append((audio, feats))
    col = T.WindowCollater(frames_w * hop, hop, ar_len, np.random.default_rng(7))
    seen_pad = seen_full = False
    short = items.pop()                            # exactly one window long
    for _ in range(50):
        b = col(items + [short])                   # ... and left out of the batch (reference train.py:987)
        assert b["x"].shape == (3, 3, frames_w) and b["y"].shape == (3, 1, frames_w * hop) and b["ar"].shape == (3, 1, ar_len)
        for i, (audio, feats) in enumerate(items):
            y = b["y"][i, 0].numpy()
            s = int(y[0]) - 1                      # audio[k] = k + 1
            assert s % hop == 0 and np.array_equal(y, audio[s:s + frames_w * hop])
            assert np.array_equal(b["x"][i].numpy().T, feats[s // hop:s // hop + frames_w])   # the frames under the window
            ar = b["ar"][i, 0].numpy()
            k = min(s, ar_len)                     # samples before the window, zero padded on the left (train.py:1084-1095)
            assert np.array_equal(ar[ar_len - k:], audio[s - k:s]) and not ar[:ar_len - k].any()
            seen_pad |= k < ar_len
            seen_full |= k == ar_len
    assert seen_pad and seen_full
    # an utterance exactly one window long is left out of the batch (reference train.py:987) — here the 5-frame item never appears
    with pytest.raises(ValueError):
        col([short])
    starts = {int(col([items[0]])["y"][0, 0, 0]) - 1 for _ in range(200)}
    assert starts == {0, hop, 2 * hop, 3 * hop}    # 9 frames, window 5: np.random.randint(0, 9 - 5) -> 0..3, upper bound exclusive


def test_window_collater_reseeds_per_dataloader_worker():
    hop, frames_w = 4, 3
    rng = np.random.default_rng(0)
    data = [(np.arange(400 * hop, dtype=np.float32), rng.standard_normal((400, 2)).astype(np.float32)) for _ in range(8)]
    col = T.WindowCollater(frames_w * hop, hop, None, seed=11)

    def epoch():
        loader = torch.utils.data.DataLoader(data, batch_size=2, collate_fn=col, num_workers=2)
        return [tuple(b["y"][:, 0, 0].tolist()) for b in loader]

    e1, e2 = epoch(), epoch()
    assert e1[0] != e1[1]          # two workers: different draws (copies of one generator would repeat them)
    assert e1 != e2                # a new epoch: new draws


def test_npy_pairs_drops_short_utterances_and_aligns_lengths(tmp_path):
    hop = 10
    a_lines, c_lines = [], []
    for utt, frames, extra in (("a", 12, 7), ("b", 3, 0), ("c", 20, -15)):
        np.save(tmp_path / f"{utt}-feats.npy", np.zeros((frames, 4), np.float32))
        np.save(tmp_path / f"{utt}-wave.npy", np.zeros(frames * hop + extra, np.float32))
        a_lines.append(f"{utt} {tmp_path / (utt + '-wave.npy')}")
        c_lines.append(f"{utt} {tmp_path / (utt + '-feats.npy')}")
    (tmp_path / "wav.scp").write_text("\n".join(a_lines) + "\n")
    (tmp_path / "feats.scp").write_text("\n".join(c_lines) + "\n")
    ds = T.NpyPairs(str(tmp_path / "wav.scp"), str(tmp_path / "feats.scp"), hop, min_frames=5)
    assert len(ds) == 2                                                   # "b" is not longer than a window
    for audio, feats in (ds[0], ds[1]):
        assert len(audio) == len(feats) * hop                             # trimmed to whole frames both ways
    assert len(ds[1][1]) == 18                                            # "c": 185 samples -> 18 frames


def test_trainer_rejects_unbuilt_options_before_touching_the_gpu():
    base = dict(generator_params={}, discriminator_params={})
    with pytest.raises(NotImplementedError, match="use_subband_stft_loss"):
        T.Trainer(dict(base, use_subband_stft_loss=True), torch.device("cpu"))
    with pytest.raises(NotImplementedError, match="generator_type"):
        T.Trainer(dict(base, generator_type="ParallelWaveGANGenerator"), torch.device("cpu"))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_main_fails_loudly_without_a_gpu(tmp_path):
    cfg = tmp_path / "c.yaml"
    cfg.write_text("batch_size: 2\n")
    with pytest.raises(SystemExit, match="MI355X"):
        T.main(["--config", str(cfg), "--outdir", str(tmp_path), "--synthetic", "4"])


def test_dump_dir_pairs_hdf5_and_npy(tmp_path):
    from articulatory_amd.utils.hdf5 import write_hdf5

    hop = 10
    for utt, frames in (("a", 12), ("b", 3), ("c", 20)):
        feats = np.full((frames, 4), frames, np.float32)
        wave = np.arange(frames * hop + 3, dtype=np.float32)
        write_hdf5(str(tmp_path / "h5" / f"{utt}.h5"), "wave", wave)
        write_hdf5(str(tmp_path / "h5" / f"{utt}.h5"), "feats", feats)
        os.makedirs(tmp_path / "npy", exist_ok=True)
        np.save(tmp_path / "npy" / f"{utt}-wave.npy", wave)
        np.save(tmp_path / "npy" / f"{utt}-feats.npy", feats)
    for fmt in ("hdf5", "npy"):
        ds = T.DumpDirPairs(str(tmp_path / ("h5" if fmt == "hdf5" else "npy")), fmt, hop, min_frames=5)
        assert len(ds) == 2
        audio, feats = ds[1]
        assert feats.shape == (20, 4) and len(audio) == 200 and audio[7] == 7


@pytest.mark.skipif(not os.path.isdir("/root/reference/egs"), reason="the reference's YAML files only exist in the build container")
@pytest.mark.parametrize("recipe,path", [("car", "egs/ema/voc1/conf/e2w_hifigan_car.yaml"), ("e2w", "egs/ema/voc1/conf/e2w_hifigan.yaml"),
                                         ("mri", "egs/mri/voc1/conf/mri2w_hifigan_car.yaml")])
def test_recipe_values_equal_the_shipped_yaml(recipe, path):
    """articulatory_amd/utils/recipes.py restates the shipped YAMLs as values (they do not travel to the GPU box): every key the Trainer
    reads equals the file's."""
    import yaml

    from articulatory_amd.utils.recipes import recipe_train_config

    with open(os.path.join("/root/reference", path)) as f:
        ref = yaml.safe_load(f)
    mine = recipe_train_config(recipe)
    for k, v in mine.items():
        if k in ("distributed", "fused_optimizers", "stft_loss_params"):
            continue
        if k == "generator_params":  # final_scale / extra_art: only in the AR YAMLs; the build's class accepts and ignores them
            v = {kk: vv for kk, vv in v.items() if kk in ref[k]}
        assert ref[k] == v, (k, ref[k], v)
    assert mine["discriminator_params"] == ref["discriminator_params"]


def test_conditioned_generators_need_conditioning_batches():
    """use_spk_id / use_ph / use_ph_loss: a batch built without the conditioning entries (a caller's own collater; WindowCollater carries them when
    its flags are on) gets a clear error instead of an AttributeError on None deep inside the step."""
    import types

    from articulatory_amd.bin.train import Trainer

    t = types.SimpleNamespace(config={"generator_params": {"use_spk_id": True, "use_ph": False}}, use_ph_loss=True)
    with pytest.raises(ValueError, match="spk_id / ph"):
        Trainer._check_conditioning(t, {"x": None, "y": None})
    Trainer._check_conditioning(t, {"x": None, "y": None, "spk_id": None, "ph": None})
    t = types.SimpleNamespace(config={"generator_params": {}}, use_ph_loss=False)
    Trainer._check_conditioning(t, {"x": None})


def test_window_collater_carries_spk_id_and_slices_ph_with_the_windows(tmp_path):
    """use_spk_id / use_ph (reference SpeechCollater, train.py:990-998, 1028-1031): ``ph`` is cut with the SAME frame starts as the features and
    the audio, ``spk_id`` rides along, and an utterance too short for a window drops out of all of them together."""
    hop, frames_w = 20, 5
    items = []
    for i, n in enumerate((9, 30, 6, 5)):
        feats = np.stack([np.arange(n, dtype=np.float32), np.full(n, 100.0 * i, np.float32)], 1)   # column 0 = the frame's index
        audio = np.arange(n * hop, dtype=np.float32)
        items.append((audio, feats, {"spk_id": 10 + i, "ph": 7 * np.arange(n + 2)}))              # ph may be longer than the features
    col = T.WindowCollater(frames_w * hop, hop, None, np.random.default_rng(3), use_spk_id=True, use_ph=True)
    for _ in range(20):
        b = col(items)
        assert b["spk_id"].dtype == torch.long and b["spk_id"].tolist() == [10, 11, 12]            # the 5-frame utterance is left out
        assert b["ph"].dtype == torch.long and b["ph"].shape == (3, frames_w)
        assert torch.equal(b["ph"], 7 * b["x"][:, 0, :].long())                                     # same frames as the features
        assert torch.equal(b["y"][:, 0, 0].long(), b["x"][:, 0, 0].long() * hop)                     # ... and as the audio
    with pytest.raises(ValueError, match="phoneme sequence"):
        col([(items[0][0], items[0][1], {"spk_id": 0, "ph": np.arange(3)})])
    plain = T.WindowCollater(frames_w * hop, hop, None, np.random.default_rng(3))                   # flags off: extras are ignored
    assert set(plain(items)) == {"x", "y"}

    # the side tables of the reference's SpeechDataset: utt2spk (ids = ranks in the sorted speaker list) and ph.scp
    d = tmp_path
    lines_w, lines_f, lines_s, lines_p = [], [], [], []
    for utt, spk, n in (("u1", "zoe", 12), ("u2", "amy", 20), ("u3", "zoe", 14), ("u4", "bob", 16)):
        np.save(d / f"{utt}-wave.npy", np.arange(n * hop, dtype=np.float32))
        np.save(d / f"{utt}-feats.npy", np.stack([np.arange(n, dtype=np.float32)] * 2, 1))
        lines_w.append(f"{utt} {d / (utt + '-wave.npy')}")
        lines_f.append(f"{utt} {d / (utt + '-feats.npy')}")
        if utt != "u4":  # u4 has no speaker / phoneme entry: it is not part of the conditioned set
            np.save(d / f"{utt}-ph.npy", np.arange(n) % 5)
            lines_s.append(f"{utt} {spk}")
            lines_p.append(f"{utt} {d / (utt + '-ph.npy')}")
    for name, lines in (("wav.scp", lines_w), ("feats.scp", lines_f), ("utt2spk", lines_s), ("ph.scp", lines_p)):
        (d / name).write_text("\n".join(lines) + "\n")
    cond = T.Conditioning(str(d / "utt2spk"), str(d / "ph.scp"))
    assert cond.spks == ["amy", "zoe"]
    ds = T.NpyPairs(str(d / "wav.scp"), str(d / "feats.scp"), hop, min_frames=5, cond=cond)
    assert len(ds) == 3
    got = {int(ds[i][2]["spk_id"]) for i in range(3)}
    assert got == {0, 1} and all(len(ds[i][2]["ph"]) == len(ds[i][1]) for i in range(3))
    dd = T.DumpDirPairs(str(d), "npy", hop, min_frames=5, cond=T.Conditioning(str(d / "utt2spk"), None, spks=["amy", "bob", "zoe"]))
    assert len(dd) == 3 and sorted(int(dd[i][2]["spk_id"]) for i in range(3)) == [0, 2, 2]         # a dev set on the training set's speaker list
    b = T.WindowCollater(frames_w * hop, hop, 8, np.random.default_rng(1), use_spk_id=True, use_ph=True)([ds[i] for i in range(3)])
    assert torch.equal(b["ph"], b["x"][:, 0, :].long() % 5) and b["ar"].shape == (3, 1, 8)
    syn = T.SyntheticPairs(3, 12, 4, hop, seed=0, num_spk=4, num_ph=9)
    sb = T.WindowCollater(frames_w * hop, hop, None, np.random.default_rng(1), use_spk_id=True, use_ph=True)([syn[i] for i in range(3)])
    assert sb["ph"].shape == (3, frames_w) and int(sb["ph"].max()) < 9 and int(sb["spk_id"].max()) < 4
