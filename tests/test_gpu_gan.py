"""The GAN training iteration (articulatory_amd/bin/train.py::Trainer.train_step, counterpart of the reference's Trainer._train_step,
articulatory/bin/train.py:241-440) on a MI355X: every logged loss of the first iteration against the CPU oracle's restatement of the same
step, the updates it makes, and the checkpoint layout.  ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch

from conftest import E2W_PARAMS
from articulatory_amd.bin.train import SyntheticPairs, Trainer, WindowCollater
from articulatory_amd.utils.synth import synth_disc_state_dict, synth_state_dict
from oracle import disc_oracle as DO
from oracle import hificar_oracle as O
from oracle.make_golden_disc import SMALL

pytestmark = pytest.mark.gpu


def make_config(use_ar=True):
    gp = dict(E2W_PARAMS, channels=128, upsample_scales=[5, 4], upsample_kernel_sizes=[10, 8], use_ar=use_ar)
    if not use_ar:
        gp["in_channels"] = 13
    adam = {"lr": 2.0e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}
    sched = {"gamma": 0.5, "milestones": [100, 200]}
    return dict(
        generator_type="HiFiGANGenerator", generator_params=gp, discriminator_type="HiFiGANMultiScaleMultiPeriodDiscriminator",
        discriminator_params=SMALL, use_stft_loss=False, use_mel_loss=True,
        mel_loss_params=dict(fs=16000, fft_size=256, hop_size=64, win_length=None, window="hann", num_mels=20, fmin=0, fmax=8000, log_base=None),
        generator_adv_loss_params={"average_by_discriminators": False}, discriminator_adv_loss_params={"average_by_discriminators": False},
        use_feat_match_loss=True, feat_match_loss_params={"average_by_discriminators": False, "average_by_layers": False, "include_final_outputs": False},
        lambda_aux=45.0, lambda_adv=1.0, lambda_feat_match=2.0, batch_size=4, batch_max_steps=400,
        generator_optimizer_type="Adam", generator_optimizer_params=adam, generator_scheduler_type="MultiStepLR", generator_scheduler_params=sched,
        generator_grad_norm=-1, discriminator_optimizer_type="Adam", discriminator_optimizer_params=adam, discriminator_scheduler_type="MultiStepLR",
        discriminator_scheduler_params=sched, discriminator_grad_norm=10.0, discriminator_train_start_steps=0, distributed=False)


def build(config):
    assert torch.cuda.is_available()
    t = Trainer(config, torch.device("cuda:0"))
    gsd = synth_state_dict(config["generator_params"], seed=31)
    dsd = synth_disc_state_dict(config["discriminator_params"], seed=32)
    t.G.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
    t.D.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    data = SyntheticPairs(4, 60, 13, 20, seed=3)
    ar_len = 512 if config["generator_params"]["use_ar"] else None
    batch = WindowCollater(400, 20, ar_len, np.random.default_rng(5))([data[i] for i in range(4)])
    return t, gsd, dsd, batch


@pytest.mark.parametrize("use_ar", [True, False])
def test_first_iteration_losses_vs_oracle(use_ar):
    config = make_config(use_ar)
    t, gsd, dsd, batch = build(config)
    t.steps = 1  # past discriminator_train_start_steps: generator AND discriminator parts
    log = {k: float(v) for k, v in t.train_step(batch).items()}
    assert "libhificar.so" in open("/proc/self/maps").read()
    # ---- the same iteration on the CPU oracle (train.py:262-424 for this configuration)
    x, y = batch["x"], batch["y"]
    ar = batch.get("ar")
    gp = config["generator_params"]
    with torch.no_grad():
        gw = O.fold_weight_norm(gsd)
        y_ = O.generator_forward(gw, gp, x, ar)
        mel = DO.mel_loss(y_, y, **config["mel_loss_params"])
        dw = DO.fold_disc_weight_norm(dsd)
        dy = torch.cat([ar, y], 2) if use_ar else y
        dy_ = torch.cat([ar, y_], 2) if use_ar else y_
        p_, p = DO.disc_forward(dw, SMALL, dy_), DO.disc_forward(dw, SMALL, dy)
        adv = DO.gen_adv_loss(p_, False)
        fm = DO.feat_match_loss(p_, p, False, False, False)
        gen = 45.0 * mel + 1.0 * (adv + 2.0 * fm)
    ref = {"train/mel_loss": float(mel), "train/adversarial_loss": float(adv), "train/feature_matching_loss": float(fm), "train/generator_loss": float(gen)}
    for k, v in ref.items():
        assert abs(log[k] - v) < 1e-4 * max(abs(v), 1e-3), (k, log[k], v)
    # the discriminator part runs on the UPDATED generator (train.py:389): finite, and the real loss is that of the untouched D
    with torch.no_grad():
        real = DO.dis_adv_loss(p_, p, False)[0]
    assert abs(log["train/real_loss"] - float(real)) < 1e-4 * abs(float(real))
    assert np.isfinite(log["train/fake_loss"]) and abs(log["train/discriminator_loss"] - log["train/real_loss"] - log["train/fake_loss"]) < 1e-5
    assert t.steps == 2


def test_training_moves_both_networks_and_checkpoint_roundtrip(tmp_path):
    config = make_config(True)
    t, gsd, dsd, batch = build(config)
    t.steps = 1
    first = {k: float(v) for k, v in t.train_step(batch).items()}
    for _ in range(24):
        log = t.train_step(batch)
    last = {k: float(v) for k, v in log.items()}
    assert all(np.isfinite(v) for v in last.values())
    assert last["train/mel_loss"] < first["train/mel_loss"]              # the generator fits the batch
    assert last["train/discriminator_loss"] < first["train/discriminator_loss"]
    g_now, d_now = t.G.state_dict(), t.D.state_dict()
    assert any(not np.allclose(g_now[k].cpu().numpy(), v) for k, v in gsd.items())
    assert any(not np.allclose(d_now[k].cpu().numpy(), v) for k, v in dsd.items())
    path = os.path.join(tmp_path, "checkpoint-26steps.pkl")
    t.save_checkpoint(path)
    state = torch.load(path, map_location="cpu")
    assert sorted(state) == ["epochs", "model", "optimizer", "scheduler", "steps"]               # train.py:147-176
    assert sorted(state["model"]) == ["discriminator", "generator"] and list(state["model"]["generator"]) == list(gsd)
    assert list(state["model"]["discriminator"]) == list(dsd)
    t2 = Trainer(config, torch.device("cuda:0"))
    t2.load_checkpoint(path)
    assert t2.steps == t.steps
    a = {k: float(v) for k, v in t.train_step(batch).items()}
    b = {k: float(v) for k, v in t2.train_step(batch).items()}
    for k in a:
        assert abs(a[k] - b[k]) <= 1e-5 * max(abs(a[k]), 1e-3), (k, a[k], b[k])  # resumed run = uninterrupted run


def test_distributed_iteration_over_rccl_world1():
    """Trainer(distributed=True): both networks all-reduce their flat gradient buffers inside backward over the "nccl" (RCCL) backend.
    One GPU here, so a world of one rank: the collectives run and the iteration equals the single-process one."""
    import torch.distributed as dist

    config = make_config(True)
    t0, _, _, batch = build(config)
    t0.steps = 1
    want = {k: float(v) for k, v in t0.train_step(batch).items()}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        t = Trainer(config, torch.device("cuda:0"), distributed=True)
        t.G.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(config["generator_params"], seed=31).items()})
        t.D.load_state_dict({k: torch.from_numpy(v) for k, v in synth_disc_state_dict(config["discriminator_params"], seed=32).items()})
        assert t.G._grad_sync is not None and t.D._grad_sync is not None
        t.steps = 1
        got = {k: float(v) for k, v in t.train_step(batch).items()}
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-5 * max(abs(want[k]), 1e-3), (k, got[k], want[k])
        a, b = t.D.state_dict(), t0.D.state_dict()
        assert all(torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7) for k in a)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["warmup_generator_only", "no_feature_matching_hinge", "stft_aux_batch1"])
def test_trainer_option_variants(variant):
    """The branches of _train_step the shipped recipes also pass through or switch on: the generator-only warm-up before
    discriminator_train_start_steps (train.py:347,386), no feature matching + hinge losses, the multi-resolution STFT auxiliary loss."""
    config = make_config(True)
    if variant == "warmup_generator_only":
        config["discriminator_train_start_steps"] = 5
    elif variant == "no_feature_matching_hinge":
        config["use_feat_match_loss"] = False
        config["generator_adv_loss_params"] = {"average_by_discriminators": True, "loss_type": "hinge"}
        config["discriminator_adv_loss_params"] = {"average_by_discriminators": True, "loss_type": "hinge"}
    else:
        config.update(use_mel_loss=False, use_stft_loss=True, stft_loss_params={"fft_sizes": [256, 512, 128], "hop_sizes": [30, 60, 12],
                                                                               "win_lengths": [150, 300, 60], "window": "hann_window"})
    t, gsd, dsd, batch = build(config)
    if variant == "stft_aux_batch1":
        batch = {k: v[:1] for k, v in batch.items()}
    d_before = {k: v.clone() for k, v in t.D.state_dict().items()}
    t.steps = 1  # (as in the reference, nothing trains at step 0: `steps > generator_train_start_steps`, train.py:266)
    logs = [{k: float(v) for k, v in t.train_step(batch).items()} for _ in range(3)]
    assert all(np.isfinite(v) for log in logs for v in log.values())
    if variant == "warmup_generator_only":
        assert all(sorted(log) == ["train/generator_loss", "train/mel_loss"] for log in logs)      # no adversarial part yet
        assert all(torch.equal(v, d_before[k]) for k, v in t.D.state_dict().items())                # the discriminator is untouched
        x, y, ar = batch["x"], batch["y"], batch["ar"]
        with torch.no_grad():
            mel = DO.mel_loss(O.generator_forward(O.fold_weight_norm(gsd), config["generator_params"], x, ar), y, **config["mel_loss_params"])
        assert abs(logs[0]["train/generator_loss"] - 45.0 * float(mel)) < 1e-4 * 45.0 * float(mel)
    elif variant == "no_feature_matching_hinge":
        assert "train/feature_matching_loss" not in logs[0] and "train/adversarial_loss" in logs[0] and "train/real_loss" in logs[0]
        x, y, ar = batch["x"], batch["y"], batch["ar"]
        with torch.no_grad():
            y_ = O.generator_forward(O.fold_weight_norm(gsd), config["generator_params"], x, ar)
            dw = DO.fold_disc_weight_norm(dsd)
            p_, p = DO.disc_forward(dw, SMALL, torch.cat([ar, y_], 2)), DO.disc_forward(dw, SMALL, torch.cat([ar, y], 2))
            adv = DO.gen_adv_loss(p_, True, "hinge")
            real = DO.dis_adv_loss(p_, p, True, "hinge")[0]
        assert abs(logs[0]["train/adversarial_loss"] - float(adv)) < 1e-4 * max(abs(float(adv)), 1e-2)
        assert abs(logs[0]["train/real_loss"] - float(real)) < 1e-4 * float(real)
    else:
        assert "train/spectral_convergence_loss" in logs[0] and "train/log_stft_magnitude_loss" in logs[0] and "train/mel_loss" not in logs[0]
        x, y, ar = batch["x"], batch["y"], batch["ar"]
        with torch.no_grad():
            y_ = O.generator_forward(O.fold_weight_norm(gsd), config["generator_params"], x, ar)
            sc, mag = DO.multi_resolution_stft_loss(y_, y, [256, 512, 128], [30, 60, 12], [150, 300, 60])
        assert abs(logs[0]["train/spectral_convergence_loss"] - float(sc)) < 1e-4 * float(sc)
        assert abs(logs[0]["train/log_stft_magnitude_loss"] - float(mag)) < 1e-4 * float(mag)


def test_iterations_are_bit_reproducible():
    """No atomics anywhere in the training kernels (split sums are reduced in a fixed order, the sub-discriminators' streams only
    overlap independent work): two runs from the same state give bit-identical losses and parameters."""
    config = make_config(True)
    runs = []
    for _ in range(2):
        t, _, _, batch = build(config)
        t.steps = 1
        logs = [{k: float(v) for k, v in t.train_step(batch).items()} for _ in range(4)]
        runs.append((logs, {k: v.clone() for k, v in t.G.state_dict().items()}, {k: v.clone() for k, v in t.D.state_dict().items()}))
    assert runs[0][0] == runs[1][0]
    assert all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
    assert all(torch.equal(runs[0][2][k], runs[1][2][k]) for k in runs[0][2])


def test_eval_step_and_best_mel_checkpoint(tmp_path):
    """eval_step = the reference's _eval_step (train.py:470-600): no updates, every loss against the oracle; eval_epoch keeps the best-mel
    checkpoint (train.py:621-626)."""
    config = make_config(True)
    t, gsd, dsd, batch = build(config)
    g_before = {k: v.clone() for k, v in t.G.state_dict().items()}
    avg = t.eval_epoch([batch, batch], str(tmp_path))
    assert all(torch.equal(v, g_before[k]) for k, v in t.G.state_dict().items()) and t.G.training and t.D.training
    x, y, ar = batch["x"], batch["y"], batch["ar"]
    with torch.no_grad():
        y_ = O.generator_forward(O.fold_weight_norm(gsd), config["generator_params"], x, ar)
        mel = DO.mel_loss(y_, y, **config["mel_loss_params"])
        dw = DO.fold_disc_weight_norm(dsd)
        p_, p = DO.disc_forward(dw, SMALL, torch.cat([ar, y_], 2)), DO.disc_forward(dw, SMALL, torch.cat([ar, y], 2))
        adv, fm = DO.gen_adv_loss(p_, False), DO.feat_match_loss(p_, p, False, False, False)
        real, fake = DO.dis_adv_loss(p_, p, False)
    ref = {"eval/mel_loss": mel, "eval/adversarial_loss": adv, "eval/feature_matching_loss": fm, "eval/generator_loss": 45.0 * mel + adv + 2.0 * fm,
           "eval/real_loss": real, "eval/fake_loss": fake, "eval/discriminator_loss": real + fake}
    assert sorted(avg) == sorted(ref)
    for k, v in ref.items():
        assert abs(avg[k] - float(v)) < 1e-4 * max(abs(float(v)), 1e-3), (k, avg[k], float(v))
    assert open(tmp_path / "best_mel_step.txt").read().strip() == "0" and os.path.exists(tmp_path / "best_mel_ckpt.pkl")
    os.remove(tmp_path / "best_mel_ckpt.pkl")
    t.eval_epoch([batch], str(tmp_path))       # not better than the best so far: no new checkpoint
    assert not os.path.exists(tmp_path / "best_mel_ckpt.pkl")


def test_train_cli_end_to_end_and_resume(tmp_path):
    """python -m articulatory_amd.bin.train on a YAML config: synthetic utterances, three iterations, checkpoint; resumed for two more."""
    import yaml

    from articulatory_amd.bin import train as T

    config = make_config(True)
    config.update(train_max_steps=3, save_interval_steps=100, log_interval_steps=1, num_workers=0, pin_memory=False)
    cfg = tmp_path / "conf.yaml"
    cfg.write_text(yaml.safe_dump(config))
    out = tmp_path / "exp"
    T.main(["--config", str(cfg), "--outdir", str(out), "--synthetic", "8", "--verbose", "0"])
    ck = out / "checkpoint-3steps.pkl"
    assert ck.exists()
    state = torch.load(ck, map_location="cpu")
    assert state["steps"] == 3 and state["optimizer"]["discriminator"]["state"]   # steps 1 and 2 were adversarial: D has Adam moments
    T.main(["--config", str(cfg), "--outdir", str(out), "--synthetic", "8", "--verbose", "0", "--resume", str(ck), "--max-steps", "5"])
    state2 = torch.load(out / "checkpoint-5steps.pkl", map_location="cpu")
    assert state2["steps"] == 5
    k = next(iter(state["model"]["generator"]))
    assert not torch.equal(state["model"]["generator"][k], state2["model"]["generator"][k])


@pytest.mark.parametrize("kind", ["spk", "ph"])
def test_train_cli_with_conditioned_generators(tmp_path, kind):
    """use_spk_id / use_ph + use_ph_loss through main(): the datasets read utt2spk / ph.scp (the reference's SpeechDataset side tables,
    audio_mel_dataset.py:403-461), the collater carries spk_id and slices ph with the windows (train.py:990-998, 1028-1031), and the iterations
    move the conditioning parameters — from files and from --synthetic utterances."""
    import yaml

    from articulatory_amd.bin import train as T

    config = make_config(True)
    if kind == "spk":
        config["generator_params"] = dict(config["generator_params"], use_spk_id=True, num_spk=3, spk_emb_size=8)
        watch = "spk_emb_mat.weight"
    else:
        config["generator_params"] = dict(config["generator_params"], in_channels=13 + 128 + 8, use_ph=True, num_ph=9, ph_emb_size=8, use_ph_loss=True)
        config["lambda_ph"] = 3.0
        watch = "ph_fc.weight"
    config.update(train_max_steps=3, save_interval_steps=100, log_interval_steps=1, num_workers=0, pin_memory=False)
    cfg = tmp_path / "conf.yaml"
    cfg.write_text(yaml.safe_dump(config))
    rng = np.random.default_rng(0)
    hop, lines = 20, {"wav.scp": [], "feats.scp": [], "utt2spk": [], "ph.scp": []}
    for i in range(8):
        n = 40 + 3 * i
        for name, arr in (("wave", (rng.standard_normal(n * hop) * 0.1).astype(np.float32)), ("feats", rng.standard_normal((n, 13)).astype(np.float32)),
                          ("ph", rng.integers(0, 9, size=n))):
            np.save(tmp_path / f"u{i}-{name}.npy", arr)
        lines["wav.scp"].append(f"u{i} {tmp_path}/u{i}-wave.npy")
        lines["feats.scp"].append(f"u{i} {tmp_path}/u{i}-feats.npy")
        lines["utt2spk"].append(f"u{i} s{i % 3}")
        lines["ph.scp"].append(f"u{i} {tmp_path}/u{i}-ph.npy")
    for name, ls in lines.items():
        (tmp_path / name).write_text("\n".join(ls) + "\n")
    side = ["--utt2spk", str(tmp_path / "utt2spk")] if kind == "spk" else ["--ph-scp", str(tmp_path / "ph.scp")]
    for tag, data in (("files", ["--audio-scp", str(tmp_path / "wav.scp"), "--feats-scp", str(tmp_path / "feats.scp")] + side), ("synth", ["--synthetic", "8"])):
        out = tmp_path / ("exp_" + tag)
        T.main(["--config", str(cfg), "--outdir", str(out), "--verbose", "0"] + data)
        state = torch.load(out / "checkpoint-3steps.pkl", map_location="cpu")
        assert state["steps"] == 3
        fresh = getattr(__import__("articulatory_amd.models", fromlist=["x"]), "HiFiGANGenerator")(**config["generator_params"]).state_dict()
        assert state["model"]["generator"][watch].shape == fresh[watch].shape
        assert bool(torch.isfinite(state["model"]["generator"][watch]).all())
    if kind == "spk":  # a speaker table that disagrees with num_spk is refused, as train.py:1584-1585 asserts
        (tmp_path / "utt2spk").write_text("\n".join(f"u{i} s{i % 2}" for i in range(8)) + "\n")
        with pytest.raises(SystemExit, match="num_spk"):
            T.main(["--config", str(cfg), "--outdir", str(tmp_path / "x"), "--verbose", "0", "--audio-scp", str(tmp_path / "wav.scp"),
                    "--feats-scp", str(tmp_path / "feats.scp"), "--utt2spk", str(tmp_path / "utt2spk")])


def test_phoneme_conditioned_iteration_with_ph_loss_vs_oracle():
    """A use_ph + use_ph_loss generator through Trainer.train_step (train.py:273-278, 327-331: y_, ph_ = generator(x, ph=ph);
    gen_loss += lambda_ph * cross_entropy(ph_, ph)): the logged losses of the first iteration against the CPU oracle, and the
    conditioning parameters move."""
    config = make_config(True)
    config["generator_params"] = dict(config["generator_params"], in_channels=13 + 128 + 8, use_ph=True, num_ph=9, ph_emb_size=8, use_ph_loss=True)
    config["lambda_ph"] = 3.0
    t = Trainer(config, torch.device("cuda:0"))
    gp = config["generator_params"]
    gsd = synth_state_dict(gp, seed=31)
    dsd = synth_disc_state_dict(config["discriminator_params"], seed=32)
    t.G.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
    t.D.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    data = SyntheticPairs(4, 60, 13, 20, seed=3)
    batch = WindowCollater(400, 20, 512, np.random.default_rng(5))([data[i] for i in range(4)])
    batch["ph"] = torch.from_numpy(np.random.default_rng(6).integers(0, 9, (4, 20)))
    t.steps = 1
    log = {k: float(v) for k, v in t.train_step(batch).items()}
    with torch.no_grad():
        y_, ph_ = O.generator_forward(O.fold_weight_norm(gsd), gp, batch["x"], batch["ar"], ph=batch["ph"])
        mel = DO.mel_loss(y_, batch["y"], **config["mel_loss_params"])
        ph_loss = torch.nn.functional.cross_entropy(ph_, batch["ph"])
        dw = DO.fold_disc_weight_norm(dsd)
        p_ = DO.disc_forward(dw, SMALL, torch.cat([batch["ar"], y_], 2))
        p = DO.disc_forward(dw, SMALL, torch.cat([batch["ar"], batch["y"]], 2))
        adv, fm = DO.gen_adv_loss(p_, False), DO.feat_match_loss(p_, p, False, False, False)
    ref = {"train/mel_loss": float(mel), "train/ph_loss": float(ph_loss), "train/adversarial_loss": float(adv), "train/feature_matching_loss": float(fm),
           "train/generator_loss": float(45.0 * mel + 3.0 * ph_loss + adv + 2.0 * fm)}
    for k, v in ref.items():
        assert abs(log[k] - v) < 1e-4 * max(abs(v), 1e-3), (k, log[k], v)
    now = t.G.state_dict()
    for k in ("ph_emb_mat.weight", "ph_fc.weight", "ph_fc.bias"):
        assert not np.allclose(now[k].cpu().numpy(), gsd[k]), k


@pytest.mark.parametrize("fused", [True, False])
def test_fused_and_foreach_optimizers_give_the_same_iterations(fused):
    """torch's fused Adam updates the parameters WITHOUT bumping their version counters, which is how the modules notice updates: the
    Trainer's optimizer post-step hooks tell them.  Three iterations with either optimizer form must log the same losses (round-3 defect:
    with fused_optimizers the second forward of an iteration ran on the weights of the step before)."""
    config = dict(make_config(True), fused_optimizers=fused)
    t, gsd, dsd, batch = build(config)
    t.steps = 1
    logs = [{k: float(v) for k, v in t.train_step(batch).items()} for _ in range(3)]
    ref_t, _, _, _ = build(dict(make_config(True), fused_optimizers=False))
    ref_t.steps = 1
    want = [{k: float(v) for k, v in ref_t.train_step(batch).items()} for _ in range(3)]
    for a, b in zip(logs, want):
        for k in b:
            assert abs(a[k] - b[k]) <= 2e-4 * max(abs(b[k]), 1e-3), (fused, k, a[k], b[k])
    assert logs[2]["train/fake_loss"] != logs[0]["train/fake_loss"]


@pytest.mark.parametrize("aux", ["mel", "stft"])
def test_auxiliary_loss_on_a_side_stream_gives_the_same_iterations(aux):
    """The auxiliary loss runs next to the discriminators' passes on a side stream (``overlap_aux_loss``, default on) and joins before the
    sum: three iterations log exactly what the serial order logs (same kernels, same summation order)."""
    base = make_config(True)
    if aux == "stft":
        base.update(use_mel_loss=False, use_stft_loss=True,
                    stft_loss_params={"fft_sizes": [256, 128], "hop_sizes": [64, 32], "win_lengths": [256, 100], "window": "hann_window"})
    runs = []
    for overlap in (True, False):
        t, _, _, batch = build(dict(base, overlap_aux_loss=overlap))
        t.steps = 1
        runs.append([{k: float(v) for k, v in t.train_step(batch).items()} for _ in range(3)])
    for a, b in zip(*runs):
        assert a == b, (a, b)
    assert ("train/mel_loss" in runs[0][0]) == (aux == "mel")


def test_early_real_gradient_gives_the_same_iterations():
    """The real pass of the discriminator update does not depend on the generator update: its backward starts on a side stream right after
    the generator loss's backward (``early_real_gradient``, default on; discriminator.py::start_real_gradient) and the update itself only
    adds the fake pass.  real + fake instead of fake + real: the same sums, so three iterations log exactly the same losses either way."""
    runs = []
    for early in (True, False):
        t, _, _, batch = build(dict(make_config(True), early_real_gradient=early))
        t.steps = 1
        logs = []
        for _ in range(3):
            logs.append({k: float(v) for k, v in t.train_step(batch).items()})
            assert t.D.__dict__.get("_early_real") is None  # consumed by the discriminator update (or never started)
        runs.append(logs)
    for a, b in zip(*runs):
        assert a == b, (a, b)
    # and it really ran early: the hook leaves its result behind until the discriminator loss's backward takes it
    t, _, _, batch = build(make_config(True))
    assert t.D.start_real_gradient() is False  # (nothing cached yet)
    real = torch.rand(2, 1, 600, device="cuda") - 0.5
    fake = (torch.rand(2, 1, 600, device="cuda") - 0.5).requires_grad_(True)
    total, _, _ = t.D.generator_loss(fake, real, lambda_adv=1.0, lambda_feat_match=2.0)
    total.backward()
    assert t.D.start_real_gradient() is True and "_early_real" in t.D.__dict__
    tot, _, _ = t.D.discriminator_loss(fake.detach(), real)
    tot.backward()
    assert "_early_real" not in t.D.__dict__
    early_grads = [p.grad.clone() for p in t.D.parameters()]
    t.D.zero_grad(set_to_none=True)
    tot, _, _ = t.D.discriminator_loss(fake.detach(), real)
    tot.backward()
    for a, p in zip(early_grads, t.D.parameters()):
        assert torch.equal(a, p.grad)


def test_iteration_with_spectrally_normalised_period_discriminators():
    """A discriminator with use_spectral_norm on its period sub-networks through Trainer.train_step: the criterion falls back from the
    fused nodes to one native forward per D(x) (the reference advances the power iteration at each of them) — the first iteration's
    generator-side losses against the CPU oracle, and both networks move."""
    from oracle.make_golden_disc_sn import SN_PERIOD

    config = make_config(True)
    config["discriminator_params"] = dict(SMALL, period_discriminator_params=SN_PERIOD)
    t = Trainer(config, torch.device("cuda:0"))
    gp, dp = config["generator_params"], config["discriminator_params"]
    gsd, dsd = synth_state_dict(gp, seed=31), synth_disc_state_dict(dp, seed=32)
    t.G.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
    t.D.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    data = SyntheticPairs(4, 60, 13, 20, seed=3)
    batch = WindowCollater(400, 20, 512, np.random.default_rng(5))([data[i] for i in range(4)])
    t.steps = 1
    log = {k: float(v) for k, v in t.train_step(batch).items()}
    with torch.no_grad():
        y_ = O.generator_forward(O.fold_weight_norm(gsd), gp, batch["x"], batch["ar"])
        state = {k: torch.from_numpy(v) for k, v in dsd.items()}
        w1, st = DO.fold_disc_spectral_norm(state, training=True)       # D(fake): first power iteration
        p_ = DO.disc_forward(w1, dp, torch.cat([batch["ar"], y_], 2))
        state.update(st)
        w2, st = DO.fold_disc_spectral_norm(state, training=True)       # D(real): second
        p = DO.disc_forward(w2, dp, torch.cat([batch["ar"], batch["y"]], 2))
        adv, fm = DO.gen_adv_loss(p_, False), DO.feat_match_loss(p_, p, False, False, False)
    assert abs(log["train/adversarial_loss"] - float(adv)) < 1e-4 * abs(float(adv))
    assert abs(log["train/feature_matching_loss"] - float(fm)) < 1e-4 * abs(float(fm))
    assert all(np.isfinite(v) for v in log.values())
    now = t.D.state_dict()
    k = "mpd.discriminators.0.convs.0.0.weight_orig"
    assert not np.allclose(now[k].cpu().numpy(), dsd[k]) and not np.allclose(now[k[:-4] + "u"].cpu().numpy(), dsd[k[:-4] + "u"])
