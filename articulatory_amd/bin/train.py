#!/usr/bin/env python3
"""GAN training of the a2w HiFi-GAN / HiFi-CAR recipes on MI355X — the counterpart of the reference's ``articulatory-train``
(articulatory/bin/train.py) for the path this package covers: HiFiGANGenerator + HiFiGANMultiScaleMultiPeriodDiscriminator with the mel
(or no) auxiliary loss, adversarial + feature-matching losses, Adam / RAdam + torch LR schedulers, the reference's checkpoint layout.

``Trainer.train_step`` follows ``Trainer._train_step`` (train.py:241-440) statement by statement for that path: generator forward
(autograd through libhificar), mel loss, D(cat([ar, y_]), D(cat([ar, y])) -> adversarial + feature matching, generator update; then the
generator forward again without a graph ("re-compute y_ which leads better quality", :389), discriminator real / fake losses,
discriminator update.  Every network, loss and gradient runs in libhificar; PyTorch holds the parameters, the optimizers and the
schedule.  Data parallelism: one process per GPU, ``sync_gradients`` on both networks (the reference's DDP wrap is disabled,
train.py:1790-1801): one all-reduce per network per step over RCCL.

Data: ``--train-dumpdir`` (the reference's dump directory: ``<utt>.h5`` files with "wave" / "feats" datasets, read by
articulatory_amd/utils/hdf5.py, or ``-wave.npy`` / ``-feats.npy`` pairs), ``--audio-scp`` / ``--feats-scp`` (``utt path`` lines of ``.npy``
arrays; libsndfile is not in this image) or ``--synthetic N`` (N random utterances, for benchmarks and smoke runs).  Windows are cut as the reference's collater does
in ``random_window`` mode (train.py:1013-1097): ``batch_max_steps`` samples, the matching frames, and the ``ar_input`` samples before
the window (zero padded on the left) as the AR context.

    python -m articulatory_amd.bin.train --config conf/e2w_hifigan_car.yaml --outdir exp/run --synthetic 256
    python -m torch.distributed.run --nproc-per-node 8 -m articulatory_amd.bin.train --config ... --outdir ...
"""
import argparse
import contextlib
import logging
import os
import time
from collections import defaultdict

import numpy as np
import torch
import yaml

from articulatory_amd.losses import MelSpectrogramLoss, MultiResolutionSTFTLoss
from articulatory_amd.models import HiFiGANGenerator


class WindowCollater:
    """random_window packaging of (audio, features) pairs (train.py:978-1035, 1071-1097 for the a2w + AR case).

    As in the reference, an utterance must be LONGER than the window (``len(art) + end_offset > start_offset``, train.py:987: shorter
    or equal ones are left out of the batch) and the first frame is drawn from ``[0, len - batch_max_frames)`` (np.random.randint's
    exclusive upper bound, train.py:1013).  In a DataLoader worker process the generator is re-seeded from the worker's own seed
    (distinct per worker, and per epoch because torch draws a new base seed for every iterator): copies of one parent generator would
    otherwise all replay the same draws."""

    def __init__(self, batch_max_steps, hop_size, ar_len=None, rng=None, seed=None, use_spk_id=False, use_ph=False):
        assert batch_max_steps % hop_size == 0
        self.batch_max_steps, self.hop_size, self.ar_len = batch_max_steps, hop_size, ar_len
        self.use_spk_id, self.use_ph = use_spk_id, use_ph  # SpeechCollater's flags (train.py:914-915): items then carry a dict with these keys
        self.batch_max_frames = batch_max_steps // hop_size
        self.seed = seed
        self.rng = rng or np.random.default_rng(seed)
        self._worker_seed = None

    def _generator(self):
        info = torch.utils.data.get_worker_info()
        if info is not None and self._worker_seed != info.seed:
            self._worker_seed = info.seed
            self.rng = np.random.default_rng([int(info.seed) & 0xFFFFFFFFFFFFFFFF, 0 if self.seed is None else int(self.seed)])
        return self.rng

    def __call__(self, items):
        """items: [(audio (T,), feats (frames, C))] or [(audio, feats, {"spk_id": int, "ph": (frames,) ints})]
        -> {"x": (B, C, frames), "y": (B, 1, T), "ar": (B, 1, ar_len), "spk_id": (B,) long, "ph": (B, frames) long}.

        ``ph`` is cut with the window's FRAME starts, exactly like the features (train.py:1028-1031, aux_context_window 0); ``spk_id`` rides
        along (train.py:990-998)."""
        rng = self._generator()
        items = [(it[0], it[1][: len(it[0]) // self.hop_size], it[2] if len(it) > 2 else {}) for it in items]
        items = [it for it in items if len(it[1]) > self.batch_max_frames]
        if not items:
            raise ValueError(f"no utterance of the batch is longer than the window ({self.batch_max_frames} frames)")
        audios, feats, extras = zip(*items)
        starts = np.array([rng.integers(0, len(c) - self.batch_max_frames) for c in feats])
        wav_starts = starts * self.hop_size
        y = np.stack([a[s:s + self.batch_max_steps] for a, s in zip(audios, wav_starts)])
        x = np.stack([c[s:s + self.batch_max_frames] for c, s in zip(feats, starts)])
        batch = {"x": torch.from_numpy(x.astype(np.float32)).transpose(2, 1).contiguous(), "y": torch.from_numpy(y.astype(np.float32)).unsqueeze(1)}
        if self.ar_len is not None:
            ars = []
            for a, s in zip(audios, wav_starts):
                ar = a[max(0, s - self.ar_len):s]
                ars.append(np.pad(ar, (self.ar_len - len(ar), 0), "constant"))
            batch["ar"] = torch.from_numpy(np.stack(ars).astype(np.float32)).unsqueeze(1)
        if self.use_spk_id:
            batch["spk_id"] = torch.tensor([int(e["spk_id"]) for e in extras], dtype=torch.long)
        if self.use_ph:
            for e, c in zip(extras, feats):
                if len(e["ph"]) < len(c):
                    raise ValueError(f"a phoneme sequence of {len(e['ph'])} frames for {len(c)} feature frames")
            batch["ph"] = torch.from_numpy(np.stack([np.asarray(e["ph"])[s:s + self.batch_max_frames] for e, s in zip(extras, starts)]).astype(np.int64))
        return batch


class Conditioning:
    """Per-utterance conditioning of the reference's SpeechDataset (audio_mel_dataset.py:403-461, 513-519): ``utt2spk`` (``utt spk`` lines;
    speaker ids are the ranks in the sorted speaker list, or in ``spks`` when a dev set re-uses the training set's list) and ``ph.scp``
    (``utt path.npy`` lines: one phoneme index per feature frame)."""

    def __init__(self, utt2spk=None, ph_scp=None, spks=None):
        def read(p):
            with open(p) as f:
                return dict(line.split(None, 1) for line in f.read().splitlines() if line.strip())

        self.utt2spk = {k: v.strip() for k, v in read(utt2spk).items()} if utt2spk else None
        self.spks = list(spks) if spks is not None else (sorted(set(self.utt2spk.values())) if self.utt2spk else None)
        self.spk2id = {s: i for i, s in enumerate(self.spks)} if self.spks is not None else None
        if self.utt2spk is not None:
            assert all(s in self.spk2id for s in self.utt2spk.values()), "utt2spk names a speaker that is not in the speaker list"
        self.ph = {k: v.strip() for k, v in read(ph_scp).items()} if ph_scp else None

    def known(self, utt):
        return (self.utt2spk is None or utt in self.utt2spk) and (self.ph is None or utt in self.ph)

    def extras(self, utt, frames):
        e = {}
        if self.utt2spk is not None:
            e["spk_id"] = self.spk2id[self.utt2spk[utt]]
        if self.ph is not None:
            e["ph"] = np.asarray(np.load(self.ph[utt])).reshape(-1)[:frames]
        return e


class NpyPairs(torch.utils.data.Dataset):
    """(audio, features) pairs from two ``utt value`` scp files (values as articulatory_amd/utils/scp.py reads them); utterances shorter than the window are dropped
    (remove_short_samples, audio_mel_dataset.py of the reference)."""

    def __init__(self, audio_scp, feats_scp, hop_size, min_frames, cond=None):
        self.cond = cond

        def read(p):
            with open(p) as f:
                return dict(line.split(None, 1) for line in f.read().splitlines() if line.strip())

        from articulatory_amd.utils.scp import load_scp_value

        self._load = load_scp_value  # values: .npy, .h5[:dataset] ("wave" / "feats" by default), .ark:offset
        a, c = read(audio_scp), read(feats_scp)
        self.items = []
        for utt in sorted(set(a) & set(c)):
            if cond is not None and not cond.known(utt):
                continue
            v = c[utt].strip()
            frames = np.load(v, mmap_mode="r").shape[0] if v.endswith(".npy") else self._load(v, "feats").shape[0]
            if frames > min_frames:  # (strictly longer than the window: audio_mel_dataset.py:57-80 keeps lengths > threshold)
                self.items.append((a[utt].strip(), v, utt))
        self.hop_size = hop_size

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        audio = np.asarray(self._load(self.items[i][0], "wave"), np.float32).reshape(-1)
        feats = np.asarray(self._load(self.items[i][1], "feats"), np.float32)
        n = min(len(audio) // self.hop_size, len(feats))
        if self.cond is not None:
            return audio[: n * self.hop_size], feats[:n], self.cond.extras(self.items[i][2], n)
        return audio[: n * self.hop_size], feats[:n]


class DumpDirPairs(torch.utils.data.Dataset):
    """The reference's dump directory (AudioMelDataset, articulatory/datasets/audio_mel_dataset.py as set up at train.py:1530-1570):
    ``format: hdf5`` -> ``<utt>.h5`` files with "wave" and "feats" datasets; ``format: npy`` -> ``<utt>-wave.npy`` + ``<utt>-feats.npy``."""

    def __init__(self, dumpdir, fmt, hop_size, min_frames, cond=None):
        import glob

        self.cond = cond

        from articulatory_amd.utils.hdf5 import read_hdf5

        self.hop_size = hop_size
        if fmt == "hdf5":
            files = sorted(glob.glob(os.path.join(dumpdir, "**", "*.h5"), recursive=True))
            self.load = lambda f: (read_hdf5(f, "wave"), read_hdf5(f, "feats"))
        elif fmt == "npy":
            files = sorted(glob.glob(os.path.join(dumpdir, "**", "*-wave.npy"), recursive=True))
            self.load = lambda f: (np.load(f), np.load(f.replace("-wave.npy", "-feats.npy")))
        else:
            raise ValueError("support only hdf5 or npy format.")
        # utterance id = the file's base name without its suffix (audio_mel_dataset.py:382-389)
        self.utt_of = lambda f: os.path.basename(f)[: -len("-wave.npy")] if fmt == "npy" else os.path.splitext(os.path.basename(f))[0]
        self.files = [f for f in files if (read_hdf5(f, "feats") if fmt == "hdf5" else np.load(f.replace("-wave.npy", "-feats.npy"), mmap_mode="r")).shape[0]
                      > min_frames and (cond is None or cond.known(self.utt_of(f)))]

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        audio, feats = self.load(self.files[i])
        audio, feats = np.asarray(audio, np.float32).reshape(-1), np.asarray(feats, np.float32)
        n = min(len(audio) // self.hop_size, len(feats))
        if self.cond is not None:
            return audio[: n * self.hop_size], feats[:n], self.cond.extras(self.utt_of(self.files[i]), n)
        return audio[: n * self.hop_size], feats[:n]


class SyntheticPairs(torch.utils.data.Dataset):
    def __init__(self, n, frames, dims, hop_size, seed=0, num_spk=0, num_ph=0):
        rng = np.random.default_rng(seed)
        self.items = [((rng.standard_normal(frames * hop_size) * 0.1).astype(np.float32), rng.standard_normal((frames, dims)).astype(np.float32))
                      for _ in range(n)]
        if num_spk or num_ph:  # random speakers / frame-level phoneme indices (conditioned recipes)
            self.items = [it + ({**({"spk_id": int(rng.integers(0, num_spk))} if num_spk else {}),
                                 **({"ph": rng.integers(0, num_ph, size=frames)} if num_ph else {})},) for it in self.items]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _optimizer(kind, params, kw, fused=True):
    cls = getattr(torch.optim, kind)  # (the reference's articulatory.optimizers re-exports torch.optim + RAdam; torch has RAdam now)
    # fused (default; config key fused_optimizers: false for torch's foreach sequence): the same Adam update as one multi-tensor kernel — the
    # foreach form's ~20 small launches per optimizer are issued slower than the GPU runs them (≈ 2 ms of idle GPU per iteration at the recipe)
    if fused and kind in ("Adam", "AdamW") and "fused" not in kw:
        kw = dict(kw, fused=True)  # one multi-tensor kernel per step instead of torch's foreach sequence (same update rule)
    return cls(params, **kw)


class Trainer:
    """The generator / discriminator / criterion / optimizer bundle of one rank."""

    def __init__(self, config, device, distributed=False):
        self.config, self.device, self.distributed = config, device, distributed
        gtype_name = config.get("generator_type", "HiFiGANGenerator")
        if gtype_name not in ("HiFiGANGenerator", "GBlockGenerator"):
            raise NotImplementedError(f"generator_type {gtype_name} is not built")
        import articulatory_amd.models as models

        dtype_name = config.get("discriminator_type", "HiFiGANMultiScaleMultiPeriodDiscriminator")
        if dtype_name not in ("HiFiGANMultiScaleMultiPeriodDiscriminator", "HiFiGANMultiScaleDiscriminator", "HiFiGANMultiPeriodDiscriminator"):
            raise NotImplementedError(f"discriminator_type {dtype_name} is not built")
        for flag in ("use_subband_stft_loss", "use_inter_loss", "use_pcd"):
            if config.get(flag, False):
                raise NotImplementedError(f"{flag} is not built (SURVEY.md §8 f1 covers the HiFi-GAN / HiFi-CAR recipes: mel or multi-resolution STFT loss)")
        gp = config["generator_params"]
        self.use_ar = bool(gp.get("use_ar", False))
        self.use_ph_loss = bool(gp.get("use_ph_loss", False))  # train.py:1735-1739: the generator's flag decides, criterion = F.cross_entropy
        self.G = getattr(models, gtype_name)(**gp, precision="f32").to(device).train()  # train.py:1649-1660
        self.D = getattr(models, dtype_name)(**config["discriminator_params"]).to(device).train()  # train.py:1661-1668
        self.mel = MelSpectrogramLoss(**config["mel_loss_params"]) if config.get("use_mel_loss", False) else None
        self.stft = MultiResolutionSTFTLoss(**config.get("stft_loss_params", {})) if config.get("use_stft_loss", False) else None  # train.py:1688
        if distributed:
            self.G.sync_gradients()
            self.D.sync_gradients()
        self.optimizer = {
            "generator": _optimizer(config.get("generator_optimizer_type", "RAdam"), self.G.parameters(), config["generator_optimizer_params"],
                                    config.get("fused_optimizers", True)),
            "discriminator": _optimizer(config.get("discriminator_optimizer_type", "RAdam"), self.D.parameters(),
                                        config["discriminator_optimizer_params"], config.get("fused_optimizers", True)),
        }
        # (the modules notice parameter updates through the tensors' version counters — which torch's fused optimizers do not bump; both
        # networks watch optimizer steps themselves: articulatory_amd/utils/optim_hook.py)
        self.scheduler = {
            k: getattr(torch.optim.lr_scheduler, config.get(f"{k}_scheduler_type", "StepLR"))(optimizer=self.optimizer[k],
                                                                                               **config[f"{k}_scheduler_params"])
            for k in ("generator", "discriminator")
        }
        self.steps = self.epochs = 0
        self.total_train_loss = defaultdict(float)

    # ------------------------------------------------------------------ one iteration (train.py:241-440)
    def _check_conditioning(self, batch):
        """The reference's collaters slice ``ph`` with the windows' frame starts and pass ``spk_id`` along (train.py:1029-1032, 248-249);
        WindowCollater does the same when built with ``use_spk_id`` / ``use_ph`` over datasets that carry the side tables (``Conditioning``)."""
        gp = self.config["generator_params"]
        need = [k for k, on in (("spk_id", gp.get("use_spk_id", False)), ("ph", gp.get("use_ph", False) or self.use_ph_loss)) if on and k not in batch]
        if need:
            raise ValueError(f"the generator is conditioned on {' / '.join(need)} (generator_params) but the batch has no such entry: "
                             f"batch keys {sorted(batch)}.  build WindowCollater with use_spk_id / use_ph over a dataset with a Conditioning table, or add the entries yourself "
                             "(ph: (B, frames) indices sliced with the window's frame starts, spk_id: (B,))")

    def train_step(self, batch):
        cfg = self.config
        x = batch["x"].to(self.device, non_blocking=True)
        y = batch["y"].to(self.device, non_blocking=True)
        ar = batch["ar"].to(self.device, non_blocking=True) if self.use_ar else None
        spk_id = batch["spk_id"].to(self.device, non_blocking=True) if "spk_id" in batch else None  # train.py:248-249
        ph = batch["ph"].to(self.device, non_blocking=True) if "ph" in batch else None
        self._check_conditioning(batch)
        log = {}
        adv_on = self.steps > cfg["discriminator_train_start_steps"]
        disc_y = (torch.cat([ar, y], dim=2) if self.use_ar else y) if adv_on else None  # train.py:340-346 (the same tensor in both parts)
        ga = cfg.get("generator_adv_loss_params", {})
        da = cfg.get("discriminator_adv_loss_params", {})
        fm = cfg.get("feat_match_loss_params", {})
        #######################
        #      Generator      #
        #######################
        if self.steps > cfg.get("generator_train_start_steps", 0):
            y_ = self.G(x, spk_id=spk_id, ar=ar, ph=ph)
            if self.use_ph_loss:
                y_, ph_ = y_
            gen_loss = 0.0
            # The auxiliary losses and the adversarial criterion both depend on y_ only: with the adversarial part on, the auxiliary loss
            # (a separate engine: framing + four GEMMs, value and gradient in one call) is enqueued on a side stream next to the
            # discriminators' passes and joined before the sum below (same terms in the same order).
            side = self._aux_side_stream() if adv_on and not self.use_ph_loss and cfg.get("overlap_aux_loss", True) else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                y_.record_stream(side)
                y.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                if self.stft is not None:  # train.py:288-297
                    sc_loss, mag_loss = self.stft(y_, y)
                    gen_loss = gen_loss + sc_loss + mag_loss
                    log["train/spectral_convergence_loss"] = sc_loss.detach()
                    log["train/log_stft_magnitude_loss"] = mag_loss.detach()
                if self.mel is not None:
                    mel_loss = self.mel(y_, y)
                    gen_loss = gen_loss + mel_loss
                    log["train/mel_loss"] = mel_loss.detach()
                gen_loss = gen_loss * cfg.get("lambda_aux", 1.0)
            aux_side = side if torch.is_tensor(gen_loss) else None
            if self.use_ph_loss:  # train.py:327-331: frame-rate phoneme logits (B, num_ph, T) against the phoneme indices
                ph_loss = torch.nn.functional.cross_entropy(ph_, ph.long())
                gen_loss = gen_loss + cfg["lambda_ph"] * ph_loss
                log["train/ph_loss"] = ph_loss.detach()
            if adv_on:
                disc_y_ = torch.cat([ar, y_], dim=2) if self.use_ar else y_
                use_fm = cfg.get("use_feat_match_loss", False)
                total, adv, fml = self._generator_criterion(
                    disc_y_, disc_y if use_fm else None, loss_type=ga.get("loss_type", "mse"),
                    average_by_discriminators=ga.get("average_by_discriminators", True), lambda_adv=cfg["lambda_adv"],
                    lambda_feat_match=cfg.get("lambda_feat_match", 0.0) if use_fm else 0.0,
                    fm_average_by_layers=fm.get("average_by_layers", True), fm_average_by_discriminators=fm.get("average_by_discriminators", True),
                    fm_include_final_outputs=fm.get("include_final_outputs", False))
                if aux_side is not None:  # the auxiliary loss joins the main stream here
                    torch.cuda.current_stream().wait_stream(aux_side)
                    gen_loss.record_stream(torch.cuda.current_stream())
                    aux_side = None
                gen_loss = gen_loss + total
                log["train/adversarial_loss"] = adv
                if use_fm:
                    log["train/feature_matching_loss"] = fml
            log["train/generator_loss"] = gen_loss.detach()
            self.optimizer["generator"].zero_grad(set_to_none=True)
            gen_loss.backward()
            if adv_on and cfg.get("early_real_gradient", True) and hasattr(self.D, "start_real_gradient"):
                # the discriminator update's real pass does not wait for the generator update: its backward starts now, on a side stream
                self.D.start_real_gradient(loss_type=da.get("loss_type", "mse"), average_by_discriminators=da.get("average_by_discriminators", True))
            if cfg.get("generator_grad_norm", -1) > 0:
                torch.nn.utils.clip_grad_norm_(self.G.parameters(), cfg["generator_grad_norm"])
            self.optimizer["generator"].step()
            self._scheduler_step("generator", gen_loss)
        #######################
        #    Discriminator    #
        #######################
        if adv_on:
            with torch.no_grad():
                y_ = self.G(x, spk_id=spk_id, ar=ar, ph=ph)  # re-compute y_ (train.py:389-400): a training-mode forward without a graph
                if self.use_ph_loss:
                    y_, _ = y_
            disc_y_ = torch.cat([ar, y_], dim=2) if self.use_ar else y_
            dis_loss, real_loss, fake_loss = self._discriminator_criterion(disc_y_, disc_y, loss_type=da.get("loss_type", "mse"),
                                                                           average_by_discriminators=da.get("average_by_discriminators", True))
            log.update({"train/real_loss": real_loss, "train/fake_loss": fake_loss, "train/discriminator_loss": dis_loss.detach()})
            self.optimizer["discriminator"].zero_grad(set_to_none=True)
            dis_loss.backward()
            if cfg.get("discriminator_grad_norm", -1) > 0:
                torch.nn.utils.clip_grad_norm_(self.D.parameters(), cfg["discriminator_grad_norm"])
            self.optimizer["discriminator"].step()
            self._scheduler_step("discriminator", dis_loss)
        self.steps += 1
        return log

    def _aux_side_stream(self):
        if getattr(self, "_aux_stream", None) is None:
            self._aux_stream = torch.cuda.Stream(device=self.device)
        return self._aux_stream

    # The GAN criterion: one fused native node per side (D.generator_loss / D.discriminator_loss).  A spectrally normalised discriminator
    # advances its power iteration at every D(x) of the reference's step (train.py:347,356,421-422: three or four different weight sets per
    # iteration), which one fused node cannot mirror: those run forward(x, native=True) per call + the reductions of articulatory_amd.losses.
    def _generator_criterion(self, disc_y_, disc_y, loss_type, average_by_discriminators, lambda_adv, lambda_feat_match, fm_average_by_layers,
                             fm_average_by_discriminators, fm_include_final_outputs):
        if not getattr(self.D, "_spectral", False):
            return self.D.generator_loss(disc_y_, disc_y, loss_type=loss_type, average_by_discriminators=average_by_discriminators, lambda_adv=lambda_adv,
                                         lambda_feat_match=lambda_feat_match, fm_average_by_layers=fm_average_by_layers,
                                         fm_average_by_discriminators=fm_average_by_discriminators, fm_include_final_outputs=fm_include_final_outputs)
        from articulatory_amd import losses as NL

        p_ = self.D(disc_y_, native=True)
        adv = NL.generator_adversarial_loss(p_, average_by_discriminators, loss_type)
        fm = torch.zeros((), device=disc_y_.device)
        if disc_y is not None:
            with torch.no_grad():
                p = self.D(disc_y, native=True)
            fm = NL.feature_match_loss(p_, p, fm_average_by_layers, fm_average_by_discriminators, fm_include_final_outputs)
        return lambda_adv * (adv + lambda_feat_match * fm), adv.detach(), fm.detach()

    def _discriminator_criterion(self, disc_y_, disc_y, loss_type, average_by_discriminators):
        if not getattr(self.D, "_spectral", False):
            return self.D.discriminator_loss(disc_y_, disc_y, loss_type=loss_type, average_by_discriminators=average_by_discriminators)
        from articulatory_amd import losses as NL

        p = self.D(disc_y, native=True)
        p_ = self.D(disc_y_.detach(), native=True)
        real, fake = NL.discriminator_adversarial_loss(p_, p, average_by_discriminators, loss_type)
        return real + fake, real.detach(), fake.detach()

    def _scheduler_step(self, which, loss):
        if self.config.get(f"{which}_scheduler_type", "StepLR") == "ReduceLROnPlateau":  # train.py:380-383,432-435
            self.scheduler[which].step(loss.detach())
        else:
            self.scheduler[which].step()

    # ------------------------------------------------------------------ evaluation (train.py:470-640)
    @torch.no_grad()
    def eval_step(self, batch):
        """The losses of one dev batch without updates (``_eval_step``): both networks in eval mode; the adversarial terms are always
        computed here (no start-step condition in the reference's evaluation either)."""
        cfg = self.config
        x = batch["x"].to(self.device, non_blocking=True)
        y = batch["y"].to(self.device, non_blocking=True)
        ar = batch["ar"].to(self.device, non_blocking=True) if self.use_ar else None
        ga, da, fm = cfg.get("generator_adv_loss_params", {}), cfg.get("discriminator_adv_loss_params", {}), cfg.get("feat_match_loss_params", {})
        log = {}
        spk_id = batch["spk_id"].to(self.device, non_blocking=True) if "spk_id" in batch else None
        ph = batch["ph"].to(self.device, non_blocking=True) if "ph" in batch else None
        self._check_conditioning(batch)
        y_ = self.G(x, spk_id=spk_id, ar=ar, ph=ph)
        if self.use_ph_loss:
            y_, ph_ = y_
        aux_loss = 0.0
        if self.stft is not None:
            sc_loss, mag_loss = self.stft(y_, y)
            aux_loss = aux_loss + sc_loss + mag_loss
            log["eval/spectral_convergence_loss"], log["eval/log_stft_magnitude_loss"] = sc_loss, mag_loss
        if self.mel is not None:
            mel_loss = self.mel(y_, y)
            aux_loss = aux_loss + mel_loss
            log["eval/mel_loss"] = mel_loss
        aux_loss = aux_loss * cfg.get("lambda_aux", 1.0)
        if self.use_ph_loss:  # train.py:551-553
            ph_loss = torch.nn.functional.cross_entropy(ph_, ph.long())
            aux_loss = aux_loss + cfg["lambda_ph"] * ph_loss
            log["eval/ph_loss"] = ph_loss
        disc_y = torch.cat([ar, y], dim=2) if self.use_ar else y
        disc_y_ = torch.cat([ar, y_], dim=2) if self.use_ar else y_
        use_fm = cfg.get("use_feat_match_loss", False)
        total, adv, fml = self._generator_criterion(
            disc_y_, disc_y if use_fm else None, loss_type=ga.get("loss_type", "mse"), average_by_discriminators=ga.get("average_by_discriminators", True),
            lambda_adv=cfg["lambda_adv"], lambda_feat_match=cfg.get("lambda_feat_match", 0.0) if use_fm else 0.0,
            fm_average_by_layers=fm.get("average_by_layers", True), fm_average_by_discriminators=fm.get("average_by_discriminators", True),
            fm_include_final_outputs=fm.get("include_final_outputs", False))
        dis_loss, real_loss, fake_loss = self._discriminator_criterion(disc_y_, disc_y, loss_type=da.get("loss_type", "mse"),
                                                                       average_by_discriminators=da.get("average_by_discriminators", True))
        log.update({"eval/adversarial_loss": adv, "eval/generator_loss": aux_loss + total, "eval/real_loss": real_loss, "eval/fake_loss": fake_loss,
                    "eval/discriminator_loss": dis_loss})
        if use_fm:
            log["eval/feature_matching_loss"] = fml
        return log

    def eval_epoch(self, loader, outdir=None):
        """``_eval_epoch``: average dev losses; the best mel loss so far keeps ``best_mel_ckpt.pkl`` / ``best_mel_step.txt`` (train.py:621-626)."""
        self.G.eval()
        self.D.eval()
        totals, n = defaultdict(float), 0
        for batch in loader:
            for k, v in self.eval_step(batch).items():
                totals[k] += float(v)
            n += 1
        self.G.train()
        self.D.train()
        if n == 0:
            return {}
        avg = {k: v / n for k, v in totals.items()}
        if outdir is not None and "eval/mel_loss" in avg and avg["eval/mel_loss"] < getattr(self, "best_mel_loss", float("inf")):
            self.best_mel_loss = avg["eval/mel_loss"]
            os.makedirs(outdir, exist_ok=True)
            with open(os.path.join(outdir, "best_mel_step.txt"), "w+") as f:
                f.write("%d\n" % self.steps)
            self.save_checkpoint(os.path.join(outdir, "best_mel_ckpt.pkl"))
        return avg

    # ------------------------------------------------------------------ checkpoints (train.py:140-238)
    def save_checkpoint(self, path):
        state = {
            "optimizer": {k: v.state_dict() for k, v in self.optimizer.items()},
            "scheduler": {k: v.state_dict() for k, v in self.scheduler.items()},
            "steps": self.steps,
            "epochs": self.epochs,
            "model": {"generator": self.G.state_dict(), "discriminator": self.D.state_dict()},
        }
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        torch.save(state, path)

    def load_checkpoint(self, path, load_only_params=False):
        state = torch.load(path, map_location="cpu")
        self.G.load_state_dict(state["model"]["generator"])
        self.D.load_state_dict(state["model"]["discriminator"])
        if not load_only_params:
            self.steps, self.epochs = state["steps"], state["epochs"]
            for k in ("generator", "discriminator"):
                self.optimizer[k].load_state_dict(state["optimizer"][k])
                self.scheduler[k].load_state_dict(state["scheduler"][k])


def hop_of(config):
    return int(np.prod(config["generator_params"]["upsample_scales"]))


def feature_dims(config):
    gp = config["generator_params"]
    # (the phoneme embeddings are appended to the input rows inside the generator, hifigan.py:217-220: they are not feature columns)
    return gp["in_channels"] - (gp.get("ar_output", 0) if gp.get("use_ar", False) else 0) - (gp.get("ph_emb_size", 0) if gp.get("use_ph", False) else 0)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", required=True)
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--audio-scp")
    ap.add_argument("--feats-scp")
    ap.add_argument("--train-dumpdir", help="dump directory of <utt>.h5 (wave + feats) or <utt>-wave.npy / <utt>-feats.npy files (config: format)")
    ap.add_argument("--dev-dumpdir", help="dev-set dump directory: evaluated every eval_interval_steps (rank 0)")
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many random utterances instead of a dataset")
    ap.add_argument("--utt2spk", help="'utt spk' lines (use_spk_id: speaker ids are ranks in the sorted speaker list)")
    ap.add_argument("--ph-scp", help="'utt path.npy' lines: one phoneme index per feature frame (use_ph / use_ph_loss)")
    ap.add_argument("--dev-utt2spk", help="the dev set's utt2spk (default: --utt2spk); speaker ids follow the training set's list")
    ap.add_argument("--dev-ph-scp", help="the dev set's ph.scp (default: --ph-scp)")
    ap.add_argument("--resume", default="")
    ap.add_argument("--max-steps", type=int, default=None, help="override train_max_steps")
    ap.add_argument("--verbose", type=int, default=1)
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO if a.verbose else logging.WARN, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    with open(a.config) as f:
        config = yaml.safe_load(f)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("articulatory_amd.bin.train needs a MI355X: the networks only exist as HIP kernels")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        from articulatory_amd.utils.affinity import pin_rank

        # ~1000 launches per iteration are enqueued from Python: every rank gets its own cores and a bounded intra-op pool — before the process
        # group exists, so that RCCL's threads are born inside the slice
        pinned = pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        torch.distributed.init_process_group("nccl")  # RCCL
        if torch.distributed.get_world_size() != world:
            raise SystemExit(f"WORLD_SIZE={world} but the process group has {torch.distributed.get_world_size()} ranks")
        logging.info(f"rank {rank}/{world}: {pinned or 'host affinity unchanged'}")
    config["distributed"] = world > 1
    hop = hop_of(config)
    frames = config["batch_max_steps"] // hop
    gp = config["generator_params"]
    ar_len = gp.get("ar_input") if gp.get("use_ar", False) else None
    # conditioned generators (train.py:1574-1576): spk_id per utterance, ph per frame, sliced with the windows by the collater
    use_spk_id = bool(gp.get("use_spk_id", False))
    use_ph = bool(gp.get("use_ph", False) or gp.get("use_ph_loss", False))
    cond = dev_cond = None
    if (use_spk_id or use_ph) and not a.synthetic:
        if use_spk_id and not a.utt2spk:
            raise SystemExit("generator_params.use_spk_id needs --utt2spk (the reference reads data/<stage>/utt2spk, audio_mel_dataset.py:412-419)")
        if use_ph and not a.ph_scp:
            raise SystemExit("generator_params.use_ph / use_ph_loss need --ph-scp (the reference reads data/<stage>/ph.scp, audio_mel_dataset.py:450-461)")
        cond = Conditioning(a.utt2spk if use_spk_id else None, a.ph_scp if use_ph else None)
        if use_spk_id and len(cond.spks) != gp["num_spk"]:  # train.py:1584-1585
            raise SystemExit(f"{len(cond.spks)} speakers in {a.utt2spk} but generator_params.num_spk = {gp['num_spk']}")
        if a.dev_dumpdir:
            dev_cond = Conditioning((a.dev_utt2spk or a.utt2spk) if use_spk_id else None, (a.dev_ph_scp or a.ph_scp) if use_ph else None, spks=cond.spks)
    if (use_spk_id or use_ph) and a.synthetic and a.dev_dumpdir:
        # synthetic training data carries its own speaker / phoneme ids; a real dev set still needs its side tables (its speaker list defines the ids)
        if (use_spk_id and not (a.dev_utt2spk or a.utt2spk)) or (use_ph and not (a.dev_ph_scp or a.ph_scp)):
            raise SystemExit("a conditioned generator with --dev-dumpdir needs the dev set's side tables: --dev-utt2spk (use_spk_id) / --dev-ph-scp "
                             "(use_ph, use_ph_loss), also with --synthetic training data")
        dev_cond = Conditioning((a.dev_utt2spk or a.utt2spk) if use_spk_id else None, (a.dev_ph_scp or a.ph_scp) if use_ph else None)
        if use_spk_id and len(dev_cond.spks) != gp["num_spk"]:
            raise SystemExit(f"{len(dev_cond.spks)} speakers in the dev set's utt2spk but generator_params.num_spk = {gp['num_spk']}")
    if a.synthetic:
        data = SyntheticPairs(a.synthetic, 4 * frames, feature_dims(config), hop, seed=rank, num_spk=gp.get("num_spk") if use_spk_id else 0,
                              num_ph=gp.get("num_ph") if use_ph else 0)
    elif a.train_dumpdir:
        data = DumpDirPairs(a.train_dumpdir, config.get("format", "hdf5"), hop, frames, cond)
    else:
        if not (a.audio_scp and a.feats_scp):
            raise SystemExit("give --train-dumpdir, or --audio-scp and --feats-scp, or --synthetic N")
        data = NpyPairs(a.audio_scp, a.feats_scp, hop, frames, cond)
    sampler = torch.utils.data.distributed.DistributedSampler(data, world, rank, shuffle=True) if world > 1 else None
    loader = torch.utils.data.DataLoader(data, batch_size=config["batch_size"], shuffle=sampler is None, sampler=sampler, drop_last=True,
                                         collate_fn=WindowCollater(config["batch_max_steps"], hop, ar_len, seed=1234 + rank, use_spk_id=use_spk_id, use_ph=use_ph),
                                         num_workers=config.get("num_workers", 0), pin_memory=config.get("pin_memory", False))
    if len(loader) == 0:
        raise SystemExit(f"fewer utterances ({len(data)}) than one batch ({config['batch_size']})")
    dev_loader = None
    if a.dev_dumpdir and rank == 0:
        dev = DumpDirPairs(a.dev_dumpdir, config.get("format", "hdf5"), hop, frames, dev_cond)
        dev_loader = torch.utils.data.DataLoader(dev, batch_size=config["batch_size"], shuffle=False, drop_last=False,
                                                 collate_fn=WindowCollater(config["batch_max_steps"], hop, ar_len, np.random.default_rng(4321),
                                                                           use_spk_id=use_spk_id, use_ph=use_ph))
    trainer = Trainer(config, device, distributed=world > 1)
    if a.resume:
        trainer.load_checkpoint(a.resume)
        logging.info(f"Successfully resumed from {a.resume}.")
    max_steps = a.max_steps if a.max_steps is not None else config["train_max_steps"]
    t0, n0 = time.time(), trainer.steps
    pending = []
    while trainer.steps < max_steps:
        if sampler is not None:
            sampler.set_epoch(trainer.epochs)
        for batch in loader:
            # the step's losses stay device scalars until the log line needs them: no host synchronisation per step (the reference's
            # ``.item()`` per loss per step, train.py:286-437, is one), so the host keeps enqueueing ahead of the GPU across iterations
            pending.append(trainer.train_step(batch))
            if trainer.steps % config.get("log_interval_steps", 100) == 0 or trainer.steps >= max_steps:
                for log in pending:
                    for k, v in log.items():
                        trainer.total_train_loss[k] += float(v)
                pending = []
            if trainer.steps % config.get("log_interval_steps", 100) == 0 and rank == 0:
                n = config.get("log_interval_steps", 100)
                logging.info(f"(Steps: {trainer.steps}) " + ", ".join(f"{k} = {v / n:.4f}" for k, v in sorted(trainer.total_train_loss.items()))
                             + f", {(time.time() - t0) / max(trainer.steps - n0, 1) * 1e3:.1f} ms/step")
                trainer.total_train_loss = defaultdict(float)
            if dev_loader is not None and trainer.steps % config.get("eval_interval_steps", 10 ** 9) == 0:
                avg = trainer.eval_epoch(dev_loader, a.outdir)
                logging.info(f"(Steps: {trainer.steps}) " + ", ".join(f"{k} = {v:.4f}" for k, v in sorted(avg.items())))
            if trainer.steps % config.get("save_interval_steps", 10 ** 9) == 0 and rank == 0:
                trainer.save_checkpoint(os.path.join(a.outdir, f"checkpoint-{trainer.steps}steps.pkl"))
            if trainer.steps >= max_steps:
                break
        trainer.epochs += 1
    if rank == 0:
        trainer.save_checkpoint(os.path.join(a.outdir, f"checkpoint-{trainer.steps}steps.pkl"))
        logging.info(f"Finished training: {trainer.steps} steps, {(time.time() - t0) / max(trainer.steps - n0, 1) * 1e3:.1f} ms/step.")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
