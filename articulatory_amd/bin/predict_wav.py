#!/usr/bin/env python3
"""Predict waveforms with a trained generator — counterpart of the reference's
egs/ema/voc1/local/predict_wav.py:24-137 (same flags, same scp / config / checkpoint conventions).

Differences, all on the device side: the autoregressive loop of every utterance runs as ONE enqueued
C-ABI call (no per-chunk host round trip), and utterances of any lengths can be batched
(``--batch-size``; the reference is strictly one utterance at a time) — each utterance's waveform is the one it
gets alone.  ``soundfile`` is not in this
image, so 16-bit PCM WAV files are written with the standard library.
"""

import argparse
import logging
import os
import wave

import numpy as np
import torch
import yaml

from articulatory_amd.bin.decode import ar_loop, ar_loop_ragged, windows
from articulatory_amd.utils import load_model
from articulatory_amd.utils.scp import load_scp_value


def write_wav(path, y, sampling_rate):
    """float waveform in [-1, 1] -> mono PCM_16 WAV (what sf.write's default subtype produces for .wav).

    Sample conversion as libsndfile does it for float32 input (src/pcm.c f2s_array: lrintf(src * 32767.f), i.e. a FLOAT32
    product rounded to nearest-even), so files match the reference's ``sf.write`` sample for sample inside [-1, 1]; outside it
    libsndfile (clipping off, its default) wraps around where this writer clips."""
    y = np.asarray(y).reshape(-1)
    if y.dtype == np.int16:  # already converted on the device (articulatory_amd.utils.pcm16)
        pcm = y.astype("<i2")
    else:
        pcm = np.clip(np.rint(y.astype(np.float32) * np.float32(32767.0)), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sampling_rate))
        f.writeframes(pcm.tobytes())


def read_scp(path):
    """kaldi-style 'utt_id value' lines (predict_wav.py:95-105); values as articulatory_amd/utils/scp.py reads them (.npy, .h5, .ark:offset)."""
    fids, featps = [], []
    with open(path, "r") as inf:
        for line in inf:
            parts = line.strip().split()
            if len(parts) < 2:
                continue
            fids.append(parts[0])
            featps.append(parts[1])
    return fids, featps


def get_parser():
    parser = argparse.ArgumentParser(description="Decode dumped features with trained generator.")
    parser.add_argument("--feats-scp", "--scp", default=None, type=str, help="kaldi-style feats.scp file.")
    parser.add_argument("--outdir", type=str, required=True, help="directory to save generated speech.")
    parser.add_argument("--checkpoint", type=str, required=True, help="checkpoint file to be loaded.")
    parser.add_argument("--config", default=None, type=str,
                        help="yaml format configuration file. if not explicitly provided, "
                             "it will be searched in the checkpoint directory. (default=None)")
    parser.add_argument("--verbose", type=int, default=1, help="logging level. higher is more logging. (default=1)")
    parser.add_argument("--batch-size", type=int, default=1,
                        help="synthesise up to this many utterances (any lengths) per device call (extension; default=1)")
    return parser


def synthesize_file_list(model, fids, featps, config, device, outdir, batch_size=1, writer=write_wav):
    """The generation loop of predict_wav.py:124-137.  ``batch_size`` > 1 (not in the reference): AR utterances are
    synthesised ``batch_size`` at a time, whatever their lengths — each exactly as if it were alone."""
    use_ar = bool(config["generator_params"].get("use_ar", False))
    written = []

    def kept():
        for fid, featp in zip(fids, featps):
            c = torch.tensor(load_scp_value(featp), dtype=torch.float).to(device)
            if c.shape[0] > 250:  # the reference skips short utterances (predict_wav.py:130)
                yield fid, c

    with torch.no_grad():
        if use_ar and batch_size > 1:
            for batch in windows(kept(), 8 * batch_size):  # one device call per window, batch_size utterances in flight
                ys = [ar_loop(model, batch[0][1], config)] if len(batch) == 1 else ar_loop_ragged(model, [c for _, c in batch], config, batch=batch_size)
                for (fid, _), y in zip(batch, ys):
                    writer(os.path.join(outdir, fid + ".wav"), y.cpu().numpy(), config["sampling_rate"])
                    written.append(fid)
            return written
        for fid, c in kept():
            if use_ar:
                y = ar_loop(model, c, config)
            else:
                if len(c.shape) == 1:
                    c = c.long()
                y = model.inference(c)
            writer(os.path.join(outdir, fid + ".wav"), y.cpu().numpy(), config["sampling_rate"])
            written.append(fid)
    return written


def main(argv=None):
    args = get_parser().parse_args(argv)
    level = logging.DEBUG if args.verbose > 1 else logging.INFO if args.verbose > 0 else logging.WARN
    logging.basicConfig(level=level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    if args.verbose <= 0:
        logging.warning("Skip DEBUG/INFO messages")
    if not os.path.exists(args.outdir):
        os.makedirs(args.outdir)
    if args.config is None:
        args.config = os.path.join(os.path.dirname(args.checkpoint), "config.yml")
    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.Loader)
    config.update(vars(args))
    fids, featps = read_scp(args.feats_scp)

    if not torch.cuda.is_available():
        raise RuntimeError("predict_wav: no GPU visible; this package has no CPU synthesis path")
    # under torchrun (one process per GPU) every rank synthesises its own share of the list and writes its own files
    from articulatory_amd.bin.shard import shard_items
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    pairs = shard_items(list(zip(fids, featps)))
    fids, featps = [p[0] for p in pairs], [p[1] for p in pairs]
    device = torch.device("cuda")
    model = load_model(args.checkpoint, config)
    logging.info(f"Loaded model parameters from {args.checkpoint}.")
    model.remove_weight_norm()
    model = model.eval().to(device)
    print(sum(p.numel() for p in model.parameters() if p.requires_grad))
    synthesize_file_list(model, fids, featps, config, device, config["outdir"], batch_size=args.batch_size)


if __name__ == "__main__":
    main()
