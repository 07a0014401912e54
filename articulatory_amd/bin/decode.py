#!/usr/bin/env python3
"""Decoding drivers — counterparts of the reference's articulatory/bin/decode.py.

* ``ar_loop`` (reference decode.py:31-100): the chunked autoregressive driver.  The reference loop is
  batch-1 Python with one generator forward and one host round trip of ``prev_samples`` per chunk; here
  the whole non-WSOLA loop (all chunks, PastFCEncoder included, feedback taken straight from the output
  buffer) is enqueued on the device by ONE C-ABI call.  ``ar_loop_batch`` runs B equal-length utterances
  side by side and ``ar_loop_ragged`` B utterances of different lengths (each exactly as if it were alone),
  which the reference cannot.  The WSOLA variant (``do_wsola``: half-overlapping chunks,
  decode.py:84-100) is a per-chunk loop over ``model.forward`` as in the reference, since each chunk's
  context comes from the *middle* of the previous chunk.
* ``main`` (reference decode.py:103-358, console script ``articulatory-decode``): same flags, config
  merge, scp / dump-dir inputs (``.npy`` features), ``--normalize-before``, PCM_16 ``<utt>_gen.wav``
  outputs and RTF report, for the a2w dataset modes of the HiFi-GAN / HiFi-CAR generator.
"""

import argparse
import glob
import logging
import os
import time

import numpy as np
import torch
import yaml

from articulatory_amd.utils.scp import is_supported, load_scp_value


def _chunk_frames(config, params_key="generator_params"):
    in_chunk_len = int(config["batch_max_steps"] / config["hop_size"])
    past_out_len = config[params_key]["ar_input"]
    return in_chunk_len, past_out_len


def ar_loop(model, x, config, do_wsola=False, modality=None, generator2=False):
    """x: (art_len, num_feats) tensor on the model's device -> (audio_len,) tensor
    (``do_wsola``: -> (list of chunk waveforms, list of chunk inputs), as the reference returns)."""
    if modality is not None or generator2:
        raise NotImplementedError("ar_loop: multi-modality / generator2 variants are not built (SURVEY.md §8 f3)")
    if config.get("dataset_mode", "a2w") == "w2a":
        raise NotImplementedError("ar_loop: w2a (inversion) models are out of scope")
    in_chunk_len, past_out_len = _chunk_frames(config)
    if x.dim() == 1:
        x = x.unsqueeze(1)
    if not do_wsola:
        c = x.transpose(0, 1).unsqueeze(0)  # (1, num_feats, art_len)
        return model.ar_synthesis(c, in_chunk_len)[0]

    # decode.py:84-100: chunks start every in_chunk_len/2 frames and are in_chunk_len (+ extra_art) long
    extra_art = config["generator_params"]["extra_art"]
    audio_chunk_len = config["batch_max_steps"]
    assert in_chunk_len % 2 == 0
    ins = [x[i:i + in_chunk_len + int(extra_art)] for i in range(0, len(x), int(in_chunk_len / 2))]
    prev_samples = torch.zeros((1, 1, past_out_len), dtype=x.dtype, device=x.device)
    outs = []
    for art_i, art in enumerate(ins):
        signal = model(art.unsqueeze(0).permute(0, 2, 1), ar=prev_samples)  # (1, 1, audio_chunk_length)
        outs.append(signal[0][0])
        if art_i < len(ins) - 1:
            prev_samples = signal[:, :, int(audio_chunk_len / 2) - past_out_len:int(audio_chunk_len / 2)]
            assert prev_samples.shape[2] == past_out_len
    return outs, ins


def ar_loop_batch(model, xs, config):
    """xs: (B, art_len, num_feats) equal-length utterances -> (B, audio_len)."""
    in_chunk_len, _ = _chunk_frames(config)
    return model.ar_synthesis(xs.permute(0, 2, 1), in_chunk_len)


def pad_utterances(xs):
    """list of (T_i, C) tensors -> ((B, T_max, C) zero-padded tensor, [T_i])."""
    lens = [int(x.shape[0]) for x in xs]
    out = torch.zeros((len(xs), max(lens), xs[0].shape[1]), dtype=xs[0].dtype, device=xs[0].device)
    for i, x in enumerate(xs):
        out[i, :lens[i]] = x
    return out, lens


def ar_loop_ragged(model, xs, config, batch=64):
    """xs: list of (T_i, num_feats) tensors of any lengths -> list of (hop * T_i,) waveforms; every utterance gets the
    result of ``ar_loop`` on it alone (its own short tail chunk included).  One device call for the whole list: at most
    ``batch`` utterances are in flight and a finished one is replaced by the next (longest first, so the tail is short)."""
    in_chunk_len, _ = _chunk_frames(config)
    order = sorted(range(len(xs)), key=lambda i: -int(xs[i].shape[0]))
    padded, lens = pad_utterances([xs[i] for i in order])
    y = model.ar_synthesis_packed(padded.permute(0, 2, 1), in_chunk_len, lens, batch=batch)
    hop = y.shape[1] // padded.shape[1]
    out = [None] * len(xs)
    for k, i in enumerate(order):
        out[i] = y[k, :hop * lens[k]]
    return out


def windows(items, n):
    """Consecutive groups of up to n items of an iterable (bounded memory for a whole-dataset decode)."""
    buf = []
    for it in items:
        buf.append(it)
        if len(buf) >= n:
            yield buf
            buf = []
    if buf:
        yield buf


def length_batches(items, batch_size, window=8):
    """Group (key, tensor) items into batches of <= batch_size with similar lengths: sort a window of
    ``window * batch_size`` items by length (bounded memory, little padding), cut it into batches."""
    buf = []
    for it in items:
        buf.append(it)
        if len(buf) >= window * batch_size:
            buf.sort(key=lambda kv: kv[1].shape[0])
            while buf:
                yield buf[:batch_size]
                buf = buf[batch_size:]
    buf.sort(key=lambda kv: kv[1].shape[0])
    while buf:
        yield buf[:batch_size]
        buf = buf[batch_size:]


# ----------------------------------------------------------------------------------------------
# articulatory-decode counterpart
# ----------------------------------------------------------------------------------------------
_A2W_MODES = ("default", "m2w", "a2w", "a2w_pcd")


def list_features(feats_scp=None, dumpdir=None, fmt="npy"):
    """(utt_id, path) pairs of a kaldi-style ``utt_id path.npy`` scp (the reference's NpyScpLoader, utils/utils.py:240-291)
    or of a dump dir of ``<utt_id>-feats.npy`` files (decode.py:207-222).  Nothing is loaded here."""
    if (feats_scp is not None) == (dumpdir is not None):
        raise ValueError("Please specify either --dumpdir or --feats-scp.")
    pairs = []
    if feats_scp is not None:
        with open(feats_scp) as f:
            for line in f:
                parts = line.strip().split()
                if len(parts) < 2:
                    continue
                if not is_supported(parts[1]):
                    raise ValueError("Not supported feats.scp type.")
                pairs.append((parts[0], parts[1]))
    else:
        if fmt == "hdf5":  # decode.py:210-212: <utt_id>.h5 files holding a "feats" dataset
            for path in sorted(glob.glob(os.path.join(dumpdir, "**", "*.h5"), recursive=True)):
                pairs.append((os.path.splitext(os.path.basename(path))[0], path))
        elif fmt == "npy":
            for path in sorted(glob.glob(os.path.join(dumpdir, "**", "*-feats.npy"), recursive=True)):
                pairs.append((os.path.basename(path)[: -len("-feats.npy")], path))
        else:
            raise ValueError("Support only hdf5 or npy format.")
    return pairs


def _load(path):
    return load_scp_value(path)  # .npy, .h5[:dataset], .ark:offset (articulatory_amd/utils/scp.py)


def npy_frames(path):
    """Frame count of a (T, C) feature file (.npy: from its header alone, no data is read)."""
    if not path.endswith(".npy"):
        return int(_load(path).shape[0])
    return int(np.load(path, mmap_mode="r").shape[0])


def load_features(pairs):
    """(utt_id, path) pairs -> (utt_id, (T, C) ndarray), one utterance at a time, as the reference's loop streams them."""
    for utt_id, path in pairs:
        yield utt_id, _load(path)


def iter_features(feats_scp=None, dumpdir=None, fmt="npy"):
    """(utt_id, (T, C) ndarray) pairs, loaded lazily."""
    return load_features(list_features(feats_scp, dumpdir, fmt))


def get_parser():
    parser = argparse.ArgumentParser(description="Decode dumped features with trained generator.")
    parser.add_argument("--feats-scp", "--scp", default=None, type=str,
                        help="kaldi-style feats.scp file. you need to specify either feats-scp or dumpdir.")
    parser.add_argument("--dumpdir", default=None, type=str,
                        help="directory including feature files. you need to specify either feats-scp or dumpdir.")
    parser.add_argument("--outdir", type=str, required=True, help="directory to save generated speech.")
    parser.add_argument("--checkpoint", type=str, required=True, help="checkpoint file to be loaded.")
    parser.add_argument("--config", default=None, type=str,
                        help="yaml format configuration file. if not explicitly provided, "
                             "it will be searched in the checkpoint directory. (default=None)")
    parser.add_argument("--normalize-before", default=False, action="store_true",
                        help="whether to perform feature normalization before input to the model.")
    parser.add_argument("--verbose", type=int, default=1, help="logging level. higher is more logging. (default=1)")
    parser.add_argument("--batch-size", type=int, default=1,
                        help="utterances synthesised per device call (any lengths; not in the reference, which is batch-1)")
    parser.add_argument("--dry-run", default=False, action="store_true",
                        help="print this rank's share of the utterance list as one JSON line and exit (no GPU needed; not in the reference)")
    return parser


def decode_dataset(model, items, config, device, outdir, normalize_before=False, writer=None, batch_size=1):
    """The generation loop of decode.py:292-351 for the a2w modes.  Returns (n_utterances, average RTF).
    ``batch_size`` > 1: ragged batches of utterances per device call (RTF = batch time / batch audio)."""
    from articulatory_amd.bin.predict_wav import write_wav

    writer = writer or write_wav
    use_ar = bool(config["generator_params"].get("use_ar", False))
    do_wsola = bool(config.get("wsola", False))
    total_rtf, n = 0.0, 0
    if batch_size > 1 and not do_wsola:
        with torch.no_grad():
            feats = ((u, torch.tensor(c, dtype=torch.float).to(device)) for u, c in items)
            # AR: a window of 8 batches per device call, continuously batched; non-AR: rectangular-ish batches
            for batch in (windows(feats, 8 * batch_size) if use_ar else length_batches(feats, batch_size)):
                start = time.time()
                xs = [c for _, c in batch]
                if use_ar:
                    ys = ar_loop_ragged(model, xs, config, batch=batch_size)
                else:
                    if normalize_before:
                        xs = [(c - model.mean) / model.scale for c in xs]
                    padded, lens = pad_utterances(xs)
                    yb = model(padded.permute(0, 2, 1), lengths=lens)
                    hop = yb.shape[2] // padded.shape[1]
                    ys = [yb[i, 0, :hop * m] for i, m in enumerate(lens)]
                ys = [y.cpu().numpy() for y in ys]  # device -> host: the synchronisation point the RTF needs
                rtf = (time.time() - start) / (sum(len(y) for y in ys) / config["sampling_rate"])
                for (utt_id, _), y in zip(batch, ys):
                    writer(os.path.join(outdir, f"{utt_id}_gen.wav"), y, config["sampling_rate"])
                    total_rtf += rtf
                    n += 1
        return n, (total_rtf / n if n else float("nan"))
    with torch.no_grad():
        for utt_id, c in items:
            c = torch.tensor(c, dtype=torch.float).to(device)
            start = time.time()
            if use_ar:
                y = ar_loop(model, c, config, do_wsola=do_wsola)
            else:
                y = model.inference(c, normalize_before=normalize_before).view(-1)
            if not do_wsola:
                y = y.cpu().numpy()  # device -> host: also the synchronisation point the RTF needs
                rtf = (time.time() - start) / (len(y) / config["sampling_rate"])
                total_rtf += rtf
                writer(os.path.join(outdir, f"{utt_id}_gen.wav"), y, config["sampling_rate"])
            else:
                signals, arts = y
                for cyi, cy in enumerate(signals):
                    cy = cy.cpu().numpy()
                    rtf = (time.time() - start) / (len(cy) / config["sampling_rate"])
                    total_rtf += rtf
                    writer(os.path.join(outdir, "%s_%d_gen.wav" % (utt_id, cyi)), cy, config["sampling_rate"])
                    np.save(os.path.join(outdir, "%s_%d.npy" % (utt_id, cyi)), arts[cyi].cpu().numpy())
            n += 1
    return n, (total_rtf / n if n else float("nan"))


def main(argv=None):
    from articulatory_amd.utils import load_model

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
    if (args.feats_scp is not None and args.dumpdir is not None) or (args.feats_scp is None and args.dumpdir is None):
        raise ValueError("Please specify either --dumpdir or --feats-scp.")
    dataset_mode = config.setdefault("dataset_mode", "default")
    if dataset_mode not in _A2W_MODES:
        raise NotImplementedError(f"dataset_mode {dataset_mode!r}: only the articulatory/mel -> waveform modes "
                                  f"{_A2W_MODES} are built (SURVEY.md §8 f3)")
    if config.get("transform") or config.get("input_transform"):
        raise NotImplementedError("feature transforms are not built")
    pairs = list_features(args.feats_scp, args.dumpdir, config.get("format", "npy"))
    logging.info(f"The number of features to be decoded = {len(pairs)}.")

    # under torchrun (one process per GPU) every rank decodes its own share of the list and writes its own files
    from articulatory_amd.bin.shard import shard_items
    # shard the (utt_id, path) list first — lengths from the .npy headers — and load each rank's utterances lazily, one at a
    # time (every rank loading the whole dataset would cost N x the I/O and a full copy in host RAM per rank)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        pairs = shard_items(pairs, length_of=lambda kv: npy_frames(kv[1]))
    if args.dry_run:  # what this rank would decode, without touching a GPU (tests/test_distributed_gloo.py runs it under an 8-rank torchrun)
        import json
        print(json.dumps({"rank": int(os.environ.get("RANK", "0")), "world_size": int(os.environ.get("WORLD_SIZE", "1")),
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "utterances": [u for u, _ in pairs],
                          "frames": int(sum(npy_frames(pth) for _, pth in pairs))}), flush=True)
        return
    if not torch.cuda.is_available():
        raise RuntimeError("decode: no GPU visible; this package has no CPU synthesis path")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    items = load_features(pairs)
    device = torch.device("cuda")
    model = load_model(args.checkpoint, config)
    logging.info(f"Loaded model parameters from {args.checkpoint}.")
    if args.normalize_before:
        assert hasattr(model, "mean"), "Feature stats are not registered."
        assert hasattr(model, "scale"), "Feature stats are not registered."
    model.remove_weight_norm()
    model = model.eval().to(device)
    print(sum(p.numel() for p in model.parameters() if p.requires_grad))
    n, rtf = decode_dataset(model, items, config, device, config["outdir"], normalize_before=args.normalize_before,
                            batch_size=args.batch_size)
    logging.info(f"Finished generation of {n} utterances (RTF = {rtf:.03f}).")


if __name__ == "__main__":
    main()
