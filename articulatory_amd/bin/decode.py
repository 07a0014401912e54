"""Autoregressive chunk driver — counterpart of ``ar_loop`` in the reference's
articulatory/bin/decode.py:31-83 (non-WSOLA, a2w branch).

The reference loop is batch-1 Python: one generator forward per chunk with a host round trip for
``prev_samples``.  Here the whole loop (all chunks, PastFCEncoder included, feedback taken straight
from the output buffer) is enqueued on the device by one C-ABI call; ``ar_loop_batch`` runs B
equal-length utterances side by side, which the reference cannot.
"""

import torch


def _chunk_frames(config, params_key="generator_params"):
    in_chunk_len = int(config["batch_max_steps"] / config["hop_size"])
    past_out_len = config[params_key]["ar_input"]
    return in_chunk_len, past_out_len


def ar_loop(model, x, config, do_wsola=False, modality=None, generator2=False):
    """x: (art_len, num_feats) tensor on the model's device -> (audio_len,) tensor."""
    if do_wsola or modality is not None or generator2:
        raise NotImplementedError("ar_loop: WSOLA / multi-modality / generator2 variants are not built (SURVEY.md §8 f3)")
    if config.get("dataset_mode", "a2w") == "w2a":
        raise NotImplementedError("ar_loop: w2a (inversion) models are out of scope")
    in_chunk_len, _ = _chunk_frames(config)
    if x.dim() == 1:
        x = x.unsqueeze(1)
    c = x.transpose(0, 1).unsqueeze(0)  # (1, num_feats, art_len)
    return model.ar_synthesis(c, in_chunk_len)[0]


def ar_loop_batch(model, xs, config):
    """xs: (B, art_len, num_feats) equal-length utterances -> (B, audio_len)."""
    in_chunk_len, _ = _chunk_frames(config)
    return model.ar_synthesis(xs.permute(0, 2, 1), in_chunk_len)
