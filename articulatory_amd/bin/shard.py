"""Utterance-batch sharding across the GPUs of one node.

Utterances never interact (no cross-item op; the AR state is per utterance), so the batch axis is
partitioned across ranks with NO data-path collective; the only exchange is one all-gather of the
finished waveforms ("waveform collection only").  One process per GPU (``torchrun``); on ROCm the
``nccl`` backend is RCCL over xGMI.  The reference has no multi-GPU inference at all (its only
distributed code is the disabled DDP wrap at articulatory/bin/train.py:1790-1801).
"""

import torch
import torch.distributed as dist


def shard_range(n_items, world_size, rank):
    """Contiguous [lo, hi) slice of n_items for this rank; requires equal shards so that the gather is a
    single fixed-size collective."""
    if n_items % world_size != 0:
        raise ValueError(f"batch of {n_items} utterances does not split evenly over {world_size} ranks; "
                         "pad or bucket the batch first")
    per = n_items // world_size
    return rank * per, (rank + 1) * per


def shard_items(items, world_size=None, rank=None, length_of=None):
    """This rank's share of a list of utterances for file-to-file decoding (each rank writes its own wav files: no
    collective at all).  With ``length_of`` utterances go longest first to the least-loaded rank, so that every rank gets
    the same amount of audio (deterministic: all ranks compute the same deal).  world_size / rank default to the
    torchrun environment."""
    import os

    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    if world_size <= 1:
        return list(items)
    items = list(items)
    if length_of is None:
        return items[rank::world_size]
    load = [0] * world_size
    mine = []
    for i in sorted(range(len(items)), key=lambda i: (-length_of(items[i]), i)):
        r = min(range(world_size), key=lambda q: (load[q], q))
        load[r] += length_of(items[i])
        if r == rank:
            mine.append(i)
    return [items[i] for i in sorted(mine)]


def synthesize_sharded(synth_fn, feats, group=None):
    """Each rank synthesises its slice of ``feats`` (B, ...) with ``synth_fn`` and every rank receives all
    waveforms (B, n_samples) in the original utterance order.

    synth_fn: (feats_shard) -> (B/W, n_samples) tensor on the compute device, e.g.
              ``lambda x: ar_loop_batch(model, x, config)``, or ``lambda x: pcm16(ar_loop_batch(model, x, config))``
              to collect PCM_16 (half the bytes on the links).
    """
    if not dist.is_available() or not dist.is_initialized():
        return synth_fn(feats)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(feats.shape[0], world, rank)
    y = synth_fn(feats[lo:hi]).contiguous()
    out = torch.empty((world * y.shape[0],) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
    if y.dtype == torch.int16:
        # PCM_16 shards (utils.pcm16): neither RCCL nor gloo has a 16-bit integer type; the gather moves bytes
        dist.all_gather_into_tensor(out.view(torch.uint8), y.view(torch.uint8), group=group)
    else:
        dist.all_gather_into_tensor(out, y, group=group)
    return out
