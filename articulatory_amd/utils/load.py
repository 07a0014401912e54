"""``load_model`` — counterpart of the reference's articulatory/utils/utils.py:294-372 for the
generator families this package registers."""

import os

import torch
import yaml


def load_model(checkpoint, config=None, stats=None):
    """Load a trained generator.

    Args:
        checkpoint (str): checkpoint written by the reference trainer
            (``{"model": {"generator": state_dict, ...}, ...}``, articulatory/bin/train.py:147-176).
        config (dict): configuration dict; read from ``dirname(checkpoint)/config.yml`` when None.
        stats (str): statistics file; ``stats.{h5,npy}`` beside the checkpoint is picked up when None.

    Returns:
        torch.nn.Module: the generator (weight norm still applied, on CPU — the caller does
        ``remove_weight_norm(); eval().to(device)`` exactly as with the reference).
    """
    if config is None:
        with open(os.path.join(os.path.dirname(checkpoint), "config.yml")) as f:
            config = yaml.load(f, Loader=yaml.Loader)

    from .. import models  # lazy, as in the reference (circular import)

    generator_type = config.get("generator_type", "ParallelWaveGANGenerator")  # the reference's default
    if not hasattr(models, generator_type):
        raise AttributeError(
            f"generator_type {generator_type!r} is not registered in articulatory_amd.models "
            f"(available: {', '.join(models.__all__)})")
    model_class = getattr(models, generator_type)
    # same typo workaround as the reference (utils.py:330-333)
    generator_params = {k.replace("upsample_kernal_sizes", "upsample_kernel_sizes"): v
                        for k, v in config["generator_params"].items()}
    model = model_class(**generator_params)
    model.load_state_dict(torch.load(checkpoint, map_location="cpu")["model"]["generator"])

    if stats is None:
        dirname = os.path.dirname(checkpoint)
        ext = "h5" if config["format"] == "hdf5" else "npy"
        if os.path.exists(os.path.join(dirname, f"stats.{ext}")):
            stats = os.path.join(dirname, f"stats.{ext}")
    if stats is not None:
        model.register_stats(stats)

    if config["generator_params"]["out_channels"] > 1:
        raise NotImplementedError("multi-band (PQMF) generators are out of scope for this package")
    return model
