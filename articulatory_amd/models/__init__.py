"""Generator registry: ``getattr(articulatory_amd.models, config["generator_type"])``.

Same lookup the reference performs on ``articulatory.models`` (articulatory/bin/train.py:1649-1653,
articulatory/utils/utils.py:325-328; plugin recipe in the reference README.md:65).
"""
from .discriminator import (  # noqa: F401
    HiFiGANMultiPeriodDiscriminator,
    HiFiGANMultiScaleDiscriminator,
    HiFiGANMultiScaleMultiPeriodDiscriminator,
    HiFiGANPeriodDiscriminator,
    HiFiGANScaleDiscriminator,
)
from .gblock import GBlockGenerator  # noqa: F401
from .hifigan import HiFiGANGenerator  # noqa: F401

__all__ = ["HiFiGANGenerator", "GBlockGenerator", "HiFiGANMultiScaleMultiPeriodDiscriminator", "HiFiGANMultiScaleDiscriminator", "HiFiGANMultiPeriodDiscriminator",
           "HiFiGANScaleDiscriminator", "HiFiGANPeriodDiscriminator"]
