"""articulatory_amd — MI355X-native HiFi-GAN / HiFi-CAR generator forward (EMA -> 16-kHz waveform).

One hot path of articulatory/articulatory, rebuilt as hand-written HIP kernels for gfx950 behind the
reference's ``generator_type`` plugin surface.  See DESIGN.md for the scope and INTEGRATION.md for how
the reference binds to it.
"""
from . import models  # noqa: F401

__version__ = "0.1.0"
