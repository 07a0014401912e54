from .load import load_model  # noqa: F401
from .pcm import pcm16  # noqa: F401
