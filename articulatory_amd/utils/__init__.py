from .load import load_model  # noqa: F401
