from . import diffusion  # noqa: F401
