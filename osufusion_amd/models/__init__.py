from . import diffusion, rectified_flow  # noqa: F401
