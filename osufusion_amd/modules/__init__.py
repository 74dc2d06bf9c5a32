from . import attention, residual, unet, utils  # noqa: F401
