from . import attention, lora_layers, residual, unet, utils  # noqa: F401
