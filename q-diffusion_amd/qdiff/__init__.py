"""qdiff — MI355X-native quantised-UNet denoising engine with the q-diffusion `qdiff` API.

Same top-level exports as the reference package (qdiff/__init__.py:1-5)."""
from .block_recon import block_reconstruction
from .layer_recon import layer_reconstruction
from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule
from .quant_model import QuantModel

__all__ = ["block_reconstruction", "layer_reconstruction", "BaseQuantBlock", "QuantModule", "QuantModel"]
