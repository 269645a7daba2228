"""UNet definitions the quantised engine runs (state-dict compatible with the reference's)."""
from . import ddim_unet, ldm_unet  # noqa: F401
