"""Drop-in for DDPM_clean_application/src/unet.py: copy this file (and diffusion_modules.py beside it) over the
reference's two files, and `from .unet import *` in its main_ddpm.py:22 resolves to the MI355X implementation."""
from diffusionmodelscustom_amd.unet import *  # noqa: F401,F403
from diffusionmodelscustom_amd.unet import __all__  # noqa: F401
