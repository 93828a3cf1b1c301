"""Drop-in for DDPM_clean_application/src/diffusion_modules.py (see unet.py beside this file)."""
from diffusionmodelscustom_amd.diffusion_modules import DiffusionUtils  # noqa: F401
