"""Put this directory on PYTHONPATH and the reference drivers' ``from diffusion_DANRA_conditional import *``
resolves to the MI355X implementation."""
from diffusionmodelscustom_amd.diffusion_DANRA_conditional import *  # noqa: F401,F403
from diffusionmodelscustom_amd.diffusion_DANRA_conditional import DiffusionUtils  # noqa: F401
