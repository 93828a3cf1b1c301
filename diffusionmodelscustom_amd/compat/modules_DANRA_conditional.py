"""Put this directory on PYTHONPATH and the reference drivers' ``from modules_DANRA_conditional import *``
resolves to the MI355X implementation."""
from diffusionmodelscustom_amd.modules_DANRA_conditional import *  # noqa: F401,F403
