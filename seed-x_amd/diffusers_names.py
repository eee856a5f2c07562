"""The three names the reference's scripts import from ``diffusers`` (eval_*.py: ``from diffusers import AutoencoderKL,
UNet2DConditionModel, EulerDiscreteScheduler``), served by this package's stand-ins."""
from .detokenizer import EulerDiscreteScheduler  # noqa: F401
from .unet import UNet2DConditionModel  # noqa: F401
from .vae import AutoencoderKL  # noqa: F401
