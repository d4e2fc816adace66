"""Network assemblies of the HIP path (same constructor signatures, forward contract and state_dict keys as
the reference's models/architectures/{deeplab,unet}.py)."""
from .deeplabv3p import DeepLab  # noqa: F401
from .unet_valid import UNet  # noqa: F401
