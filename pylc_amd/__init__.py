"""pylc_amd -- MI355X (gfx950) native training / inference path for PyLC's segmentation networks.

The package is a thin host-side mirror of the reference's duck-typed contract:
``UNet`` / ``DeepLab`` (``nn.Module``s with the reference's constructor signatures and state_dict keys),
``MultiLoss`` and ``Model``.  All arithmetic runs in libpylc_hip.so (include/pylc_hip.h); importing
the package without the built library fails loudly -- there is no CPU / eager fallback.
"""
from . import lib  # noqa: F401  (raises if libpylc_hip.so is missing)
from .runtime import runtime  # noqa: F401
from .nets import DeepLab, UNet  # noqa: F401
from . import torch_ops  # noqa: F401  (registers torch.ops.pylc_hip.*)
