"""torch.autograd bindings of the HIP kernels (host-side plumbing only: shapes, buffers, streams).

Tensor convention: every activation is a 4-D tensor of logical shape [B, C, H, W] whose MEMORY is
NHWC (``channels_last``), possibly a channel slice of a wider buffer (pitch > C).  Conv weights are
logical [Cout, Cin, kh, kw] with KRSC memory.  Nothing here computes on the CPU or through ATen math
kernels; if libpylc_hip.so is missing, importing ``pylc_amd.lib`` already failed.

Layout of the package (one flat namespace: everything is re-exported here, `ops.conv2d`, `ops.bn_act`, ...):
    _core.py   tensor layout, fp16-plane tensors, range tags, streams / side stream, timing hooks, gradient links
    conv.py    Conv2dFn, conv2d, ConvTranspose2x2Fn, conv_bn_act_eval (+ the fp16-plane inference form)
    bn.py      BnActFn, GroupBnActFn, bn_act, the SyncBN message driver
    dw.py      DwConv3x3Fn, dwconv3x3, the inference half form
    misc.py    ReLU, dropout, max-pool, concat, bilinear, global average pool, image pack, MultiLoss
The few module-level SWITCHES that callers re-bind (`ops.PLANES_MIN_PIXELS = 0`, `ops.bn_timing = []`, `ops.mark_hook = fn`) live in
_core and are forwarded by properties of this module, so that an assignment here reaches the code that reads them."""
import sys
import types

from . import _core, conv, bn, dw, misc
from ._core import *       # noqa: F401,F403
from .conv import *        # noqa: F401,F403
from .bn import *          # noqa: F401,F403
from .dw import *          # noqa: F401,F403
from .misc import *        # noqa: F401,F403

_FORWARDED = ('PLANES_MIN_PIXELS', 'bn_timing', 'mark_hook', '_timer')


def _forward(name):
    return property(lambda self: getattr(_core, name), lambda self, value: setattr(_core, name, value))


class _OpsModule(types.ModuleType):
    pass


for _n in _FORWARDED:
    setattr(_OpsModule, _n, _forward(_n))
    globals().pop(_n, None)
sys.modules[__name__].__class__ = _OpsModule
