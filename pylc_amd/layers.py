"""Layer modules of the HIP path: parameter containers with the reference's state_dict names/shapes whose
forward is a HIP kernel launch (pylc_amd.ops).  Replaces nn.Conv2d / nn.BatchNorm2d / nn.ReLU / nn.Dropout
as used by models/backbone/*.py, models/modules/aspp.py, models/decoder.py, models/architectures/unet.py."""
import math

import torch
from torch import nn

from . import ops
from .runtime import runtime


class Conv2d(nn.Module):
    """Dense conv; weight is logical [Cout, Cin, k, k] with KRSC memory (what the MFMA kernel streams)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, bias=False, init='resnet', bn=False):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.feeds_bn = bn and cout % 4 == 0      # a BatchNorm consumes the output: emit its statistics from the conv epilogue
        self.stride, self.padding, self.dilation = stride, padding, dilation
        w = torch.empty(cout, k, k, cin)
        fan_in = cin * k * k
        if init == 'resnet':        # N(0, sqrt(2/(k*k*cout))): resnet.py:139-141, xception.py:243-245
            w.normal_(0, math.sqrt(2.0 / (k * k * cout)))
        elif init == 'kaiming':     # kaiming_normal_ fan_in, gain sqrt(2): aspp.py:34,93, decoder.py:55
            w.normal_(0, math.sqrt(2.0 / fan_in))
        else:                       # torch Conv2d default (kaiming_uniform_(a=sqrt(5))): U-Net never calls initialize(), unet.py:81
            w.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        self.weight = nn.Parameter(w.permute(0, 3, 1, 2))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in)))
        else:
            self.register_parameter('bias', None)

    def forward(self, x, res_link=None, out=None):
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.feeds_bn and self.training, res_link, out)

    def takes_planes(self):
        """Static part of ops.conv_takes_planes: a BatchNorm that feeds ONLY convs for which this holds may write fp16 planes."""
        return self.cout % 4 == 0 and self.cin % 8 == 0

    def extra_repr(self):
        return '%d, %d, k=%d, s=%d, p=%d, d=%d%s' % (self.cin, self.cout, self.k, self.stride, self.padding, self.dilation,
                                                      '' if self.bias is None else ', bias')


class DepthwiseConv3x3(nn.Module):
    """groups=C 3x3 conv with xception.py's fixed_padding folded in; weight [C,1,3,3]."""

    def __init__(self, c, stride=1, dilation=1):
        super().__init__()
        self.c, self.stride, self.dilation = c, stride, dilation
        self.weight = nn.Parameter(torch.empty(c, 1, 3, 3).normal_(0, math.sqrt(2.0 / (9 * c))))

    def forward(self, x, res_link=None):
        return ops.dwconv3x3(x, self.weight, self.stride, self.dilation, res_link, want_stats=self.training)     # always followed by a BatchNorm


class BatchNorm2d(nn.Module):
    """BatchNorm2d with the following ReLU / residual add fused into the same pass.  Synchronises its statistics
    over runtime.sync_group when one is set (the RCCL replacement of models/sync_batchnorm)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.zeros((), dtype=torch.long))
        self._nbt_pending = 0       # training forwards not yet added to the buffer (flushed when the state is read)

    @classmethod
    def evaluate(cls, c):          # the reference's U-Net calls normalizer.evaluate(out_size) (unet.py:113,117)
        return cls(c)

    def forward(self, y, residual=None, relu=False, res_link=None, out_planes=False, drop=None, into=None, sole=False, defer=False):
        """out_planes: every consumer of the output is a conv with takes_planes() (or a BatchNorm residual input) -- write fp16 planes.
        drop: the Dropout module that follows the activation in the reference, fused into this pass."""
        if self.training:
            self._nbt_pending += 1          # no per-layer device add: 113 tiny launches per step otherwise
        return ops.bn_act(y, self.weight, self.bias, self.running_mean, self.running_var, residual, relu,
                          self.training, self.eps, self.momentum, runtime.sync_group if (self.training and runtime.sync_bn) else None,
                          runtime.bn_clamp_eps, res_link, out_planes, drop.spec() if drop is not None else None, into, sole, defer)

    def train(self, mode=True):
        if mode:
            self._coef_key = None            # the running statistics are about to move
        return super().train(mode)

    def eval_coeffs(self):
        """[scale | shift] of the eval-mode affine (pylc_bn_eval_coeffs), computed once per set of weights for the fused inference kernels:
        the cache is dropped when the module re-enters training mode and keyed on the tensors' versions and the flat arena's generation
        (the HIP kernels write parameters and running statistics through raw pointers)."""
        from .lib import lib, check, ptr, stream
        arena = getattr(self.weight, '_pylc_arena', None)
        arena = arena() if arena is not None else None
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version, self.weight.data_ptr(),
               arena.generation if arena is not None else None)
        if getattr(self, '_coef_key', None) != key:
            c = self.num_features
            coef = torch.empty(2 * c, device=self.weight.device)
            check(lib.pylc_bn_eval_coeffs(ptr(self.running_mean), ptr(self.running_var), ptr(self.weight), ptr(self.bias), self.eps, c,
                                          ptr(coef[:c]), ptr(coef[c:]), stream()))
            self._coef, self._coef_key = coef, key
        return self._coef

    def eval_coeff_ranges(self):
        """int32[2]: float bits of max|scale|, max|shift| of eval_coeffs() -- what the fused inference conv needs to bound a plane output
        (ops.conv_bn_act_eval_planes); cached with the coefficients."""
        coef = self.eval_coeffs()
        if getattr(self, '_coef_rng_key', None) is not self._coef_key or getattr(self, '_coef_rng', None) is None:
            from .lib import lib, check, ptr, stream
            c = self.num_features
            rng = torch.zeros(2, dtype=torch.int32, device=coef.device)
            check(lib.pylc_amax(ptr(coef[:c]), 1, c, c, ptr(rng[0:1]), stream()))
            check(lib.pylc_amax(ptr(coef[c:]), 1, c, c, ptr(rng[1:2]), stream()))
            self._coef_rng, self._coef_rng_key = rng, self._coef_key
        return self._coef_rng

    def flush_counter(self):
        if self._nbt_pending:
            self.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        super()._load_from_state_dict(*args, **kwargs)

    def extra_repr(self):
        return '%d' % self.num_features


def conv_bn(conv, bn, x, residual=None, relu=False, conv_link=None, out_planes=False, drop=None, into=None, sole=False):
    """[drop](bn(conv(x), residual, relu)).  In inference (eval mode, autograd off) the BatchNorm, the residual add and the ReLU run
    inside the conv epilogue (ops.conv_bn_act_eval); otherwise the two modules are called as usual (the conv output has ONE consumer,
    the BatchNorm: that is what lets its backward hand dy back as fp16 planes)."""
    thin = conv.cin % 4 != 0 and not ops.is_planes(x) and x.shape[1] == ((conv.cin + 3) & ~3)          # the stem on the 4-channel input pack
    if (not bn.training and not torch.is_grad_enabled() and runtime.fuse_eval_bn and (conv.cin % 4 == 0 or thin)
            and (residual is None or ops.is_planes(residual) or ops.pitch_of(ops.as_nhwc(residual)) == ((conv.cout + 3) & ~3))):
        return ops.conv_bn_act_eval(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, bn.running_mean,
                                    bn.running_var, bn.weight, bn.bias, bn.eps, residual, relu, into, coef=bn.eval_coeffs(),
                                    coef_ranges=bn.eval_coeff_ranges(), out_planes=out_planes)
    # sole: the caller states that the result has exactly ONE consumer, a conv -- whose dgrad may then take this BatchNorm's backward sums
    return bn(conv(x, res_link=conv_link), residual=residual, relu=relu, out_planes=out_planes, drop=drop, into=into, sole=sole)


def bn_group(entries):
    """The BatchNorms of PARALLEL branches as one autograd node (ops.GroupBnActFn): under SyncBN their statistics share one all-reduce per
    direction.  entries: [(BatchNorm2d module, its input y, keyword arguments of BatchNorm2d.forward), ...]; returns the outputs in order.
    Per layer the kernels and their order are exactly those of BatchNorm2d.forward."""
    specs = []
    for bn, y, kw in entries:
        if bn.training:
            bn._nbt_pending += 1
        drop = kw.get('drop')
        specs.append(dict(y=y, gamma=bn.weight, beta=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var, training=bn.training,
                          eps=bn.eps, momentum=bn.momentum, clamp_eps=runtime.bn_clamp_eps, residual=kw.get('residual'), relu=kw.get('relu', False),
                          res_link=kw.get('res_link'), out_planes=kw.get('out_planes', False), drop=drop.spec() if drop is not None else None,
                          into=kw.get('into'), sole=kw.get('sole', False)))
    training = entries[0][0].training
    return ops.bn_act_group(specs, runtime.sync_group if (training and runtime.sync_bn) else None)


class Dropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.training and runtime.dropout_enabled and self.p > 0:
            return ops.dropout(x, self.p, runtime.next_seed())
        return x

    def spec(self):
        """(p, seed) for a BatchNorm pass that applies this dropout itself (BatchNorm2d.forward(drop=...)), None when inactive."""
        if self.training and runtime.dropout_enabled and self.p > 0:
            return (self.p, runtime.next_seed())
        return None


class Named(nn.Module):
    """A container whose children carry explicit (numeric-string) names, to reproduce the reference's
    nn.Sequential state_dict keys (e.g. decoder.last_conv.4.weight) without its parameter-free members."""

    def __init__(self, **children):
        super().__init__()
        for k, m in children.items():
            self.add_module(k.lstrip('_'), m)

    def child(self, name):
        return self._modules[str(name)]
