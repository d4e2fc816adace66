"""Modified Aligned Xception encoder on the HIP kernels.

Mirrors models/backbone/xception.py: fixed_padding :16-22, SeparableConv2d :25-39, Block :42-99,
AlignedXception :102-239.  The reference's in-place-ReLU aliasing (Block.relu is ReLU(inplace=True) and is
rep[0] when start_with_relu, so the skip branch reads ReLU(inp): SURVEY.md appendix D.1) is reproduced
explicitly: `inp` is replaced by relu(inp) before BOTH branches."""
import os

import torch
from torch import nn

from .. import ops
from ..layers import Conv2d, DepthwiseConv3x3, BatchNorm2d, conv_bn
from ..lib import lib, check, ptr, stream
from ..runtime import runtime


def _inference():
    return not torch.is_grad_enabled() and runtime.fuse_eval_bn


class SeparableConv2d(nn.Module):
    """depthwise 3x3 ('SAME' padding folded in) -> BN -> pointwise 1x1 (xception.py:34-39)."""

    def __init__(self, cin, cout, stride=1, dilation=1):
        super().__init__()
        self.conv1 = DepthwiseConv3x3(cin, stride, dilation)
        self.bn = BatchNorm2d(cin)
        self.pointwise = Conv2d(cin, cout, 1, bn=True)         # every SeparableConv2d is followed by a BatchNorm (Block / bn3-5)

    def forward(self, x, link=None):
        # the inner BatchNorm feeds the pointwise conv only: it writes the conv's operand format (fp16 planes) directly
        return self.pointwise(self.bn(self.conv1(x, res_link=link), out_planes=self.pointwise.takes_planes(), sole=True))

    def train(self, mode=True):
        if mode:
            self._fold_key = None           # running statistics / weights are about to move: refold at the next inference forward
        return super().train(mode)

    def _folded(self):
        """Inference: the eval-mode inner BatchNorm is a per-channel affine in front of the pointwise conv -- W (s (.) x + t) = (W diag s) x
        + W t -- so it is folded into that conv's filter and a bias (pylc_conv1x1_fold_input_affine), once per set of weights."""
        bn, w = self.bn, self.pointwise.weight
        arena = getattr(w, '_pylc_arena', None)
        arena = arena() if arena is not None else None
        # (the HIP optimiser and BatchNorm kernels write parameters and running statistics through raw pointers, so tensor versions alone do
        # not see a training step: the arena's generation counts optimiser steps, and train() below drops the cache whenever the module
        # re-enters training mode -- the only mode that moves the running statistics)
        key = (w._version, w.data_ptr(), bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               arena.generation if arena is not None else None)
        if getattr(self, '_fold_key', None) != key:
            cout, cin = w.shape[:2]
            coef = torch.empty(2 * cin, device=w.device)
            check(lib.pylc_bn_eval_coeffs(ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.weight), ptr(bn.bias), bn.eps, cin,
                                          ptr(coef[:cin]), ptr(coef[cin:]), stream()))
            w2 = torch.empty_like(w)                   # same KRSC memory ([Cout][Cin] for a 1x1 filter)
            b2 = torch.empty(cout, device=w.device)
            amax = torch.empty(1, dtype=torch.int32, device=w.device)
            check(lib.pylc_conv1x1_fold_input_affine(ptr(w), ptr(coef[:cin]), ptr(coef[cin:]), None, cout, cin, ptr(w2), ptr(b2), ptr(amax), stream()))
            w2._pylc_wamax = amax                      # the filter range ops.weight_amax looks up
            if ops.ranges_needed() and runtime.fold_planes:
                # the folded filter's fp16 planes, as FlatArena prepares them for the arena's filters: the conv kernel then copies
                # filter tiles instead of splitting fp32 values at every reduction step (pylc_weight_prepare, one table entry)
                import ctypes as C
                from .. import lib as L
                kp = (cout + 3) & ~3
                n_fwd, n_t = 2 * cout * cin, 2 * cin * kp
                planes = torch.zeros((n_fwd + n_t + 7) & ~7, dtype=torch.float16, device=w.device)
                entry = L.WPrepEntry(0, 0, n_fwd, 0, cout, 1, cin, 0)
                table = torch.frombuffer(bytearray(bytes(entry)), dtype=torch.uint8).clone().to(w.device)
                tiles = ((cout + 31) // 32) * ((cin + 31) // 32)
                check(lib.pylc_weight_prepare(ptr(w2), ptr(table), 1, tiles, ptr(amax), ptr(planes), 0, stream()))      # (separate plane arrays)
                w2._pylc_planes = (planes[:n_fwd], planes[n_fwd:n_fwd + n_t])
                self._fold_table = table               # (read by the launch above: kept until the cache is dropped)
            self._fold, self._fold_key = (w2, b2), key
        return self._fold

    def fused_eval(self, x, bn_out, relu=False, residual=None):
        """Inference: act(bn_out(pointwise(bn(depthwise(x)))) (+ residual)) in two kernels -- the depthwise conv, and the pointwise conv with the
        inner BatchNorm folded into its filter and the outer BatchNorm, the residual add and the ReLU in its epilogue (ops.conv_bn_act_eval)
        -- instead of four passes."""
        w2, b2 = self._folded()
        if ops.ranges_needed() and runtime.eval_planes and not runtime.no_planes:
            # precision mode 3: the whole group on one-plane fp16 tensors -- depthwise half -> half, pointwise conv planes -> planes with the
            # outer BatchNorm / residual / ReLU in its epilogue (ops.conv_bn_act_eval_planes)
            yh = ops.dwconv3x3_eval_half(x, self.conv1.weight, self.conv1.stride, self.conv1.dilation)
            if yh is not None:
                return ops.conv_bn_act_eval(yh, w2, b2, 1, 0, 1, bn_out.running_mean, bn_out.running_var, bn_out.weight, bn_out.bias,
                                            bn_out.eps, residual, relu, coef=bn_out.eval_coeffs(), coef_ranges=bn_out.eval_coeff_ranges(),
                                            out_planes=True)
        y = self.conv1(x)
        if ops.ranges_needed():
            # the pointwise conv's arithmetic needs a bound of |y|: 9 max|w_dw| max|x| from the two operand ranges instead of a pass over y
            bound = torch.empty(1, dtype=torch.int32, device=y.device)
            check(lib.pylc_range_product(ptr(ops.amax_of(x)), ptr(ops.weight_amax(self.conv1.weight)), 9.0, None, ptr(bound), stream()))
            ops.tag_amax(y, bound)
        return ops.conv_bn_act_eval(y, w2, b2, 1, 0, 1, bn_out.running_mean, bn_out.running_var, bn_out.weight, bn_out.bias,
                                    bn_out.eps, residual, relu, coef=bn_out.eval_coeffs())


class Block(nn.Module):
    def __init__(self, inpl, planes, reps, stride=1, dilation=1, start_with_relu=True, grow_first=True, is_last=False):
        super().__init__()
        if planes != inpl or stride != 1:
            self.skip = Conv2d(inpl, planes, 1, stride, bn=True)
            self.skipbn = BatchNorm2d(planes)
        else:
            self.skip = None
        seq, filters = [], inpl
        if grow_first:
            seq += ['relu', SeparableConv2d(inpl, planes, 1, dilation), BatchNorm2d(planes)]
            filters = planes
        for _ in range(reps - 1):
            seq += ['relu', SeparableConv2d(filters, filters, 1, dilation), BatchNorm2d(filters)]
        if not grow_first:
            seq += ['relu', SeparableConv2d(inpl, planes, 1, dilation), BatchNorm2d(planes)]
        if stride != 1:
            seq += ['relu', SeparableConv2d(planes, planes, 2, 1), BatchNorm2d(planes)]
        if stride == 1 and is_last:
            seq += ['relu', SeparableConv2d(planes, planes, 1, 1), BatchNorm2d(planes)]
        if not start_with_relu:
            seq = seq[1:]
        self.start_with_relu = start_with_relu
        self.rep = nn.Module()                    # children named by their index in the reference's nn.Sequential
        self.plan = []
        for i, item in enumerate(seq):
            if item == 'relu':
                self.plan.append(('relu', None))
            else:
                self.rep.add_module(str(i), item)
                self.plan.append(('bn' if isinstance(item, BatchNorm2d) else 'sep', str(i)))

    def forward(self, inp, input_relud=False, relu_out=False):
        """input_relud: the caller already applied this block's leading ReLU (fused into the pass that produced `inp`: because of the
        aliasing quirk below nothing ever reads the un-rectified tensor).  relu_out: apply the NEXT block's leading ReLU (or the
        explicit one of xception.py:200 / :222) in this block's last pass."""
        if self.start_with_relu and not input_relud:
            inp = ops.relu(inp)                   # aliasing quirk: both branches see relu(inp)
        if not self.training and _inference():
            return self._forward_inference(inp, relu_out)
        # `inp` has two consumers -- the first depthwise conv of `rep` and the skip path (the 1x1 skip conv, or the residual input of the
        # last BatchNorm): their gradients meet in one buffer instead of an autograd add pass (ops.ResidualLink)
        link = ops.grad_link(inp)
        if self.skip is not None:
            skip = self.skipbn(self.skip(inp, res_link=link))
        else:
            skip = inp
        x = inp
        n = len(self.plan)
        i = 1 if self.start_with_relu else 0
        first_sep = True
        while i < n:
            kind, name = self.plan[i]
            if kind == 'sep':
                x = getattr(self.rep, name)(x, link if first_sep else None)
                first_sep = False
            elif kind == 'bn':
                # precision mode 3 with half activations (ops.half_acts): a BatchNorm whose output a depthwise conv reads writes ONE fp16 plane
                # -- what the stride-1 / dilation-1 depthwise kernels read -- instead of fp32
                half = ops.half_dw()
                if i == n - 1:          # the branch ends in a BatchNorm: `rep(inp) + skip` (xception.py:97) is its apply pass
                    return getattr(self.rep, name)(x, residual=skip, relu=relu_out,   # (y*scale + shift) + skip: the same two fp32 operations
                                                   res_link=link if self.skip is None else None, out_planes=half)
                fuse = self.plan[i + 1][0] == 'relu'                   # BN followed by the shared ReLU -> one pass
                # (its only consumer is the depthwise conv of the next separable conv: `sole`)
                # ... which can also apply this BatchNorm itself: no apply pass, no output tensor (ops.bn_act(defer=))
                x = getattr(self.rep, name)(x, relu=fuse, out_planes=half, sole=True, defer=half)
                if fuse:
                    i += 1
            else:
                x = ops.relu(x)
            i += 1
        return ops.relu(x + skip) if relu_out else x + skip


    def _forward_inference(self, inp, relu_out):
        """Eval mode without autograd: every (separable conv, BatchNorm[, ReLU]) group of the plan is SeparableConv2d.fused_eval; the last
        one takes the skip branch as the residual of its epilogue."""
        skip = inp if self.skip is None else conv_bn(self.skip, self.skipbn, inp)
        x, n = inp, len(self.plan)
        i = 1 if self.start_with_relu else 0
        while i < n:
            kind, name = self.plan[i]
            if kind == 'relu':
                x = ops.relu(x)
                i += 1
                continue
            assert kind == 'sep' and self.plan[i + 1][0] == 'bn'
            sep, bn = getattr(self.rep, name), getattr(self.rep, self.plan[i + 1][1])
            if i + 1 == n - 1:                    # rep(inp) + skip (xception.py:97), then the next block's leading ReLU
                return sep.fused_eval(x, bn, relu=relu_out, residual=skip)
            fuse = self.plan[i + 2][0] == 'relu'
            x = sep.fused_eval(x, bn, relu=fuse)
            i += 3 if fuse else 2
        raise AssertionError('an Xception block ends in a BatchNorm')


class AlignedXception(nn.Module):
    def __init__(self, output_stride=16):
        super().__init__()
        if output_stride == 16:
            b3s, mid_d, exit_d = 2, 1, (1, 2)
        elif output_stride == 8:
            b3s, mid_d, exit_d = 1, 2, (2, 4)
        else:
            raise NotImplementedError(output_stride)
        self.conv1 = Conv2d(3, 32, 3, 2, 1, bn=True)
        self.bn1 = BatchNorm2d(32)
        self.conv2 = Conv2d(32, 64, 3, 1, 1, bn=True)
        self.bn2 = BatchNorm2d(64)
        self.block1 = Block(64, 128, 2, 2, 1, start_with_relu=False)
        self.block2 = Block(128, 256, 2, 2, 1, start_with_relu=False)
        self.block3 = Block(256, 728, 2, b3s, 1, is_last=True)
        for i in range(4, 20):
            setattr(self, 'block%d' % i, Block(728, 728, 3, 1, mid_d))
        self.block20 = Block(728, 1024, 2, 1, exit_d[0], grow_first=False, is_last=True)
        self.conv3 = SeparableConv2d(1024, 1536, 1, exit_d[1])
        self.bn3 = BatchNorm2d(1536)
        self.conv4 = SeparableConv2d(1536, 1536, 1, exit_d[1])
        self.bn4 = BatchNorm2d(1536)
        self.conv5 = SeparableConv2d(1536, 2048, 1, exit_d[1])
        self.bn5 = BatchNorm2d(2048)

    def forward(self, x4, keep_planes=False):
        """keep_planes (as ResNet101.forward): the caller's convs read fp16 planes, so the last BatchNorm writes them; any other caller gets
        fp32 tensors."""
        x = self.bn1(self.conv1(x4), relu=True, out_planes=self.conv2.takes_planes() and runtime.gap_planes, sole=True)       # read by conv2 only
        x = self.bn2(self.conv2(x), relu=True, out_planes=ops.half_dw())       # read by block1's first depthwise conv and its skip conv
        # every block output is read through a ReLU only (xception.py:200 explicitly; blocks 3..20 through the in-place ReLU that
        # leads their `rep` and aliases the skip input, :53-97; the exit flow's :222), so the ReLU runs in the producing BatchNorm pass
        low = self.block1(x, relu_out=True)       # xception.py:199-202
        x = self.block2(low, relu_out=True)
        for i in range(3, 21):
            x = getattr(self, 'block%d' % i)(x, input_relud=True, relu_out=True)
        if not self.training and _inference():
            x = self.conv3.fused_eval(x, self.bn3, relu=True)
            x = self.conv4.fused_eval(x, self.bn4, relu=True)
            return self.conv5.fused_eval(x, self.bn5, relu=True), low
        x = self.bn3(self.conv3(x), relu=True, out_planes=ops.half_dw(), sole=True, defer=True)       # read by the next separable conv's depthwise kernel only
        x = self.bn4(self.conv4(x), relu=True, out_planes=ops.half_dw(), sole=True, defer=True)
        x = self.bn5(self.conv5(x), relu=True, out_planes=keep_planes and runtime.gap_planes)       # the ASPP's convs and its image pool (ops.GapFn) read planes
        if not keep_planes:
            return ops.export_activation(x), ops.export_activation(low)
        return x, low
