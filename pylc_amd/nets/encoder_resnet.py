"""ResNet-101 encoder (output stride 16/8, multi-grid layer4) on the HIP kernels.

Mirrors models/backbone/resnet.py: Bottleneck :16-53, strides/dilations :60-69, _make_layer :88-103,
_make_MG_unit :105-122, forward :124-135, init :137-147.  Each conv -> BN -> ReLU triple is two launches
(implicit-GEMM conv, fused BN-apply+ReLU[+residual]) plus the BN statistics reduction."""
import torch
from torch import nn

from .. import ops
from ..layers import Conv2d, BatchNorm2d, Named, conv_bn


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, dilation, project):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bn=True)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, dilation, dilation, bn=True)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bn=True)
        self.bn3 = BatchNorm2d(planes * 4)
        if project:
            self.downsample = Named(_0=Conv2d(inplanes, planes * 4, 1, stride, bn=True), _1=BatchNorm2d(planes * 4))
        else:
            self.downsample = None
        # activation formats: a BatchNorm writes fp16 planes when the conv(s) reading its output copy operand tiles (layers.Conv2d.takes_planes);
        # out_planes (the block output: next block's conv1 / downsample / residual input, ASPP) is set by ResNet101
        self.pl1, self.pl2 = self.conv2.takes_planes(), self.conv3.takes_planes()
        self.out_planes = False

    def forward(self, x):
        # x's gradient has two producers: conv1's dgrad and either bn3's residual branch (identity blocks) or the downsample
        # conv's dgrad (projection blocks; the low-level features add the decoder's conv1).  The link makes them sum into one
        # buffer in their own epilogues instead of leaving several tensors for autograd to add
        link = ops.grad_link(x)
        out = conv_bn(self.conv1, self.bn1, x, relu=True, conv_link=link, out_planes=self.pl1, sole=True)       # read by conv2 only
        out = conv_bn(self.conv2, self.bn2, out, relu=True, out_planes=self.pl2, sole=True)                     # read by conv3 only
        if self.downsample is not None:
            res = conv_bn(self.downsample.child(0), self.downsample.child(1), x, conv_link=link)
            return self._tail(out, res, None)
        return self._tail(out, x, link)

    def _tail(self, out, res, link):
        # relu(bn3(conv3(.)) + residual): one BatchNorm pass in training, the conv epilogue alone in inference
        if link is None:
            return conv_bn(self.conv3, self.bn3, out, residual=res, relu=True, out_planes=self.out_planes)
        return self.bn3(self.conv3(out), residual=res, relu=True, res_link=link, out_planes=self.out_planes)


class ResNet101(nn.Module):
    LAYERS = ((64, 3), (128, 4), (256, 23))
    MULTI_GRID = (1, 2, 4)

    def __init__(self, output_stride=16):
        super().__init__()
        if output_stride == 16:
            strides, dils = (1, 2, 2, 1), (1, 1, 1, 2)
        elif output_stride == 8:
            strides, dils = (1, 2, 1, 1), (1, 1, 2, 4)
        else:
            raise NotImplementedError(output_stride)
        self.conv1 = Conv2d(3, 64, 7, 2, 3, bn=True)
        self.bn1 = BatchNorm2d(64)
        inpl = 64
        for li, (planes, n) in enumerate(self.LAYERS):
            blocks = []
            for b in range(n):
                s = strides[li] if b == 0 else 1
                blocks.append(Bottleneck(inpl, planes, s, dils[li], b == 0 and (s != 1 or inpl != planes * 4)))
                inpl = planes * 4
            setattr(self, 'layer%d' % (li + 1), nn.Sequential(*blocks))
        blocks = []
        for b, mg in enumerate(self.MULTI_GRID):
            s = strides[3] if b == 0 else 1
            blocks.append(Bottleneck(inpl, 512, s, mg * dils[3], b == 0 and (s != 1 or inpl != 2048)))
            inpl = 2048
        self.layer4 = nn.Sequential(*blocks)
        # block outputs: planes when the next block's first conv takes them; the last block feeds the ASPP convs (2048 -> 256)
        chain = [b for l in (self.layer1, self.layer2, self.layer3, self.layer4) for b in l]
        for cur, nxt in zip(chain, chain[1:]):
            cur.out_planes = nxt.conv1.takes_planes()
        chain[-1].out_planes = True

    def forward(self, x4, keep_planes=False):
        """x4: normalised image packed to 4 NHWC channels. Returns (features/16, low-level features/4).

        keep_planes: in training graphs both results are fp16-PLANE tensors (ops.is_planes: float32-typed, bytes = two fp16 planes),
        which only pylc_amd's own kernels can read.  DeepLab -- whose ASPP / decoder convs copy those planes as operand tiles -- opts in;
        any other caller gets ordinary fp32 NHWC tensors (one conversion pass each), so that a foreign op on the features cannot
        silently compute on reinterpreted bytes.  (Hooks on the INNER modules -- Bottleneck outputs -- still see the raw format: read
        them through ops.as_nhwc.)"""
        x = conv_bn(self.conv1, self.bn1, x4, relu=True)          # (inference: BatchNorm + ReLU in the stem conv's epilogue)
        # (inference: the pooled tensor is read by layer1's conv1 and downsample conv only -- written as their plane operand directly; training
        #  keeps the fp32 tensor, whose gradient the two dgrads accumulate)
        blk = self.layer1[0]
        pool_planes = (not torch.is_grad_enabled()) and blk.conv1.takes_planes() and (blk.downsample is None or blk.downsample.child(0).takes_planes())
        x = ops.maxpool(x, 3, 2, 1, out_planes=pool_planes)
        low = self.layer1(x)
        x = self.layer4(self.layer3(self.layer2(low)))
        if not keep_planes:
            return ops.export_activation(x), ops.export_activation(low)
        return x, low
