"""U-Net (valid 3x3 convs, depth 5, bilinear up-sampling path) on the HIP kernels.

Mirrors models/architectures/unet.py: UNet :19-104, UNetConvBlock :107-126, UNetUpBlock :129-155 with the
arguments models/model.py:140-147 passes (padding=False, up_mode='upsample', dropout=0.5)."""
import torch
from torch import nn

from .. import ops
from ..layers import Conv2d, BatchNorm2d, Dropout, Named


class UNetConvBlock(nn.Module):
    def __init__(self, cin, cout, padding, dropout):
        super().__init__()
        p = int(padding)
        self.block = Named(_0=Conv2d(cin, cout, 3, 1, p, bias=True, init='torch', bn=True), _1=BatchNorm2d(cout),
                           _3=Conv2d(cout, cout, 3, 1, p, bias=True, init='torch', bn=True), _4=BatchNorm2d(cout))
        self.drop = Dropout(dropout or 0.0)

    def forward(self, x):
        b = self.block
        x = b.child(1)(b.child(0)(x), relu=True)
        x = b.child(4)(b.child(3)(x), relu=True)
        return self.drop(x)


class UNetUpBlock(nn.Module):
    def __init__(self, cin, cout, padding, dropout):
        super().__init__()
        self.up = Named(_1=Conv2d(cin, cout, 1, bias=True, init='torch'))
        self.conv_block = UNetConvBlock(cin, cout, padding, dropout)

    def forward(self, x, bridge):
        # unet.py:145-152: cat([up, center_crop(bridge)], 1).  The 1x1 conv writes `up` straight into the concat buffer, the
        # crop is copied behind it, and the crop's gradient is summed by the max-pool backward of the same skip tensor
        b, _, h, w = x.shape
        conv = self.up.child(1)
        holder = [ops.empty_nhwc(b, conv.cout + bridge.shape[1], 2 * h, 2 * w, x.device)]
        up = conv(ops.bilinear(x, 2 * h, 2 * w), out=holder)
        return self.conv_block(ops.crop_concat(up, bridge, holder, ops.grad_link(bridge)))


class UNet(nn.Module):
    def __init__(self, in_channels=1, n_classes=2, depth=5, wf=6, padding=False, up_mode='upsample', dropout=None,
                 activ_func=None, normalizer=None):
        super().__init__()
        if up_mode != 'upsample':
            raise ValueError("only up_mode='upsample' (the reference default, config.py:231) is built")
        self.depth = depth
        self.in_channels = in_channels
        prev = in_channels
        self.encoder = nn.ModuleList()
        for i in range(depth):
            self.encoder.append(UNetConvBlock(prev, 2 ** (wf + i), padding, dropout))
            prev = 2 ** (wf + i)
        self.decoder = nn.ModuleList()
        for i in reversed(range(depth - 1)):
            self.decoder.append(UNetUpBlock(prev, 2 ** (wf + i), padding, dropout))
            prev = 2 ** (wf + i)
        self.last = Conv2d(prev, n_classes, 1, bias=True, init='torch')

    def forward(self, x):
        c = x.shape[1]
        if c % 4 != 0:
            x = ops.pack_nchw(x, (c + 3) & ~3)
        skips = []
        for i, down in enumerate(self.encoder):
            x = down(x)
            if i != self.depth - 1:
                skips.append(x)
                x = ops.maxpool(x, 2, 2, 0, link=ops.grad_link(x))
        for i, up in enumerate(self.decoder):
            x = up(x, skips[-i - 1])
        return self.last(x)
