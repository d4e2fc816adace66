"""U-Net (valid 3x3 convs, depth 5, bilinear up-sampling path) on the HIP kernels.

Mirrors models/architectures/unet.py: UNet :19-104, UNetConvBlock :107-126, UNetUpBlock :129-155 with the
arguments models/model.py:140-147 passes (padding=False, up_mode='upsample', dropout=0.5)."""
import torch
from torch import nn

from .. import ops
from ..layers import Conv2d, BatchNorm2d, Dropout, Named, conv_bn


class UNetConvBlock(nn.Module):
    def __init__(self, cin, cout, padding, dropout):
        super().__init__()
        p = int(padding)
        self.block = Named(_0=Conv2d(cin, cout, 3, 1, p, bias=True, init='torch', bn=True), _1=BatchNorm2d(cout),
                           _3=Conv2d(cout, cout, 3, 1, p, bias=True, init='torch', bn=True), _4=BatchNorm2d(cout))
        self.drop = Dropout(dropout or 0.0)

    def forward(self, x, out_planes=False):
        """out_planes: the block's output has ONE reader, a conv that takes fp16 planes (the 1x1 conv of the next up block)."""
        b = self.block
        # (layers.conv_bn: in inference BatchNorm + ReLU run in the conv epilogue and the tensors between the convs are fp16 planes; in
        #  training it is bn(conv(x), ...) as written in the reference)
        x = conv_bn(b.child(0), b.child(1), x, relu=True, out_planes=b.child(3).takes_planes(), sole=True)      # feeds the second conv only
        # the block's nn.Dropout (unet.py:120) runs in the BatchNorm passes
        return conv_bn(b.child(3), b.child(4), x, relu=True, drop=self.drop, out_planes=out_planes, sole=out_planes)


class UpConv2x2(nn.Module):
    """nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2) (unet.py:132-133, up_mode='upconv'): every input pixel becomes a 2x2 output
    block, y[b, o, 2i+r, 2j+s] = sum_c x[b, c, i, j] W[c, o, r, s] + bias[o].  That is the data gradient of a 2x2 / stride-2 conv with
    the same weight tensor ([cin, cout, 2, 2] = that conv's [Cout, Cin, kh, kw]), so the forward runs pylc_conv2d_dgrad (one launch per
    output parity class), the input gradient pylc_conv2d_fwd and the weight gradient pylc_conv2d_wgrad (ops.ConvTranspose2x2Fn).
    state_dict keys as the reference's: up.weight [cin, cout, 2, 2], up.bias [cout]."""

    def __init__(self, cin, cout):
        super().__init__()
        bound = 1.0 / (cout * 4) ** 0.5             # torch's ConvTranspose2d default init: fan_in = weight.size(1) * kh * kw
        self.weight = nn.Parameter(torch.empty(cin, 2, 2, cout).uniform_(-bound, bound).permute(0, 3, 1, 2))     # KRSC memory
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        self.cout = cout

    def forward(self, x):
        return ops.conv_transpose2x2(x, self.weight, self.bias)


class UNetUpBlock(nn.Module):
    def __init__(self, cin, cout, padding, dropout, up_mode='upsample'):
        super().__init__()
        self.up_mode = up_mode
        if up_mode == 'upconv':
            self.up = UpConv2x2(cin, cout)
        else:
            self.up = Named(_1=Conv2d(cin, cout, 1, bias=True, init='torch'))
        self.conv_block = UNetConvBlock(cin, cout, padding, dropout)

    def forward(self, x, bridge, out_planes=False):
        if self.up_mode == 'upconv':
            # unet.py:145-152 with the transposed conv: its output and the centre crop of the bridge are concatenated (ops.cat_channels
            # carries the ranges; the slice write of the 'upsample' path needs a conv that writes into a buffer, which dgrad does not)
            up = self.up(x)
            th, tw = up.shape[2:]
            h0, w0 = (bridge.shape[2] - th) // 2, (bridge.shape[3] - tw) // 2
            return self.conv_block(ops.cat_channels((up, ops.as_nhwc(bridge)[:, :, h0:h0 + th, w0:w0 + tw])))
        # unet.py:145-152: cat([up, center_crop(bridge)], 1).  The 1x1 conv writes `up` straight into the concat buffer, the
        # crop is copied behind it, and the crop's gradient is summed by the max-pool backward of the same skip tensor
        # nn.Upsample(bilinear, align_corners) then Conv2d(k = 1) (unet.py:135-138): both are linear and the conv acts per pixel, so they
        # commute exactly (the interpolation weights of a pixel sum to 1, which carries the bias through): the 1x1 conv runs on the LOW
        # resolution tensor -- a quarter of the pixels -- and the interpolation on its half as many channels, written straight into the
        # concat buffer.  Same function, same parameters and gradients up to fp32 rounding.
        b, _, h, w = x.shape
        conv = self.up.child(1)
        z = ops.bound_conv_output(conv(x), x, conv.weight, conv.bias)     # (range bound instead of a range pass: only the concat kernel reads z)
        link = ops.grad_link(bridge)
        cat = ops.upsample2_crop_concat(z, bridge, link)       # training: interpolation + crop written as the next conv's fp16-plane operand
        if cat is None:
            holder = [ops.empty_nhwc(b, conv.cout + bridge.shape[1], 2 * h, 2 * w, x.device)]
            cat = ops.crop_concat(ops.bilinear(z, 2 * h, 2 * w, into=(holder, 0)), bridge, holder, link)
        return self.conv_block(cat, out_planes=out_planes)


class UNet(nn.Module):
    def __init__(self, in_channels=1, n_classes=2, depth=5, wf=6, padding=False, up_mode='upsample', dropout=None,
                 activ_func=None, normalizer=None):
        super().__init__()
        if up_mode not in ('upsample', 'upconv'):
            raise ValueError("up_mode must be 'upsample' (the reference default, config.py:231) or 'upconv' (unet.py:132-138)")
        self.depth = depth
        self.in_channels = in_channels
        prev = in_channels
        self.encoder = nn.ModuleList()
        for i in range(depth):
            self.encoder.append(UNetConvBlock(prev, 2 ** (wf + i), padding, dropout))
            prev = 2 ** (wf + i)
        self.decoder = nn.ModuleList()
        for i in reversed(range(depth - 1)):
            self.decoder.append(UNetUpBlock(prev, 2 ** (wf + i), padding, dropout, up_mode))
            prev = 2 ** (wf + i)
        self.last = Conv2d(prev, n_classes, 1, bias=True, init='torch')

    def forward(self, x):
        c = x.shape[1]
        if c % 4 != 0:
            x = ops.pack_nchw(x, (c + 3) & ~3)
        skips = []
        # a block output read only by the next up block's 1x1 conv leaves its BatchNorm as fp16 planes (no range pass, no conversion)
        nxt = [up.up.child(1).takes_planes() if up.up_mode != 'upconv' else False for up in self.decoder]
        for i, down in enumerate(self.encoder):
            last = i == self.depth - 1
            x = down(x, out_planes=last and bool(nxt) and nxt[0])
            if not last:
                skips.append(x)
                x = ops.maxpool(x, 2, 2, 0, link=ops.grad_link(x), out_planes=self.encoder[i + 1].block.child(0).takes_planes())
        for i, up in enumerate(self.decoder):
            x = up(x, skips[-i - 1], out_planes=i + 1 < len(nxt) and nxt[i + 1])
        return self.last(x)
