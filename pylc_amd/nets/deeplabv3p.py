"""DeepLabV3+ (ASPP + decoder) assembly on the HIP kernels.

Mirrors models/architectures/deeplab.py:17-39, models/modules/aspp.py:15-100, models/decoder.py:15-62.
Constructor signature and state_dict keys are the reference's, so models/model.py:165-174 can build it and
reference checkpoints load unchanged."""
import torch
from torch import nn

from .. import ops
from ..layers import Conv2d, BatchNorm2d, Dropout, Named, conv_bn, bn_group
from ..runtime import runtime
from .encoder_resnet import ResNet101
from .encoder_xception import AlignedXception


class _ASPPBranch(nn.Module):
    def __init__(self, cin, cout, k, dilation):
        super().__init__()
        self.atrous_conv = Conv2d(cin, cout, k, 1, 0 if k == 1 else dilation, dilation, init='kaiming', bn=True)
        self.bn = BatchNorm2d(cout)

    def forward(self, x, link=None, into=None):
        return conv_bn(self.atrous_conv, self.bn, x, relu=True, conv_link=link, into=into)


class ASPP(nn.Module):
    def __init__(self, output_stride=16, inplanes=2048):
        super().__init__()
        dil = (1, 6, 12, 18) if output_stride == 16 else (1, 12, 24, 36)
        self.aspp1 = _ASPPBranch(inplanes, 256, 1, dil[0])
        self.aspp2 = _ASPPBranch(inplanes, 256, 3, dil[1])
        self.aspp3 = _ASPPBranch(inplanes, 256, 3, dil[2])
        self.aspp4 = _ASPPBranch(inplanes, 256, 3, dil[3])
        self.global_avg_pool = Named(_1=Conv2d(inplanes, 256, 1, init='kaiming', bn=True), _2=BatchNorm2d(256))
        self.conv1 = Conv2d(1280, 256, 1, init='kaiming', bn=True)
        self.bn1 = BatchNorm2d(256)
        self.dropout = Dropout(0.5)

    def forward(self, x):
        h, w = x.shape[2:]
        link = ops.grad_link(x)                  # the four branch dgrads and the pooling branch's gradient sum into one buffer (ops.ResidualLink)
        g = ops.global_avg_pool(x, link)
        # torch.cat (aspp.py:80) by slice: every branch's BatchNorm pass -- in inference the fused conv epilogue -- and the image-pool branch's
        # interpolation write their 256 channels straight into the 1280-channel buffer the projection conv reads
        buf = [ops.empty_nhwc(x.shape[0], 1280, h, w, x.device)]
        if self.training and torch.is_grad_enabled() and runtime.sync_group is not None and runtime.sync_bn and runtime.coalesce_sync_bn:
            # SyncBN: the five branches are parallel, so their BatchNorms' statistics travel in ONE all-reduce per direction (10 -> 2
            # collectives per step): all five convs first, then the BatchNorms as one node (layers.bn_group)
            pool_conv, pool_bn = self.global_avg_pool.child(1), self.global_avg_pool.child(2)
            brs = (self.aspp1, self.aspp2, self.aspp3, self.aspp4)
            ys = [br.atrous_conv(x, res_link=link) for br in brs]
            outs = bn_group([(br.bn, y, dict(relu=True, into=(buf, 256 * i))) for i, (br, y) in enumerate(zip(brs, ys))] +
                            [(pool_bn, pool_conv(g), dict(relu=True))])
            branches = outs[:4] + [ops.bilinear(outs[4], h, w, into=(buf, 1024))]
        else:
            g = conv_bn(self.global_avg_pool.child(1), self.global_avg_pool.child(2), g, relu=True)
            branches = [self.aspp1(x, link, (buf, 0)), self.aspp2(x, link, (buf, 256)), self.aspp3(x, link, (buf, 512)),
                        self.aspp4(x, link, (buf, 768)), ops.bilinear(g, h, w, into=(buf, 1024))]
        y = ops.concat_slices(buf, branches)
        return conv_bn(self.conv1, self.bn1, y, relu=True, drop=self.dropout)      # dropout fused into the BatchNorm passes


class Decoder(nn.Module):
    def __init__(self, n_classes, low_level_inplanes):
        super().__init__()
        self.conv1 = Conv2d(low_level_inplanes, 48, 1, init='kaiming', bn=True)
        self.bn1 = BatchNorm2d(48)
        self.last_conv = Named(_0=Conv2d(304, 256, 3, 1, 1, init='kaiming', bn=True), _1=BatchNorm2d(256),
                               _4=Conv2d(256, 256, 3, 1, 1, init='kaiming', bn=True), _5=BatchNorm2d(256),
                               _8=Conv2d(256, n_classes, 1, bias=True, init='kaiming'))
        self.drop3, self.drop7 = Dropout(0.5), Dropout(0.1)
        # Launch ORDER of the backward (ops.Conv2dFn.backward, runtime.wgrad_hold): the 304->256 conv's wgrad fills every CU for ~2.9 ms on
        # the side queue; launched right after its dgrad it sits in front of the next kernels of the main queue -- the 48-channel BatchNorm
        # backward and the 256->48 dgrad of conv1 (decoder.py:27), 0.17 ms of work that then takes 2.1 ms.  runtime.decoder_hold = (2, 0) holds that
        # wgrad until the conv backward after conv1's begins.  MEASURED (round 4, profiles/r04_wgrad_hold_ab.txt): the 256->48 row drops to
        # its isolated time, the wait moves to the next short kernel (the ASPP's 1280->256 dgrad: 0.10 -> 1.90 ms) and the step is 0.3 %
        # SLOWER (403.9 / 404.2 vs 405.1 / 405.9 tiles/s): the GPU is never idle during that wait -- the wgrad is work the step has to do
        # anyway -- so the order of the queue does not change the sum.  Default: no hold.
        import os
        self.last_conv.child(0).weight._pylc_wgrad_hold = int(runtime.decoder_hold[0])
        self.last_conv.child(4).weight._pylc_wgrad_hold = int(runtime.decoder_hold[1])

    def forward(self, x, low):
        # torch.cat (decoder.py:47) by slice: the up-sampled ASPP output and the reduced low-level features are written into the two channel
        # ranges of one buffer (training: by the BatchNorm pass; inference: by the fused conv epilogue)
        b, _, lh, lw = low.shape
        cx, cl = x.shape[1], self.conv1.cout
        buf = [ops.empty_nhwc(b, cx + cl, lh, lw, x.device)]
        up = ops.bilinear(x, lh, lw, into=(buf, 0))
        low = conv_bn(self.conv1, self.bn1, low, relu=True, conv_link=ops.grad_link(low), into=(buf, cx))
        x = ops.concat_slices(buf, (up, low))
        lc = self.last_conv
        x = conv_bn(lc.child(0), lc.child(1), x, relu=True, out_planes=lc.child(4).takes_planes(), drop=self.drop3, sole=True)
        x = conv_bn(lc.child(4), lc.child(5), x, relu=True, drop=self.drop7)
        return lc.child(8)(x)


class DeepLab(nn.Module):
    """DeepLab(activ_func=, normalizer=, backbone=, output_stride=16, n_classes=, in_channels=, freeze_bn=, pretrained=)
    -- the call made at models/model.py:166-173.  activ_func / normalizer are accepted for signature
    compatibility; the HIP path always runs ReLU and (Sync)BatchNorm, which is also all the reference can select
    (SURVEY.md section 5 'Config / flag system')."""

    def __init__(self, activ_func=None, normalizer=None, backbone='resnet', output_stride=16, n_classes=9, in_channels=3,
                 freeze_bn=False, pretrained=False):
        super().__init__()
        if backbone == 'resnet':
            self.backbone = ResNet101(output_stride)
            low_c = 256
        elif backbone == 'xception':
            self.backbone = AlignedXception(output_stride)
            low_c = 128
        else:
            raise ValueError('backbone %r not available (resnet | xception)' % backbone)
        self.aspp = ASPP(output_stride)
        self.decoder = Decoder(n_classes, low_c)
        self.in_channels = in_channels
        self.n_classes = n_classes
        if pretrained:
            self.load_pretrained_backbone()

    PRETRAINED_PATH = './data/models/resnet101-5d3b4d8f.pth'       # config.py:188 `defaults.pretrained`

    def load_pretrained_backbone(self, path=None):
        """resnet.py:149-158 (`_load_pretrained_model`): read the torchvision ResNet-101 ImageNet state dict and copy every entry whose key
        the backbone also has; entries it does not have (fc.*) are ignored, backbone tensors the file lacks keep their initialisation.
        Like the reference, a missing file is an error (torch.load raises).  Only the ResNet backbone has a pretrained path (the
        Xception branch, xception.py:247-283, points at a URL the reference never downloads)."""
        import torch
        if not isinstance(self.backbone, ResNet101):
            raise ValueError('pretrained weights exist for the resnet backbone only')
        path = path or self.PRETRAINED_PATH
        pretrain = torch.load(path, map_location='cpu', weights_only=True)
        own = self.backbone.state_dict()
        picked = {k: v for k, v in pretrain.items() if k in own}
        own.update(picked)
        self.backbone.load_state_dict(own)
        return sorted(picked)

    def forward(self, x):
        """x: normalised fp32 [B,3,H,W] (any layout), or the 4-channel NHWC pack made by ops.image_pack.
        Returns logits [B,n_classes,H,W] (NHWC memory)."""
        h, w = x.shape[2:]
        x4 = x if x.shape[1] == 4 else ops.pack_nchw(x, 4)
        f, low = self.backbone(x4, keep_planes=True)
        y = self.decoder(self.aspp(f), low)
        return ops.bilinear(y, h, w)
