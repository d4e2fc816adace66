"""CPU restatement of Model.train / Model.eval / Model.test (TEST INFRASTRUCTURE).

Follows models/model.py: normalize_image :416-445, train :282-336, eval :338-365,
test :367-382, init_optim :238-254 (AdamW lr 1e-4, wd 5e-5; config.py:193,205),
clip_grad_norm_(0.5) :326.
"""
import numpy as np
import torch

from .nets import deeplab_forward, unet_forward
from .loss import multiloss, ce_loss, dice_loss, focal_loss

PX_RGB_MEAN = (132.47, 144.47, 149.45)      # config.py:171
PX_RGB_STD = (24.85, 22.04, 18.77)          # config.py:172
UNET_CROP = 94                              # config.py:228-236 (crop_left/up = 94, right/down = 418 at 512 px)


class StepConfig:
    def __init__(self, arch='deeplab', backbone='resnet', n_classes=9, ch=3,
                 px_mean=PX_RGB_MEAN, px_std=PX_RGB_STD,
                 loss_weights=(0.5, 0.5, 0.5), class_weights=None, weighted=False,
                 lr=1e-4, weight_decay=5e-5, clip=0.5, dropout=True):
        self.arch, self.backbone, self.n_classes, self.ch = arch, backbone, n_classes, ch
        self.px_mean, self.px_std = tuple(px_mean), tuple(px_std)
        self.loss_weights = tuple(loss_weights)
        self.class_weights = None if class_weights is None else torch.as_tensor(class_weights, dtype=torch.float32)
        self.weighted = weighted
        self.lr, self.weight_decay, self.clip = lr, weight_decay, clip
        self.dropout = dropout


def normalize_image(img, px_mean, px_std):
    """model.py:434-445 (non-default branches): ((x - mean)/std)/255 per channel; grayscale uses
    the mean of px_mean / px_std."""
    if img.shape[1] == 1:
        mean = float(np.mean(np.asarray(px_mean, np.float32)))      # fp32 statistics: see make_golden shim 6
        std = float(np.mean(np.asarray(px_std, np.float32)))
        return (img - mean) / std / 255
    m = torch.tensor(px_mean, dtype=img.dtype)[None, :, None, None]
    s = torch.tensor(px_std, dtype=img.dtype)[None, :, None, None]
    return ((img - m) / s) / 255


def _prep(cfg, x, y=None):
    x = normalize_image(x, cfg.px_mean, cfg.px_std)
    if y is not None and cfg.arch == 'unet':
        out = x.shape[2] - 2 * UNET_CROP                      # model.py:306-307 generalised to any tile size
        y = y[:, UNET_CROP:UNET_CROP + out, UNET_CROP:UNET_CROP + out]
    if cfg.ch == 1 and cfg.arch == 'deeplab':
        x = torch.cat((x, x, x), 1)                           # model.py:310-311
    return x, y


def forward(sd, cfg, x, training, taps=None, momentum=0.1):
    if cfg.arch == 'unet':
        return unet_forward(sd, x, training, cfg.dropout, taps, momentum=momentum)
    return deeplab_forward(sd, x, cfg.backbone, training, cfg.dropout, taps, momentum=momentum)


def calibrate_bn(sd, cfg, x):
    """Fixture helper (not a reference function): one training-mode forward with BN momentum 1 so the
    running statistics of formula weights equal the batch statistics of `x` -- eval-mode activations
    then stay O(1) as in a trained model.  num_batches_tracked is reset to 0 afterwards."""
    xin, _ = _prep(cfg, x)
    with torch.no_grad():
        forward(sd, cfg, xin, True, momentum=1.0)
    for k, t in sd.items():
        if k.endswith('num_batches_tracked'):
            t.zero_()
    return sd


def trainable(sd):
    return [t for t in sd.values() if t.is_floating_point() and t.requires_grad]


def make_optimizer(sd, cfg):
    """model.py:240-245 over net.parameters() (all conv + BN affine tensors)."""
    for k, t in sd.items():
        if t.is_floating_point() and not (k.endswith('running_mean') or k.endswith('running_var')):
            t.requires_grad_(True)
    return torch.optim.AdamW(trainable(sd), lr=cfg.lr, weight_decay=cfg.weight_decay)


def train_step(sd, opt, cfg, x, y):
    """One Model.train step (model.py:300-328). Returns (ce, dice, focal, total, logits)."""
    x, y = _prep(cfg, x, y)
    logits = forward(sd, cfg, x, True)
    total, ce, dsc, fl = multiloss(logits, y, cfg.loss_weights, cfg.class_weights, cfg.weighted)
    opt.zero_grad()
    total.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(trainable(sd), cfg.clip)
    opt.step()
    return ce.item(), dsc.item(), fl.item(), total.item(), logits.detach(), float(gnorm)


def eval_step(sd, cfg, x, y):
    """Model.eval (model.py:338-365): eval-mode forward, three losses separately."""
    x, y = _prep(cfg, x, y)
    with torch.no_grad():
        logits = forward(sd, cfg, x, False)
        cw = cfg.class_weights if cfg.weighted else None
        return logits, ce_loss(logits, y, cw).item(), dice_loss(logits, y).item(), focal_loss(logits, y).item()


def test_step(sd, cfg, x):
    """Model.test (model.py:367-382)."""
    x, _ = _prep(cfg, x)
    with torch.no_grad():
        return forward(sd, cfg, x, False)
