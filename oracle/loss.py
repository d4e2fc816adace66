"""CPU restatement of the reference MultiLoss (TEST INFRASTRUCTURE).

Follows models/modules/loss.py: ce_loss :66-69, forward :107-112, dice_loss :137-146,
focal_loss :174-189; constants from config.py:201-208 (loss weights 0.5/0.5/0.5,
dice_smooth 1, fl_alpha 0.25, fl_gamma 2, fl_reduction 'mean') and loss.py:50 (eps 1e-8).
Written in log-softmax / closed form rather than the reference's
softmax + one_hot chain; make_golden.py asserts both agree to fp32 rounding.
"""
import torch
import torch.nn.functional as F

DICE_SMOOTH = 1.0
FL_ALPHA = 0.25
FL_GAMMA = 2.0
FL_EPS = 1e-8


def ce_loss(logits, target, class_weights=None):
    """nn.CrossEntropyLoss([weights]) over [B,C,H,W] logits / [B,H,W] int64 targets (loss.py:66-69)."""
    logp = F.log_softmax(logits, dim=1)
    nll = -logp.gather(1, target.unsqueeze(1)).squeeze(1)
    if class_weights is None:
        return nll.mean()
    w = class_weights.to(logits.dtype)[target]
    return (w * nll).sum() / w.sum()


def dice_loss(logits, target):
    """loss.py:137-146: sums over (B,H,W) per class, smooth = 1, mean over classes."""
    c = logits.shape[1]
    p = F.softmax(logits, dim=1)
    onehot = F.one_hot(target, c).permute(0, 3, 1, 2).to(p.dtype)
    inter = (p * onehot).sum(dim=(0, 2, 3))
    card = p.sum(dim=(0, 2, 3)) + onehot.sum(dim=(0, 2, 3))
    return (1 - (2 * inter + DICE_SMOOTH) / (card + DICE_SMOOTH)).mean()


def focal_loss(logits, target):
    """loss.py:174-189: q = softmax(z)_t + 1e-8 ; mean(-alpha (1-q)^gamma log q)."""
    q = F.softmax(logits, dim=1).gather(1, target.unsqueeze(1)).squeeze(1) + FL_EPS
    return (-FL_ALPHA * torch.pow(1 - q, FL_GAMMA) * torch.log(q)).mean()


def multiloss(logits, target, weights=(0.5, 0.5, 0.5), class_weights=None, weighted=False):
    """loss.py:107-112 -> (total, ce, dice, focal)."""
    ce = ce_loss(logits, target, class_weights if weighted else None)
    dsc = dice_loss(logits, target)
    fl = focal_loss(logits, target)
    return weights[0] * ce + weights[1] * dsc + weights[2] * fl, ce, dsc, fl
