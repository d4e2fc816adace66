"""MultiLoss on the fused HIP loss head -- same constructor, attributes and callables as the reference's
models/modules/loss.py:23-216 (the duck-typed `crit` contract of models/model.py:196-209,317-319,360-362)."""
import numpy as np
import torch
from torch import nn

from . import ops
from .runtime import runtime


class MultiLoss(nn.Module):
    """MultiLoss(loss_weights={'weighted','weights','ce','dice','focal'}, schema={'n_classes','class_codes','class_labels'}).

    forward(pred, target) -> 0-d tensor (ce_w*CE + dice_w*Dice + focal_w*Focal) with grad; side effects
    .ce / .dsc / .fl hold the three terms (loss.py:107-112).  One kernel pass computes all three."""

    def __init__(self, loss_weights, schema):
        super().__init__()
        self.n_classes = schema['n_classes']
        self.codes = schema.get('class_codes')
        self.categories = schema.get('class_labels')
        self.weighted = bool(loss_weights['weighted'])
        w = loss_weights.get('weights')
        # the reference crashes on weights=None (loss.py:46,60-61); here None means uniform weights
        w = np.ones(self.n_classes, np.float32) if w is None else np.asarray(w, np.float32)
        if w.shape != (self.n_classes,):
            raise ValueError('class weights must have length n_classes=%d' % self.n_classes)
        self.register_buffer('weights', torch.from_numpy(w.copy()))
        self.dsc_weight = float(loss_weights['dice'])
        self.ce_weight = float(loss_weights['ce'])
        self.fl_weight = float(loss_weights['focal'])
        self.eps = 1e-8
        self.ce = self.dsc = self.fl = 0.

    def _check(self, pred, target):
        if not torch.is_tensor(pred):
            raise TypeError('Input type is not a torch.Tensor. Got {}'.format(type(pred)))
        if pred.dim() != 4 or pred.size(1) != self.n_classes:
            raise ValueError('Invalid input shape, we expect Bx{}xHxW. Got: {}'.format(self.n_classes, tuple(pred.shape)))
        if pred.size(0) != target.size(0) or pred.size(2) != target.size(1) or pred.size(3) != target.size(2):
            raise ValueError('Expected prediction {} to match target {}.'.format(tuple(pred.shape), tuple(target.shape)))
        if pred.device != target.device:
            raise ValueError('input and target must be in the same device. Got: {} and {}'.format(pred.device, target.device))

    def _all(self, pred, target, w_ce, w_d, w_f):
        self._check(pred, target)
        cw = self.weights if self.weighted else None
        return ops.multiloss(pred, target, cw, w_ce, w_d, w_f, runtime.sync_group)

    def forward(self, pred, target):
        losses = self._all(pred, target, self.ce_weight, self.dsc_weight, self.fl_weight)
        self.ce, self.dsc, self.fl = losses[1].detach(), losses[2].detach(), losses[3].detach()
        return losses[0]

    # the validation path calls the three terms separately (model.py:360-362)
    def ce_loss(self, pred, target):
        return self._all(pred, target, 1.0, 0.0, 0.0)[0]

    def dice_loss(self, pred, target):
        return self._all(pred, target, 0.0, 1.0, 0.0)[0]

    def focal_loss(self, pred, target):
        return self._all(pred, target, 0.0, 0.0, 1.0)[0]

    def all_losses(self, pred, target):
        """(total, ce, dice, focal) as one [4] tensor from a single pass."""
        return self._all(pred, target, self.ce_weight, self.dsc_weight, self.fl_weight)

    def print_settings(self):
        hline = '_' * 40
        print('{:30s}{:<10s}'.format('Loss', 'Weight'))
        print(hline)
        print('{:30s}{:<10f}'.format('Cross-entropy', self.ce_weight))
        print('\tCE losses {}weighted by class.'.format('' if self.weighted else 'not '))
        print('{:30s}{:<10f}'.format('Dice Coefficient', self.dsc_weight))
        print('{:30s}{:<10f}'.format('Focal Loss', self.fl_weight))
        if self.codes and self.categories:
            print('\n{:8s}{:22s}{:<10s}'.format('Class', 'Label', 'Weight'))
            print(hline)
            for i, w in enumerate(self.weights.tolist()):
                print('{:8s}{:22s}{:<10f}'.format(str(self.codes[i]), str(self.categories[i]), w))
