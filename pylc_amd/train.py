"""Epoch driver of the HIP path -- the counterpart of the reference's train.py:72-156 loop body (train_epoch, validate,
StepLR stepping, checkpoint on every validation + best-model copy on a new best validation Dice).  The CLI / HDF5 set-up
above that loop (train.py:22-70) is out of scope: callers pass any re-iterable of (img, mask) batches, e.g. a list or a
factory returning a pylc_amd.data.TileFeeder."""
import os

import torch


def _batches(src):
    return src() if callable(src) else src


def train_epoch(model, batches):
    """train.py:95-122."""
    model.net.train()
    for x, y in _batches(batches):
        model.train(x, y)
    return model


def validate(model, batches, save_dir=None):
    """train.py:125-156: eval-mode pass, log, checkpoint."""
    model.net.eval()
    with torch.no_grad():
        for x, y in _batches(batches):
            model.eval(x, y)
        model.log()
        if save_dir is not None:
            model.save(save_dir)
    return model


def trainer(model, train_batches, valid_batches, n_epochs, save_dir=None):
    """train.py:72-92, including its resume quirk `range(offset, n_epochs - offset)`."""
    model.net.train()
    offset = model.epoch
    for epoch in range(offset, n_epochs - offset):
        model.loss.lr += [(model.iter, model.get_lr())]
        if epoch == 0:
            validate(model, valid_batches, save_dir)
        train_epoch(model, train_batches)
        validate(model, valid_batches, save_dir)
        model.sched.step()
        model.epoch += 1
    return model
