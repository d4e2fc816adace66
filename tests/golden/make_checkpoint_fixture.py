#!/usr/bin/env python3
"""Container-only (needs /root/reference): write tests/golden/ref_checkpoint_tiny.pth -- a checkpoint and a best-model file produced by the
REFERENCE's own Checkpoint.save (models/modules/checkpoint.py:51-67), pickled `config.Parameters` meta and torch.optim.AdamW state
included -- so that `-m "not gpu"` exercises pylc_amd.checkpoint's reader on a reference-written file on every box.

The reference's networks weigh 116-238 MB as files, so the file holds a stand-in net: the first conv / BatchNorm pair of the reference's
ResNet (`backbone.conv1`, `backbone.bn1`: the reference's own module classes, key names and shapes), stepped once by the reference's
optimiser (Model.init_optim).  Everything format-related -- dict keys, the meta class path, numpy-typed meta fields, the AdamW
state_dict layout -- is the reference's; only the size is not.  Regeneration: the net is seeded before the reference builds it, the
scratch directory and `meta.seed` are pinned, so every tensor, `data.pkl` and the JSON digests come out byte-identical; the one record that
differs between two runs is torch.save's own random `.data/serialization_id` (40 digits, not data).  `ref_losses_tiny.pth` is the loss log
written by the reference's RunningLoss.save with validation rows / best_dice as the numpy scalars Model.eval produces.  (The full-size cross-load in both directions: check_checkpoint_compat.py.)"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from make_golden import enter_reference, build_reference_model  # noqa: E402


def main():
    # a FIXED scratch directory: the reference's meta pickles its save / output paths, and a random mkdtemp name would make the files differ
    # from run to run in those bytes alone
    import shutil
    import tempfile
    fixed = os.path.join(tempfile.gettempdir(), 'pylc_ref_fixture')
    shutil.rmtree(fixed, ignore_errors=True)
    tempfile.mkdtemp = lambda *a, **k: (os.makedirs(fixed), fixed)[1]
    enter_reference()
    from oracle import step as ostep
    from tests import _data as D
    from torch import nn
    torch.manual_seed(11)               # BEFORE the reference initialises its net: the fixture regenerates byte for byte
    ref = build_reference_model('deeplab', 'resnet', 9, 3, ostep.PX_RGB_MEAN, ostep.PX_RGB_STD, D.class_weights(9), False)

    class Stem(nn.Module):              # the reference's own stem modules under the reference's own key names
        def __init__(self, backbone):
            super().__init__()
            self.backbone = nn.Module()
            self.backbone.conv1, self.backbone.bn1 = backbone.conv1, backbone.bn1

        def forward(self, x):
            return self.backbone.bn1(self.backbone.conv1(x))
    ref.net = Stem(ref.net.backbone)
    ref.optim = ref.init_optim()        # model.py:238-254 over the stand-in's parameters
    out = ref.net(torch.randn(2, 3, 32, 32))
    out.square().mean().backward()
    ref.optim.step()
    ref.epoch, ref.iter = 3, 41
    ref.meta.seed = 11                                               # config.py draws it from torch.random.seed(): pinned, so data.pkl regenerates byte for byte
    ref.meta.px_mean = np.asarray(ref.meta.px_mean, np.float64)      # profile.py hands numpy-typed statistics to the meta
    ref.meta.m2 = np.float64(0.125)                                  # (profile.py's m2 / jsd are numpy scalars before the JSON round trip)
    # the loss log the reference way: training rows are python floats (`.item()`, model.py:319), validation rows and best_dice are
    # what Model.eval appends -- `.cpu().numpy()` values (model.py:360-363) averaged by RunningLoss.log (loss.py:284-293): NUMPY scalars
    ref.loss.train += [(0, 2.25, 0.875, 0.5), (20, 1.75, 0.8125, 0.4375)]
    for ce, dice, fl in ((1.9, 0.85, 0.45), (1.7, 0.80, 0.40)):
        ref.loss.intv += [(torch.tensor(ce).cpu().numpy(), torch.tensor(dice).cpu().numpy(), torch.tensor(fl).cpu().numpy())]
    ref.loss.log(41, training=False)
    ref.loss.lr += [(0, 1e-4)]
    ref.loss.save()                                                  # loss.py:296-305
    assert ref.loss.is_best
    ref.checkpoint.save(ref, is_best=True)                           # checkpoint.py:51-67: checkpoint.pth AND the best-model file
    import shutil
    shutil.copy(ref.loss.log_file, os.path.join(HERE, 'ref_losses_tiny.pth'))
    shutil.copy(ref.checkpoint.checkpoint_file, os.path.join(HERE, 'ref_checkpoint_tiny.pth'))
    shutil.copy(ref.checkpoint.model_file, os.path.join(HERE, 'ref_model_tiny.pth'))
    sd = ref.net.state_dict()
    st = ref.optim.state_dict()
    expect = {'epoch': 3, 'iter': 41, 'keys': [[k, list(v.shape)] for k, v in sd.items()],
              'digest': {k: D.digest(v) for k, v in sd.items() if v.is_floating_point()},
              'exp_avg_digest': {str(i): D.digest(s['exp_avg']) for i, s in st['state'].items()},
              'exp_avg_sq_digest': {str(i): D.digest(s['exp_avg_sq']) for i, s in st['state'].items()},
              'losses': {'train': [list(r) for r in ref.loss.train], 'valid': [[float(v) for v in r] for r in ref.loss.valid],
                         'best_dice': float(ref.loss.best_dice), 'valid_types': [type(v).__name__ for v in ref.loss.valid[0]]},
              'lr': st['param_groups'][0]['lr'], 'meta': {'arch': ref.meta.arch, 'backbone': ref.meta.backbone, 'n_classes': ref.meta.n_classes,
                                                          'ch': ref.meta.ch, 'lr': ref.meta.lr, 'weight_decay': ref.meta.weight_decay}}
    with open(os.path.join(HERE, 'ref_checkpoint_tiny.json'), 'w') as f:
        json.dump(expect, f)
    print('wrote', os.path.getsize(os.path.join(HERE, 'ref_checkpoint_tiny.pth')), 'bytes;', expect['keys'])


if __name__ == '__main__':
    main()
