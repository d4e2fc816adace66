"""Every ATen operator one training step of the headline configuration dispatches, by call site (TorchDispatchMode): what is left of torch
in the step besides allocation.   usage: python tools/aten_ops_in_step.py [--batch 8] [--config c3]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        site = '?'
        for fr in reversed(traceback.extract_stack(limit=14)):
            if 'pylc_amd' in fr.filename and 'tools' not in fr.filename:
                site = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                break
        self.seen[(str(func), site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--config', default='c3')
    a = ap.parse_args()
    import bench
    from pylc_amd.model import Model, Meta
    cfg = bench.CONFIGS[a.config]
    dev = torch.device('cuda', 0)
    w_ce, w_dice, w_focal = cfg['losses']
    meta = Meta(arch=cfg['arch'], backbone=cfg['backbone'], ch=cfg['ch'], n_classes=cfg['classes'], report=10 ** 9,
                ce_weight=w_ce, dice_weight=w_dice, focal_weight=w_focal)
    model = Model(meta, dev).build()
    x, y = bench.synth(0, a.batch, cfg['ch'], cfg['tile'], cfg['classes'], dev)
    for _ in range(3):
        model.train(x, y)
    torch.cuda.synchronize()
    with Log() as log:
        model.train(x, y)
    torch.cuda.synchronize()
    skip = ('aten.empty', 'aten.as_strided', 'aten.view', 'aten.detach', 'aten.slice', 'aten.alias', 'aten.select', 'aten._unsafe_view',
            'aten.permute', 'aten.reshape', 'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.t.', 'aten.transpose')
    total = 0
    for (op, site), n in sorted(log.seen.items(), key=lambda kv: -kv[1]):
        if any(op.startswith(s) for s in skip):
            continue
        total += n
        print('%5d  %-42s %s' % (n, op, site))
    print('total device-touching ATen calls per step (views / allocations excluded):', total)


if __name__ == '__main__':
    main()
