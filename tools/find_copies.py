"""Which Python lines issue the device copies / fills / adds of one training step (torch.profiler with stacks, all threads -- the autograd
engine's included, which a TorchDispatchMode does not see).   usage: python tools/find_copies.py [--batch 8] [--config c3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model.train(x, y)
        torch.cuda.synchronize()
    want = ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::_to_copy', 'aten::cat',
            'aten::maximum', 'aten::mul', 'aten::zeros', 'aten::stack', 'aten::sum', 'aten::item', 'aten::_local_scalar_dense')
    rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key in want]
    for e in sorted(rows, key=lambda e: -e.count):
        stack = [s for s in e.stack if 'pylc_amd' in s or 'bench' in s][:3]
        print('%5d  %-22s dev %.1f us  %s' % (e.count, e.key, e.device_time_total, ' <- '.join(s.split('/')[-1] for s in stack) or '(autograd engine / no python frame)'))
    print('---- device activities')
    for e in sorted(prof.key_averages(), key=lambda e: -e.count):
        if e.device_time_total > 0 and 'pylc' not in e.key and not e.key.startswith('aten::'):
            print('%5d  %-60s %.1f us' % (e.count, e.key[:60], e.device_time_total))


if __name__ == '__main__':
    main()
