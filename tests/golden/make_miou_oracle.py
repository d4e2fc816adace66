#!/usr/bin/env python3
"""Oracle side of tests/test_mode3_gpu.py::test_mode3_miou_parity_after_training, computed once (3 minutes of CPU training that the GPU
test run would otherwise repeat): the fp32 CPU oracle (pinned to the reference by make_golden.py) trains DeepLabV3+/Xception for N steps
on the seeded learnable tiles, dropout live, and is scored with the reference's weighted-Jaccard "mIoU" on held-out tiles.

    python tests/golden/make_miou_oracle.py        ->  tests/golden/miou_xception_oracle.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import torch  # noqa: E402

CONFIG = {'n_steps': 40, 'b': 4, 'hw': 64, 'ncls': 11, 'lr': 1e-3, 'torch_seed': 9, 'init_seed': 12, 'train_seed0': 2000, 'valid_seed': 6000,
          'valid_b': 8}


def run(cfg=CONFIG):
    import oracle
    from oracle import step as ostep
    from tests import _data as D
    torch.manual_seed(cfg['torch_seed'])
    spec = oracle.state_spec('deeplab', 'xception', cfg['ncls'], 3)
    w0 = oracle.init_state(spec, seed=cfg['init_seed'])
    gray = lambda t: t[:, :1].contiguous()
    batches = [D.learnable_tiles(cfg['train_seed0'] + i, cfg['b'], cfg['hw'], cfg['ncls']) for i in range(cfg['n_steps'])]
    batches = [(gray(x), y) for x, y in batches]
    xv, yv = D.learnable_tiles(cfg['valid_seed'], cfg['valid_b'], cfg['hw'], cfg['ncls'])
    xv = gray(xv)
    c = ostep.StepConfig('deeplab', 'xception', cfg['ncls'], 1, lr=cfg['lr'], dropout=True)
    sd = {k: v.clone() for k, v in w0.items()}
    opt = ostep.make_optimizer(sd, c)
    for x, y in batches:
        o = ostep.train_step(sd, opt, c, x, y)
    c.dropout = False
    xin, _ = ostep._prep(c, xv)
    with torch.no_grad():
        pred = ostep.forward(sd, c, xin, True).argmax(1).numpy()
    return {'miou_batch_stat': float(oracle.weighted_jaccard(yv.numpy(), pred, cfg['ncls'])), 'last_losses': [float(v) for v in o[:3]]}


if __name__ == '__main__':
    out = {'config': CONFIG, 'torch': torch.__version__, 'threads': torch.get_num_threads()}
    out.update(run())
    print(out)
    with open(os.path.join(HERE, 'miou_xception_oracle.json'), 'w') as f:
        json.dump(out, f)
