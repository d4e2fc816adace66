"""What would a captured HIP graph of the training step buy?  Captures ONE Model.train step of the headline configuration with
torch.cuda.graph (our kernels are plain launches on the capturing stream; the wgrad side stream forks and joins inside the capture) and
replays it: step time eager vs replayed.  An experiment, not the product path: the dropout seeds and the optimiser's step count are baked
into the captured launches.   usage: python tools/graph_probe.py [--config c3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c3')
    a = ap.parse_args()
    import bench
    from pylc_amd.model import Model, Meta
    cfg = bench.CONFIGS[a.config]
    if cfg['precision'] != 2:
        os.environ.setdefault('PYLC_CONV_PRECISION', str(cfg['precision']))
    dev = torch.device('cuda', 0)
    w_ce, w_dice, w_focal = cfg['losses']
    meta = Meta(arch=cfg['arch'], backbone=cfg['backbone'], ch=cfg['ch'], n_classes=cfg['classes'], report=10 ** 9,
                ce_weight=w_ce, dice_weight=w_dice, focal_weight=w_focal)
    model = Model(meta, dev).build()
    x, y = bench.synth(0, cfg['batch'], cfg['ch'], cfg['tile'], cfg['classes'], dev)
    for _ in range(8):
        model.train(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.train(x, y)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    print('eager: %.2f ms/step  %.1f tiles/s' % (eager * 1e3, cfg['batch'] / eager), flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model.train(x, y)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            model.train(x, y)
    except Exception as e:
        print('capture failed: %s: %s' % (type(e).__name__, str(e)[:400]))
        return
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 10
    print('graph replay: %.2f ms/step  %.1f tiles/s  (%.1f %% of eager)' % (rep * 1e3, cfg['batch'] / rep, 100 * rep / eager), flush=True)
    print('last loss after replays:', [float(v) for v in (model.crit.ce, model.crit.dsc, model.crit.fl)])


if __name__ == '__main__':
    main()
