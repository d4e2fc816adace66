"""Does a LOW-priority wgrad side stream beat one queue?  The whole training step runs on a HIGH-priority stream (torch.cuda.Stream(priority=-1));
the wgrad side stream (PYLC_SIDE_STREAM=1) keeps the default (low) priority, so its workgroups should only be dispatched where the main queue
leaves CUs idle (launch gaps, ramp-down tails).  Prints ms per step for: one queue / side stream at equal priority / side stream below the step.
usage: python tools/prio_probe.py [--config c3]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c3')
    a = ap.parse_args()
    import bench
    from pylc_amd.model import Model, Meta
    from pylc_amd.runtime import runtime
    cfg = bench.CONFIGS[a.config]
    if cfg['precision'] != 2:
        os.environ.setdefault('PYLC_CONV_PRECISION', str(cfg['precision']))
    dev = torch.device('cuda', 0)
    w_ce, w_dice, w_focal = cfg['losses']
    meta = Meta(arch=cfg['arch'], backbone=cfg['backbone'], ch=cfg['ch'], n_classes=cfg['classes'], report=10 ** 9,
                ce_weight=w_ce, dice_weight=w_dice, focal_weight=w_focal)
    model = Model(meta, dev).build()
    x, y = bench.synth(0, cfg['batch'], cfg['ch'], cfg['tile'], cfg['classes'], dev)

    def run(name, side, hi):
        runtime.wgrad_side_stream = side
        st = torch.cuda.Stream(priority=-1) if hi is True else (torch.cuda.Stream() if hi == 'plain' else torch.cuda.current_stream())
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(5):
                model.train(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model.train(x, y)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
        torch.cuda.current_stream().wait_stream(st)
        print('%-52s %.2f ms/step  %.1f tiles/s' % (name, dt * 1e3, cfg['batch'] / dt), flush=True)

    for _ in range(2):
        run('one queue', False, False)
        run('one queue, on a high-priority stream', False, True)
        run('one queue, on a created stream of default priority', False, 'plain')
        if a.config == 'c5':
            run('side stream, equal priority (null stream)', True, False)
            run('side stream, step on a created stream of default priority', True, 'plain')
            run('side stream BELOW the step (step on a high-priority stream)', True, True)


if __name__ == '__main__':
    main()
