"""Reference-vs-reference noise of the true-size oracle steps of tests/test_truesize_oracle_gpu.py: the SAME oracle step (oracle.step.train_step,
pinned bit-exactly to the reference by make_golden.py) run with 1 and with 8 CPU threads -- two summation orders of one algorithm.  Per
pinned gradient tensor the largest elementwise difference (relative to the tensor's largest entry) goes to tests/golden/truesize_conditioning.json;
the GPU test uses it as the floor of its elementwise bound, exactly as test_nets_gpu.py uses the fixtures' `cond_maxdiff` (DESIGN.md section 3:
a 101-layer train-mode network turns last-bit conv differences into per-cent gradient differences through ReLU flips).
Run in the build container:  python tests/golden/make_truesize_conditioning.py      (needs no reference import: the oracle only)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import oracle
    from oracle import step as ostep
    from tests import _data as D
    from tests.test_truesize_oracle_gpu import CASES, grad_keys
    out = {}
    for tag, (arch, backbone, n_cls, ch, b, hw, salt, seed) in CASES.items():
        if arch == 'deeplab' and backbone == 'xception':
            continue                                       # eval only
        cfg = ostep.StepConfig(arch, backbone, n_cls, ch, dropout=False)
        spec = oracle.state_spec(arch, backbone, n_cls, 3 if arch == 'deeplab' else ch)
        x = D.tiles(seed, b, ch, hw, hw)
        y = D.blob_masks(seed + 1, b, hw, hw, n_cls, cell=32)
        torch.set_num_threads(8)
        w = ostep.calibrate_bn(oracle.formula_state(spec, salt=salt), cfg, x.clone())
        keys = grad_keys(tag, w)
        runs = []
        for threads in (8, 1):
            torch.set_num_threads(threads)
            sd = {k: v.clone() for k, v in w.items()}
            opt = ostep.make_optimizer(sd, cfg)
            res = ostep.train_step(sd, opt, cfg, x.clone(), y.clone())
            runs.append((res, {k: sd[k].grad.double().clone() for k in keys}))
            print(tag, threads, 'threads:', res[:3], res[5], flush=True)
        rec = {'loss_maxdiff': max(abs(a - b) for a, b in zip(runs[0][0][:3], runs[1][0][:3])),
               'gnorm_rel': abs(runs[0][0][5] - runs[1][0][5]) / runs[0][0][5], 'grads': {}}
        for k in keys:
            a, c = runs[0][1][k], runs[1][1][k]
            rec['grads'][k] = {'absmax': float(a.abs().max()), 'cond_maxdiff': float((a - c).abs().max()),
                               'cos': float((a * c).sum() / (a.norm() * c.norm()))}
            print('   %-44s cond %.3g of |g|max' % (k, rec['grads'][k]['cond_maxdiff'] / rec['grads'][k]['absmax']), flush=True)
        out[tag] = rec
    json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'truesize_conditioning.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
