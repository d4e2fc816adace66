#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
Recipe = SURVEY.md section 8c: scratch cwd with a `schemas` symlink, stub modules for the
absent cv2 / h5py / seaborn, no bytecode written into the reference tree, and the shims for
the reference's own bugs (U-Net `normalizer.evaluate`, list-typed `meta.weights`, U-Net crop).

For every network it (1) checks the oracle's key/shape spec against the reference
state_dict, (2) loads name-keyed formula weights into the reference, (3) asserts the oracle
reproduces the reference (eval logits, train-step losses, gradients, post-AdamW parameters,
BN running stats), and (4) writes small fixtures holding OUTPUTS only -- inputs and weights
are regenerated from seeds by tests/_data.py and oracle.formula_state.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py --only-grads deeplab_resnet deeplab_xception unet     # the elementwise-gradient fixtures only
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def enter_reference():
    scratch = tempfile.mkdtemp(prefix='pylc_ref_')
    os.symlink(os.path.join(REF, 'schemas'), os.path.join(scratch, 'schemas'))
    os.chdir(scratch)
    cv2 = types.ModuleType('cv2')
    for i, n in enumerate(('INTER_NEAREST', 'INTER_LINEAR', 'INTER_CUBIC', 'INTER_AREA', 'INTER_LANCZOS4',
                           'IMREAD_COLOR', 'IMREAD_GRAYSCALE', 'COLOR_BGR2RGB', 'COLOR_RGB2BGR',
                           'COLOR_BGR2GRAY', 'BORDER_REFLECT', 'BORDER_CONSTANT', 'BORDER_REFLECT_101')):
        setattr(cv2, n, i)
    cv2.resize = lambda img, *a, **k: img
    sys.modules['cv2'] = cv2
    sys.modules['h5py'] = types.ModuleType('h5py')
    sns = types.ModuleType('seaborn')
    sns.heatmap = lambda *a, **k: None
    sns.set = lambda *a, **k: None
    sys.modules['seaborn'] = sns
    sys.path.insert(0, REF)
    return scratch


def build_reference_model(arch, backbone, n_classes, ch, px_mean, px_std, class_weights, weighted,
                          loss_weights=(0.5, 0.5, 0.5), tile=512):
    from config import defaults
    from models.model import Model
    from torch import nn
    m = Model()
    meta = m.meta
    meta.arch, meta.backbone, meta.ch, meta.n_classes = arch, backbone, ch, n_classes
    meta.pretrained = False
    meta.px_mean, meta.px_std = list(px_mean), list(px_std)
    if ch == 1:
        # shim 6: model.py:433-435 computes (float32 array - np.mean(list)) -- a float64 scalar, which
        # NumPy >= 2 (NEP 50) promotes to float64 and the fp32 conv then rejects.  float32 statistics
        # reproduce what NumPy 1.x (value-based casting) computed.
        meta.px_mean, meta.px_std = np.asarray(px_mean, np.float32), np.asarray(px_std, np.float32)
    meta.weights = [float(w) for w in class_weights]
    meta.weighted = weighted
    meta.ce_weight, meta.dice_weight, meta.focal_weight = loss_weights
    meta.id = None
    if arch == 'unet':
        class BN(nn.BatchNorm2d):           # shim 1: unet.py:113,117 call normalizer.evaluate(c)
            @classmethod
            def evaluate(cls, c):
                return cls(c)
        m.normalizers['batch'] = BN
        out = tile - 188                     # shim 3: crop hard-wired to 512 px (config.py:228-236)
        meta.crop_left = meta.crop_up = 94
        meta.crop_right = meta.crop_down = 94 + out
    m.build()
    for mod in m.net.modules():              # parity runs: dropout off (RNG streams differ)
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    assert defaults is meta
    return m


def tensor_digests(named):
    from tests._data import digest
    return {k: digest(v) for k, v in named}


def golden_net(tag, arch, backbone, b, ch, hw, n_classes, out_dir):
    import oracle
    from oracle import step as ostep
    from tests import _data as D
    print('==', tag)
    cw = D.class_weights(n_classes)
    px_mean, px_std = ostep.PX_RGB_MEAN, ostep.PX_RGB_STD
    ref = build_reference_model(arch, backbone, n_classes, ch, px_mean, px_std, cw, False, tile=hw)
    ref_sd = ref.net.state_dict()
    spec = oracle.state_spec(arch, backbone, n_classes, 3 if arch == 'deeplab' else ch)
    ref_keys = [(k, list(v.shape)) for k, v in ref_sd.items()]
    my_keys = [(k, list(s)) for k, (s, _) in spec.items()]
    assert ref_keys == my_keys, 'state_dict spec mismatch'
    x = D.tiles(100, b, ch, hw, hw)
    y = D.blob_masks(101, b, hw, hw, n_classes, cell=8)
    cfg = ostep.StepConfig(arch, backbone, n_classes, ch, px_mean, px_std, dropout=False)
    w = ostep.calibrate_bn(oracle.formula_state(spec, salt=1), cfg, x.clone())
    ref.net.load_state_dict(w)

    # ---- eval forward (Model.test semantics) ------------------------------------------------
    ref.net.eval()
    ref_logits = ref.test(x.clone())[0]
    sd = {k: v.clone() for k, v in w.items()}
    taps = {}
    xin, _ = ostep._prep(cfg, x.clone())
    with torch.no_grad():
        my_logits = ostep.forward(sd, cfg, xin, False, taps)
    err = (ref_logits - my_logits).abs().max().item()
    print('  eval logits max|diff| = %.3g  (|logits| max %.3g)' % (err, ref_logits.abs().max().item()))
    assert err <= 2e-5 * max(1.0, ref_logits.abs().max().item())
    # conditioning of the fixture itself: the same oracle on 1 CPU thread (different summation order)
    torch.set_num_threads(1)
    with torch.no_grad():
        one = ostep.forward({k: v.clone() for k, v in w.items()}, cfg, xin, False)
    torch.set_num_threads(8)
    cond = (one - my_logits).abs().max().item()
    print('  conditioning: 1-thread vs 8-thread oracle eval logits max|diff| = %.3g' % cond)
    assert cond < 2e-4, 'fixture is ill-conditioned'
    top2 = ref_logits.topk(2, dim=1).values
    fix = {
        'eval_logits': ref_logits.numpy().astype(np.float32),
        'eval_argmax': ref_logits.argmax(1).numpy().astype(np.uint8),
        'eval_margin': (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32),
    }
    _, rce, rdice, rfl = None, None, None, None
    ref_eval = ref.eval(x.clone(), y.clone())
    e_ce, e_dice, e_fl = [float(v) for v in ref.loss.intv[-1]]
    ref.loss.intv = []
    _, o_ce, o_dice, o_fl = ostep.eval_step(sd, cfg, x.clone(), y.clone())
    assert max(abs(e_ce - o_ce), abs(e_dice - o_dice), abs(e_fl - o_fl)) < 2e-6, (e_ce, o_ce, e_dice, o_dice, e_fl, o_fl)
    meta = {'eval_losses': [e_ce, e_dice, e_fl], 'conditioning_eval_logits': cond}
    meta['taps'] = {k: [float(v.mean()), float(v.std()), float(v.abs().max())] for k, v in taps.items()}

    # ---- two training steps (Model.train semantics) -----------------------------------------
    ref.net.train()
    opt = ostep.make_optimizer(sd, cfg)
    steps = []
    for it in range(2):
        ref.train(x.clone(), y.clone())
        r = [float(v) for v in ref.loss.intv[-1]] if ref.loss.intv else None
        if r is None:                                   # log() at iter%report==0 clears intv
            r = [float(ref.crit.ce), float(ref.crit.dsc), float(ref.crit.fl)]
        o = ostep.train_step(sd, opt, cfg, x.clone(), y.clone())
        d = max(abs(r[i] - o[i]) for i in range(3))
        print('  step %d  ref (ce,dice,focal)=%s  max|diff|=%.3g  gnorm=%.5g' % (it, r, d, o[5]))
        assert d < (5e-6 if it == 0 else 2e-4)      # step 1 inherits the AdamW |g|~eps sensitivity noted below
        steps.append({'ce': r[0], 'dice': r[1], 'focal': r[2], 'grad_norm_preclip': o[5]})
        if it == 0:
            gref = {k: p.grad for k, p in ref.net.named_parameters()}
            worst = 0.0
            gmax = max(g.abs().max().item() for g in gref.values())
            # parameters whose true gradient is identically 0 (an additive bias that a train-mode
            # BatchNorm removes again): both sides hold pure summation noise there
            zero_keys = [k for k in gref if (arch == 'unet' and (k.endswith('block.0.bias') or k.endswith('block.3.bias') or k.endswith('up.1.bias')))
                         or (backbone == 'xception' and k.startswith('backbone.') and k.endswith('.bn.bias'))]
            meta['zero_grad_keys'] = zero_keys
            for k, g in gref.items():
                if k in zero_keys:
                    continue
                og = sd[k].grad
                rel = (g - og).abs().max().item() / (g.abs().max().item() + 1e-3 * gmax)   # some true grads are 0 (a bias ahead of a train-mode BN): pure summation noise
                if rel > 1e-3:
                    print('    grad mismatch %-50s rel %.3g  |g|max %.3g' % (k, rel, g.abs().max().item()))
                worst = max(worst, rel)
            print('  step 0  worst relative grad diff = %.3g' % worst)
            assert worst < 2e-3
            meta['grad_digest_step0'] = tensor_digests(gref.items())
            # elementwise gradients of six tensors spread over the depth of the network (four conv filters, two BatchNorm vectors, each
            # at most 40k elements): a permutation or a sign error inside a tensor leaves the digests above unchanged, these not
            convs = [k for k, g in gref.items() if k not in zero_keys and g.dim() == 4 and g.numel() <= 40000]
            vecs = [k for k, g in gref.items() if k not in zero_keys and g.dim() == 1 and g.numel() >= 32]
            picked = [convs[round(i * (len(convs) - 1) / 3)] for i in range(4)] + [vecs[len(vecs) // 5], vecs[-3]]
            # plus named tensors from the middle and the far end of the network; of the large ones the first `rows` filters only
            named = {'deeplab_resnet': [('backbone.layer3.22.conv2.weight', 8), ('aspp.aspp4.atrous_conv.weight', 2), ('backbone.layer3.22.bn3.weight', 0)],
                     'deeplab_xception': [('backbone.block12.rep.4.conv1.weight', 0), ('backbone.block12.rep.4.pointwise.weight', 8),
                                          ('aspp.aspp4.atrous_conv.weight', 2)],
                     'unet': [('encoder.3.block.3.weight', 4), ('decoder.0.conv_block.block.0.weight', 2)]}[tag]
            rows = {k: 0 for k in picked}
            rows.update({k: r for k, r in named if k not in zero_keys})
            cut = lambda k, t: t[:rows[k]] if rows[k] else t
            grad_fix = {'g::' + k: cut(k, gref[k]).detach().numpy().astype(np.float32).copy() for k in rows}
            grad_meta = {k: {'absmax': float(cut(k, gref[k]).abs().max()), 'rows': rows[k],
                             'oracle_maxdiff': float((cut(k, gref[k]) - cut(k, sd[k].grad)).abs().max())} for k in rows}
            new_sd = ref.net.state_dict()
            pw, pk = 0.0, None
            for k, v in new_sd.items():
                if v.is_floating_point():
                    e = (v - sd[k].detach()).abs().max().item()
                    if e > pw:
                        pw, pk = e, k
            # the first AdamW update is lr * g/(|g| + 1e-8): for |g| ~ 1e-8 an fp32-rounding-level
            # gradient difference moves the update by a fraction of lr (1e-4), hence the loose bound
            print('  step 0  worst post-AdamW/BN-stat diff = %.3g (%s)' % (pw, pk))
            assert pw < 2.5e-4
            meta['state_digest_step0'] = tensor_digests((k, v) for k, v in new_sd.items() if v.is_floating_point())
    meta['train_steps'] = steps
    # ---- conditioning of the training fixture: the same oracle on ONE CPU thread (different summation order).
    # Tests widen their tolerances to a multiple of these reference-vs-reference differences.
    torch.set_num_threads(1)
    sd1 = {k: v.clone() for k, v in w.items()}
    opt1 = ostep.make_optimizer(sd1, cfg)
    cond_steps, grad_cond = [], {}
    for it in range(2):
        o1 = ostep.train_step(sd1, opt1, cfg, x.clone(), y.clone())
        cond_steps.append({'loss': max(abs(o1[i] - steps[it][n]) for i, n in enumerate(('ce', 'dice', 'focal'))),
                           'gnorm_rel': abs(o1[5] - steps[it]['grad_norm_preclip']) / steps[it]['grad_norm_preclip']})
        if it == 0:
            coef = min(1.0, cfg.clip / (o1[5] + 1e-6))
            for k, g in gref.items():
                g1 = sd1[k].grad          # already clipped in place by clip_grad_norm_
                grad_cond[k] = float((g1 - g).norm() / (g.norm() + 1e-3 * gmax))
                if k in grad_meta:
                    grad_meta[k]['cond_maxdiff'] = float((cut(k, g1) - cut(k, g)).abs().max())       # reference-vs-reference noise, elementwise
    torch.set_num_threads(8)
    meta['conditioning_train'] = cond_steps
    meta['grad_conditioning_step0'] = grad_cond
    print('  conditioning (1 vs 8 threads): %s ; worst per-param grad rel l2 diff %.3g' % (cond_steps, max(grad_cond.values())))
    meta['keys'] = ref_keys
    meta['config'] = {'arch': arch, 'backbone': backbone, 'b': b, 'ch': ch, 'hw': hw, 'n_classes': n_classes,
                      'tile_seed': 100, 'mask_seed': 101, 'mask_cell': 8, 'weight_salt': 1,
                      'torch': torch.__version__}
    np.savez_compressed(os.path.join(out_dir, tag + '_grads.npz'), **grad_fix)
    with open(os.path.join(out_dir, tag + '_grads.json'), 'w') as f:
        json.dump(grad_meta, f)
    print('  elementwise gradient fixtures: %s' % {k: (v['absmax'], v['oracle_maxdiff'], v['cond_maxdiff']) for k, v in grad_meta.items()})
    if ONLY_GRADS:
        return
    np.savez_compressed(os.path.join(out_dir, tag + '.npz'), **fix)
    with open(os.path.join(out_dir, tag + '.json'), 'w') as f:
        json.dump(meta, f)


def golden_multiloss(out_dir):
    import oracle
    from tests import _data as D
    from models.modules.loss import MultiLoss
    print('== multiloss')
    out = {}
    arrays = {}
    for n_cls in (9, 11):
        rs = np.random.RandomState(300 + n_cls)
        z = torch.from_numpy((rs.standard_normal((2, n_cls, 40, 36)) * 3).astype(np.float32))
        t = D.blob_masks(301 + n_cls, 2, 40, 36, n_cls, cell=4)
        cw = D.class_weights(n_cls)
        for weighted in (False, True):
            crit = MultiLoss({'weighted': weighted, 'weights': [float(v) for v in cw], 'ce': 0.5, 'dice': 0.5, 'focal': 0.5},
                             {'n_classes': n_cls, 'class_codes': ['c%d' % i for i in range(n_cls)],
                              'class_labels': ['l%d' % i for i in range(n_cls)]})
            zr = z.clone().requires_grad_(True)
            tot = crit.forward(zr, t)
            tot.backward()
            zo = z.clone().requires_grad_(True)
            o_tot, o_ce, o_d, o_f = oracle.multiloss(zo, t, (0.5, 0.5, 0.5), torch.from_numpy(cw), weighted)
            o_tot.backward()
            assert abs(float(tot) - float(o_tot)) < 1e-6
            assert (zr.grad - zo.grad).abs().max().item() < 1e-9 + 1e-4 * zr.grad.abs().max().item()
            key = 'c%d_%s' % (n_cls, 'w' if weighted else 'u')
            out[key] = {'total': float(tot), 'ce': float(crit.ce), 'dice': float(crit.dsc), 'focal': float(crit.fl)}
            arrays[key + '_grad'] = zr.grad.numpy()
            print('  %s: %s' % (key, out[key]))
    out['config'] = {'shape': [2, 'n_cls', 40, 36], 'logit_seed': '300+n_cls', 'logit_scale': 3,
                     'mask_seed': '301+n_cls', 'mask_cell': 4}
    np.savez_compressed(os.path.join(out_dir, 'multiloss.npz'), **arrays)
    with open(os.path.join(out_dir, 'multiloss.json'), 'w') as f:
        json.dump(out, f)


def golden_stitch(out_dir):
    """tools.reconstruct on synthetic logits (cv2.resize stubbed to identity) vs oracle.stitch_classes."""
    import oracle
    from utils.tools import reconstruct
    print('== stitch')
    palette = np.random.RandomState(77).randint(0, 256, (9, 3)).astype(np.uint8)
    out, arrays = {}, {}
    for tag, rows, cols, tile, stride in (('half', 3, 4, 16, 8), ('full', 2, 3, 16, 16)):
        rs = np.random.RandomState(500 + rows)
        tiles = (rs.standard_normal((rows * cols, 9, tile, tile)) * 2).astype(np.float32)
        olap = tile - stride
        w, h = cols * stride + olap, rows * stride + olap
        meta = types.SimpleNamespace(extract={'w_fitted': w, 'h_fitted': h, 'w_scaled': w, 'h_scaled': h, 'offset': 0},
                                     tile_size=tile, stride=stride, palette_rgb=[list(map(int, p)) for p in palette], n_classes=9)
        batches = [torch.from_numpy(tiles[k:k + 5].copy()) for k in range(0, len(tiles), 5)]     # reconstruct mutates its input
        ref_rgb = reconstruct(batches, meta)
        mine = oracle.stitch_classes(tiles, rows, cols, tile, stride)
        assert ref_rgb.shape == (h, w, 3)
        assert np.array_equal(ref_rgb.astype(np.uint8), palette[mine]), 'oracle stitch != reference reconstruct'
        arrays[tag + '_mask'] = mine
        out[tag] = {'rows': rows, 'cols': cols, 'tile': tile, 'stride': stride, 'logit_seed': 500 + rows, 'logit_scale': 2}
        print('  %s: %dx%d tiles -> mask %s identical to the reference' % (tag, rows, cols, mine.shape))
    np.savez_compressed(os.path.join(out_dir, 'stitch.npz'), palette=palette, **arrays)
    with open(os.path.join(out_dir, 'stitch.json'), 'w') as f:
        json.dump(out, f)


def golden_driver(out_dir):
    """train.py's train_epoch / validate cadence + RunningLoss bookkeeping on a 3-batch synthetic dataset."""
    import oracle
    from oracle import step as ostep
    from tests import _data as D
    import train as ref_train
    print('== driver')
    n_cls, b, hw, n_epochs, report = 9, 3, 96, 2, 2
    cw = D.class_weights(n_cls)
    ref = build_reference_model('deeplab', 'resnet', n_cls, 3, ostep.PX_RGB_MEAN, ostep.PX_RGB_STD, cw, False)
    ref.meta.report = report
    spec = oracle.state_spec('deeplab', 'resnet', n_cls, 3)
    tr = [D.learnable_tiles(700 + i, b, hw, n_cls) for i in range(3)]
    va = [D.learnable_tiles(800 + i, b, hw, n_cls) for i in range(2)]
    cfg = ostep.StepConfig('deeplab', 'resnet', n_cls, 3, dropout=False)
    w = ostep.calibrate_bn(oracle.formula_state(spec, salt=6), cfg, tr[0][0].clone())
    ref.net.load_state_dict(w)
    ref.net.train()
    best = []
    orig_save = ref.save
    def save_spy():
        best.append(bool(ref.loss.is_best))
        orig_save()
    ref.save = save_spy
    for epoch in range(n_epochs):                      # train.py:72-92 (the loop body; the CLI/DB setup above it is out of scope)
        ref.loss.lr += [(ref.iter, ref.get_lr())]
        if epoch == 0:
            ref = ref_train.validate(ref, [(x.clone(), y.clone()) for x, y in va], len(va))
        ref = ref_train.train_epoch(ref, [(x.clone(), y.clone()) for x, y in tr], len(tr))
        ref = ref_train.validate(ref, [(x.clone(), y.clone()) for x, y in va], len(va))
        ref.sched.step()
        ref.epoch += 1
    sd = {k: v.clone() for k, v in w.items()}
    log = oracle.run_training(sd, cfg, tr, va, n_epochs, report=report)
    f = lambda rows: [[float(v) for v in r] for r in rows]
    rt, rv = f(ref.loss.train), f(ref.loss.valid)
    assert [r[0] for r in rt] == [r[0] for r in log.train] and [r[0] for r in rv] == [r[0] for r in log.valid]
    err = max(abs(a - b) for A, B in ((rt, log.train), (rv, log.valid)) for ra, rb in zip(A, B) for a, b in zip(ra, rb))
    print('  train entries %s, valid entries %s, max|diff| vs oracle driver %.3g, best events %s' % ([r[0] for r in rt], [r[0] for r in rv], err, best))
    assert err < 1e-3 and best == log.best_events
    assert abs(ref.get_lr() - 1e-4 * 0.9 ** n_epochs) < 1e-12
    # conditioning of the 6-step trajectory: the same (bit-pinned) algorithm with a different fp32 summation order
    torch.set_num_threads(1)
    log1 = oracle.run_training({k: v.clone() for k, v in w.items()}, cfg, tr, va, n_epochs, report=report)
    torch.set_num_threads(8)
    cond = max(abs(a - b) for A, B in ((log1.train, log.train), (log1.valid, log.valid)) for ra, rb in zip(A, B) for a, b in zip(ra, rb))
    print('  conditioning (1 thread vs 8 threads, same algorithm): max|diff| %.3g' % cond)
    with open(os.path.join(out_dir, 'driver.json'), 'w') as fh:
        json.dump({'train': rt, 'valid': rv, 'best_events': best, 'best_dice': float(ref.loss.best_dice), 'conditioning': cond,
                   'lr': f(ref.loss.lr), 'final_lr': ref.get_lr(),
                   'config': {'n_classes': n_cls, 'b': b, 'hw': hw, 'n_epochs': n_epochs, 'report': report, 'weight_salt': 6,
                              'train_seeds': [700, 701, 702], 'valid_seeds': [800, 801]}}, fh)


def golden_fp64(tag, arch, backbone, b, ch, hw, n_classes, out_dir):
    """fp64 TRUTH for the fixture of `tag`: the reference's own modules cast to double, on the same (fp32-valued) weights and inputs --
    eval logits and the clipped step-0 gradients of the tensors <tag>_grads.json names.  Stored next to them: how far the reference's
    fp32 results (the committed fixtures) are from this truth, per tensor (max and rms), which is the yardstick of
    tests/test_nets_gpu.py::test_error_against_fp64_truth -- |HIP - fp64| <= 1.25 x |reference_fp32 - fp64| on the logits, <= 5 x per gradient
    tensor and <= 2.5 x on their median (why not a flat 2 x: DESIGN.md section 3 (v), the ReLU-mask flips of either fp32-grade arithmetic)."""
    import oracle
    from oracle import step as ostep
    from tests import _data as D
    print('== fp64 truth:', tag)
    cw = D.class_weights(n_classes)
    ref = build_reference_model(arch, backbone, n_classes, ch, ostep.PX_RGB_MEAN, ostep.PX_RGB_STD, cw, False, tile=hw)
    spec = oracle.state_spec(arch, backbone, n_classes, 3 if arch == 'deeplab' else ch)
    x = D.tiles(100, b, ch, hw, hw)
    y = D.blob_masks(101, b, hw, hw, n_classes, cell=8)
    cfg = ostep.StepConfig(arch, backbone, n_classes, ch, dropout=False)
    w = ostep.calibrate_bn(oracle.formula_state(spec, salt=1), cfg, x.clone())      # the SAME fp32 weights / running statistics as golden_net
    ref.net.load_state_dict(w)
    ref.net.double()
    x64, y64 = ostep._prep(cfg, x.clone().double(), y.clone())                      # normalize_image / crop / x3 stack (model.py:300-311) in double
    arr32 = np.load(os.path.join(out_dir, tag + '.npz'))
    g32 = np.load(os.path.join(out_dir, tag + '_grads.npz'))
    gmeta = json.load(open(os.path.join(out_dir, tag + '_grads.json')))

    def dist(a32, a64):
        d = np.asarray(a32, np.float64) - a64
        return {'max': float(np.abs(d).max()), 'rms': float(np.sqrt((d * d).mean())), 'absmax': float(np.abs(a64).max())}

    ref.net.eval()
    with torch.no_grad():
        logits64 = ref.net(x64).numpy()
    meta = {'eval_logits': dist(arr32['eval_logits'], logits64)}
    fix = {'eval_logits': logits64}
    print('  eval logits: reference fp32 vs fp64 %s' % meta['eval_logits'])
    ref.net.train()
    for p in ref.net.parameters():
        p.grad = None
    loss = ref.crit(ref.net(x64), y64)                                              # model.py:314-317
    loss.backward()
    gnorm = float(torch.nn.utils.clip_grad_norm_(ref.net.parameters(), 0.5))        # model.py:326
    meta['loss'] = float(loss)
    meta['grad_norm_preclip'] = gnorm
    params = dict(ref.net.named_parameters())
    meta['grads'] = {}
    for k, m in gmeta.items():
        g = params[k].grad.numpy()
        if m.get('rows'):
            g = g[:m['rows']]
        fix['g::' + k] = g.copy()
        meta['grads'][k] = dist(g32['g::' + k], g)
        print('  grad %-44s reference fp32 vs fp64 %s' % (k, meta['grads'][k]))
    np.savez_compressed(os.path.join(out_dir, tag + '_fp64.npz'), **fix)
    with open(os.path.join(out_dir, tag + '_fp64.json'), 'w') as f:
        json.dump(meta, f)


ONLY_GRADS = False


def main():
    global ONLY_GRADS
    torch.set_num_threads(8)
    enter_reference()
    if '--only-grads' in sys.argv:          # write <net>_grads.npz|json only, leave the other fixtures as they are
        ONLY_GRADS = True
        sys.argv.remove('--only-grads')
    if '--only-fp64' in sys.argv:           # write <net>_fp64.npz|json only (needs the fp32 fixtures of the same nets)
        sys.argv.remove('--only-fp64')
        for tag, args in (('deeplab_resnet', ('deeplab', 'resnet', 2, 3, 96, 9)), ('deeplab_xception', ('deeplab', 'xception', 2, 1, 96, 11)),
                          ('unet', ('unet', 'resnet', 2, 3, 256, 9))):
            if not sys.argv[1:] or tag in sys.argv[1:]:
                golden_fp64(tag, *args, HERE)
        return
    which = sys.argv[1:] or ['multiloss', 'stitch', 'driver', 'deeplab_resnet', 'deeplab_xception', 'unet']
    if 'stitch' in which:
        golden_stitch(HERE)
    if 'driver' in which:
        golden_driver(HERE)
    if 'multiloss' in which:
        golden_multiloss(HERE)
    if 'deeplab_resnet' in which:
        golden_net('deeplab_resnet', 'deeplab', 'resnet', 2, 3, 96, 9, HERE)
    if 'deeplab_xception' in which:
        golden_net('deeplab_xception', 'deeplab', 'xception', 2, 1, 96, 11, HERE)
    if 'unet' in which:
        golden_net('unet', 'unet', 'resnet', 2, 3, 256, 9, HERE)


if __name__ == '__main__':
    main()
