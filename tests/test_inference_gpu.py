"""Sliding-window inference on the GPU (-m gpu): tile extraction, stitching (reference reconstruct() semantics incl. its
overlap quirks), colourize + nearest resize -- against the committed reference fixture (tests/golden/stitch.*) and the
CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('tag', ['half', 'full'])
def test_stitch_matches_reference_fixture(dev, tag):
    import oracle
    from pylc_amd import inference
    cfg = json.load(open(os.path.join(HERE, 'stitch.json')))[tag]
    arr = np.load(os.path.join(HERE, 'stitch.npz'))
    rows, cols, tile, stride = cfg['rows'], cfg['cols'], cfg['tile'], cfg['stride']
    tiles = (np.random.RandomState(cfg['logit_seed']).standard_normal((rows * cols, 9, tile, tile)) * cfg['logit_scale']).astype(np.float32)
    mask = inference.stitch_logits(torch.from_numpy(tiles).to(dev), rows, cols, tile, stride).cpu().numpy()
    ref = arr[tag + '_mask']
    assert mask.shape == ref.shape
    scores = oracle.stitch_scores(tiles, rows, cols, tile, stride)
    top2 = np.sort(scores, axis=0)[-2:]
    decided = (top2[1] - top2[0]) > 1e-5            # expf vs np.exp may differ in the last ulp at exact near-ties
    assert decided.mean() > 0.999
    assert np.array_equal(mask[decided], ref[decided])
    # colourize + nearest resize (cv2.INTER_NEAREST semantics)
    pal = arr['palette']
    for oh, ow in ((ref.shape[0], ref.shape[1]), (ref.shape[0] * 2 + 3, ref.shape[1] * 3 - 5), (ref.shape[0] // 2, ref.shape[1] // 2 + 1)):
        rgb = inference.colourize(torch.from_numpy(ref).to(dev), pal, oh, ow).cpu().numpy()
        assert np.array_equal(rgb, oracle.colourize_resize(ref, pal, oh, ow))


def test_pack_tiles_matches_split_and_normalize(dev):
    import ctypes as C
    import oracle
    from oracle import step as ostep
    from pylc_amd import ops
    from pylc_amd.lib import lib, check, ptr, stream
    from tests import _data as D
    img = D.tiles(9, 1, 3, 96, 128)[0]
    tile, stride = 32, 16
    tiles, rows, cols = oracle.split_tiles(img.numpy(), tile, stride)
    want = oracle.normalize_image(torch.from_numpy(tiles), ostep.PX_RGB_MEAN, ostep.PX_RGB_STD)
    m = (C.c_float * 3)(*ostep.PX_RGB_MEAN); s = (C.c_float * 3)(*ostep.PX_RGB_STD)
    d = img.to(dev)
    got = ops.empty_nhwc(rows * cols, 4, tile, tile, dev)
    check(lib.pylc_image_pack_tiles(ptr(d), 3, 96, 128, tile, stride, 0, rows * cols, m, s, ptr(got), stream()))
    assert (got[:, :3].cpu() - want).abs().max().item() < 1e-7 and float(got[:, 3].abs().max()) == 0.0
    part = ops.empty_nhwc(5, 4, tile, tile, dev)                   # a batch from the middle of the tile list
    check(lib.pylc_image_pack_tiles(ptr(d), 3, 96, 128, tile, stride, 7, 5, m, s, ptr(part), stream()))
    assert torch.equal(part, got[7:12])


def test_predict_image_matches_oracle(dev):
    """Whole test.py path: split (stride tile/2) -> DeepLab eval forward -> stitch -> class mask."""
    import oracle
    from oracle import step as ostep
    from pylc_amd import inference, runtime
    from pylc_amd.model import Model, Meta
    from tests import _data as D
    runtime.dropout_enabled = False
    tile, stride = 64, 32
    img = D.learnable_tiles(21, 1, 192, 9, cell=16)[0][0, :, :128, :]          # [3,128,192]
    tiles_np, rows, cols = oracle.split_tiles(img.numpy(), tile, stride)
    tiles = torch.from_numpy(tiles_np)
    cfg = ostep.StepConfig('deeplab', 'resnet', 9, 3, dropout=False)
    w = ostep.calibrate_bn(oracle.formula_state(oracle.state_spec('deeplab', 'resnet', 9, 3), salt=2), cfg, tiles.clone())
    logits = ostep.test_step({k: v.clone() for k, v in w.items()}, cfg, tiles.clone()).numpy()
    scores = oracle.stitch_scores(logits, rows, cols, tile, stride)
    want = scores.argmax(0).astype(np.uint8)
    model = Model(Meta(), dev).build()
    model.net.load_state_dict(w)
    got = inference.predict_image(model, img, tile, stride, batch=8).cpu().numpy()
    assert got.shape == want.shape == (128, 192)
    top2 = np.sort(scores, axis=0)[-2:]
    # the HIP logits differ from the CPU ones by <= 1e-3 (fp32 summation order): compare where the stitched scores are
    # separated by more than that; interiors hold logits (O(1) margins), overlaps hold probabilities (smaller margins)
    decided = (top2[1] - top2[0]) > 4e-3
    agree = (got == want).mean()
    print('predict_image: %.2f%% pixels agree, %.1f%% decided' % (100 * agree, 100 * decided.mean()))
    assert decided.mean() > 0.4 and agree > 0.97
    assert np.array_equal(got[decided], want[decided])


def test_gpu_metrics_match_oracle(dev):
    """Confusion-matrix kernel + closed-form scores vs the oracle (itself equal to sklearn, tests/test_cpu_oracle.py)."""
    import oracle
    from oracle import metrics as om
    from pylc_amd import metrics
    rs = np.random.RandomState(3)
    for n_cls, n in ((9, 1 << 20), (11, 777777)):
        yt = rs.randint(0, n_cls, n)
        yp = np.where(rs.rand(n) < 0.6, yt, rs.randint(0, n_cls - 2, n))
        for tdt, pdt in ((torch.int64, torch.uint8), (torch.uint8, torch.uint8), (torch.int64, torch.int64)):
            cm = metrics.confusion_matrix(torch.from_numpy(yt).to(dev, tdt), torch.from_numpy(yp).to(dev, pdt), n_cls)
            want = om.confusion_matrix(yt, yp, n_cls)
            assert np.array_equal(cm.cpu().numpy(), want)
        got = metrics.scores(cm)
        ref = om.scores_from_confusion(want)
        assert abs(got['iou'] - oracle.weighted_jaccard(yt, yp, n_cls)) < 1e-12
        for k in ('f1', 'iou', 'mcc'):
            assert abs(got[k] - ref[k]) < 1e-12
        assert np.abs(got['cmatrix'] - ref['cmatrix']).max() < 1e-12


def test_uint8_feed_equals_float_path(dev):
    """uint8 tiles through the pinned, double-buffered feeder give bit-identical network inputs and training losses."""
    from pylc_amd import runtime
    from pylc_amd.data import TileFeeder
    from pylc_amd.model import Model, Meta
    from tests import _data as D
    import oracle
    runtime.dropout_enabled = False
    batches = [D.learnable_tiles(40 + i, 2, 64, 9) for i in range(4)]
    u8 = [(x.to(torch.uint8).numpy(), y.to(torch.uint8).numpy()) for x, y in batches]
    w = oracle.formula_state(oracle.state_spec('deeplab', 'resnet', 9, 3), salt=4)
    res = []
    for feed in (batches, TileFeeder(u8, dev)):
        model = Model(Meta(), dev).build()
        model.net.load_state_dict(w)
        out = []
        for x, y in feed:
            model.train(x, y)
            out.append((float(model.crit.ce), float(model.crit.dsc), float(model.crit.fl)))
        res.append(out)
    assert len(res[1]) == 4 and res[0] == res[1]
    x = batches[0][0]
    a = Model(Meta(), dev).pack_input(x)
    b = Model(Meta(), dev).pack_input(x.to(torch.uint8))
    assert torch.equal(a, b)


def test_fused_eval_batchnorm_is_bit_identical(dev):
    """Inference runs eval-mode BatchNorm, the residual add and the ReLU inside the conv epilogue (layers.conv_bn); the logits
    must equal, bit for bit, the ones from conv -> separate BatchNorm pass."""
    import oracle
    from oracle import step as ostep
    from pylc_amd import runtime
    from pylc_amd.model import Model, Meta
    from tests import _data as D
    x = D.tiles(91, 2, 3, 128, 160)
    cfg = ostep.StepConfig('deeplab', 'resnet', 9, 3, dropout=False)
    w = ostep.calibrate_bn(oracle.formula_state(oracle.state_spec('deeplab', 'resnet', 9, 3), salt=4), cfg, x.clone())
    model = Model(Meta(), dev).build()
    model.net.load_state_dict(w)
    model.net.eval()
    out = {}
    for fuse in (True, False):
        runtime.fuse_eval_bn = fuse
        out[fuse] = model.test(x)[0].clone()
    runtime.fuse_eval_bn = True
    assert torch.equal(out[True], out[False])
    want = ostep.test_step({k: v.clone() for k, v in w.items()}, cfg, x.clone())
    assert (out[True].float().cpu() - want).abs().max().item() < 1e-3
