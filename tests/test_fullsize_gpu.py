"""BASELINE.json configurations at FULL size on the MI355X (-m gpu), checked through size-independent properties
(the CPU oracle cannot run these sizes in seconds): shapes, finiteness, loss decomposition, BN-induced invariants,
and a descent check over a few steps."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _finite(model):
    g = model.arena.g
    return bool(torch.isfinite(g).all()) and bool(torch.isfinite(model.arena.p).all())


def test_config2_unet_512_bs16_ce_only(dev):
    """configs[1]: U-Net, 3-ch 512x512, 9 classes, bs=16, CE loss only."""
    from pylc_amd.model import Model, Meta
    from pylc_amd import runtime
    from tests import _data as D
    runtime.dropout_enabled = True
    x, y = D.learnable_tiles(31, 16, 512, 9, cell=32)
    model = Model(Meta(arch='unet', ce_weight=1.0, dice_weight=0.0, focal_weight=0.0, lr=1e-3), dev).build()
    logits = model.test(x[:2])[0]
    assert tuple(logits.shape) == (2, 9, 324, 324)                      # 512 - 188 (valid convs), unet.py:91-104
    losses = []
    for _ in range(4):
        model.train(x, y)
        losses.append(float(model.crit.ce))
        assert _finite(model)
    assert abs(losses[0] - math.log(9)) < 1.0 and losses[-1] < losses[0]   # starts near ln(9), descends
    # a conv bias that feeds a training-mode BatchNorm has an identically zero gradient
    gb = model.net.encoder[0].block.child(0).bias.grad
    gw = model.net.encoder[0].block.child(0).weight.grad
    assert float(gb.abs().max()) < 1e-4 * max(float(gw.abs().max()), 1e-12) + 1e-6
    # total == ce when dice/focal weights are zero
    yv = model.crit_target = model.crop_target(y.to(dev))
    with torch.no_grad():
        out = model.net(model.pack_input(x))
        all4 = model.crit.all_losses(out, yv)
    assert abs(float(all4[0]) - float(all4[1])) < 1e-6


def test_config5_xception_1024_gray_bs8(dev):
    """configs[4] (single-GPU part): DeepLabV3+/Xception, 1-channel 1024x1024 tiles, bs=8, full multi-loss."""
    from pylc_amd.model import Model, Meta
    from pylc_amd import runtime
    from tests import _data as D
    runtime.dropout_enabled = True
    x3, y = D.learnable_tiles(32, 8, 1024, 11, cell=64)
    x = x3[:, :1].contiguous()
    model = Model(Meta(backbone='xception', ch=1, n_classes=11, lr=1e-3), dev).build()
    first = None
    for it in range(3):
        model.train(x, y)
        tot = 0.5 * (float(model.crit.ce) + float(model.crit.dsc) + float(model.crit.fl))
        first = tot if first is None else first
        assert _finite(model)
    assert tot < first
    model.net.eval()                      # test.py:41 switches to eval before Model.test (a batch of 1 cannot train BatchNorm)
    logits = model.test(x[:1])[0]
    assert tuple(logits.shape) == (1, 11, 1024, 1024)
    # Dice is bounded in [0, 1]; the three reported terms recombine to the optimised total (loss.py:112)
    with torch.no_grad():
        l4 = model.crit.all_losses(model.net(model.pack_input(x[:2])), y[:2].to(dev))
    assert 0.0 <= float(l4[2]) <= 1.0
    assert abs(float(l4[0]) - 0.5 * float(l4[1] + l4[2] + l4[3])) < 1e-5


@pytest.mark.parametrize('n_cls', [9, 11])
def test_config3_config4_r101_512_bs32_deterministic(dev, n_cls):
    """configs[2] (9 classes, the headline) and configs[3] (11 classes, schema_b) at full size (DeepLabV3+/R101, 512x512, bs 32):
    the step is bit-reproducible -- every
    reduction has a fixed order, the operand ranges are order-independent maxima, the wgrad side stream and the prepared
    filter planes are synchronised -- and descends.  Two models from the same seed, three steps each."""
    from pylc_amd.model import Model, Meta
    from pylc_amd import runtime
    from tests import _data as D
    runtime.dropout_enabled = True
    x, y = D.learnable_tiles(33, 32, 512, n_cls, cell=32)
    runs = []
    for _ in range(2):
        torch.manual_seed(1234)
        runtime.manual_seed(7)
        model = Model(Meta(n_classes=n_cls, lr=1e-3), dev).build()
        losses = []
        for _ in range(3):
            model.train(x, y)
            losses.append((float(model.crit.ce), float(model.crit.dsc), float(model.crit.fl)))
            assert _finite(model)
        runs.append((losses, model.arena.p.clone(), model.arena.g.clone()))
        del model
    assert runs[0][0] == runs[1][0]                                       # losses bit-identical
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])      # all 59 M parameters and gradients
    assert abs(runs[0][0][0][0] - math.log(n_cls)) < 1.2 and sum(runs[0][0][-1]) < sum(runs[0][0][0])


def test_one_queue_and_side_stream_schedules_are_bit_identical(dev):
    """Round 5: the f16x3 step runs ONE queue (filter gradients on the compute stream, their split-K slab sums in one launch at the end of the
    backward pass -- runtime.side_stream_on(), ops.flush_slab_sums); the side-stream schedule of rounds 1-5 (a sum behind every wgrad) stays
    for precision mode 3.  Same kernels either way: two training steps of DeepLabV3+/R101 (256x256, bs 4) must leave the same bits in every
    parameter and gradient."""
    from pylc_amd.model import Model, Meta
    from pylc_amd import runtime
    from tests import _data as D
    x, y = D.learnable_tiles(35, 4, 256, 9, cell=32)
    prev = runtime.wgrad_side_stream
    runs = []
    try:
        for side in (False, True):
            runtime.wgrad_side_stream = side
            torch.manual_seed(1234)
            runtime.manual_seed(7)
            model = Model(Meta(n_classes=9, lr=1e-3), dev).build()
            for _ in range(2):
                model.train(x, y)
            torch.cuda.synchronize()
            runs.append((float(model.crit.ce), model.arena.p.clone(), model.arena.g.clone()))
            del model
    finally:
        runtime.wgrad_side_stream = prev
    assert runs[0][0] == runs[1][0]
    assert float(runs[0][2].abs().max()) > 0
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


def test_config5_inference_leg_full_res_image(dev):
    """configs[4], second half: sliding-window inference over a full-resolution image at the reference's size (pylc_gpu.ipynb:1057-1063:
    a 3453x4940 photograph fitted to 3072 x 4096; test.py:63 stride = tile / 2): one grayscale 3072 x 4096 image, 1024^2 tiles, stride
    512 -> 5 x 7 = 35 tiles in batches of 8 (test.py:69).  Properties: shape, class range, determinism, and agreement of the whole
    pylc_amd.inference.predict_image path (tile cutter + normalisation + batches 8,8,8,8,3 + stitch) with per-tile Model.test + stitch_logits."""
    import numpy as np
    from pylc_amd.model import Model, Meta
    from pylc_amd import inference
    torch.manual_seed(5)
    model = Model(Meta(backbone='xception', ch=1, n_classes=11), dev).build()
    model.net.eval()
    h, w, tile, stride = 3072, 4096, 1024, 512
    rs = np.random.RandomState(8)
    # a smooth synthetic "photograph" (random low-frequency field + noise) so that neighbouring tiles see correlated content
    low = torch.from_numpy(rs.uniform(0, 255, (1, 1, h // 128, w // 128)).astype(np.float32))
    img = torch.nn.functional.interpolate(low, size=(h, w), mode='bilinear', align_corners=True)[0]
    img = (img + torch.from_numpy(rs.normal(0, 8, (1, h, w)).astype(np.float32))).clamp(0, 255).round()
    mask = inference.predict_image(model, img, tile, stride, batch=8)
    assert tuple(mask.shape) == (h, w) and mask.dtype == torch.uint8 and int(mask.max()) < 11
    again = inference.predict_image(model, img, tile, stride, batch=8)
    assert torch.equal(mask, again)                                         # deterministic
    # the same through the per-tile public API: Model.test on every tile (batches of 5), logits stitched by stitch_logits
    rows, cols = inference.tile_grid(h, w, tile, stride)
    assert (rows, cols) == (5, 7)
    tiles = torch.stack([img[:, r * stride:r * stride + tile, c * stride:c * stride + tile] for r in range(rows) for c in range(cols)])
    logits = torch.cat([model.test(tiles[k:k + 5])[0].float() for k in range(0, rows * cols, 5)])
    ref = inference.stitch_logits(logits, rows, cols, tile, stride)
    agree = float((ref == mask).float().mean())
    print('full-res inference: %d tiles, %.4f%% of %d pixels agree with per-tile Model.test + stitch' % (rows * cols, 100 * agree, h * w))
    # batch composition differs (8,8,8,8,3 vs 5 x 7): eval-mode BatchNorm is per-sample, the conv kernels' per-tensor operand scale is per
    # batch -- fp32-grade either way, so only argmax near-ties may differ
    assert agree > 0.9999


def test_ranges_taken_inside_the_step_at_full_size(dev):
    """Size-independent checks of the range shortcuts on the configs[2] network (59 M parameters, 2048-channel backbone output):
    after an optimiser step the arena's parameter ranges -- written by the AdamW kernel itself (pylc_adamw_step_ranges) -- equal what
    pylc_amax_segments reads back from the updated arena, bit for bit; the image pool over the backbone's fp16 planes equals the pool of
    the converted tensor; and one step takes no more than three stand-alone range passes."""
    from pylc_amd.model import Model, Meta
    from pylc_amd import ops
    from pylc_amd.lib import lib, check, ptr, stream
    from tests import _data as D
    x, y = D.learnable_tiles(33, 8, 512, 9, cell=32)
    prev = lib.pylc_get_conv_precision()
    check(lib.pylc_set_conv_precision(2))
    try:
        model = Model(Meta(lr=1e-3), dev).build()
        model.train(x, y)
        ops.amax_passes[:] = [0, 0]
        model.train(x, y)
        assert ops.amax_passes[0] <= 3, ops.amax_passes
        a = model.arena
        ref = torch.zeros_like(a.amax)
        check(lib.pylc_amax_segments(ptr(a.p), ptr(a._segments), len(a.params), ptr(ref), stream()))
        assert torch.equal(a.amax, ref)
        want = torch.stack([p.detach().abs().max() for p in a.params])
        assert torch.equal(a.amax.view(torch.float32), want)
        f = ops.to_planes(torch.randn(8, 2048, 32, 32, device=dev).contiguous(memory_format=torch.channels_last) * 2.0)
        assert torch.equal(ops.global_avg_pool(f), ops.global_avg_pool(ops.from_planes(f)))
    finally:
        check(lib.pylc_set_conv_precision(prev))
