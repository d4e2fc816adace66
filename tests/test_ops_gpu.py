"""Parity of every HIP kernel family against the fp32/fp64 PyTorch CPU op it replaces (-m gpu).

The conv kernels run on the exact-fp32 matrix pipe, so they are compared with an fp64 CPU convolution at a
tolerance of a few fp32 ulps of the accumulated magnitude; integer-like outputs (pool indices) are exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)


def to_dev_nhwc(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


CONV_CASES = [
    # cin, cout, k, stride, pad, dil, B, H, W, bias
    (64, 64, 3, 1, 1, 1, 2, 20, 24, False),       # layer1 3x3
    (64, 256, 1, 1, 0, 1, 2, 16, 16, False),      # 1x1 expand
    (128, 128, 3, 2, 1, 1, 2, 18, 22, False),     # strided 3x3 (odd/even sizes)
    (128, 128, 3, 2, 1, 1, 1, 17, 19, False),     # strided, odd input
    (256, 512, 1, 2, 0, 1, 2, 16, 16, False),     # downsample 1x1 s2
    (96, 64, 3, 1, 6, 6, 2, 16, 16, False),       # atrous, taps partly out of bounds (tap skipping)
    (64, 64, 3, 1, 18, 18, 1, 16, 16, False),     # atrous d=18 on a 16^2 map: only the centre tap is ever valid
    (304, 256, 3, 1, 1, 1, 1, 12, 12, False),     # Cin not a multiple of 32 (decoder concat)
    (40, 48, 1, 1, 0, 1, 2, 9, 7, False),         # decoder.conv1-like, Cout=48
    (256, 9, 1, 1, 0, 1, 2, 12, 12, True),        # 9-class head with bias (pitch 12)
    (64, 11, 1, 1, 0, 1, 1, 10, 10, True),        # 11-class head
    (64, 64, 3, 1, 0, 1, 2, 14, 14, True),        # U-Net valid conv with bias
    (128, 64, 3, 1, 0, 1, 1, 20, 20, True),
    (728, 728, 1, 1, 0, 1, 1, 8, 8, False),       # Xception pointwise (Cin % 32 != 0)
    (32, 2048, 1, 1, 0, 1, 4, 1, 1, False),       # ASPP image-pool branch: 1x1 spatial
    # >= 192 tiles of 256x128: these dispatch to the 8-wave 256x128 tile (fwd and dgrad)
    (128, 128, 3, 1, 1, 1, 4, 112, 112, False),
    (72, 256, 3, 1, 12, 12, 8, 64, 64, False),    # atrous with tap skipping, Cin % 16 != 0
    (256, 256, 1, 1, 0, 1, 8, 64, 64, True),
    (128, 128, 3, 2, 1, 1, 4, 225, 225, False),   # stride-2 dgrad parity classes on the big-tile path, odd size
]


@pytest.fixture(params=[2, 1, 0], ids=['f16x3', 'bf16x6', 'f32mfma'])
def conv_mode(request):
    from pylc_amd.lib import lib, check
    prev = lib.pylc_get_conv_precision()
    check(lib.pylc_set_conv_precision(request.param))
    yield request.param
    check(lib.pylc_set_conv_precision(prev))


def test_conv_precision_modes(dev):
    """The split arithmetics (bf16x6, f16x3) must be as accurate as the exact-fp32 matrix pipe: all are compared with an
    fp64 CPU convolution on the same data, with errors measured relative to sum|a*b| (the natural fp32 error scale)."""
    from pylc_amd import ops
    from pylc_amd.lib import lib, check
    b, cin, cout, hw = 2, 256, 256, 24
    x = rnd(101, b, cin, hw, hw)
    x[0, :, :4, :4] *= 1e4          # wide dynamic range inside one dot product
    x[1, :, 5:9, 5:9] *= 1e-4
    wt = rnd(102, cout, cin, 3, 3, scale=0.05)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    scale = F.conv2d(x.double().abs(), wt.double().abs(), None, 1, 1)
    errs = {}
    prev = lib.pylc_get_conv_precision()
    for mode in (0, 1, 2):
        check(lib.pylc_set_conv_precision(mode))
        y = ops.conv2d(to_dev_nhwc(x, dev), to_dev_nhwc(wt, dev), None, 1, 1, 1).double().cpu()
        e = ((y - ref).abs() / scale)
        errs[mode] = (e.mean().item(), e.max().item())
    check(lib.pylc_set_conv_precision(prev))
    print('conv error / sum|ab|: f32-mfma mean %.3g max %.3g | bf16x6 mean %.3g max %.3g | f16x3 mean %.3g max %.3g' % (errs[0] + errs[1] + errs[2]))
    assert errs[0][1] < 1e-6 and errs[1][1] < 1e-6 and errs[2][1] < 1e-6      # a few fp32 ulps of sum|ab|
    assert errs[1][0] < 1.5 * errs[0][0] + 1e-9                 # bf16x6 is no worse than the fp32 chain on average
    assert errs[2][0] < 1.5 * errs[0][0] + 1e-9                 # nor is f16x3


def test_conv_f16x3_dynamic_range(dev):
    """f16x3 scales by the tensor maximum, so its guarantee is per tensor: every pixel whose values lie within 2^29 of
    the largest one keeps fp32-grade relative accuracy (the gradient of a well-classified pixel next to a misclassified
    one), smaller ones degrade gracefully towards an absolute floor of 2^-51 of the maximum.  1x1 conv: each output pixel
    depends on one input pixel, so the error can be read per pixel."""
    from pylc_amd import ops
    from pylc_amd.lib import lib, check
    b, cin, cout, hw = 1, 256, 128, 32
    g = torch.Generator().manual_seed(7)
    expo = torch.linspace(0, -14, hw * hw).reshape(1, 1, hw, hw)         # pixel magnitudes from 1 down to 1e-14
    x = torch.randn(b, cin, hw, hw, generator=g) * 10.0 ** expo
    wt = torch.randn(cout, cin, 1, 1, generator=g) * 0.05
    ref = F.conv2d(x.double(), wt.double())
    scale = F.conv2d(x.double().abs(), wt.double().abs())
    prev = lib.pylc_get_conv_precision()
    out = {}
    for mode in (0, 2):
        check(lib.pylc_set_conv_precision(mode))
        out[mode] = ops.conv2d(to_dev_nhwc(x, dev), to_dev_nhwc(wt, dev), None, 1, 0, 1).double().cpu()
    check(lib.pylc_set_conv_precision(prev))
    rel = {m: ((out[m] - ref).abs() / scale).amax(dim=1).flatten() for m in out}       # per pixel, worst channel
    mag = (10.0 ** expo).flatten()
    inside = mag >= 2.0 ** -24          # elements ~N(0, mag) vs a tensor maximum of ~4.5: comfortably inside the 2^28 window
    print('per-pixel error / sum|ab|, pixels within 2^24 of the largest: f32-mfma %.3g, f16x3 %.3g; f16x3 at 1e-12 of the max: %.3g'
          % (rel[0][inside].max(), rel[2][inside].max(), rel[2][(mag < 2e-12) & (mag > 5e-13)].max()))
    assert rel[2][inside].max() < 1e-6
    assert rel[2][inside].max() < 2.0 * rel[0][inside].max()
    amax = x.abs().max().item()
    floor = 2.0 ** -50 * amax * wt.abs().sum(dim=1).max().item()          # absolute error bound for the tiny pixels
    assert ((out[2] - ref).abs().amax(dim=1).flatten()[~inside] < floor + 1e-6 * scale.amax(dim=1).flatten()[~inside]).all()


def test_amax_kernels(dev):
    """pylc_amax / pylc_amax_segments return the exact bit pattern of max|x| (pitched NHWC views included)."""
    from pylc_amd import ops
    from pylc_amd.lib import lib, check, ptr, stream
    x = rnd(5, 2, 20, 9, 7)
    x[1, 3, 4, 5] = -37.5
    xd = to_dev_nhwc(x, dev)
    got = ops.amax_of(xd)
    assert got.view(torch.float32).item() == 37.5
    buf = ops.zeros_nhwc(2, 48, 9, 7, dev)                      # a concat buffer: the view's pitch is 48
    view = buf[:, 8:28]
    view.copy_(xd)
    assert ops.amax_of(view).view(torch.float32).item() == 37.5
    odd = to_dev_nhwc(rnd(6, 1, 9, 5, 5), dev)                  # 9 channels in a 12-float pitch
    assert ops.amax_of(odd).view(torch.float32).item() == odd.abs().max().item()
    flat = torch.randn(1000, device=dev)
    offs = torch.tensor([0, 4, 300, 1000], dtype=torch.int64, device=dev)
    out = torch.empty(3, dtype=torch.int32, device=dev)
    check(lib.pylc_amax_segments(ptr(flat), ptr(offs), 3, ptr(out), stream()))
    want = [flat[0:4].abs().max().item(), flat[4:300].abs().max().item(), flat[300:].abs().max().item()]
    assert out.view(torch.float32).tolist() == want
    # many segments of very different sizes (a parameter arena: 64-float vectors next to multi-million-float filters), a non-zero start
    sizes = [64, 8, 20000, 256, 3, 8192, 8193, 1, 70000, 512, 40, 16384]
    edges = np.concatenate([[12], 12 + np.cumsum(sizes)])
    big = torch.randn(int(edges[-1]) + 5, device=dev)
    offs = torch.tensor(edges, dtype=torch.int64, device=dev)
    out = torch.empty(len(sizes), dtype=torch.int32, device=dev)
    check(lib.pylc_amax_segments(ptr(big), ptr(offs), len(sizes), ptr(out), stream()))
    assert out.view(torch.float32).tolist() == [big[a:b].abs().max().item() for a, b in zip(edges[:-1], edges[1:])]
    xd.add_(1.0)                                                # an in-place change invalidates the cached range
    assert ops.amax_of(xd).view(torch.float32).item() == xd.abs().max().item()


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_bwd(dev, case, conv_mode):
    from pylc_amd import ops
    cin, cout, k, stride, pad, dil, b, h, w, bias = case
    x = rnd(1, b, cin, h, w)
    wt = rnd(2, cout, cin, k, k, scale=(2.0 / (cin * k * k)) ** 0.5)
    bs = rnd(3, cout, scale=0.1) if bias else None
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    br = bs.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride, pad, dil)
    dy = rnd(4, *yr.shape)
    yr.backward(dy.double())

    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = to_dev_nhwc(wt, dev).requires_grad_(True)
    bd = bs.to(dev).requires_grad_(True) if bias else None
    y = ops.conv2d(xd, wd, bd, stride, pad, dil)
    assert tuple(y.shape) == tuple(yr.shape)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    assert rel_err(y, yr) < 2e-6
    assert rel_err(xd.grad, xr.grad) < 4e-6
    assert rel_err(wd.grad, wr.grad) < 5e-6
    if bias:
        assert rel_err(bd.grad, br.grad) < 2e-6
        # channels [cout, roundup4) of the padded logits buffer must be exact zeros
        cp = (cout + 3) & ~3
        if cp > cout:
            full = torch.as_strided(y, (y.shape[0], cp, y.shape[2], y.shape[3]), y.stride(), y.storage_offset())
            assert float(full[:, cout:].detach().abs().max()) == 0.0


@pytest.mark.parametrize('k,stride,pad,hw', [(7, 2, 3, 40), (3, 1, 0, 30), (3, 2, 1, 33)])
def test_conv_thin_input(dev, k, stride, pad, hw, conv_mode):
    """3-channel image convs (ResNet stem 7x7/2, U-Net first 3x3, Xception stem 3x3/2) through the 4-channel pack."""
    from pylc_amd import ops
    b, cout = 2, 64
    x = rnd(5, b, 3, hw, hw)
    wt = rnd(6, cout, 3, k, k, scale=0.1)
    wr = wt.double().requires_grad_(True)
    yr = F.conv2d(x.double(), wr, None, stride, pad)
    dy = rnd(7, *yr.shape)
    yr.backward(dy.double())
    x4 = ops.pack_nchw(x.to(dev), 4)
    wd = to_dev_nhwc(wt, dev).requires_grad_(True)
    y = ops.conv2d(x4, wd, None, stride, pad, 1)
    y.backward(dy.to(dev))
    assert rel_err(y, yr) < 2e-6
    assert rel_err(wd.grad, wr.grad) < 5e-6


@pytest.mark.parametrize('b,hw', [(2, (64, 128)), (3, (96, 64)), (1, (512, 512))])
def test_stem_patch_kernel(dev, b, hw):
    """conv_stem.hip (round 6): the ResNet stem (7x7 / stride 2 / pad 3, 3 -> 64 through the 4-channel pack; resnet.py:72) as a patch kernel --
    16 x 32 output patches, the input window split once into LDS planes, filter rows padded to eight taps.  Against float64 on the operands
    as given (the f16x3 bound of test_conv_precision_modes: max error 1e-6 of sum |a||b| -- the reduction ORDER differs from the generic
    thin-input path's, the products do not), against that generic path (pylc_debug_pp_flags bit 25) to rounding level, the BatchNorm
    statistics partials against sums over the output, borders (padding) and several patches per block included; two runs bit-identical."""
    from pylc_amd import ops
    from pylc_amd.lib import lib
    if lib.pylc_get_conv_precision() != 2:
        pytest.skip('the patch kernel is the f16x3 stem')
    h, w_ = hw
    x = rnd(5, b, 3, h, w_, scale=2.0)
    wt = rnd(6, 64, 3, 7, 7, scale=0.1)
    x4 = ops.pack_nchw(x.to(dev), 4)
    wd = to_dev_nhwc(wt, dev)
    outs = {}
    with torch.no_grad():
        for flags in (33554432, 0, 0):
            lib.pylc_debug_pp_flags(flags)
            y = ops.conv2d(x4, wd, None, 2, 3, 1, want_stats=True)
            torch.cuda.synchronize()
            outs.setdefault(flags, []).append((y.clone(), y._pylc_sums.clone()))
    lib.pylc_debug_pp_flags(0)
    (y_gen, s_gen), = outs[33554432]
    (y0, s0), (y1, s1) = outs[0]
    assert torch.equal(y0, y1) and torch.equal(s0, s1)
    assert s0.shape[0] == b * (h // 2 // 16) * (w_ // 2 // 32), 'the launch did not take the patch kernel'
    yr = F.conv2d(x.double(), wt.double(), None, 2, 3)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, 2, 3)
    err = ((y0.double().cpu() - yr).abs() / bound).max().item()
    err_gen = ((y_gen.double().cpu() - yr).abs() / bound).max().item()
    print('stem patch kernel %s x %d: max error / sum|a||b| %.3g (generic path %.3g); patch vs generic max|diff| %.3g of max|y| %.3g'
          % (hw, b, err, err_gen, (y0 - y_gen).abs().max().item(), y0.abs().max().item()))
    assert err < 1e-6 and err < 2 * err_gen + 1e-7
    yd = y0.double()
    sums = s0.double().sum(0)
    assert rel_err(sums[:64], yd.sum((0, 2, 3))) < 1e-5 and rel_err(sums[64:128], (yd * yd).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize('c,stride,dil,hw', [(64, 1, 1, 18), (128, 2, 1, 18), (728, 1, 1, 9), (1024, 1, 2, 10), (128, 2, 1, 17),
                                             (32, 1, 1, (7, 71)), (8, 1, 1, (33, 1)), (1536, 1, 1, (5, 40)), (256, 1, 1, (1, 9)),
                                             (16, 1, 1, (66, 130))])
def test_dwconv(dev, c, stride, dil, hw):
    """Depthwise 3x3 with fixed_padding; the stride-1/dilation-1 cases run the register-window strip kernels (odd heights,
    rows of several segments, one-pixel rows/columns, more channel vectors than threads), the others the general kernels."""
    from pylc_amd import ops
    b = 2
    hw = hw if isinstance(hw, tuple) else (hw, hw)
    x = rnd(8, b, c, *hw)
    wt = rnd(9, c, 1, 3, 3, scale=0.3)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    total = 2 * dil
    xp = F.pad(xr, (total // 2, total - total // 2, total // 2, total - total // 2))   # xception.py fixed_padding
    yr = F.conv2d(xp, wr, None, stride, 0, dil, groups=c)
    dy = rnd(10, *yr.shape)
    yr.backward(dy.double())
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = wt.to(dev).requires_grad_(True)
    y = ops.dwconv3x3(xd, wd, stride, dil)
    assert tuple(y.shape) == tuple(yr.shape)
    y.backward(dy.to(dev))
    assert rel_err(y, yr) < 2e-6
    assert rel_err(xd.grad, xr.grad) < 2e-6
    assert rel_err(wd.grad, wr.grad) < 5e-6
    # the statistics partials of the BatchNorm that follows (strip-kernel shapes only): same output bits, sums == sums over the output
    y2 = ops.dwconv3x3(xd, wd, stride, dil, want_stats=True)
    assert torch.equal(y2, y)
    part = getattr(y2, '_pylc_sums', None)
    assert (part is not None) == (stride == 1 and dil == 1 and hw[1] >= 2)
    if part is not None:
        sums = part.double().sum(0)
        yd = y.detach().double()
        assert rel_err(sums[:c], yd.sum((0, 2, 3))) < 1e-5 and rel_err(sums[c:], (yd * yd).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize('c,b,hw,relu,res', [(64, 4, 16, True, False), (256, 2, 9, True, True), (48, 3, 11, True, False),
                                             (728, 2, 6, False, False), (2048, 2, 4, True, True), (256, 5, 1, True, False)])
def test_bn_train(dev, c, b, hw, relu, res):
    from pylc_amd import ops
    y = rnd(11, b, c, hw, hw, scale=2.0) + 0.5
    g, be = 1 + 0.1 * rnd(12, c), 0.1 * rnd(13, c)
    rm, rv = 0.1 * rnd(14, c), 1 + 0.1 * rnd(15, c).abs()
    r = rnd(16, b, c, hw, hw) if res else None
    yr, gr, ber = y.double().requires_grad_(True), g.double().requires_grad_(True), be.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    rmr, rvr = rm.double().clone(), rv.double().clone()
    o = F.batch_norm(yr, rmr, rvr, gr, ber, True, 0.1, 1e-5)
    if res:
        o = o + rr
    if relu:
        o = F.relu(o)
    do = rnd(17, *o.shape)
    o.backward(do.double())
    yd = to_dev_nhwc(y, dev).requires_grad_(True)
    gd, bed = g.to(dev).requires_grad_(True), be.to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev), rv.to(dev)
    rd = to_dev_nhwc(r, dev).requires_grad_(True) if res else None
    od = ops.bn_act(yd, gd, bed, rmd, rvd, rd, relu, True)
    od.backward(do.to(dev))
    assert rel_err(od, o) < 5e-6
    assert rel_err(rmd, rmr) < 1e-6 and rel_err(rvd, rvr) < 2e-6
    assert rel_err(yd.grad, yr.grad) < 2e-5
    assert rel_err(gd.grad, gr.grad) < 2e-5 and rel_err(bed.grad, ber.grad) < 2e-5
    if res:
        assert rel_err(rd.grad, rr.grad) < 1e-6


def test_bn_eval(dev):
    from pylc_amd import ops
    c, b, hw = 128, 2, 7
    y = rnd(18, b, c, hw, hw)
    g, be, rm, rv = 1 + 0.1 * rnd(19, c), 0.1 * rnd(20, c), 0.1 * rnd(21, c), 1 + 0.1 * rnd(22, c).abs()
    o = F.relu(F.batch_norm(y.double(), rm.double(), rv.double(), g.double(), be.double(), False, 0.1, 1e-5))
    od = ops.bn_act(to_dev_nhwc(y, dev), g.to(dev), be.to(dev), rm.to(dev), rv.to(dev), None, True, False)
    assert rel_err(od, o) < 2e-6


@pytest.mark.parametrize('k,s,p,hw,c', [(3, 2, 1, 16, 64), (2, 2, 0, 21, 64), (2, 2, 0, 12, 128), (3, 2, 1, 15, 8)])
def test_maxpool(dev, k, s, p, hw, c):
    from pylc_amd import ops
    x = F.relu(rnd(23, 2, c, hw, hw))          # post-ReLU: exact-zero ties exercise the first-max rule
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, k, s, p)
    dy = rnd(24, *yr.shape)
    yr.backward(dy)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    y = ops.maxpool(xd, k, s, p)
    y.backward(dy.to(dev))
    assert torch.equal(y.cpu(), yr.detach())
    assert rel_err(xd.grad, xr.grad) < 1e-6


@pytest.mark.parametrize('c,h,w,oh,ow', [(256, 8, 8, 32, 32), (12, 16, 16, 64, 64), (256, 1, 1, 8, 8), (64, 12, 12, 24, 24),
                                         (9, 10, 12, 40, 48), (128, 7, 5, 14, 10),          # up-sampling: the separable backward
                                         (32, 16, 16, 24, 40), (16, 20, 20, 10, 30)])       # below x2 / down-sampling: the one-pass backward
def test_bilinear(dev, c, h, w, oh, ow):
    from pylc_amd import ops
    x = rnd(25, 2, c, h, w)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, size=(oh, ow), mode='bilinear', align_corners=True)
    dy = rnd(26, *yr.shape)
    yr.backward(dy.double())
    cp = (c + 3) & ~3
    xd = ops.zeros_nhwc(2, c, h, w, dev, cp)
    xd.copy_(x.to(dev))
    xd.requires_grad_(True)
    y = ops.bilinear(xd, oh, ow)
    y.backward(dy.to(dev))
    assert rel_err(y, yr) < 2e-6
    assert rel_err(xd.grad, xr.grad) < 5e-6


def test_gap(dev):
    from pylc_amd import ops
    x = rnd(27, 3, 2048, 6, 5)
    xr = x.double().requires_grad_(True)
    yr = F.adaptive_avg_pool2d(xr, 1)
    dy = rnd(28, *yr.shape)
    yr.backward(dy.double())
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    y = ops.global_avg_pool(xd)
    y.backward(dy.to(dev))
    assert rel_err(y, yr) < 2e-6 and rel_err(xd.grad, xr.grad) < 1e-6


def test_relu_dropout(dev):
    from pylc_amd import ops
    x = rnd(29, 2, 64, 9, 9)
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    y = ops.relu(xd)
    y.backward(torch.ones_like(y))
    assert torch.equal(y.cpu(), F.relu(x)) and torch.equal(xd.grad.cpu(), (x > 0).float())
    big = torch.ones(4, 256, 32, 32, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for p in (0.5, 0.1):
        d = ops.dropout(big, p, 1234)
        keep = (d != 0).float().mean().item()
        assert abs(keep - (1 - p)) < 0.01
        assert abs(d.max().item() - 1 / (1 - p)) < 1e-5
        d2 = ops.dropout(big, p, 1234)
        assert torch.equal(d, d2)                  # same seed -> same mask (backward regenerates it)
        d.sum().backward()
        assert torch.equal((big.grad != 0), (d != 0))
        big.grad = None
        assert not torch.equal(d, ops.dropout(big, p, 99))


@pytest.mark.parametrize('n_cls,weighted', [(9, False), (9, True), (11, False), (11, True)])
def test_multiloss_golden(dev, n_cls, weighted):
    """Against the committed reference outputs (tests/golden/multiloss.*), and the oracle on the same inputs."""
    import json, os
    from pylc_amd import ops
    import oracle
    from tests import _data as D
    here = os.path.join(os.path.dirname(__file__), 'golden')
    gold = json.load(open(os.path.join(here, 'multiloss.json')))
    arr = np.load(os.path.join(here, 'multiloss.npz'))
    key = 'c%d_%s' % (n_cls, 'w' if weighted else 'u')
    rs = np.random.RandomState(300 + n_cls)
    z = torch.from_numpy((rs.standard_normal((2, n_cls, 40, 36)) * 3).astype(np.float32))
    t = D.blob_masks(301 + n_cls, 2, 40, 36, n_cls, cell=4)
    cw = torch.from_numpy(D.class_weights(n_cls))
    zd = to_dev_nhwc(z, dev).requires_grad_(True)
    losses = ops.multiloss(zd, t.to(dev), cw.to(dev) if weighted else None, 0.5, 0.5, 0.5)
    losses[0].backward()
    got = losses.detach().cpu().tolist()
    for i, name in enumerate(('total', 'ce', 'dice', 'focal')):
        assert abs(got[i] - gold[key][name]) < 2e-6 * max(1.0, abs(gold[key][name])), (name, got[i], gold[key][name])
    g = torch.from_numpy(arr[key + '_grad'])
    assert rel_err(zd.grad, g) < 1e-5
    o = oracle.multiloss(z, t, (0.5, 0.5, 0.5), cw, weighted)
    assert abs(got[0] - o[0].item()) < 2e-6 * max(1.0, abs(got[0]))


def test_multiloss_saturated(dev):
    """Huge logit gaps: -log p_t must come from log-sum-exp, not log(softmax) (no inf / nan)."""
    from pylc_amd import ops
    import oracle
    z = torch.zeros(1, 9, 4, 4)
    z[:, 0] = 200.0
    t = torch.ones(1, 4, 4, dtype=torch.int64)
    zd = to_dev_nhwc(z, dev).requires_grad_(True)
    losses = ops.multiloss(zd, t.to(dev), None, 0.5, 0.5, 0.5)
    losses[0].backward()
    o = oracle.multiloss(z, t)
    assert torch.isfinite(losses).all() and torch.isfinite(zd.grad).all()
    assert abs(losses[1].item() - o[1].item()) < 1e-3      # ce = 200
    assert abs(losses[2].item() - o[2].item()) < 1e-5


def test_image_pack(dev):
    from pylc_amd import ops
    import oracle
    from tests import _data as D
    for ch in (3, 1):
        img = D.tiles(31, 2, ch, 20, 24)
        want = oracle.normalize_image(img, oracle.step.PX_RGB_MEAN, oracle.step.PX_RGB_STD)
        if ch == 1:
            want = torch.cat((want, want, want), 1)
            m = float(np.mean(np.asarray(oracle.step.PX_RGB_MEAN, np.float32)))
            s = float(np.mean(np.asarray(oracle.step.PX_RGB_STD, np.float32)))
            got = ops.image_pack(img.to(dev), [m] * 3, [s] * 3)
        else:
            got = ops.image_pack(img.to(dev), oracle.step.PX_RGB_MEAN, oracle.step.PX_RGB_STD)
        assert got.shape[1] == 4 and float(got[:, 3].abs().max()) == 0.0
        assert (got[:, :3].cpu() - want).abs().max().item() < 1e-7


def test_adamw_clip(dev):
    import ctypes as C
    from pylc_amd.lib import lib, check, ptr, stream
    n = 100003
    p0, g0 = rnd(32, n), rnd(33, n, scale=0.01)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-4, weight_decay=5e-5)
    p, g = p0.to(dev), g0.to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    out2 = torch.empty(2, device=dev)
    ws = torch.empty(lib.pylc_sqnorm_workspace_floats(n), device=dev)
    for step in range(1, 4):
        pr.grad = g0.clone() * step
        norm_ref = torch.nn.utils.clip_grad_norm_([pr], 0.5)
        opt.step()
        gs = (g * step).contiguous()
        check(lib.pylc_grad_norm_clip(ptr(gs), n, 0.5, ptr(out2), ptr(ws), stream()))
        check(lib.pylc_adamw_step(ptr(p), ptr(gs), ptr(m), ptr(v), n, ptr(out2), 1e-4, 0.9, 0.999, 1e-8, 5e-5, step, stream()))
        assert abs(out2[0].item() - norm_ref.item()) < 1e-5 * norm_ref.item()
        assert (p.cpu() - pr.detach()).abs().max().item() < 2e-7


@pytest.mark.parametrize('shape', [(64, 128, 3, 1, 2, 30, 26), (4, 64, 7, 2, 2, 40, 40), (256, 256, 1, 1, 8, 64, 64), (128, 48, 1, 1, 3, 17, 9)])
def test_conv_fused_bn_statistics(dev, shape, conv_mode):
    """Per-channel (sum, sum of squares) emitted by the conv epilogue == the same reduction over the stored output."""
    from pylc_amd import ops
    cin, cout, k, stride, b, h, w = shape
    x = rnd(61, b, cin, h, w)
    wt = rnd(62, cout, cin if cin != 4 else 3, k, k, scale=0.1)
    xd = to_dev_nhwc(x, dev) if cin != 4 else ops.pack_nchw(x[:, :3].to(dev), 4)
    y = ops.conv2d(xd, to_dev_nhwc(wt, dev), None, stride, k // 2, 1, want_stats=True)
    part = y._pylc_sums                                  # per-tile partials [rows][2*cout]
    sums = part.double().sum(0)
    yd = y.double()
    ref_s, ref_ss = yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))
    assert rel_err(sums[:cout], ref_s) < 1e-5 and rel_err(sums[cout:2 * cout], ref_ss) < 1e-5
    plain = ops.conv2d(xd, to_dev_nhwc(wt, dev), None, stride, k // 2, 1)
    assert torch.equal(plain, y)
    # the one-launch combine + coefficients equals the two-launch path bit for bit
    from pylc_amd.lib import lib, check, ptr, stream
    g, be = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    n = float(y.shape[0] * y.shape[2] * y.shape[3])
    a = [torch.empty(cout, device=dev) for _ in range(4)]
    bb = [torch.empty(cout, device=dev) for _ in range(4)]
    two = torch.empty(2 * cout, device=dev)
    check(lib.pylc_bn_stats_from_partial(ptr(part), part.shape[0], cout, ptr(two), stream()))
    check(lib.pylc_bn_finalize(ptr(two), n, cout, ptr(g), ptr(be), 1e-5, 0.1, 0, None, None, *[ptr(t) for t in a], stream()))
    check(lib.pylc_bn_finalize_from_partial(ptr(part), part.shape[0], n, cout, ptr(g), ptr(be), 1e-5, 0.1, 0, None, None,
                                            *[ptr(t) for t in bb], stream()))
    assert all(torch.equal(u, v) for u, v in zip(a, bb))


def test_prepared_filter_planes_match_inline_split(dev):
    """FlatArena prepares every conv filter once per step as two fp16 planes (forward and dgrad layouts); the conv kernels
    that copy those planes must give bit-identical results to the ones that split the fp32 filter themselves."""
    from pylc_amd import ops, layers
    from pylc_amd.lib import lib, check
    from pylc_amd.optim import FlatArena
    if lib.pylc_get_conv_precision() != 2:
        pytest.skip('prepared planes belong to the f16x3 arithmetic')
    from pylc_amd import runtime
    torch.manual_seed(3)
    runtime.no_planes = True            # this test is about the FILTER planes: activations stay fp32 (tests/test_planes_gpu.py covers the rest)
    for cin, cout, k, pad, b, hw in ((64, 128, 3, 1, 16, 64), (256, 136, 1, 0, 8, 128)):
        conv = layers.Conv2d(cin, cout, k, 1, pad, 1).to(dev)
        x = to_dev_nhwc(rnd(11, b, cin, hw, hw), dev).requires_grad_(True)
        dy = to_dev_nhwc(rnd(12, b, cout, hw, hw), dev)
        y0 = conv(x)
        y0.backward(dy)
        dx0, dw0 = x.grad.clone(), conv.weight.grad.clone()
        x.grad = None
        arena = FlatArena(conv)                      # re-homes the weight and attaches the prepared planes
        assert getattr(conv.weight, '_pylc_planes', None) is not None
        y1 = conv(x)
        y1.backward(dy)
        ops.sync_side_streams()
        assert torch.equal(y0, y1) and torch.equal(dx0, x.grad)
        assert torch.equal(dw0, conv.weight.grad)
        # the planes follow the weights: change them, refresh, compare against a fresh inline split
        with torch.no_grad():
            conv.weight.mul_(3.0)
        arena.refresh_ranges()
        y2 = conv(x)
        ref = layers.Conv2d(cin, cout, k, 1, pad, 1).to(dev)
        with torch.no_grad():
            ref.weight.copy_(conv.weight)
        assert torch.equal(y2, ref(x))
    runtime.no_planes = False


def test_filter_gradients_are_summed_when_backward_returns(dev):
    """on the one-queue schedule a conv's wgrad leaves split-K slabs and ONE launch sums the slabs of all layers
    (ops.flush_slab_sums).  A caller of layers.Conv2d + optim.FlatArena who never heard of that -- no ops.sync_side_streams(), no Model.train
    -- must still read SUMMED gradients from .grad as soon as backward() returns: the first noted sum queues an autograd-engine callback.
    Large pixel counts so that the split plan has several slabs; compared with the gradient of the same layer without an arena (which
    sums its slabs inside the call)."""
    from pylc_amd import ops, layers
    from pylc_amd.optim import FlatArena
    from pylc_amd import runtime
    if not runtime.batch_slab_sums or runtime.side_stream_on():
        pytest.skip('the batched slab sum belongs to the one-queue schedule')
    torch.manual_seed(5)
    for cin, cout, k, pad, b, hw in ((64, 64, 3, 1, 8, 128), (256, 64, 1, 0, 8, 128)):
        conv = layers.Conv2d(cin, cout, k, 1, pad, 1).to(dev)
        x = to_dev_nhwc(rnd(21, b, cin, hw, hw), dev).requires_grad_(True)
        dy = to_dev_nhwc(rnd(22, b, cout, hw, hw), dev)
        conv(x).backward(dy)
        dw0 = conv.weight.grad.clone()
        arena = FlatArena(conv)
        arena.g.zero_()
        for _ in range(2):                                # the second pass re-uses the cached table
            arena.g.fill_(float('nan'))
            conv(x).backward(dy)
            assert not ops._core._pending_slab_sums.get(dev.index or 0), 'a slab sum was still pending after backward() returned'
            assert torch.equal(conv.weight.grad, dw0)
        del arena


def test_conv_multi_round_launches_are_bit_identical(dev):
    """Launches with more than one 256x128 tile per CU: one block per tile (default) against persistent blocks walking a
    strided share of the tiles (debug flag 64) -- not a bit may change in the forward with fused BatchNorm statistics, the
    dgrad, or a dgrad accumulating into another conv's gradient, on row-aligned, ragged (M and N not multiples of the tile)
    and tap-skipping atrous geometries; every configuration runs twice (repeat launches are deterministic)."""
    from pylc_amd import ops
    from pylc_amd.lib import lib
    if lib.pylc_get_conv_precision() != 2:
        pytest.skip('variants of the f16x3 ping-pong kernel')
    cases = [(8, 128, 256, 3, 1, 1, 96),        # 576 tiles, 36 K-steps
             (16, 256, 1024, 1, 0, 1, 48),      # 1152 tiles of a short-K 1x1
             (5, 64, 200, 3, 1, 1, 90),         # ragged: M = 40500, N = 200
             (32, 64, 384, 3, 12, 12, 32)]      # atrous taps that fall into the padding are skipped per tile
    for b, cin, cout, k, pad, dil, hw in cases:
        x = to_dev_nhwc(rnd(31, b, cin, hw, hw), dev).requires_grad_(True)
        w1 = to_dev_nhwc(rnd(32, cout, cin, k, k, scale=0.05), dev).requires_grad_(True)
        w2 = to_dev_nhwc(rnd(33, cout, cin, k, k, scale=0.05), dev).requires_grad_(True)
        dy1 = to_dev_nhwc(rnd(34, b, cout, hw, hw), dev)
        dy2 = to_dev_nhwc(rnd(35, b, cout, hw, hw), dev)
        res = {}
        try:
            for flags in (0, 0, 64, 64):
                lib.pylc_debug_pp_flags(flags)
                x.grad = w1.grad = w2.grad = None
                link = ops.ResidualLink()
                y1 = ops.conv2d(x, w1, None, 1, pad, dil, want_stats=True, res_link=link)
                y2 = ops.conv2d(x, w2, None, 1, pad, dil, res_link=link)
                torch.autograd.backward([y1, y2], [dy1, dy2])       # the second dgrad to run accumulates into the first one's buffer
                ops.sync_side_streams()
                torch.cuda.synchronize()
                got = (y1.detach().clone(), y1._pylc_sums.clone(), y2.detach().clone(), x.grad.clone())
                if flags in res:
                    assert all(torch.equal(p, q) for p, q in zip(res[flags], got)), ('repeat launch differs', flags, hw)
                res[flags] = got
        finally:
            lib.pylc_debug_pp_flags(0)
        for p, q in zip(res[0], res[64]):
            assert torch.equal(p, q), (b, cin, cout, k, dil, hw)
        assert torch.isfinite(res[0][3]).all() and res[0][3].abs().max().item() > 0


def test_conv_kernel_variants_are_bit_identical(dev):
    """The scheduling variants of one arithmetic must not change a single bit: the 32x32x16-MFMA forward / dgrad kernels
    (ping-pong with swizzled or padded LDS rows, lock-step -- in the same reduction order) among themselves, and the wgrad fast paths vs the general path
    (same tiles, same reduction order).  The default 16x16x32-MFMA kernel sums each 32-deep step in one instruction instead
    of two, so it agrees with them to fp32 rounding only."""
    from pylc_amd import ops
    from pylc_amd.lib import lib, check
    if lib.pylc_get_conv_precision() != 2:
        pytest.skip('variants of the f16x3 kernels')
    for hw in (64, 44):            # 64: row-aligned reduction tiles (uniform-geometry wgrad); 44: per-row counters
        x = to_dev_nhwc(rnd(21, 8, 128, hw, hw), dev).requires_grad_(True)
        w = to_dev_nhwc(rnd(22, 256, 128, 3, 3, scale=0.05), dev).requires_grad_(True)
        dy = to_dev_nhwc(rnd(23, 8, 256, hw, hw), dev)
        res = {}
        for name, big, flags in (('default', 2, 0), ('pp32_swizzled', 2, 256), ('pp32_padded', 2, 128), ('lockstep', 1, 0),
                                 ('pp32_chunk_inner', 2, 256 | 4096), ('general_wgrad', 2, 8)):
            lib.pylc_debug_set_big_tile(big)
            lib.pylc_debug_pp_flags(flags)
            x.grad = w.grad = None
            y = ops.conv2d(x, w, None, 1, 1, 1)
            y.backward(dy)
            ops.sync_side_streams()
            torch.cuda.synchronize()
            res[name] = (y.detach().clone(), x.grad.clone(), w.grad.clone())
        lib.pylc_debug_set_big_tile(2)
        lib.pylc_debug_pp_flags(0)
        for a, b in zip(res['pp32_swizzled'], res['pp32_padded']):
            assert torch.equal(a, b), (hw, 'pp32_padded')
        # the ping-pong kernel walks a 3x3 reduction taps-innermost, the lock-step kernel channel-chunks-innermost: same bits
        # once the ping-pong kernel is told to use that order too (flag 4096), fp32 rounding apart otherwise
        for a, b in zip(res['pp32_chunk_inner'], res['lockstep']):
            assert torch.equal(a, b), (hw, 'lockstep')
        for a, b in zip(res['pp32_swizzled'][:2], res['pp32_chunk_inner'][:2]):
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item(), hw
        assert all(torch.equal(a, b) for a, b in zip(res['default'], res['general_wgrad'])), hw
        for a, b in zip(res['default'][:2], res['pp32_swizzled'][:2]):
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item(), hw
        assert torch.equal(res['default'][2], res['pp32_swizzled'][2])      # dw comes from the wgrad kernel either way


def test_gradient_link_equals_autograd_sum(dev):
    """A tensor read by several convs (projection blocks, low-level features, ASPP): with ops.grad_link the dgrads sum into
    one buffer in their epilogues; the result is bit-identical to letting autograd add the separate gradients (same
    arrival order, fp32 adds), including a stride-2 consumer whose dgrad touches only one pixel parity class."""
    from pylc_amd import ops
    torch.manual_seed(5)
    x0 = to_dev_nhwc(rnd(71, 4, 64, 32, 32), dev)
    ws = [to_dev_nhwc(rnd(72 + i, co, 64, k, k, scale=0.1), dev).requires_grad_(True) for i, (co, k) in enumerate(((32, 1), (128, 1), (64, 3), (48, 1)))]
    geo = ((1, 0, 1), (2, 0, 1), (1, 2, 2), (1, 0, 1))      # (stride, pad, dilation)

    def run(linked, extra=False):
        h = (x0 * 1.0).requires_grad_(True)              # non-leaf consumer input, as inside a network
        h.retain_grad()
        link = ops.grad_link(h) if linked else None
        outs = [ops.conv2d(h, w, None, s, p, d, res_link=link) for w, (s, p, d) in zip(ws, geo)]
        loss = sum((o * o).sum() for o in outs)
        if extra:
            loss = loss + (h * h).sum()                  # a consumer that is not a conv: autograd adds its part to the buffer
        for w in ws:
            w.grad = None
        loss.backward()
        assert link is None or (link.pending == 0 and link.buf is None)
        return h.grad.clone(), [w.grad.clone() for w in ws]

    g_ref, gw_ref = run(False)
    g_lnk, gw_lnk = run(True)
    assert torch.equal(g_ref, g_lnk)
    assert all(torch.equal(a, b) for a, b in zip(gw_ref, gw_lnk))
    g_ref, _ = run(False, True)
    g_lnk, _ = run(True, True)
    assert rel_err(g_lnk, g_ref) < 1e-6                  # same terms, different association of the fp32 adds


@pytest.mark.parametrize('hh', [22, 23])
def test_unet_skip_connection_fusion(dev, hh):
    """(hh = 23: an odd skip tensor, as the 121-pixel level of the 512^2 U-Net -- its last row / column lies in no pooling window.)
    unet.py:95-101,145-152: the skip tensor feeds the 2x2 max-pool and, centre-cropped, the decoder concat.  The 1x1 `up`
    conv writes into the concat buffer (conv2d(out=)), ops.crop_concat copies the crop behind it, and the crop's gradient is
    summed by the max-pool backward (pylc_maxpool_bwd_add) -- against torch's cat / slicing / max_pool2d on the CPU."""
    from pylc_amd import ops
    b, c, th = 2, 8, 12
    skip = rnd(81, b, c, hh, hh)
    z = rnd(82, b, 16, th, th)
    wt = rnd(83, c, 16, 1, 1, scale=0.3)
    bias = rnd(84, c)
    g_cat, g_pool = rnd(85, b, 2 * c, th, th), rnd(86, b, c, hh // 2, hh // 2)
    # reference
    sr, zr, wr, br = (t.double().requires_grad_(True) for t in (skip, z, wt, bias))
    o = (hh - th) // 2
    cat_r = torch.cat([F.conv2d(zr, wr, br), sr[:, :, o:o + th, o:o + th]], 1)
    pool_r = F.max_pool2d(sr, 2)
    ((cat_r * g_cat.double()).sum() + (pool_r * g_pool.double()).sum()).backward()
    # HIP path
    leaf = to_dev_nhwc(skip, dev).requires_grad_(True)
    sd = leaf * 1.0                                      # a non-leaf skip tensor, as in the network
    zd = to_dev_nhwc(z, dev).requires_grad_(True)
    wd = to_dev_nhwc(wt, dev).requires_grad_(True)
    bd = bias.to(dev).requires_grad_(True)
    pooled = ops.maxpool(sd, 2, 2, 0, link=ops.grad_link(sd))
    holder = [ops.empty_nhwc(b, 2 * c, th, th, dev)]
    up = ops.conv2d(zd, wd, bd, 1, 0, 1, out=holder)
    cat = ops.crop_concat(up, sd, holder, ops.grad_link(sd))
    assert cat.data_ptr() == holder[0].data_ptr() and rel_err(cat, cat_r) < 2e-6 and rel_err(pooled, pool_r) < 1e-7
    ((cat * to_dev_nhwc(g_cat, dev)).sum() + (pooled * to_dev_nhwc(g_pool, dev)).sum()).backward()
    assert rel_err(leaf.grad, sr.grad) < 1e-6
    assert rel_err(zd.grad, zr.grad) < 4e-6 and rel_err(wd.grad, wr.grad) < 5e-6 and rel_err(bd.grad, br.grad) < 2e-6
    # without a pool on the same tensor the crop gradient is materialised (zero-padded)
    leaf2 = to_dev_nhwc(skip, dev).requires_grad_(True)
    holder = [ops.empty_nhwc(b, 2 * c, th, th, dev)]
    up = ops.conv2d(zd, wd, bd, 1, 0, 1, out=holder)
    cat = ops.crop_concat(up, leaf2 * 1.0, holder, None)
    (cat * to_dev_nhwc(g_cat, dev)).sum().backward()
    ref = torch.zeros_like(skip)
    ref[:, :, o:o + th, o:o + th] = g_cat[:, c:]
    assert rel_err(leaf2.grad, ref) < 1e-7


@pytest.mark.parametrize('cin,cout,b,h,w', [(128, 64, 2, 12, 10), (64, 64, 1, 7, 9)])
def test_conv_transpose2x2(dev, cin, cout, b, h, w):
    """U-Net up_mode='upconv' (unet.py:132-133): nn.ConvTranspose2d(k=2, s=2) as the dgrad / fwd / wgrad of a 2x2 stride-2 conv."""
    from pylc_amd import ops
    x = rnd(31, b, cin, h, w)
    wt = rnd(32, cin, cout, 2, 2, scale=0.1)
    bias = rnd(33, cout, scale=0.1)
    xr, wr, br = x.double().requires_grad_(True), wt.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, br, stride=2)
    do = rnd(34, *ref.shape)
    ref.backward(do.double())
    xd = to_dev_nhwc(x, dev).requires_grad_(True)
    wd = to_dev_nhwc(wt, dev).requires_grad_(True)
    bd = bias.to(dev).requires_grad_(True)
    y = ops.conv_transpose2x2(xd, wd, bd)
    y.backward(to_dev_nhwc(do, dev))
    ops.sync_side_streams()
    assert tuple(y.shape) == (b, cout, 2 * h, 2 * w)
    assert rel_err(y, ref) < 3e-6
    assert rel_err(xd.grad, xr.grad) < 3e-6 and rel_err(wd.grad, wr.grad) < 3e-6 and rel_err(bd.grad, br.grad) < 3e-6


def test_unet_upconv_mode_matches_torch(dev):
    """The whole U-Net with up_mode='upconv' against a plain torch restatement of unet.py built from the same state dict (forward, fp64)."""
    from pylc_amd import UNet, runtime
    runtime.dropout_enabled = False
    torch.manual_seed(5)
    net = UNet(in_channels=3, n_classes=9, up_mode='upconv', dropout=0.5).to(dev).eval()
    sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    x = rnd(41, 1, 3, 188 + 16, 188 + 16)

    def block(t, p):
        for i in ('0', '3'):
            t = F.conv2d(t, sd[p + '.block.%s.weight' % i], sd[p + '.block.%s.bias' % i])
            j = str(int(i) + 1)
            t = F.relu(F.batch_norm(t, sd[p + '.block.%s.running_mean' % j], sd[p + '.block.%s.running_var' % j], sd[p + '.block.%s.weight' % j],
                                    sd[p + '.block.%s.bias' % j], False, 0.1, 1e-5))
        return t
    t, skips = x.double(), []
    for i in range(5):
        t = block(t, 'encoder.%d' % i)
        if i != 4:
            skips.append(t)
            t = F.max_pool2d(t, 2)
    for i in range(4):
        up = F.conv_transpose2d(t, sd['decoder.%d.up.weight' % i], sd['decoder.%d.up.bias' % i], stride=2)
        br = skips[-i - 1]
        dy, dx = (br.shape[2] - up.shape[2]) // 2, (br.shape[3] - up.shape[3]) // 2
        t = block(torch.cat([up, br[:, :, dy:dy + up.shape[2], dx:dx + up.shape[3]]], 1), 'decoder.%d.conv_block' % i)
    ref = F.conv2d(t, sd['last.weight'], sd['last.bias'])
    with torch.no_grad():
        got = net(x.to(dev))
    assert tuple(got.shape) == tuple(ref.shape)
    assert rel_err(got, ref) < 2e-5


def test_image_pack_grayscale_default_branch(dev):
    """Model.normalize_image(default=True) on a 1-channel image omits the division by 255 (models/model.py:428-430): reproduced."""
    from pylc_amd.model import Model, Meta
    m = Model(Meta(ch=1, backbone='xception', n_classes=11), dev)
    x = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (2, 1, 16, 20)).astype(np.float32))
    got = m.pack_input(x, default=True)
    want = (x.numpy().astype('float32') - m.meta.px_grayscale_mean) / m.meta.px_grayscale_std          # the reference's expression
    for c in range(3):
        assert np.abs(got[:, c].cpu().numpy() - want[:, 0]).max() <= 1e-6 * np.abs(want).max()
    assert float(got[:, 3].abs().max()) == 0.0
    x8 = x.to(torch.uint8)
    assert torch.equal(m.pack_input(x8, default=True), got)
    ref255 = ((x.numpy() - np.mean(np.asarray(m.meta.px_mean, np.float32))) / np.mean(np.asarray(m.meta.px_std, np.float32))) / 255
    assert np.abs(m.pack_input(x)[:, 0].cpu().numpy() - ref255[:, 0]).max() <= 1e-6 * np.abs(ref255).max()


def test_concat_by_slice_equals_cat(dev):
    """aspp.py:80 / decoder.py:47: BatchNorm passes and the interpolation write their channel ranges of one buffer (`into=`) and
    ops.concat_slices() hands the buffer on -- values, range tag and every gradient equal to producing the parts and torch.cat."""
    from pylc_amd import ops, runtime
    prev = runtime.dropout_enabled
    runtime.dropout_enabled = True
    b, h, w = 2, 12, 20
    ya, yb = rnd(1, b, 64, h, w, scale=2.0), rnd(2, b, 48, h, w) + 0.3
    g = rnd(3, b, 32, 5, 7)
    ga, be = 1 + 0.1 * rnd(4, 64), 0.1 * rnd(5, 64)
    gb, bb = 1 + 0.1 * rnd(6, 48), 0.1 * rnd(7, 48)
    do = rnd(8, b, 144, h, w).to(dev).contiguous(memory_format=torch.channels_last)
    got = {}
    for mode in ('cat', 'slices'):
        cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
        leaves = [cl(ya), ga.to(dev), be.to(dev), cl(yb), gb.to(dev), bb.to(dev), cl(g)]
        for t in leaves:
            t.requires_grad_(True)
        stats = [torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(48, device=dev), torch.ones(48, device=dev)]
        if mode == 'cat':
            parts = [ops.bn_act(leaves[0], leaves[1], leaves[2], stats[0], stats[1], None, True, True),
                     ops.bilinear(leaves[6], h, w),
                     ops.bn_act(leaves[3], leaves[4], leaves[5], stats[2], stats[3], None, True, True, drop=(0.5, 77))]
            out = ops.cat_channels(parts)
        else:
            buf = [ops.empty_nhwc(b, 144, h, w, dev)]
            parts = [ops.bn_act(leaves[0], leaves[1], leaves[2], stats[0], stats[1], None, True, True, into=(buf, 0)),
                     ops.bilinear(leaves[6], h, w, into=(buf, 64)),
                     ops.bn_act(leaves[3], leaves[4], leaves[5], stats[2], stats[3], None, True, True, drop=(0.5, 77), into=(buf, 96))]
            out = ops.concat_slices(buf, parts)
            assert out.data_ptr() == buf[0].data_ptr()
        tag = ops.amax_of(out).clone() if ops.ranges_needed() else None
        (out * 1.0).backward(do)
        torch.cuda.synchronize()
        got[mode] = [out.detach().clone(), tag] + [t.grad.clone() for t in leaves] + [s.clone() for s in stats]
    runtime.dropout_enabled = prev
    for i, (a, c) in enumerate(zip(got['cat'], got['slices'])):
        if a is not None:
            assert torch.equal(a, c), i
    with pytest.raises(Exception):
        buf = [ops.empty_nhwc(b, 144, h, w, dev)]
        ops.concat_slices(buf, [ops.bilinear(g.to(dev).contiguous(memory_format=torch.channels_last), h, w, into=(buf, 0))])        # does not cover the buffer


@pytest.mark.parametrize('path', ['standalone', 'conv_fused', 'conv_bias'])
def test_bn_train_mean_1e3_sigma(dev, path):
    """|mean| / sigma ~ 10^3 per channel: sum(x^2)/n - mean^2 from fp32 sums carries no variance bits there; the finalize kernels
    re-measure such channels in a second pass (bn.hip: kRefineRatio), as torch.nn.BatchNorm2d's two-pass / Welford statistics do.
    Both statistics sources: the standalone reduction over y, and the per-tile partials of the conv epilogue."""
    from pylc_amd import ops
    c, b, hw = 64, 4, 24
    if path == 'standalone':
        y = (1000.0 + rnd(71, b, c, hw, hw)) * (1 + torch.arange(c).float().view(1, c, 1, 1) / c)       # mean 1000..2000, sigma 1..2
        y[:, 5] = rnd(72, b, hw, hw) * 3.0 + 0.25                                                           # one well-conditioned channel
        yd = to_dev_nhwc(y, dev).requires_grad_(True)
        y_ref = y.double()
    elif path == 'conv_bias':
        # the common case (U-Net's first layer on the 1/255-scaled input): a bias that dwarfs the spread.  The conv epilogue takes its
        # statistics of (y - bias), so this costs no second pass
        x = 0.01 * rnd(78, b, 64, hw, hw)
        wt = 0.05 * rnd(79, c, 64, 3, 3)
        bias = (5.0 + rnd(80, c).abs()).to(dev)
        conv_out = ops.conv2d(to_dev_nhwc(x, dev), to_dev_nhwc(wt, dev), bias, 1, 0, 1, want_stats=True)
        assert getattr(conv_out._pylc_sums, '_pylc_shift', None) is not None
        y_ref = conv_out.detach().double().cpu()
        ratio = (y_ref.mean((0, 2, 3)).abs() / y_ref.std((0, 2, 3))).min().item()
        assert ratio > 300, ratio
        yd = conv_out.detach().requires_grad_(True)
        yd._pylc_sums = conv_out._pylc_sums
    else:
        # conv outputs with a large common mode: positive inputs x positive filters (mean ~ 64 * 9 * 1.0 = 576, sigma ~ 0.3); no padding,
        # so that every output sums all 576 products
        x = 1.0 + 0.01 * rnd(73, b, 64, hw, hw)
        wt = 1.0 + 0.01 * rnd(74, c, 64, 3, 3)
        conv_out = ops.conv2d(to_dev_nhwc(x, dev), to_dev_nhwc(wt, dev), None, 1, 0, 1, want_stats=True)
        assert getattr(conv_out, '_pylc_sums', None) is not None
        y_ref = conv_out.detach().double().cpu()
        ratio = (y_ref.mean((0, 2, 3)).abs() / y_ref.std((0, 2, 3))).min().item()
        assert ratio > 300, ratio
        yd = conv_out.detach().requires_grad_(True)
        yd._pylc_sums = conv_out._pylc_sums
    g, be = 1 + 0.1 * rnd(75, c), 0.1 * rnd(76, c)
    yr, gr, ber = y_ref.clone().requires_grad_(True), g.double().requires_grad_(True), be.double().requires_grad_(True)
    rmr, rvr = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    z = F.batch_norm(yr, rmr, rvr, gr, ber, True, 0.1, 1e-5)
    o = F.relu(z)
    do = rnd(77, *o.shape)
    # elements within the input-rounding distance of the ReLU kink may legitimately fall on either side: they take no part in the
    # gradient comparison (their upstream gradient is zeroed on both sides)
    do = do * (z.detach().abs() > 2e-3).float()
    o.backward(do.double())
    gd, bed = g.to(dev).requires_grad_(True), be.to(dev).requires_grad_(True)
    rmd, rvd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    od = ops.bn_act(yd, gd, bed, rmd, rvd, None, True, True)
    od.backward(to_dev_nhwc(do, dev))
    # y itself is fp32: (y - mean) carries |mean| 2^-24 ~ 1e-4 sigma of input rounding, which bounds what any fp32 BatchNorm can return
    assert rel_err(od, o) < 3e-4, rel_err(od, o)
    assert rel_err(rvd, rvr) < 1e-4, rel_err(rvd, rvr)                  # the variance itself: no cancellation left
    assert rel_err(rmd, rmr) < 1e-6
    assert rel_err(yd.grad, yr.grad) < 2e-3 and rel_err(gd.grad, gr.grad) < 5e-4 and rel_err(bed.grad, ber.grad) < 1e-5
    # torch's own fp32 BatchNorm on the device, for scale: ours must not be worse than 4x its error
    yt = yd.detach().clone().requires_grad_(True)
    ot = F.relu(F.batch_norm(yt, torch.zeros(c, device=dev), torch.ones(c, device=dev), g.to(dev), be.to(dev), True, 0.1, 1e-5))
    assert rel_err(od, o) <= 4 * rel_err(ot, o) + 1e-5, (rel_err(od, o), rel_err(ot, o))


def test_range_tags_go_stale_with_the_tensor(dev):
    """An operand range tag (ops.tag_amax) is trusted only while the tensor's version counter is the one it was attached at: an
    in-place change after tagging -- autograd's gradient accumulation does that -- must send amax_of() back to a read pass, and the
    tag must not travel on through inherit_amax / cat_channels.  A too-small range would overflow the fp16 pieces (inf/NaN), so the
    conv on the modified tensor is checked against fp64 as well."""
    from pylc_amd import ops
    from pylc_amd.lib import lib, check
    prev = lib.pylc_get_conv_precision()
    check(lib.pylc_set_conv_precision(2))
    try:
        bits = lambda t: float(t.view(torch.float32).item())
        a = to_dev_nhwc(rnd(81, 2, 32, 12, 12), dev)
        b = to_dev_nhwc(rnd(82, 2, 32, 12, 12) * 0.5, dev)
        passes0 = ops.amax_passes[0]
        ra = bits(ops.amax_of(a))
        assert abs(ra - a.abs().max().item()) == 0 and ops.amax_passes[0] == passes0 + 1
        assert bits(ops.amax_of(a)) == ra and ops.amax_passes[0] == passes0 + 1           # fresh tag: reused, no pass
        ops.amax_of(b)
        cat = ops.cat_channels([a, b])
        assert bits(ops.amax_of(cat)) == ra and ops.amax_passes[0] == passes0 + 2         # max of the parts' tags, no pass
        up = ops.bilinear(a, 24, 24)
        assert getattr(up, '_pylc_amax', None) is not None and bits(ops.amax_of(up)) == ra
        # in-place change: 1000x larger values, version bumped
        a.mul_(1000.0)
        assert bits(ops.amax_of(a)) == a.abs().max().item() and ops.amax_passes[0] == passes0 + 3   # stale tag ignored: a read pass
        a2 = to_dev_nhwc(rnd(83, 2, 32, 12, 12), dev)
        ops.amax_of(a2)
        a2.add_(500.0)
        cat2 = ops.cat_channels([a2, b])                                                  # one part stale: the result carries no tag
        assert getattr(cat2, '_pylc_amax', None) is None
        assert bits(ops.amax_of(cat2)) == cat2.abs().max().item()
        up2 = ops.bilinear(a2, 24, 24)                                                    # stale source: nothing inherited
        assert getattr(up2, '_pylc_amax', None) is None
        # and the conv that consumes the modified tensor is right (a stale range of ~4 against values of ~4000 would saturate fp16)
        wt = rnd(84, 64, 32, 3, 3, scale=0.05)
        y = ops.conv2d(a, to_dev_nhwc(wt, dev), None, 1, 1, 1)
        ref = F.conv2d(a.double().cpu(), wt.double(), None, 1, 1, 1)
        assert bool(torch.isfinite(y).all()) and rel_err(y, ref) < 3e-6
    finally:
        check(lib.pylc_set_conv_precision(prev))


def test_depthwise_gradient_link(dev):
    """xception.py:88-97: a block input feeds the first depthwise conv of `rep` AND the skip path; with ops.grad_link the two gradients meet
    in one buffer (the depthwise dgrad accumulates: pylc_dwconv3x3_dgrad_acc) -- equal to autograd's sum, for the strip kernel (stride 1)
    and the generic one (stride 2), with an identity skip through a BatchNorm residual and with a 1x1 skip conv."""
    from pylc_amd import ops
    for stride, hw in ((1, 24), (2, 25)):
        x0 = rnd(91, 2, 64, hw, hw)
        wd0 = rnd(92, 64, 1, 3, 3)
        ws0 = rnd(93, 64, 64, 1, 1, scale=0.1)
        g0, b0 = 1 + 0.1 * rnd(94, 64), 0.1 * rnd(95, 64)
        res = {}
        for mode in ('autograd', 'link'):
            x = to_dev_nhwc(x0, dev).requires_grad_(True)
            wd = wd0.to(dev).requires_grad_(True)
            ws = to_dev_nhwc(ws0, dev).requires_grad_(True)
            g, be = g0.to(dev).requires_grad_(True), b0.to(dev).requires_grad_(True)
            link = ops.grad_link(x) if mode == 'link' else None
            y = ops.dwconv3x3(x, wd, stride, 1, link)
            skip = ops.conv2d(x, ws, None, stride, 0, 1, res_link=link)
            out = ops.bn_act(y, g, be, torch.zeros(64, device=dev), torch.ones(64, device=dev), skip, True, True)
            out.backward(to_dev_nhwc(rnd(96, *out.shape), dev))
            ops.sync_side_streams()
            torch.cuda.synchronize()
            res[mode] = (out.detach().clone(), x.grad.clone(), wd.grad.clone(), ws.grad.clone())
            if link is not None:
                assert link.pending == 0 and link.buf is None
        for a, c in zip(res['autograd'], res['link']):
            assert rel_err(c, a) < 1e-6
    # identity skip: the BatchNorm's residual gradient is parked, the depthwise dgrad adds to it
    x0 = rnd(97, 2, 32, 20, 20)
    wd0 = rnd(98, 32, 1, 3, 3)
    res = {}
    for mode in ('autograd', 'link'):
        x = to_dev_nhwc(x0, dev).requires_grad_(True)
        wd = wd0.to(dev).requires_grad_(True)
        g, be = torch.ones(32, device=dev, requires_grad=True), torch.zeros(32, device=dev, requires_grad=True)
        link = ops.grad_link(x) if mode == 'link' else None
        y = ops.dwconv3x3(x, wd, 1, 1, link)
        out = ops.bn_act(y, g, be, torch.zeros(32, device=dev), torch.ones(32, device=dev), x, True, True, res_link=link)
        out.backward(to_dev_nhwc(rnd(99, *out.shape), dev))
        torch.cuda.synchronize()
        res[mode] = (out.detach().clone(), x.grad.clone(), wd.grad.clone())
    for a, c in zip(res['autograd'], res['link']):
        assert rel_err(c, a) < 1e-6


@pytest.mark.parametrize('hh,nplanes', [(22, 2), (23, 2), (24, 1)])
def test_upsample2_crop_concat_planes(dev, hh, nplanes):
    """ops.upsample2_crop_concat: cat([upsample_x2(z), center_crop(bridge)], 1) written as ONE fp16-plane tensor (unet.py:135-152) against
    torch on the CPU -- values to the planes format's resolution, gradients of z and of the bridge (directly, and summed by the max-pool
    backward through the shared link)."""
    from pylc_amd import ops
    from pylc_amd.lib import lib, check
    prev = lib.pylc_get_conv_precision()
    check(lib.pylc_set_conv_precision(3 if nplanes == 1 else 2))
    prev_min = ops.PLANES_MIN_PIXELS
    ops.PLANES_MIN_PIXELS = 0
    try:
        b, c1, c2, h = 2, 16, 8, 6
        z, bridge = rnd(91, b, c1, h, h), rnd(92, b, c2, hh, hh)
        g_cat, g_pool = rnd(93, b, c1 + c2, 2 * h, 2 * h), rnd(94, b, c2, hh // 2, hh // 2)
        zr, br = z.double().requires_grad_(True), bridge.double().requires_grad_(True)
        o = (hh - 2 * h) // 2
        cat_r = torch.cat([F.interpolate(zr, scale_factor=2, mode='bilinear', align_corners=True), br[:, :, o:o + 2 * h, o:o + 2 * h]], 1)
        ((cat_r * g_cat.double()).sum() + (F.max_pool2d(br, 2) * g_pool.double()).sum()).backward()
        zd = to_dev_nhwc(z, dev).requires_grad_(True)
        leaf = to_dev_nhwc(bridge, dev).requires_grad_(True)
        bd = leaf * 1.0
        link = ops.grad_link(bd)
        pooled = ops.maxpool(bd, 2, 2, 0, link=link)
        cat = ops.upsample2_crop_concat(zd, bd, link)
        assert cat is not None and ops.is_planes(cat)
        tol = (2.0 ** -9 if nplanes == 1 else 2.0 ** -20) * float(cat_r.detach().abs().max())
        assert (ops.as_nhwc(cat).double().cpu() - cat_r.detach()).abs().max().item() <= tol
        ((ops.export_activation(cat) * to_dev_nhwc(g_cat, dev)).sum() + (pooled * to_dev_nhwc(g_pool, dev)).sum()).backward()       # (the differentiable conversion)
        assert rel_err(zd.grad, zr.grad) < 2e-6 and rel_err(leaf.grad, br.grad) < 1e-6
    finally:
        ops.PLANES_MIN_PIXELS = prev_min
        check(lib.pylc_set_conv_precision(prev))


def test_maxpool_planes_output(dev):
    """ops.maxpool(out_planes=True): the pooled tensor written as fp16 planes equals the fp32 result to the format's resolution, and the
    backward is the ordinary one."""
    from pylc_amd import ops
    prev_min = ops.PLANES_MIN_PIXELS
    ops.PLANES_MIN_PIXELS = 0
    try:
        x = F.relu(rnd(33, 2, 16, 21, 14))
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool2d(xr, 2)
        dy = rnd(34, *yr.shape)
        yr.backward(dy)
        xd = to_dev_nhwc(x, dev).requires_grad_(True)
        y = ops.maxpool(xd, 2, 2, 0, out_planes=True)
        assert ops.is_planes(y)
        got = ops.export_activation(y)
        assert (got.cpu() - yr.detach()).abs().max().item() <= 2.0 ** -20 * float(x.abs().max())
        got.backward(dy.to(dev))
        assert rel_err(xd.grad, xr.grad) < 1e-6
    finally:
        ops.PLANES_MIN_PIXELS = prev_min
