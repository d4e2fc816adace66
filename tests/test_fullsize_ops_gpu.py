"""Numerical parity at BASELINE layer sizes (-m gpu), SAMPLED against fp64.

The op-level tests (tests/test_ops_gpu.py, tests/test_planes_gpu.py) compare whole tensors with an fp64 reference at sizes a CPU
convolution finishes in seconds (largest: 4 x 128 x 112^2); the full-size tests (tests/test_fullsize_gpu.py) are property tests.  A
grid-size-dependent indexing bug -- 32-bit offsets, tile / split-K arithmetic, the ragged halo patches of the U-Net sizes, the XCD remap at
thousands of blocks -- would pass both.  Here every kernel family of the hot path runs at the true size of a BASELINE.json layer
(configs[1]: U-Net bs 16 x 512^2; configs[2]: DeepLabV3+/R101 bs 32 x 512^2; configs[4]: Xception bs 8 x 1024^2) through the product's
own operator path (layers.Conv2d / ops.* -> C ABI), and >= 2048 output elements per tensor -- drawn uniformly over the WHOLE index space,
plus the corners -- are recomputed in float64 on the device from gathered receptive fields:

    conv forward   y[b,o,p,q]   = sum_{c,r,s} x[b,c,p*st-pad+r*d,q*st-pad+s*d] w[o,c,r,s]          (models/backbone/resnet.py:21-26,92,
    conv dgrad     dx[b,c,h,w]  = sum_{o,r,s} dy[b,o,h+pad-r*d,w+pad-s*d]   w[o,c,r,s]  (stride 1)   models/modules/aspp.py:18,
    conv wgrad     dw[o,c,r,s]  = sum_{b,p,q} dy[b,o,p,q] x[b,c,p-pad+r*d,q-pad+s*d]                  models/architectures/unet.py:107-126)

Bound, as tests/test_ops_gpu.py::test_conv_precision_modes: |HIP - fp64| <= 1e-6 x sum|a||b| of the same products (a few fp32 ulps of the
accumulated magnitude; f16x3 measures 2.7e-7 worst case) -- for wgrad's 524288-term sums 2e-6 (split-K slabs are added in fp32).
BatchNorm (bn.hip) and MultiLoss (loss.hip) are compared with complete fp64 evaluations on the device (the tensors fit).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_SAMPLES = 2048


def _rand(seed, *shape, scale=1.0, dev=None):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=dev) * scale


def _nhwc(seed, b, c, h, w, dev, scale=1.0):
    """logical [B,C,H,W] with channels_last memory, filled on the device"""
    t = _rand(seed, b, h, w, c, scale=scale, dev=dev)
    return t.permute(0, 3, 1, 2)


def _samples(seed, n, dev, *dims):
    """n index tuples uniform over dims, the 2^len(dims) corners first (tile / patch edges, first and last block of the grid)"""
    g = torch.Generator().manual_seed(seed)
    idx = [torch.randint(0, d, (n,), generator=g) for d in dims]
    k = 0
    for corner in np.ndindex(*([2] * len(dims))):
        for j, c in enumerate(corner):
            idx[j][k] = (dims[j] - 1) * c
        k += 1
    return [i.to(dev) for i in idx]


@pytest.fixture
def f16x3_full(dev):
    """default arithmetic, product thresholds (every tensor here is far above ops.PLANES_MIN_PIXELS)"""
    from pylc_amd.lib import lib, check
    from pylc_amd import runtime
    prev, prev_drop = lib.pylc_get_conv_precision(), runtime.dropout_enabled
    check(lib.pylc_set_conv_precision(2))
    runtime.dropout_enabled = False
    yield
    runtime.dropout_enabled = prev_drop
    check(lib.pylc_set_conv_precision(prev))
    torch.cuda.empty_cache()


# name, cin, cout, k, stride, pad, dil, B, H, W, bias
CONV_FULL = [
    ('deeplab decoder 256->256 3x3 @128^2 bs32 (decoder.py:30)', 256, 256, 3, 1, 1, 1, 32, 128, 128, False),
    ('resnet layer3 1024->256 1x1 @32^2 bs32 (resnet.py:21)', 1024, 256, 1, 1, 0, 1, 32, 32, 32, False),
    ('resnet layer3 256->1024 1x1 @32^2 bs32 (resnet.py:26)', 256, 1024, 1, 1, 0, 1, 32, 32, 32, False),
    ('aspp 2048->256 3x3 d=12 @32^2 bs32 (aspp.py:18,64)', 2048, 256, 3, 1, 12, 12, 32, 32, 32, False),
    ('unet 64->64 3x3 valid @510^2 bs16 (unet.py:110)', 64, 64, 3, 1, 0, 1, 16, 510, 510, True),
    ('unet 128->128 3x3 valid, ragged @252^2 bs16 (unet.py:115)', 128, 128, 3, 1, 0, 1, 16, 252, 252, True),
    ('resnet layer1 64->256 1x1 @128^2 bs32 (resnet.py:26)', 64, 256, 1, 1, 0, 1, 32, 128, 128, False),
    # round 5: the instantiations the stride-1 list left out -- stride 2 (one dgrad launch per output parity class),
    # the strided 1x1 projection, the thin-input 7x7 stem, the other ASPP rates (per-tile tap skipping), the 304-channel decoder input
    ('resnet layer2 128->128 3x3 stride 2 128^2->64^2 bs32 (resnet.py:22-23)', 128, 128, 3, 2, 1, 1, 32, 128, 128, False),
    ('resnet layer2 projection 256->512 1x1 stride 2 128^2->64^2 bs32 (resnet.py:90-94)', 256, 512, 1, 2, 0, 1, 32, 128, 128, False),
    ('resnet stem 3->64 7x7 stride 2 512^2->256^2 bs32 (resnet.py:72)', 3, 64, 7, 2, 3, 1, 32, 512, 512, False),
    ('aspp 2048->256 3x3 d=6 @32^2 bs32 (aspp.py:18,61)', 2048, 256, 3, 1, 6, 6, 32, 32, 32, False),
    ('aspp 2048->256 3x3 d=18 @32^2 bs32 (aspp.py:18,67)', 2048, 256, 3, 1, 18, 18, 32, 32, 32, False),
    ('deeplab decoder 304->256 3x3 @128^2 bs32 (decoder.py:30)', 304, 256, 3, 1, 1, 1, 32, 128, 128, False),
]


def _conv_case_sampled(dev, case, tol_fwd, tol_dx, tol_dw):
    """One layer at true size through layers.Conv2d -> the C ABI; sampled outputs of y, dx, dw against float64 from gathered receptive
    fields, errors relative to sum |a||b| of the same products.  Any stride (dgrad: the taps whose output coordinate is integral)."""
    from pylc_amd import ops, layers, optim
    name, cin, cout, k, st, pad, dil, B, H, W, bias = case
    torch.manual_seed(7)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil, bias=bias, init='kaiming').to(dev)
    arena = optim.FlatArena(conv)
    thin = cin % 4 != 0                       # the stem: the product feeds the 4-channel image pack (pylc_image_pack); no input gradient
    xc = 4 if thin else cin
    x = _nhwc(11, B, xc, H, W, dev, scale=1.5)
    if not thin:
        x.requires_grad_(True)
    planes0 = ops.planes_marked[0]
    y = conv(x)
    OH, OW = y.shape[2], y.shape[3]
    assert (OH, OW) == ((H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1)
    dy = _nhwc(12, B, cout, OH, OW, dev)
    y.backward(dy)
    ops.sync_side_streams()
    torch.cuda.synchronize()
    if not thin:
        assert ops.planes_marked[0] > planes0, 'the layer did not take the fp16-plane kernels it takes in the training step'
    yv, dw = y.detach(), conv.weight.grad.detach()
    xv, w = x.detach(), conv.weight.detach()
    w64 = w.double()
    if thin:                                  # reference filter zero-padded to the pack's channel count
        w64 = torch.cat([w64, torch.zeros(cout, xc - cin, k, k, dtype=torch.float64, device=dev)], 1)
    taps = torch.arange(k, device=dev)

    # ---- forward: gather [n, cin, k, k] receptive fields ----
    b, o, p, q = _samples(21, N_SAMPLES, dev, B, cout, OH, OW)
    ih = p[:, None] * st - pad + taps[None, :] * dil                           # [n, k]
    iw = q[:, None] * st - pad + taps[None, :] * dil
    okh, okw = (ih >= 0) & (ih < H), (iw >= 0) & (iw < W)
    patch = xv[b[:, None, None, None], torch.arange(xc, device=dev)[None, :, None, None],
               ih.clamp(0, H - 1)[:, None, :, None], iw.clamp(0, W - 1)[:, None, None, :]].double()
    patch = patch * (okh[:, None, :, None] & okw[:, None, None, :])
    wsel = w64[o]                                                              # [n, cin, k, k]
    ref = (patch * wsel).sum((1, 2, 3))
    mag = (patch.abs() * wsel.abs()).sum((1, 2, 3))
    if bias:
        ref = ref + conv.bias.detach().double()[o]
        mag = mag + conv.bias.detach().double().abs()[o]
    err = ((yv[b, o, p, q].double() - ref).abs() / mag).max().item()
    print('%s: fwd  max |y - fp64| / sum|ab| = %.3g over %d samples' % (name, err, N_SAMPLES))
    assert err < tol_fwd

    # ---- dgrad: dx[b,c,h,w] = sum_{o,r,s} dy[b,o,(h+pad-r d)/st,(w+pad-s d)/st] w[o,c,r,s] over the taps with integral coordinates ----
    if not thin:
        dx = x.grad
        b, c, h, wq = _samples(22, N_SAMPLES, dev, B, cin, H, W)
        nh = h[:, None] + pad - taps[None, :] * dil
        nw = wq[:, None] + pad - taps[None, :] * dil
        oh, ow = torch.div(nh, st, rounding_mode='floor'), torch.div(nw, st, rounding_mode='floor')
        okh = (nh >= 0) & (nh % st == 0) & (oh < OH)
        okw = (nw >= 0) & (nw % st == 0) & (ow < OW)
        gp = dy[b[:, None, None, None], torch.arange(cout, device=dev)[None, :, None, None],
                oh.clamp(0, OH - 1)[:, None, :, None], ow.clamp(0, OW - 1)[:, None, None, :]].double()
        gp = gp * (okh[:, None, :, None] & okw[:, None, None, :])
        wsel = w64[:, c].permute(1, 0, 2, 3)                                   # [n, cout, k, k]
        ref = (gp * wsel).sum((1, 2, 3))
        mag = (gp.abs() * wsel.abs()).sum((1, 2, 3))
        live = mag > 0                        # (stride 2, 1x1: three of four input pixels receive no gradient at all -- exact zeros)
        assert (dx[b, c, h, wq][~live] == 0).all()
        err = ((dx[b, c, h, wq].double() - ref).abs()[live] / mag[live]).max().item()
        print('%s: dgrad max |dx - fp64| / sum|ab| = %.3g (%d live samples)' % (name, err, int(live.sum())))
        assert err < tol_dx

    # ---- wgrad: 16 output x 16 input channels (first, last and random ones) x all taps, each a sum over all B*OH*OW pixels ----
    g = torch.Generator().manual_seed(23)
    osel = torch.unique(torch.cat([torch.tensor([0, cout - 1]), torch.randint(0, cout, (14,), generator=g)])).to(dev)
    csel = torch.unique(torch.cat([torch.tensor([0, cin - 1]), torch.randint(0, cin, (14,), generator=g)])).to(dev)
    dys = dy[:, osel].double()                                                 # [B, no, OH, OW]
    xs = torch.nn.functional.pad(xv[:, csel].double(), (pad, pad, pad, pad))
    worst = 0.0
    for r in range(k):
        for s in range(k):
            xw = xs[:, :, r * dil:r * dil + (OH - 1) * st + 1:st, s * dil:s * dil + (OW - 1) * st + 1:st]
            ref = torch.einsum('bopq,bcpq->oc', dys, xw)
            mag = torch.einsum('bopq,bcpq->oc', dys.abs(), xw.abs())
            got = dw[osel][:, csel][:, :, r, s].double()
            worst = max(worst, ((got - ref).abs() / mag).max().item())
    print('%s: wgrad max |dw - fp64| / sum|ab| = %.3g over %d elements' % (name, worst, len(osel) * len(csel) * k * k))
    assert worst < tol_dw
    if bias:      # bias gradient = column sums of dy
        ref = dy.double().sum((0, 2, 3))
        assert ((conv.bias.grad.double() - ref).abs() / dy.double().abs().sum((0, 2, 3))).max().item() < 1e-6
    del arena


@pytest.mark.parametrize('case', CONV_FULL, ids=[c[0].split(' (')[0] for c in CONV_FULL])
def test_conv_fwd_dgrad_wgrad_sampled_fp64(dev, f16x3_full, case):
    _conv_case_sampled(dev, case, 1e-6, 1e-6, 2e-6)


# precision mode 3 (BASELINE configs[4]: Xception, 1024^2 tiles, bs 8): ONE fp16 plane per operand.  Each operand element carries a relative
# rounding of at most 2^-12 (round to nearest, 11 significant bits, scaled per tensor so that nothing is subnormal near the maximum), so
# every product is off by at most (2^-11 + 2^-24) |a||b| and a sum by that times sum|a||b| -- a strict bound, not a statistical one
# (test_planes_gpu.py::test_conv_mode3_single_plane_against_fp64 measures 1e-4..3.7e-4 of the largest output at small sizes); fp32 accumulation
# and the split-K slabs add the f16x3 bound on top.
MODE3_TOL = 2.0 ** -11 * 1.01 + 2e-6
CONV_FULL_MODE3 = [
    ('xception middle-flow pointwise 728->728 1x1 @64^2 bs8 (xception.py:25-39)', 728, 728, 1, 1, 0, 1, 8, 64, 64, False),
    ('deeplab decoder 256->256 3x3 @256^2 bs8 (decoder.py:30 at 1024^2 tiles)', 256, 256, 3, 1, 1, 1, 8, 256, 256, False),
    ('xception entry pointwise 128->128 1x1 @512^2 bs8 (xception.py:32,122)', 128, 128, 1, 1, 0, 1, 8, 512, 512, False),
    ('xception exit pointwise 1536->2048 1x1 @64^2 bs8 (xception.py:157)', 1536, 2048, 1, 1, 0, 1, 8, 64, 64, False),
]


@pytest.fixture
def mode3_full(dev):
    from pylc_amd.lib import lib, check
    from pylc_amd import runtime
    prev, prev_drop = lib.pylc_get_conv_precision(), runtime.dropout_enabled
    check(lib.pylc_set_conv_precision(3))
    runtime.dropout_enabled = False
    yield
    runtime.dropout_enabled = prev_drop
    check(lib.pylc_set_conv_precision(prev))
    torch.cuda.empty_cache()


@pytest.mark.parametrize('case', CONV_FULL_MODE3, ids=[c[0].split(' (')[0] for c in CONV_FULL_MODE3])
def test_conv_mode3_fwd_dgrad_wgrad_sampled_fp64(dev, mode3_full, case):
    """gg_pl_kernel<1,*> / gg_plh_kernel<1> (128-byte LDS rows) and wgrad_pl's one-plane instantiations at configs[4]'s sizes."""
    from pylc_amd import ops
    assert ops.nplanes() == 1
    _conv_case_sampled(dev, case, MODE3_TOL, MODE3_TOL, MODE3_TOL)


@pytest.mark.parametrize('c,b,h,w,stride,dil', [(128, 8, 512, 512, 1, 1), (128, 8, 512, 512, 2, 1), (1536, 8, 64, 64, 1, 2), (728, 8, 64, 64, 1, 1)],
                         ids=['entry 128 @512^2 stride 1', 'entry 128 @512^2 stride 2', 'exit 1536 @64^2 dilation 2', 'middle 728 @64^2'])
def test_half_depthwise_tiled_kernels_fullsize(dev, c, b, h, w, stride, dil):
    """dw_tile_kernel / dw_tileg_kernel (one-plane fp16 tensors: stride 1, stride 2, dilation 2; xception.py:25-39 SeparableConv2d with
    fixed_padding) at the sizes of configs[4] (bs 8, 1024^2 tiles) -- the complete check of tests/test_round3_gpu.py (forward + statistics,
    fresh / accumulating / fp32 / masked-residual data gradients, filter gradient, all against a float64 depthwise conv on the device of the
    values the kernels see), run at true size."""
    from tests.test_round3_gpu import test_half_depthwise_kernels_against_fp64 as dw_case
    dw_case(dev, c, b, h, w, stride, dil, 3)
    torch.cuda.empty_cache()


def test_dgrad_with_masked_residual_gradient_fullsize(dev, f16x3_full):
    """pylc_conv2d_dgrad_add at the size of a layer3 block's first conv (1024 -> 256 1x1 @32^2 bs 32, resnet.py:20,36-51): the data gradient of
    the conv PLUS the block's residual gradient relu'(out) * dout, formed in the dgrad epilogue from dout and the 1-bit ReLU mask that the
    block's last BatchNorm parked on the gradient link (ops.ResidualLink.masked).  2048 sampled elements against float64."""
    from pylc_amd import ops, layers, optim
    B, cin, cout, H, W = 32, 1024, 256, 32, 32
    torch.manual_seed(9)
    conv = layers.Conv2d(cin, cout, 1, 1, 0, 1, init='kaiming').to(dev)
    arena = optim.FlatArena(conv)
    x = ops.to_planes(_nhwc(61, B, cin, H, W, dev, scale=1.5)).requires_grad_(True)
    link = ops.ResidualLink()
    y = ops.conv2d(x, conv.weight, None, 1, 0, 1, res_link=link)
    assert link.armed
    dy = _nhwc(62, B, cout, H, W, dev)
    dout = ops.empty_nhwc(B, cin, H, W, dev)
    dout.copy_(_nhwc(63, B, cin, H, W, dev))
    keep = torch.rand(B, H, W, cin, generator=torch.Generator(device=dev).manual_seed(64), device=dev) > 0.45      # NHWC element order
    mask = (keep.view(-1, 8).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(1).to(torch.uint8)   # element e -> bit e & 7 of byte e >> 3
    link.masked = (dout, mask)
    y.backward(ops.to_planes(dy))
    ops.sync_side_streams()
    torch.cuda.synchronize()
    assert link.masked is None, 'the dgrad did not consume the parked (dout, mask) pair in its epilogue'
    dx = x.grad
    assert dx is not None and not ops.is_planes(dx)
    w64 = conv.weight.detach().double()[:, :, 0, 0]                            # [cout, cin]
    b, c, h, wq = _samples(65, N_SAMPLES, dev, B, cin, H, W)
    g = dy[b, :, h, wq].double()                                               # [n, cout]
    ref = (g * w64[:, c].t()).sum(1)
    mag = (g.abs() * w64[:, c].t().abs()).sum(1)
    res = dout[b, c, h, wq].double() * keep[b, h, wq, c]
    err = ((dx[b, c, h, wq].double() - (ref + res)).abs() / (mag + res.abs())).max().item()
    print('dgrad + masked residual 1024<-256 @32^2 bs32: max |dx - fp64| / (sum|ab| + |res|) = %.3g' % err)
    assert err < 1e-6
    del arena


def test_depthwise_728_at_64sq_sampled_fp64(dev, f16x3_full):
    """Xception middle flow at configs[4]'s size (728 channels, 64 x 64 map of a 1024^2 tile, bs 8; xception.py:34-39): depthwise 3x3 with
    fixed_padding -- forward, input gradient and filter gradient of the product's default (fp32 tensor) path."""
    from pylc_amd import ops
    B, C, H, W = 8, 728, 64, 64
    x = _nhwc(31, B, C, H, W, dev).requires_grad_(True)
    wt = (_rand(32, C, 1, 3, 3, dev=dev) * 0.3).requires_grad_(True)
    y = ops.dwconv3x3(x, wt, 1, 1)
    dy = _nhwc(33, B, C, H, W, dev)
    y.backward(dy)
    ops.sync_side_streams()
    torch.cuda.synchronize()
    x64 = torch.nn.functional.pad(x.detach().double(), (1, 1, 1, 1))
    w64 = wt.detach().double()
    ref = torch.zeros(B, C, H, W, dtype=torch.float64, device=dev)
    mag = torch.zeros_like(ref)
    refw = torch.zeros(C, 3, 3, dtype=torch.float64, device=dev)
    dy64 = dy.double()
    dyp = torch.nn.functional.pad(dy64, (1, 1, 1, 1))
    refdx = torch.zeros_like(ref)
    for r in range(3):
        for s in range(3):
            win = x64[:, :, r:r + H, s:s + W]
            ref += win * w64[:, 0, r, s][None, :, None, None]
            mag += win.abs() * w64[:, 0, r, s].abs()[None, :, None, None]
            refw[:, r, s] = (dy64 * win).sum((0, 2, 3))
            refdx += dyp[:, :, 2 - r:2 - r + H, 2 - s:2 - s + W] * w64[:, 0, r, s][None, :, None, None]
    b, c, p, q = _samples(34, N_SAMPLES, dev, B, C, H, W)
    e_y = ((y.detach()[b, c, p, q].double() - ref[b, c, p, q]).abs() / (mag[b, c, p, q] + 1e-30)).max().item()
    e_dx = ((x.grad[b, c, p, q].double() - refdx[b, c, p, q]).abs().max() / refdx.abs().max()).item()
    e_dw = ((wt.grad[:, 0].double() - refw).abs().max() / refw.abs().max()).item()
    print('depthwise 728 @64^2 bs8: y %.3g (of sum|ab|), dx %.3g, dw %.3g (of the largest entry)' % (e_y, e_dx, e_dw))
    assert e_y < 1e-6 and e_dx < 2e-6 and e_dw < 5e-6          # tests/test_ops_gpu.py::test_dwconv's bounds


def test_batchnorm_524288x256_against_fp64(dev, f16x3_full):
    """One training-mode BatchNorm + ReLU at the size of the decoder's 256-channel 128^2 maps at bs 32 (524288 rows x 256 channels,
    decoder.py:31; models/sync_batchnorm/batchnorm.py:48-125 is the math): batch statistics (the reduction pass), output, running
    statistics, input gradient, dgamma / dbeta -- against fp64 on the device."""
    from pylc_amd import ops, layers, optim
    B, C, H, W = 32, 256, 128, 128
    torch.manual_seed(3)
    bn = layers.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * _rand(41, C, dev=dev))
        bn.bias.copy_(0.1 * _rand(42, C, dev=dev))
    arena = optim.FlatArena(bn)
    y = (_nhwc(43, B, C, H, W, dev, scale=2.0) + 0.5).requires_grad_(True)
    out = bn(y, relu=True)
    dout = _nhwc(44, B, C, H, W, dev)
    out_f = ops.as_nhwc(out) if ops.is_planes(out) else out
    out.backward(dout)
    torch.cuda.synchronize()
    y64 = y.detach().double()
    mean = y64.mean((0, 2, 3))
    var = y64.var((0, 2, 3), unbiased=False)
    n = B * H * W
    g64, b64 = bn.weight.detach().double(), bn.bias.detach().double()
    xhat = (y64 - mean[None, :, None, None]) / torch.sqrt(var + 1e-5)[None, :, None, None]
    o64 = torch.relu(xhat * g64[None, :, None, None] + b64[None, :, None, None])
    b, c, p, q = _samples(45, N_SAMPLES, dev, B, C, H, W)
    e_out = ((out_f.detach()[b, c, p, q].double() - o64[b, c, p, q]).abs().max() / o64.abs().max()).item()
    e_rm = ((bn.running_mean.double() - 0.1 * mean).abs().max() / mean.abs().max()).item()
    e_rv = ((bn.running_var.double() - (0.9 + 0.1 * var * n / (n - 1))).abs().max()).item()
    gmask = dout.double() * (o64 > 0)
    dbeta = gmask.sum((0, 2, 3))
    dgamma = (gmask * xhat).sum((0, 2, 3))
    dy64 = (g64 / torch.sqrt(var + 1e-5))[None, :, None, None] * (gmask - dbeta[None, :, None, None] / n - xhat * dgamma[None, :, None, None] / n)
    # elements whose pre-activation sits within fp32 rounding of zero may take the other side of the ReLU: leave them out of the dy check
    safe = (xhat[b, c, p, q] * g64[c] + b64[c]).abs() > 1e-5
    e_dy = ((y.grad[b, c, p, q].double() - dy64[b, c, p, q]).abs()[safe].max() / dy64.abs().max()).item()
    e_dg = ((bn.weight.grad.double() - dgamma).abs().max() / dgamma.abs().max()).item()
    e_db = ((bn.bias.grad.double() - dbeta).abs().max() / dbeta.abs().max()).item()
    print('BatchNorm 524288 x 256: out %.3g, running mean %.3g var %.3g, dy %.3g, dgamma %.3g, dbeta %.3g' % (e_out, e_rm, e_rv, e_dy, e_dg, e_db))
    assert e_out < 5e-6 and e_rm < 1e-6 and e_rv < 2e-6          # tests/test_ops_gpu.py::test_bn_train's bounds
    assert e_dy < 2e-5 and e_dg < 2e-5 and e_db < 2e-5
    del arena


def test_multiloss_32x9x512sq_against_fp64(dev, f16x3_full):
    """The loss head at configs[2]'s size (32 x 9 x 512 x 512 logits; models/modules/loss.py:71-194): the three losses and 2048 sampled
    gradient elements against the oracle's formulas evaluated in float64 on the device."""
    from pylc_amd import ops
    import oracle
    B, C, H, W = 32, 9, 512, 512
    z = (_nhwc(51, B, C, H, W, dev, scale=3.0)).requires_grad_(True)
    t = torch.randint(0, C, (B, H, W), generator=torch.Generator().manual_seed(52)).to(dev)
    cw = torch.linspace(0.5, 2.0, C, device=dev)
    for weighted in (False, True):
        z.grad = None
        losses = ops.multiloss(z, t, cw if weighted else None, 0.5, 0.5, 0.5)
        losses[0].backward()
        torch.cuda.synchronize()
        z64 = z.detach().double().contiguous().requires_grad_(True)
        tot, ce, dsc, fl = oracle.multiloss(z64, t, (0.5, 0.5, 0.5), cw.double(), weighted)
        tot.backward()
        got = losses.detach().double().cpu().tolist()
        want = [v.item() for v in (tot, ce, dsc, fl)]
        b, c, p, q = _samples(53, N_SAMPLES, dev, B, C, H, W)
        e_g = ((z.grad[b, c, p, q].double() - z64.grad[b, c, p, q]).abs().max() / z64.grad.abs().max()).item()
        print('multiloss 32x9x512^2 weighted=%s: (total, ce, dice, focal) %s vs fp64 %s; grad %.3g' % (weighted, got, want, e_g))
        for a, r in zip(got, want):
            assert abs(a - r) < 2e-6 * max(1.0, abs(r)), (got, want)          # tests/test_ops_gpu.py::test_multiloss_golden's bound
        assert e_g < 1e-5
        del z64, tot
