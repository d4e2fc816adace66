"""fp16-plane activations (-m gpu): the kernels that read / write the pre-split format against the fp32-operand kernels.

The planes kernels (conv_pl.hip, wgrad_pl.hip) form the same two fp16 pieces per element and issue the same MFMAs in the same order
as conv_igemm.hip's f16x3 kernels, so with equal range scalars their results must be BIT-IDENTICAL; the BatchNorm kernels that
write planes scale with a range BOUND (Samuelson) instead of the exact maximum, so they agree to the format's 2^-22.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['interleaved', 'separate'])
def plane_format(request, monkeypatch):
    """Both plane formats for one test: the product's chunk-interleaved planes, and the round-4 format -- separate filter planes (env, read when
    the arena prepares them) and separate pixel planes (pylc_set_planes_interleave is process-wide: restored afterwards)."""
    import os
    from pylc_amd.lib import lib
    if request.param == 'separate':
        monkeypatch.setenv('PYLC_NO_FILTER_INTERLEAVE', '1')
        lib.pylc_set_planes_interleave(0)
    yield request.param
    lib.pylc_set_planes_interleave(0 if os.environ.get('PYLC_NO_PLANE_INTERLEAVE') else 1)


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)


def nhwc(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture
def f16x3(dev):
    from pylc_amd.lib import lib, check
    from pylc_amd import runtime
    from pylc_amd import ops
    prev, prev_min, prev_drop = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS, runtime.dropout_enabled
    check(lib.pylc_set_conv_precision(2))
    ops.PLANES_MIN_PIXELS = 0              # the cases here are small: take the planes kernels at every size
    runtime.dropout_enabled = True         # (whole-network parity tests switch dropout off process-wide)
    # the bit-identity cases compare with the fp32-operand wgrad (32x32x16 MFMAs): the planes wgrad runs in that form here; its default
    # 16x16x32 form (same products, 32 pixels per instruction) has test_wgrad_16x16x32_form_matches_32x32x16_form
    lib.pylc_debug_wgrad_m16(0)
    yield
    lib.pylc_debug_wgrad_m16(2)
    ops.PLANES_MIN_PIXELS = prev_min
    runtime.dropout_enabled = prev_drop
    check(lib.pylc_set_conv_precision(prev))
    lib.pylc_debug_pp_flags(0)
    runtime.no_planes = False


def test_planes_round_trip(dev, f16x3):
    from pylc_amd import ops
    x = nhwc(rnd(1, 3, 72, 9, 11, scale=3.0), dev)
    x[0, :, 0, 0] = 0.0
    x[1, 5, 2, 3] = 1e-6            # 2^-21 below the maximum: still a normal low piece
    p = ops.to_planes(x)
    assert ops.is_planes(p) and not ops.is_planes(x)
    back = ops.from_planes(p)
    assert rel(back, x) < 2.0 ** -22
    assert torch.equal(back[0, :, 0, 0], torch.zeros(72, device=dev))
    # as_nhwc() is the safety net of every op that does not know the format
    assert torch.equal(ops.as_nhwc(p), back)


# cin, cout, k, stride, pad, dil, B, H, W            (wgrad tile configuration, geometry path)
PLANES_CONV_CASES = [
    (256, 256, 3, 1, 1, 1, 2, 32, 32),      # 128x128 wgrad tiles, row-aligned reduction tiles
    (128, 512, 1, 1, 0, 1, 2, 32, 64),      # 1x1, identity gather
    (256, 48, 1, 1, 0, 1, 2, 32, 32),       # narrow output (decoder conv1): one half-empty column tile; 64x64 wgrad tiles
    (64, 64, 3, 1, 1, 1, 2, 32, 32),        # layer1 3x3
    (256, 32, 1, 1, 0, 1, 2, 32, 32),       # 32x128 wgrad tiles
    (304, 256, 3, 1, 1, 1, 1, 20, 44),      # Cin % 32 != 0, OW % 32 != 0: per-row counters in wgrad, partial last chunk
    (128, 128, 3, 2, 1, 1, 2, 45, 45),      # stride 2: dgrad by parity classes, odd size, pixel tail
    (72, 200, 3, 1, 6, 6, 3, 30, 30),       # atrous with tap skipping, N and M tails
    (2048, 256, 3, 1, 12, 12, 2, 32, 32),   # ASPP branch on a 32^2 map
    (32, 64, 3, 1, 1, 1, 2, 48, 48),        # Xception conv2: a single channel chunk
    (64, 128, 1, 2, 0, 1, 2, 48, 48),       # Xception block skip: 1x1 stride 2 (dgrad: three empty parity classes)
    (64, 128, 1, 1, 0, 1, 2, 48, 48),       # Xception pointwise
    (728, 728, 1, 1, 0, 1, 2, 16, 16),      # middle flow: Cin, Cout % 32 != 0, both tails
]


@pytest.mark.parametrize('case', PLANES_CONV_CASES)
def test_conv_on_planes_is_bit_identical(dev, f16x3, case):
    """conv2d on a filter with prepared planes: fp32 operands (in-kernel split, 256x128 ping-pong kernel, wgrad_split_kernel) vs
    fp16-plane operands (conversion pass + conv_pl.hip + wgrad_pl.hip) -- y, dx and dw bit for bit; y also against fp64."""
    from pylc_amd import ops, layers, optim, runtime
    from pylc_amd.lib import lib
    cin, cout, k, st, pad, dil, B, H, W = case
    torch.manual_seed(3)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil).to(dev)
    arena = optim.FlatArena(conv)
    assert conv.takes_planes()
    x = nhwc(rnd(5, B, cin, H, W, scale=2.0), dev).requires_grad_(True)
    out = {}
    for mode in ('fp32', 'planes'):
        runtime.no_planes = mode == 'fp32'
        lib.pylc_debug_pp_flags(1024 if mode == 'fp32' else 0)        # fp32 operands: force the 16x16x32-MFMA kernel at every size
        x.grad = None
        arena.g.zero_()
        y = conv(x)
        dy = nhwc(rnd(6, *y.shape), dev)
        y.backward(dy)
        ops.sync_side_streams()
        torch.cuda.synchronize()
        out[mode] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.detach().clone())
    runtime.no_planes = False
    lib.pylc_debug_pp_flags(0)
    ref = torch.nn.functional.conv2d(x.detach().double().cpu(), conv.weight.detach().double().cpu(), None, st, pad, dil)
    assert rel(out['planes'][0], ref) < 3e-6
    # bit for bit wherever the fp32-operand path ran the same MFMA shape (16x16x32: its 256x128 kernel, i.e. more than 64 output
    # columns -- cout in the forward, cin in dgrad; narrower products run its 32x32x16 kernels and agree to fp32 rounding); wgrad always
    exact = {'y': cout > 64, 'dx': cin > 64, 'dw': True}
    for name, a, b in zip(('y', 'dx', 'dw'), out['fp32'], out['planes']):
        if exact[name]:
            assert torch.equal(a, b), (name, rel(a, b))
        else:
            assert rel(a, b) < 2e-6, (name, rel(a, b))


@pytest.mark.parametrize('c,b,hw,relu,res,drop', [(64, 4, 16, True, False, 0.0), (256, 2, 12, True, True, 0.0), (48, 3, 20, True, False, 0.5),
                                                  (2048, 2, 4, True, True, 0.0), (1024, 2, 16, False, False, 0.1)])
def test_bn_planes_output_and_fused_dropout(dev, f16x3, c, b, hw, relu, res, drop):
    _bn_planes_case(dev, c, b, hw, relu, res, drop, 2.0 ** -21, 2e-5)


@pytest.mark.parametrize('c,b,hw,relu,res,drop', [(64, 4, 16, True, False, 0.0), (256, 2, 12, True, True, 0.0), (48, 3, 20, True, False, 0.5)])
def test_bn_single_plane_mode3(dev, fp16_single, c, b, hw, relu, res, drop):
    """The same BatchNorm passes in precision mode 3: ONE fp16 plane out (and as the residual input) -- values to fp16's 2^-11, the
    gradients (which see the residual / mask through one plane) to 2e-3, running statistics untouched by the output format."""
    from pylc_amd import runtime
    prev = runtime.dropout_enabled
    runtime.dropout_enabled = True
    try:
        _bn_planes_case(dev, c, b, hw, relu, res, drop, 2.0 ** -10, 2e-3)
    finally:
        runtime.dropout_enabled = prev


def _bn_planes_case(dev, c, b, hw, relu, res, drop, val_tol, grad_tol):
    """BatchNorm(+residual, ReLU, dropout) writing fp16 planes (range from the Samuelson bound) vs the fp32 kernels: values to the
    format's accuracy, parameter / input / residual gradients to fp32 rounding; the residual may itself arrive as planes.  The fused
    dropout draws the same mask as ops.dropout with the same seed."""
    from pylc_amd import ops
    y0 = rnd(11, b, c, hw, hw, scale=2.0) + 0.5
    g0, be0 = 1 + 0.1 * rnd(12, c), 0.1 * rnd(13, c)
    r0 = rnd(16, b, c, hw, hw) if res else None
    do = nhwc(rnd(17, b, c, hw, hw), dev)
    seed = 0x1234ABCD
    got = {}
    for mode in ('fp32', 'planes'):
        yd = nhwc(y0, dev).requires_grad_(True)
        gd, bed = g0.to(dev).requires_grad_(True), be0.to(dev).requires_grad_(True)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        rd = nhwc(r0, dev).requires_grad_(True) if res else None
        r_in = rd
        if res and mode == 'planes':
            r_in = ops.to_planes(rd.detach())          # a block input that travels as planes (no gradient through the conversion)
        if mode == 'fp32':
            o = ops.bn_act(yd, gd, bed, rm, rv, r_in, relu, True)
            if drop > 0:
                o = ops.dropout(o, drop, seed)
        else:
            o = ops.bn_act(yd, gd, bed, rm, rv, r_in, relu, True, out_planes=True, drop=(drop, seed) if drop > 0 else None)
            assert ops.is_planes(o)
        val = ops.as_nhwc(o)
        o.backward(do)
        torch.cuda.synchronize()
        got[mode] = (val.detach().clone(), yd.grad.clone(), gd.grad.clone(), bed.grad.clone(), rm.clone(), rv.clone())
    a, p = got['fp32'], got['planes']
    assert rel(p[0], a[0]) < val_tol
    if drop > 0:        # same elements dropped (a value below the plane's resolution may round to zero)
        assert torch.equal(p[0] == 0, a[0] == 0) or ((p[0] == 0) != (a[0] == 0)).float().mean().item() < (1e-6 if val_tol < 1e-5 else 2e-3)
    if res and val_tol > 1e-5:
        # one plane: the residual enters rounded to 2^-11, so elements of BN(y) + res within that of zero fall on the other side of the
        # ReLU (their gradient differs by all of g): compare in the L2 sense
        l2 = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm())
        assert l2(p[1], a[1]) < 5e-2 and l2(p[2], a[2]) < 5e-2 and l2(p[3], a[3]) < 5e-2
    else:
        assert rel(p[1], a[1]) < grad_tol and rel(p[2], a[2]) < grad_tol and rel(p[3], a[3]) < grad_tol
    assert torch.equal(p[4], a[4]) and torch.equal(p[5], a[5])          # running statistics do not depend on the output format


def test_bottleneck_block_planes_vs_fp32(dev, f16x3):
    """One projection + one identity ResNet block, training mode: every activation between BatchNorm and conv as planes (outputs,
    residual inputs, the dy handed from BatchNorm backward to dgrad / wgrad) vs all-fp32 -- outputs and all gradients agree to fp32
    rounding (the formats hold the same values to 2^-22; nothing else differs)."""
    from pylc_amd import ops, optim, runtime
    from pylc_amd.nets.encoder_resnet import Bottleneck
    torch.manual_seed(7)
    blocks = torch.nn.Sequential(Bottleneck(128, 64, 1, 1, True), Bottleneck(256, 64, 1, 1, False)).to(dev)
    for m in blocks.modules():
        if hasattr(m, 'num_features'):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    arena = optim.FlatArena(blocks)
    blocks[0].out_planes = True
    blocks[1].out_planes = False
    x0 = rnd(21, 4, 128, 24, 32, scale=1.5)
    do = nhwc(rnd(22, 4, 256, 24, 32), dev)
    res = {}
    for mode in ('fp32', 'planes'):
        runtime.no_planes = mode == 'fp32'
        blocks.train()
        for m in blocks.modules():
            if hasattr(m, 'running_mean'):
                m.running_mean.zero_(); m.running_var.fill_(1.0)
        arena.g.zero_()
        x = nhwc(x0, dev).requires_grad_(True)
        y = blocks(x)
        assert not ops.is_planes(y)
        y.backward(do)
        ops.sync_side_streams()
        torch.cuda.synchronize()
        res[mode] = (y.detach().clone(), x.grad.clone(), arena.g.clone())
    runtime.no_planes = False
    assert ops.plane_conversions[0] >= 0
    for name, a, b in zip(('y', 'dx', 'param grads'), res['fp32'], res['planes']):
        assert rel(b, a) < 3e-5, (name, rel(b, a))


@pytest.fixture
def fp16_single(dev):
    from pylc_amd.lib import lib, check
    from pylc_amd import runtime, ops
    prev, prev_min = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS
    check(lib.pylc_set_conv_precision(3))
    ops.PLANES_MIN_PIXELS = 0
    yield
    ops.PLANES_MIN_PIXELS = prev_min
    check(lib.pylc_set_conv_precision(prev))
    runtime.no_planes = False


@pytest.mark.parametrize('case', PLANES_CONV_CASES)
def test_conv_mode3_single_plane_against_fp64(dev, fp16_single, case):
    """Precision mode 3: the same kernels with ONE fp16 plane per operand (NTERMS = 1) -- y, dx, dw against fp64 on the operands as given:
    each operand carries a relative rounding of 2^-12, a sum of K products sqrt(2) 2^-12 of its typical term, so errors relative to the
    largest output stay below 1e-3 at every K of these cases (measured 1e-4..4e-4)."""
    from pylc_amd import ops, layers, optim
    cin, cout, k, st, pad, dil, B, H, W = case
    torch.manual_seed(3)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil).to(dev)
    arena = optim.FlatArena(conv)
    assert ops.nplanes() == 1
    x = nhwc(rnd(5, B, cin, H, W, scale=2.0), dev).requires_grad_(True)
    y = conv(x)
    dy = nhwc(rnd(6, *y.shape), dev)
    y.backward(dy)
    ops.sync_side_streams()
    torch.cuda.synchronize()
    xr = x.detach().double().cpu().requires_grad_(True)
    wr = conv.weight.detach().double().cpu().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, None, st, pad, dil)
    yr.backward(dy.double().cpu())
    e = (rel(y.detach(), yr.detach()), rel(x.grad, xr.grad), rel(conv.weight.grad.detach(), wr.grad))
    print('mode 3 %s: y %.2e dx %.2e dw %.2e' % (case, *e))
    assert max(e) < 1e-3, e


def test_filter_plane_format_travels_with_the_planes(dev):
    """the layout of a parameter's prepared filter planes (chunk-interleaved or separate plane arrays) is part of its planes object
    (optim.FlatArena._publish_planes), set from what the last prepare launch wrote.  Precision mode 2 -> 3 -> 2 WITHOUT an optimiser step, with
    and without a refresh in between: every forward must equal the result of the same mode on freshly prepared planes, bit for bit."""
    from pylc_amd import ops, layers, optim
    from pylc_amd.lib import lib, check
    prev, prev_min = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS
    ops.PLANES_MIN_PIXELS = 0
    try:
        check(lib.pylc_set_conv_precision(2))
        torch.manual_seed(3)
        conv = layers.Conv2d(64, 128, 3, 1, 1, 1).to(dev)
        arena = optim.FlatArena(conv)
        assert ops.filter_planes_fmt(conv.weight._pylc_planes) == 3        # both layouts interleaved (64 and 128 channels)
        x = nhwc(rnd(5, 2, 64, 32, 32, scale=2.0), dev)
        run = lambda: ops.conv2d(ops.to_planes(x), conv.weight, None, 1, 1, 1).clone()      # the plane kernels (conv_pl.hip) on the mode's own pixel planes
        with torch.no_grad():
            y2 = run()
            check(lib.pylc_set_conv_precision(3))                          # no refresh: mode 3 on interleaved filter planes
            y3_stale = run()
            arena.refresh_ranges()                                         # mode 3 prepares separate plane arrays
            assert ops.filter_planes_fmt(conv.weight._pylc_planes) == 0
            y3 = run()
            check(lib.pylc_set_conv_precision(2))                          # no refresh: mode 2 on separate filter planes
            y2_stale = run()
            arena.refresh_ranges()
            assert ops.filter_planes_fmt(conv.weight._pylc_planes) == 3
            y2_again = run()
        assert torch.equal(y3_stale, y3) and torch.equal(y2_stale, y2) and torch.equal(y2_again, y2)
        assert not torch.equal(y2, y3)
    finally:
        ops.PLANES_MIN_PIXELS = prev_min
        check(lib.pylc_set_conv_precision(prev))


PERSIST_OFF = 524288          # pylc_debug_pp_flags bit 19: 1x1 launches on the per-tile kernel (the bit-identity reference of gg_plp_kernel)


@pytest.mark.parametrize('shape', [(16, 64, 256, 1024), (16, 64, 1024, 256), (8, 128, 64, 256), (24, 64, 128, 512), (16, 64, 96, 256), (20, 64, 256, 1024)])
@pytest.mark.parametrize('mode', [2, 3])
def test_persistent_1x1_kernel_is_bit_identical(dev, shape, mode, plane_format):
    """conv_pl.hip gg_plp_kernel: 1x1 / stride-1 launches without an edge and with more 128 x 128 tiles than the chip holds blocks run as ONE
    stream of K-steps per block -- the next tile's first two operand stages requested before this tile's stores, which then stay in flight
    behind a counted vmcnt.  Forward (+ BatchNorm statistics partials), plain dgrad, dgrad + residual source and dgrad + masked residual
    source: every output bit-identical to the per-tile kernel (pylc_debug_pp_flags bit 19), run twice; tile counts that are not a multiple of
    the 512 blocks (640, 768: some blocks walk one tile more), two K-steps per tile (Cin = 64), an odd number of K-steps (Cin = 96)."""
    import ctypes as C
    from pylc_amd import ops, layers, optim
    from pylc_amd.lib import lib, check, ptr, stream
    B, H, cin, cout = shape
    prev, prev_min = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS
    check(lib.pylc_set_conv_precision(mode))
    ops.PLANES_MIN_PIXELS = 0
    try:
        torch.manual_seed(2)
        conv = layers.Conv2d(cin, cout, 1, 1, 0, 1, bn=True).to(dev)
        arena = optim.FlatArena(conv)
        x = nhwc(rnd(31, B, cin, H, H, scale=2.0), dev)
        xp = ops.to_planes(x)
        dy = nhwc(rnd(32, B, cout, H, H), dev)
        dyp = ops.to_planes(dy)
        res = nhwc(rnd(33, B, cin, H, H), dev)
        mask = torch.randint(0, 256, (B * H * H * cin // 8,), dtype=torch.uint8, device=dev)
        d = ops._conv_desc(x, cin, cout, 1, 1, 1, 0, 1, cin, cout)
        d.x_fmt, d.dy_fmt = 1, 1
        d.x_amax, d.w_amax, d.dy_amax = ptr(ops.planes_amax(xp)), ptr(ops.weight_amax(conv.weight)), ptr(ops.planes_amax(dyp))
        d.w_planes_t = ptr(conv.weight._pylc_planes[1])
        d.w_planes_fmt = ops.filter_planes_fmt(conv.weight._pylc_planes)
        dx = ops.empty_nhwc(B, cin, H, H, dev)

        def fwd():
            y = ops.conv2d(xp, conv.weight, None, 1, 0, 1, want_stats=True)
            return [y, y._pylc_sums]

        def dgrad():
            check(lib.pylc_conv2d_dgrad(C.byref(d), ptr(dyp), None, ptr(dx), 0, stream()))
            return [dx]

        def dgrad_res():
            check(lib.pylc_conv2d_dgrad_add(C.byref(d), ptr(dyp), None, ptr(dx), 0, ptr(res), None, stream()))
            return [dx]

        def dgrad_masked():
            check(lib.pylc_conv2d_dgrad_add(C.byref(d), ptr(dyp), None, ptr(dx), 0, ptr(res), ptr(mask), stream()))
            return [dx]

        kinds = [('fwd', fwd), ('dgrad', dgrad), ('dgrad + residual', dgrad_res), ('dgrad + masked residual', dgrad_masked)]
        with torch.no_grad():
            for name, fn in kinds:
                outs = []
                for flags in (PERSIST_OFF, 0, 0):
                    lib.pylc_debug_pp_flags(flags)
                    dx.fill_(float('nan'))
                    r = [t.clone() for t in fn()]
                    torch.cuda.synchronize()
                    outs.append(r)
                for a_, b_, c_ in zip(*outs):
                    assert not torch.isnan(b_).any(), name
                    assert a_.shape == b_.shape and torch.equal(a_, b_) and torch.equal(b_, c_), name
        if mode == 2:
            ref = torch.nn.functional.conv2d(x.double(), conv.weight.detach().double())
            lib.pylc_debug_pp_flags(0)
            with torch.no_grad():
                assert rel(fwd()[0], ref) < 3e-6
    finally:
        lib.pylc_debug_pp_flags(0)
        ops.PLANES_MIN_PIXELS = prev_min
        check(lib.pylc_set_conv_precision(prev))


def test_persistent_1x1_kernel_in_a_bottleneck_chain(dev, f16x3, plane_format):
    """The three launch kinds gg_plp_kernel takes inside the training graph -- forward with statistics, plain dgrad (conv3: 1024 -> 256 channels
    of gradient), and the dgrad that adds the ReLU-masked residual gradient in its epilogue (conv1 of an identity bottleneck,
    pylc_conv2d_dgrad_add; resnet.py:36-51) -- on three layer3-shaped bottlenecks at 16 x 64 x 64 pixels (512 pixel tiles x 8 channel tiles:
    several tiles per block): block output, input gradient and every parameter gradient bit-identical to the per-tile kernels."""
    from pylc_amd import ops, optim, runtime
    from pylc_amd.lib import lib
    from pylc_amd.nets.encoder_resnet import Bottleneck
    prev_drop = runtime.dropout_enabled
    runtime.dropout_enabled = False
    try:
        torch.manual_seed(4)
        net = torch.nn.Sequential(Bottleneck(1024, 256, 1, 1, False), Bottleneck(1024, 256, 1, 1, False), Bottleneck(1024, 256, 1, 2, False)).to(dev)
        for b in net:
            b.out_planes = True
        arena = optim.FlatArena(net)
        net.train()
        x0 = nhwc(rnd(1, 16, 1024, 64, 64), dev)
        dout = nhwc(rnd(2, 16, 1024, 64, 64), dev)
        got = {}
        for flags in (PERSIST_OFF, 0):
            lib.pylc_debug_pp_flags(flags)
            arena.g.zero_()
            x = x0.clone().requires_grad_(True)
            out = ops.export_activation(net(x))
            out.backward(dout)
            ops.sync_side_streams()
            torch.cuda.synchronize()
            got[flags] = (out.detach().clone(), x.grad.clone(), arena.g.clone())
        for name, a, b in zip(('out', 'dx', 'parameter gradients'), got[PERSIST_OFF], got[0]):
            assert torch.equal(a, b), name
        assert float(got[0][1].abs().sum()) > 0
    finally:
        lib.pylc_debug_pp_flags(0)
        runtime.dropout_enabled = prev_drop


@pytest.mark.parametrize('case', [(256, 256, 3, 1, 1, 1, 2, 32, 32), (128, 512, 1, 1, 0, 1, 2, 32, 64), (304, 256, 3, 1, 1, 1, 1, 20, 44),
                                  (136, 200, 3, 1, 2, 2, 3, 17, 23), (2048, 256, 3, 1, 12, 12, 2, 32, 32), (128, 128, 3, 2, 1, 1, 2, 45, 45)])
def test_wgrad_16x16x32_form_matches_32x32x16_form(dev, f16x3, case):
    """wgrad_pl.hip's default form for 128 x 128 f16x3 tiles (v_mfma_f32_16x16x32_f16 on swizzled unpadded LDS rows, pylc_debug_wgrad_m16)
    against the 32x32x16 form: the same products, so the results differ by isolated last-place roundings of the folded cross terms
    (<= 2 ulp of the largest gradient), and both sit at the same distance from fp64."""
    from pylc_amd import ops, layers, optim
    from pylc_amd.lib import lib
    cin, cout, k, st, pad, dil, B, H, W = case
    torch.manual_seed(3)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil).to(dev)
    arena = optim.FlatArena(conv)
    x = nhwc(rnd(5, B, cin, H, W, scale=2.0), dev).requires_grad_(True)
    dws = {}
    try:
        for m16 in (0, 2):
            lib.pylc_debug_wgrad_m16(m16)
            arena.g.zero_()
            x.grad = None
            y = conv(x)
            dy = nhwc(rnd(6, *y.shape), dev)
            y.backward(dy)
            ops.sync_side_streams()
            torch.cuda.synchronize()
            dws[m16] = conv.weight.grad.detach().clone()
    finally:
        lib.pylc_debug_wgrad_m16(0)
    wr = conv.weight.detach().double().cpu().requires_grad_(True)
    torch.nn.functional.conv2d(x.detach().double().cpu(), wr, None, st, pad, dil).backward(dy.double().cpu())
    e0, e2 = rel(dws[0], wr.grad), rel(dws[2], wr.grad)
    print('wgrad vs fp64: 32x32x16 %.2e, 16x16x32 %.2e; forms differ by %.2e of max|dw|' % (e0, e2, rel(dws[2], dws[0])))
    assert rel(dws[2], dws[0]) < 2.5e-7
    assert e2 < max(3e-6, 1.2 * e0)


def test_batched_slab_sums_are_bit_identical(dev, f16x3):
    """One queue: the split-K slabs of every wgrad of a backward pass stay in their weights' own workspaces and ONE launch sums them all
    (pylc_conv2d_wgrad_slabs + pylc_splitk_reduce_batch, ops.flush_slab_sums from ops.sync_side_streams) -- the gradients must be the bits
    of the per-layer sums (pylc_conv2d_wgrad), for several layers per pass and over repeated passes (cached table)."""
    from pylc_amd import ops, layers, optim, runtime
    torch.manual_seed(7)
    net = torch.nn.Sequential(layers.Conv2d(64, 128, 3, 1, 1, 1), layers.Conv2d(128, 128, 1, 1, 0, 1), layers.Conv2d(128, 72, 3, 1, 2, 2)).to(dev)
    arena = optim.FlatArena(net)
    x = nhwc(rnd(5, 8, 64, 48, 48, scale=2.0), dev)
    grads = {}
    prev_side, prev_batch = runtime.wgrad_side_stream, runtime.batch_slab_sums
    try:
        runtime.wgrad_side_stream = False
        for batch in (False, True, True):
            runtime.batch_slab_sums = batch
            arena.g.zero_()
            y = net(x)
            noted = []
            orig = ops._core.add_slab_sum
            ops._core.add_slab_sum = lambda d, e: (noted.append(1), orig(d, e))[1]
            try:
                y.backward(nhwc(rnd(6, *y.shape), dev))
            finally:
                ops._core.add_slab_sum = orig
            # batched: every wgrad left its slabs for the ONE launch -- which the autograd-engine callback of the first note has already made
            # when backward() returns (round 6; sync_side_streams finds nothing left to flush)
            assert len(noted) == (3 if batch else 0), noted
            assert not ops._core._pending_slab_sums.get(dev.index)
            ops.sync_side_streams()
            torch.cuda.synchronize()
            grads.setdefault(batch, []).append(arena.g.clone())
    finally:
        runtime.wgrad_side_stream, runtime.batch_slab_sums = prev_side, prev_batch
    assert float(grads[False][0].abs().max()) > 0
    assert torch.equal(grads[False][0], grads[True][0]) and torch.equal(grads[True][0], grads[True][1])


def test_one_accumulator_wgrad_is_fp32_grade(dev, f16x3):
    """wgrad_pl.hip ACC1 (off by default: pylc_debug_wgrad_acc1): the 128 x 128 wgrad with ONE accumulator set under 128 registers --
    cross terms scaled by 2^-11 in registers and added into the same fp32 accumulator.  Not bit-identical to the two-accumulator form,
    but the same accuracy class: against fp64 within 2x the default kernel's error."""
    from pylc_amd import ops, layers, optim
    from pylc_amd.lib import lib
    torch.manual_seed(3)
    conv = layers.Conv2d(256, 256, 3, 1, 1, 1).to(dev)
    arena = optim.FlatArena(conv)
    x = nhwc(rnd(5, 2, 256, 32, 32, scale=2.0), dev).requires_grad_(True)
    errs = {}
    try:
        for on in (0, 1):
            lib.pylc_debug_wgrad_acc1(on)
            arena.g.zero_()
            x.grad = None
            y = conv(x)
            dy = nhwc(rnd(6, *y.shape), dev)
            y.backward(dy)
            ops.sync_side_streams()
            torch.cuda.synchronize()
            wr = conv.weight.detach().double().cpu().requires_grad_(True)
            yr = torch.nn.functional.conv2d(x.detach().double().cpu(), wr, None, 1, 1, 1)
            yr.backward(dy.double().cpu())
            errs[on] = rel(conv.weight.grad.detach(), wr.grad)
    finally:
        lib.pylc_debug_wgrad_acc1(0)
    print('wgrad error vs fp64: two accumulators %.2e, one accumulator %.2e' % (errs[0], errs[1]))
    assert errs[0] < 3e-6 and errs[1] < max(3e-6, 2 * errs[0])


@pytest.mark.parametrize('case', [(128, 128, 3, 1, 0, 1, 2, 40, 40), (128, 256, 1, 1, 0, 1, 2, 32, 32), (64, 128, 3, 1, 0, 1, 2, 36, 36)])
def test_conv_with_bias_on_planes(dev, f16x3, case):
    """nn.Conv2d(bias=True) (the U-Net's convs, unet.py:112,116) on fp16-plane operands: the bias is added by the planes kernels' epilogue,
    its gradient is a column sum over the dy planes (pylc_planes_colsum) -- y / dx / dw bit-identical to the fp32-operand kernels, db to
    fp32 summation-order noise and against fp64."""
    from pylc_amd import ops, layers, optim, runtime
    from pylc_amd.lib import lib
    cin, cout, k, st, pad, dil, B, H, W = case
    torch.manual_seed(4)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil, bias=True).to(dev)
    conv.bias.data.uniform_(-0.5, 0.5)
    arena = optim.FlatArena(conv)
    assert conv.takes_planes()
    x = nhwc(rnd(5, B, cin, H, W, scale=2.0), dev).requires_grad_(True)
    out = {}
    for mode in ('fp32', 'planes'):
        runtime.no_planes = mode == 'fp32'
        lib.pylc_debug_pp_flags(1024 if mode == 'fp32' else 0)
        x.grad = None
        arena.g.zero_()
        y = conv(x)
        if mode == 'planes':
            assert getattr(y, '_pylc_dy_pl', False)
        dy = nhwc(rnd(6, *y.shape), dev)
        if mode == 'planes':
            dy = ops.to_planes(dy)                      # as a BatchNorm backward hands it over
        y.backward(dy)
        ops.sync_side_streams()
        torch.cuda.synchronize()
        out[mode] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.detach().clone(), conv.bias.grad.detach().clone())
    runtime.no_planes = False
    lib.pylc_debug_pp_flags(0)
    exact = {'y': cout > 64, 'dx': cin > 64, 'dw': True}        # as test_conv_on_planes_is_bit_identical: same MFMA shape on both sides
    for name, a, b in zip(('y', 'dx', 'dw'), out['fp32'], out['planes']):
        if exact[name]:
            assert torch.equal(a, b), (name, rel(a, b))
        else:
            assert rel(a, b) < 2e-6, (name, rel(a, b))
    db_ref = nhwc(rnd(6, *out['fp32'][0].shape), dev).double().sum((0, 2, 3)).cpu()
    assert rel(out['planes'][3], db_ref) < 1e-5 and rel(out['fp32'][3], db_ref) < 1e-5
    ref = torch.nn.functional.conv2d(x.detach().double().cpu(), conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), st, pad, dil)
    assert rel(out['planes'][0], ref) < 3e-6


@pytest.mark.parametrize('case', [(128, 128, 8, 64, 64), (64, 256, 4, 128, 128), (256, 136, 8, 48, 80),
                                  (64, 64, 4, 128, 128), (128, 64, 4, 124, 124), (64, 48, 4, 128, 128),      # these three: <= 64 output channels, the NARROW form
                                  # >= 512 blocks of 256 pixels x 64 channels: gg_plhn_kernel's column tiles (forward and dgrad), a last column tile of 8 channels, ragged patches
                                  (128, 192, 8, 128, 128), (128, 136, 8, 128, 136), (192, 128, 9, 122, 126)])
def test_halo_kernel_is_bit_identical(dev, f16x3, case):
    """The 3x3 halo variants (conv_pl.hip: a 16x16 output patch's 18x18 input rows DMA'd once per channel chunk -- gg_plh_kernel, eight waves on 256 pixels x
    128 channels, for launches of at least half a round of such tiles; gg_plhn_kernel, four waves on 256 pixels x 64 channels and two blocks per CU, for
    every launch of at least 512 such tiles): forward and dgrad (flipped taps)
    against the fp32-operand kernel bit for bit, against the non-halo planes kernel (debug flag 16384) and against gg_plh_kernel alone (flag 134217728), y also against fp64."""
    from pylc_amd import ops, layers, optim, runtime
    from pylc_amd.lib import lib
    cin, cout, B, H, W = case
    torch.manual_seed(5)
    conv = layers.Conv2d(cin, cout, 3, 1, 1, 1).to(dev)
    arena = optim.FlatArena(conv)
    x = nhwc(rnd(7, B, cin, H, W, scale=2.0), dev).requires_grad_(True)
    out = {}
    for mode, flags, nopl in (('fp32', 1024, True), ('halo', 0, False), ('nohalo', 16384, False), ('plh', 134217728, False)):
        runtime.no_planes = nopl
        lib.pylc_debug_pp_flags(flags)
        x.grad = None
        arena.g.zero_()
        y = conv(x)
        dy = nhwc(rnd(8, *y.shape), dev)
        y.backward(dy)
        ops.sync_side_streams()
        torch.cuda.synchronize()
        out[mode] = (y.detach().clone(), x.grad.clone())
    runtime.no_planes = False
    lib.pylc_debug_pp_flags(0)
    assert torch.equal(out['halo'][0], out['nohalo'][0]) and torch.equal(out['halo'][1], out['nohalo'][1])
    assert torch.equal(out['halo'][0], out['plh'][0]) and torch.equal(out['halo'][1], out['plh'][1])
    if cout > 64:
        assert torch.equal(out['halo'][0], out['fp32'][0])
    if cin > 64:
        assert torch.equal(out['halo'][1], out['fp32'][1])
    ref = torch.nn.functional.conv2d(x.detach().double().cpu(), conv.weight.detach().double().cpu(), None, 1, 1, 1)
    assert rel(out['halo'][0], ref) < 3e-6


@pytest.mark.parametrize('case', [(128, 192, 8, 128, 128, 1), (128, 136, 8, 128, 136, 1), (64, 128, 9, 124, 126, 0)])
def test_halo_column_tiles_statistics(dev, f16x3, case):
    """BatchNorm statistics from gg_plhn_kernel's 64-wide column tiles (each block combines and stores the partials of ITS 64 columns only): the column sums /
    sums of squares of the launch against gg_plh_kernel's (flag 134217728) and the per-tap kernel's (16384) to fp32 summation order, and against y itself in fp64."""
    from pylc_amd import ops, layers, optim
    from pylc_amd.lib import lib
    cin, cout, B, H, W, pad = case
    torch.manual_seed(5)
    conv = layers.Conv2d(cin, cout, 3, 1, pad, 1, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    xp = ops.to_planes(nhwc(rnd(7, B, cin, H, W, scale=2.0), dev))
    out = {}
    try:
        with torch.no_grad():
            for name, flags in (('default', 0), ('plh', 134217728), ('per tap', 16384)):
                lib.pylc_debug_pp_flags(flags)
                y = ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True)
                torch.cuda.synchronize()
                out[name] = (y.clone(), y._pylc_sums.double().sum(0))
    finally:
        lib.pylc_debug_pp_flags(0)
    y = out['default'][0]
    yd = y.double()
    direct = torch.cat([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])
    for name in ('plh', 'per tap'):
        assert torch.equal(y, out[name][0]), name
        assert rel(out['default'][1], out[name][1]) < 2e-6, (name, rel(out['default'][1], out[name][1]))
    got = out['default'][1]
    assert got.numel() == 2 * cout
    assert rel(got, direct) < 2e-6, rel(got, direct)
    del arena


@pytest.mark.parametrize('case', [(256, 256, 3, 1, 1, 1, 2, 32, 32), (72, 200, 3, 1, 6, 6, 3, 30, 30), (1024, 256, 1, 1, 0, 1, 2, 32, 32),
                                  (128, 128, 3, 2, 1, 1, 2, 45, 45)])
def test_256_row_planes_kernel_is_bit_identical(dev, f16x3, case):
    """The 256 x 128 instantiation of gg_pl_kernel (8 waves, three LDS stages, counted vmcnt, wave-pair de-phasing) is picked for long
    reductions on grids of >= 256 tiles -- larger than this file's cases, so it is forced here (debug flag 8192) and compared with the
    128-row instantiation (flag 2048): forward and dgrad bit for bit, incl. row / channel tails, atrous tap skipping, stride-2 dgrad."""
    from pylc_amd import ops, layers, optim
    from pylc_amd.lib import lib
    cin, cout, k, st, pad, dil, B, H, W = case
    torch.manual_seed(6)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil).to(dev)
    arena = optim.FlatArena(conv)
    x = nhwc(rnd(9, B, cin, H, W, scale=2.0), dev).requires_grad_(True)
    out = {}
    try:
        for name, flags in (('128', 2048 | 16384), ('256', 8192 | 16384), ('256-lockstep', 8192 | 16384 | 32768)):
            lib.pylc_debug_pp_flags(flags)
            x.grad = None
            arena.g.zero_()
            y = conv(x)
            dy = nhwc(rnd(10, *y.shape), dev)
            y.backward(dy)
            ops.sync_side_streams()
            torch.cuda.synchronize()
            out[name] = (y.detach().clone(), x.grad.clone())
    finally:
        lib.pylc_debug_pp_flags(0)
    for name in ('256', '256-lockstep'):
        assert torch.equal(out[name][0], out['128'][0]) and torch.equal(out[name][1], out['128'][1]), name
