"""Inference on fp16-plane tensors (-m gpu): pylc_conv2d_fwd_bnact_ex / pylc_dwconv3x3_fwd_h_eval behind ops.conv_bn_act_eval and the nets'
eval paths (Model.eval / Model.test, models/model.py:338-382).  The plane form must compute what the fp32-tensor form computes -- same
kernels' arithmetic, the tensors between them rounded to the operand format (2^-22 relative for the two-plane f16x3 format, 2^-11 for the
one-plane format of precision mode 3) -- and every plane tensor must carry a scale bound that covers it and a true maximum that is one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)


def nhwc(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


@pytest.fixture(params=[2, 3], ids=['f16x3', 'mode3'])
def mode(request, dev):
    from pylc_amd.lib import lib, check
    from pylc_amd import ops, runtime
    prev, prev_min, prev_ep = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS, runtime.eval_planes
    check(lib.pylc_set_conv_precision(request.param))
    ops.PLANES_MIN_PIXELS = 0
    yield request.param
    ops.PLANES_MIN_PIXELS = prev_min
    runtime.eval_planes = prev_ep
    check(lib.pylc_set_conv_precision(prev))


# cin, cout, k, stride, pad, dil, B, H, W, residual ('none' | 'fp32' | 'planes'), relu
CASES = [
    (256, 1024, 1, 1, 0, 1, 2, 32, 32, 'planes', True),      # bottleneck conv3 + identity residual (resnet.py:47-51)
    (1024, 256, 1, 1, 0, 1, 2, 32, 32, 'none', True),        # conv1
    (256, 256, 3, 1, 1, 1, 2, 32, 32, 'none', True),         # conv2: the halo kernel
    (64, 64, 3, 1, 1, 1, 2, 48, 48, 'none', True),           # layer1 3x3: narrow wave layout
    (256, 64, 1, 1, 0, 1, 2, 40, 36, 'fp32', False),         # narrow 1x1, ragged M, fp32 residual, no ReLU
    (128, 128, 3, 2, 1, 1, 2, 45, 45, 'none', True),         # stride 2
    (2048, 256, 3, 1, 12, 12, 2, 32, 32, 'none', True),      # ASPP branch: tap skipping
    (728, 728, 1, 1, 0, 1, 2, 16, 16, 'planes', False),      # Xception pointwise + skip, channel tails
    (64, 64, 3, 1, 1, 1, 8, 128, 128, 'none', True),         # 64 channels on enough patches for the halo kernel's narrow form (the U-Net's first blocks)
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('out_planes', [True, False])
def test_fused_inference_conv_on_planes(dev, mode, case, out_planes):
    from pylc_amd import ops, layers, optim, runtime
    cin, cout, k, st, pad, dil, B, H, W, res_kind, relu = case
    torch.manual_seed(5)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil, bn=True).to(dev)
    bn = layers.BatchNorm2d(cout).to(dev)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * rnd(1, cout).to(dev))
        bn.bias.copy_(0.2 * rnd(2, cout).to(dev))
        bn.running_mean.copy_(0.1 * rnd(3, cout).to(dev))
        bn.running_var.copy_(1 + 0.2 * rnd(4, cout).abs().to(dev))
    arena = optim.FlatArena(torch.nn.ModuleList([conv, bn]))
    conv.eval(); bn.eval()
    x = nhwc(rnd(6, B, cin, H, W, scale=1.5), dev)
    oh, ow = ops.conv_out_size(H, k, st, pad, dil), ops.conv_out_size(W, k, st, pad, dil)
    res = nhwc(rnd(7, B, cout, oh, ow), dev) if res_kind != 'none' else None
    with torch.no_grad():
        runtime.eval_planes = False
        want = layers.conv_bn(conv, bn, x, residual=res, relu=relu)
        assert not ops.is_planes(want)
        runtime.eval_planes = True
        n0 = ops.eval_plane_convs[0]
        xp = ops.to_planes(x)
        rp = ops.to_planes(res) if res_kind == 'planes' else res
        got = layers.conv_bn(conv, bn, xp, residual=rp, relu=relu, out_planes=out_planes)
        torch.cuda.synchronize()
        assert ops.eval_plane_convs[0] == n0 + 1, 'the plane form did not run'
        assert ops.is_planes(got) == out_planes
        true = ops.amax_of(got).view(torch.float32).item()
        vals = ops.as_nhwc(got) if out_planes else got
        peak = want.abs().max().item()
        # operands: x (and a plane residual) rounded to the operand format; output: rounded once more when it leaves as planes
        eps = 2.0 ** -21 if mode == 2 else 2.0 ** -9
        err = (vals - want).abs().max().item()
        assert err <= (8 if (out_planes or mode == 3) else 4) * eps * max(peak, 1.0), (err, peak)
        assert abs(true - vals.abs().max().item()) <= 4 * eps * peak + 1e-30          # the TRUE maximum travels with the tensor
        if out_planes:
            bound = ops.planes_amax(got).view(torch.float32).item()
            assert bound >= true and bound <= 2.0 ** 14 * max(true, 1e-30)             # a bound, and not absurdly loose
    del arena


# cin, cout, k, pad, B, H, W, residual as planes, relu
LEAN_EP_CASES = [
    (256, 1024, 1, 0, 4, 32, 32, True, True),        # bottleneck conv3 + identity residual: whole-line residual loads and stores
    (1024, 256, 1, 0, 4, 32, 32, False, True),       # conv1, the 256-row kernel's shapes
    (256, 256, 3, 1, 4, 32, 32, False, True),        # conv2: the halo kernel
    (64, 64, 3, 1, 2, 64, 64, False, True),          # layer1 3x3, narrow wave layout
    (64, 64, 3, 1, 8, 128, 128, False, True),        # ... on the halo kernel's narrow form
    (256, 64, 1, 0, 2, 64, 64, False, False),        # narrow 1x1, no ReLU
    (128, 512, 1, 0, 2, 40, 36, True, False),        # ragged M: the last tile takes the general path, the others the lean one
    (512, 96, 1, 0, 2, 32, 32, True, True),          # Cout = 96: the second wave column is half outside -> general path beside lean waves
    (128, 128, 3, 0, 2, 68, 68, False, True, True),  # the U-Net's valid 3x3 conv WITH a bias (unet.py:111): 66 x 66 outputs, ragged halo patches
    (64, 128, 1, 0, 2, 64, 64, False, True, True),   # bias on the 1x1 kernels
    (256, 1024, 1, 0, 32, 32, 32, True, True),       # BASELINE size: configs[2]'s conv3 at batch 32 (2048 tiles, four rounds of blocks)
    (728, 728, 1, 0, 8, 64, 64, True, True, True),   # BASELINE size: configs[4]'s middle-flow pointwise conv at 1024^2 / 16, batch 8
]


@pytest.mark.parametrize('prec', [2, 3], ids=['f16x3', 'mode3'])
@pytest.mark.parametrize('case', LEAN_EP_CASES + [(728, 728, 1, 0, 2, 32, 32, True, False), (728, 728, 1, 0, 2, 32, 32, True, True, True)])      # + the Xception pointwise conv with its skip, without / with the folded bias
def test_lean_inference_epilogue_is_bit_identical(dev, case, prec):
    """pl_epilogue_lean_ep / pl_epilogue_lean_ep_half (conv_pl.hip: edge-free tiles of the fused inference conv, whole-line plane stores and
    residual loads, two-plane and one-plane tensors) against the general epilogue of the same kernels (pylc_debug_pp_flags bit 3): the same
    plane bytes, scale bound and true maximum."""
    from pylc_amd import ops, layers, optim, runtime
    from pylc_amd.lib import lib, check
    cin, cout, k, pad, B, H, W, res_planes, relu = case[:9]
    bias = len(case) > 9 and case[9]
    prev, prev_min, prev_ep = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS, runtime.eval_planes
    check(lib.pylc_set_conv_precision(prec))
    ops.PLANES_MIN_PIXELS = 0
    runtime.eval_planes = True
    try:
        torch.manual_seed(5)
        conv = layers.Conv2d(cin, cout, k, 1, pad, 1, bias=bias, init='torch' if bias else 'resnet', bn=True).to(dev)
        bn = layers.BatchNorm2d(cout).to(dev)
        with torch.no_grad():
            bn.weight.copy_(1 + 0.3 * rnd(1, cout).to(dev))
            bn.bias.copy_(0.2 * rnd(2, cout).to(dev))
            bn.running_mean.copy_(0.1 * rnd(3, cout).to(dev))
            bn.running_var.copy_(1 + 0.2 * rnd(4, cout).abs().to(dev))
            if bias:
                conv.bias.copy_(0.5 * rnd(8, cout).to(dev))
        arena = optim.FlatArena(torch.nn.ModuleList([conv, bn]))
        conv.eval(); bn.eval()
        x = nhwc(rnd(6, B, cin, H, W, scale=1.5), dev)
        oh = ops.conv_out_size(H, k, 1, pad, 1)
        res = nhwc(rnd(7, B, cout, oh, oh), dev) if res_planes else None
        out = []
        with torch.no_grad():
            xp = ops.to_planes(x)
            rp = ops.to_planes(res) if res_planes else None
            for flags in (8, 0):
                lib.pylc_debug_pp_flags(flags)
                n0 = ops.eval_plane_convs[0]
                got = layers.conv_bn(conv, bn, xp, residual=rp, relu=relu, out_planes=True)
                torch.cuda.synchronize()
                assert ops.eval_plane_convs[0] == n0 + 1 and ops.is_planes(got)
                raw = got.permute(0, 2, 3, 1).reshape(-1).view(torch.int16)
                if prec == 3:
                    raw = raw[:got.numel()]                  # one plane: the second half of the buffer is never written
                out.append((raw.clone(), ops.amax_of(got).clone(), ops.planes_amax(got).clone()))
        assert torch.equal(out[0][0], out[1][0]), 'plane bytes differ: %d words' % (out[0][0] != out[1][0]).sum().item()
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
        del arena
    finally:
        lib.pylc_debug_pp_flags(0)
        ops.PLANES_MIN_PIXELS = prev_min
        runtime.eval_planes = prev_ep
        check(lib.pylc_set_conv_precision(prev))


def test_resnet_eval_on_planes_matches_fp32_tensors(dev):
    """DeepLabV3+/R101 in eval mode, 2 x 3 x 192 x 160 tiles: logits of the plane-tensor inference path against the fp32-tensor path
    (PYLC_RUNTIME=eval_planes=0, the round-3 path, itself pinned to the reference by the fixtures) -- both fp32-grade, so they agree far inside the
    1e-3 bar, argmax identical off near-ties -- and the plane path really ran (counter), with at most a handful of format conversions."""
    import oracle
    from oracle import step as ostep
    from pylc_amd import ops, runtime
    from pylc_amd.lib import lib, check
    from pylc_amd.model import Model, Meta
    from tests import _data as D
    prev, prev_min, prev_ep = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS, runtime.eval_planes
    check(lib.pylc_set_conv_precision(2))
    ops.PLANES_MIN_PIXELS = 0
    try:
        cfg = ostep.StepConfig('deeplab', 'resnet', 9, 3, dropout=False)
        x = D.tiles(5, 2, 3, 192, 160)
        w = ostep.calibrate_bn(oracle.formula_state(oracle.state_spec('deeplab', 'resnet', 9, 3), salt=8), cfg, x.clone())
        model = Model(Meta(), dev).build()
        model.net.load_state_dict(w)
        model.net.eval()
        out = {}
        for on in (False, True):
            runtime.eval_planes = on
            ops.eval_plane_convs[0] = 0
            ops.plane_conversions[:] = [0, 0]
            out[on] = model.test(x)[0].float().cpu()
            n_pl, n_conv = ops.eval_plane_convs[0], ops.plane_conversions[0]
        assert n_pl >= 100, n_pl                                   # 104 convs of the R101 + ASPP + decoder behind a BatchNorm
        assert n_conv <= 8, n_conv
        diff = (out[True] - out[False]).abs().max().item()
        print('R101 eval: plane tensors vs fp32 tensors max |logit diff| %.3g over %d plane convs, %d conversions' % (diff, n_pl, n_conv))
        assert diff < 2e-4
        top2 = out[False].topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert torch.equal(out[True].argmax(1)[clear], out[False].argmax(1)[clear])
        want = ostep.test_step({k: v.clone() for k, v in w.items()}, cfg, x.clone())
        assert (out[True] - want).abs().max().item() < 1e-3        # and against the CPU oracle: the north_star's bar
    finally:
        runtime.eval_planes = prev_ep
        ops.PLANES_MIN_PIXELS = prev_min
        check(lib.pylc_set_conv_precision(prev))


def test_xception_eval_on_half_planes_mode3(dev):
    """DeepLabV3+/Xception, precision mode 3, eval: every separable conv as depthwise (half -> half, pylc_dwconv3x3_fwd_h_eval) + pointwise
    conv with the folded inner BatchNorm and the outer BatchNorm / residual / ReLU in its epilogue on ONE-PLANE tensors, against the same
    mode with fp32 tensors between the kernels (the round-3 path): logits agree to the one-plane format's accuracy, argmax off near-ties."""
    import oracle
    from oracle import step as ostep
    from pylc_amd import ops, runtime
    from pylc_amd.lib import lib, check
    from pylc_amd.model import Model, Meta
    from tests import _data as D
    prev, prev_min, prev_ep = lib.pylc_get_conv_precision(), ops.PLANES_MIN_PIXELS, runtime.eval_planes
    check(lib.pylc_set_conv_precision(3))
    ops.PLANES_MIN_PIXELS = 0
    try:
        cfg = ostep.StepConfig('deeplab', 'xception', 11, 1, dropout=False)
        x = D.tiles(6, 2, 1, 256, 192)
        w = ostep.calibrate_bn(oracle.formula_state(oracle.state_spec('deeplab', 'xception', 11, 1), salt=9), cfg, x.clone())
        model = Model(Meta(backbone='xception', ch=1, n_classes=11), dev).build()
        model.net.load_state_dict(w)
        model.net.eval()
        out = {}
        for on in (False, True):
            runtime.eval_planes = on
            ops.eval_plane_convs[0] = 0
            out[on] = model.test(x)[0].float().cpu()
            n_pl = ops.eval_plane_convs[0]
        assert n_pl >= 60, n_pl
        scale = out[False].abs().max().item()
        diff = (out[True] - out[False]).abs().max().item()
        print('Xception eval (mode 3): half tensors vs fp32 tensors max |logit diff| %.3g (logit scale %.3g) over %d plane convs' % (diff, scale, n_pl))
        # one-plane tensors round the residual stream and the depthwise inputs to fp16 (2^-11) at every block -- 20 blocks, ~70 layers -- where
        # the fp32-tensor path rounds conv operands only: a few per cent of the logit scale on this random-weight net (measured 3.3 %); the
        # statistical bar for the 16-bit mode is argmax agreement off near-ties (and mIoU: tests/test_mode3_gpu.py, tests/test_fullsize_gpu.py)
        assert diff < 6e-2 * max(scale, 1.0)
        top2 = out[False].topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2.2 * diff            # a flip needs the two logits to move towards each other by the margin
        assert clear.float().mean().item() > 0.15
        assert torch.equal(out[True].argmax(1)[clear], out[False].argmax(1)[clear])
    finally:
        runtime.eval_planes = prev_ep
        ops.PLANES_MIN_PIXELS = prev_min
        check(lib.pylc_set_conv_precision(prev))
