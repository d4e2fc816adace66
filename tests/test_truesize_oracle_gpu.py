"""Whole networks at BASELINE tile sizes against the CPU oracle (-m gpu; VERDICT r5 "missing #2").

Every other HIP-vs-oracle / HIP-vs-fixture NETWORK test runs at <= 96^2 (the reference fixtures), 80 x 112 or 64^2; the true-size tests
of tests/test_fullsize_gpu.py are property tests and those of tests/test_fullsize_ops_gpu.py check one layer at a time.  A defect of the
COMPOSITION that depends on the size -- a plane-format hand-over, a range bound, the concat-by-slice offsets, the max-pool / bilinear
plane writers at 128^2 / 512^2, a tile remap at thousands of blocks -- is invisible to all of them.  Here the oracle (oracle.step, pinned
bit-exactly to the reference: tests/golden/make_golden.py) runs the SAME step at the BASELINE tile size with a batch the host finishes
in seconds, and the HIP path is compared with it output for output:

  (i)   DeepLabV3+/ResNet-101, 3-ch 512^2, 9 classes, bs 2 (configs[2]'s tile; models/model.py:282-336 train, :367-382 test)
  (ii)  U-Net, 3-ch 512^2 -> 324^2, bs 1 (configs[1]'s tile; unet.py:91-104)
  (iii) DeepLabV3+/Aligned-Xception, 1-ch 1024^2, 11 classes, bs 2, eval logits in f16x3 and in precision mode 3 (configs[4]'s tile)

Every layer of these runs takes the fp16-plane kernels the bench measures (the fixture `every_layer_on_planes` lowers the pixel threshold to what
bs 2 leaves of the 32^2 maps).  Tolerances are the north_star's
(logits / loss 1e-3, argmax exact off near-ties) and, for gradients, the bound of tests/test_nets_gpu.py::test_train_steps_match_reference
with the oracle's own reference-vs-reference noise at THIS size as its floor (tests/golden/make_truesize_conditioning.py ->
truesize_conditioning.json; cosine > 0.999: a transposed filter, a permuted channel or a wrong offset is a ~100 % difference)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
LOSS_TOL = 1e-3
GRAD_CAP = 0.05

R101_GRADS = ['backbone.conv1.weight', 'backbone.layer1.0.conv2.weight', 'backbone.layer2.3.conv3.weight', 'backbone.layer3.11.conv2.weight',
              'backbone.layer4.2.conv2.weight', 'aspp.aspp3.atrous_conv.weight', 'decoder.last_conv.0.weight', 'decoder.last_conv.8.weight',
              'backbone.layer3.22.bn3.weight', 'decoder.bn1.bias']

# tag -> (arch, backbone, classes, input channels, batch, tile, weight salt, data seed)
CASES = {'r101_512': ('deeplab', 'resnet', 9, 3, 2, 512, 11, 71),
         'unet_512': ('unet', None, 9, 3, 1, 512, 12, 73),
         'xception_1024': ('deeplab', 'xception', 11, 1, 2, 1024, 13, 75)}      # (bs 2: the fixture's BatchNorm calibration is a training-mode forward, and the ASPP's image-pool BatchNorm sees one value per tile)


def grad_keys(tag, w):
    """The tensors compared elementwise: R101 -- ten spread over stem, the four stages, ASPP and decoder; U-Net -- first / second / middle /
    last conv filters and two BatchNorm vectors."""
    if tag == 'r101_512':
        return R101_GRADS
    convs = [k for k, v in w.items() if k.endswith('weight') and v.dim() == 4]
    bns = [k for k, v in w.items() if k.endswith('weight') and v.dim() == 1]
    return [convs[0], convs[1], convs[len(convs) // 2], convs[-3], convs[-1], bns[0], bns[-1]]


def conditioning(tag):
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'truesize_conditioning.json')))[tag]


def _setup(tag, dev):
    arch, backbone, n_cls, ch, b, hw, salt, seed = CASES[tag]
    import oracle
    from oracle import step as ostep
    from pylc_amd.model import Model, Meta
    from pylc_amd import runtime
    from tests import _data as D
    runtime.dropout_enabled = False
    cfg = ostep.StepConfig(arch, backbone, n_cls, ch, dropout=False)
    spec = oracle.state_spec(arch, backbone, n_cls, 3 if arch == 'deeplab' else ch)
    x = D.tiles(seed, b, ch, hw, hw)
    y = D.blob_masks(seed + 1, b, hw, hw, n_cls, cell=32)
    w = ostep.calibrate_bn(oracle.formula_state(spec, salt=salt), cfg, x.clone())
    model = Model(Meta(arch=arch, backbone=backbone, ch=ch, n_classes=n_cls), dev).build()
    model.net.load_state_dict(w)
    return model, cfg, w, x, y


def _check_eval(model, cfg, w, x, tol, tag, min_decided=0.9, min_agree=None, rms_tol=None):
    """Eval logits against the oracle: max |diff| < tol, argmax identical wherever the oracle's top-1 / top-2 margin exceeds 2 tol (a logit error
    below tol cannot flip such a pixel), that set being at least `min_decided` of the pixels."""
    from oracle import step as ostep
    model.net.eval()
    got = model.test(x)[0].float().cpu()
    want = ostep.test_step({k: v.clone() for k, v in w.items()}, cfg, x.clone())
    assert tuple(got.shape) == tuple(want.shape)
    err = (got - want).abs().max().item()
    rms = (got - want).double().pow(2).mean().sqrt().item()
    top2 = want.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 2 * tol
    agree = (got.argmax(1) == want.argmax(1)).float().mean().item()
    print('%s eval logits %s max|diff| %.3g rms %.3g (|logits| max %.3g); decided %.4f; argmax agreement %.6f'
          % (tag, tuple(got.shape), err, rms, want.abs().max().item(), decided.float().mean().item(), agree))
    assert err < tol
    if rms_tol is not None:
        assert rms < rms_tol
    assert decided.float().mean().item() > min_decided
    assert torch.equal(got.argmax(1)[decided], want.argmax(1)[decided])
    if min_agree is not None:
        assert agree > min_agree
    model.net.train()
    return err


def _check_train_step(model, cfg, w, x, y, keys, head_key, tag, cond):
    from oracle import step as ostep
    from pylc_amd import ops
    ops.planes_marked[0] = 0
    model.net.train()
    model.train(x, y)
    torch.cuda.synchronize()
    got = [float(model.crit.ce), float(model.crit.dsc), float(model.crit.fl)]
    gnorm, coef = model.optim.norm.cpu().tolist()
    sd = {k: v.clone() for k, v in w.items()}
    opt = ostep.make_optimizer(sd, cfg)
    ce, dsc, fl, _, _, ref_norm = ostep.train_step(sd, opt, cfg, x.clone(), y.clone())
    print('%s train step: HIP (%.6f %.6f %.6f) oracle (%.6f %.6f %.6f); |g| %.5f vs %.5f; %d plane tensors'
          % (tag, got[0], got[1], got[2], ce, dsc, fl, gnorm, ref_norm, ops.planes_marked[0]))
    for a, b in zip(got, (ce, dsc, fl)):
        assert abs(a - b) < LOSS_TOL, (got, (ce, dsc, fl))
    assert abs(gnorm - ref_norm) < max(2e-3, 4 * cond['gnorm_rel']) * ref_norm
    assert ops.planes_marked[0] > 50, 'the step did not run on the fp16-plane kernels'
    params = dict(model.net.named_parameters())
    for k in keys:
        ref_g = sd[k].grad.double()                               # clipped in place by clip_grad_norm_ (model.py:326)
        got_g = (params[k].grad.double() * coef).cpu()
        assert got_g.shape == ref_g.shape, k
        amax = ref_g.abs().max().item()
        err = (got_g - ref_g).abs().max().item()
        cos = float((got_g * ref_g).sum() / (got_g.norm() * ref_g.norm()))
        # the bound of test_nets_gpu.py::test_train_steps_match_reference -- the oracle's own 1-vs-8-thread noise on this tensor (x 8), at least
        # 2 % and at most GRAD_CAP of the tensor's largest entry -- but never tighter than 4 x that noise: with TWO tiles the 32^2 maps of
        # layer4 / the ASPP hold 2048 pixels, one ReLU flip moves a filter gradient by per cents, and the reference differs from ITSELF by 4.8 %
        # on layer4.2.conv2 and 2.7 % on the ASPP's d = 12 filter (tests/golden/truesize_conditioning.json).  The cosine bound stays.
        c = cond['grads'][k]
        tol = max(min(max(2e-2 * amax, 8 * c['cond_maxdiff']), GRAD_CAP * amax), 4 * c['cond_maxdiff'])
        print('%s grad %-40s max|diff| %.3g = %.4f of |g|max %.3g (bound %.4f; oracle 1-vs-8 threads %.4f)   cos %.7f'
              % (tag, k, err, err / amax, amax, tol / amax, c['cond_maxdiff'] / c['absmax'], cos))
        assert err <= tol and cos > 0.999, (k, err, tol, amax, cos)
    d = (model.net.state_dict()[head_key].cpu() - sd[head_key].detach()).abs().max().item()
    print('%s post-AdamW max|diff| on %s = %.3g' % (tag, head_key, d))
    assert d < 2.5e-4                                                # lr 1e-4: one AdamW step moves an element by <= 1e-4 (+ decay)
    # BatchNorm running statistics after the step (momentum 0.1 over the batch statistics)
    new = model.net.state_dict()
    for k in [k for k in sd if k.endswith('running_var')][::17]:
        assert (new[k].cpu() - sd[k]).abs().max().item() < 1e-4 * max(1.0, sd[k].abs().max().item()), k


@pytest.fixture
def every_layer_on_planes():
    """bs 2 puts the 32^2 maps of layer3 / layer4 / the ASPP at 2048 pixels, below the product's plane threshold (8192; at bs 32 they hold
    32768): lower it so that every layer takes the kernel family it takes in the BASELINE configuration."""
    from pylc_amd import ops
    prev = ops.PLANES_MIN_PIXELS
    ops.PLANES_MIN_PIXELS = 1024
    yield
    ops.PLANES_MIN_PIXELS = prev


def test_r101_512_bs2_step_against_oracle(dev, every_layer_on_planes):
    model, cfg, w, x, y = _setup('r101_512', dev)
    _check_eval(model, cfg, w, x, LOGIT_TOL, 'R101 512^2')
    _check_train_step(model, cfg, w, x, y, grad_keys('r101_512', w), 'decoder.last_conv.8.weight', 'R101 512^2', conditioning('r101_512'))


def test_unet_512_bs1_step_against_oracle(dev, every_layer_on_planes):
    model, cfg, w, x, y = _setup('unet_512', dev)
    keys = grad_keys('unet_512', w)
    _check_eval(model, cfg, w, x, LOGIT_TOL, 'U-Net 512^2')
    _check_train_step(model, cfg, w, x, y, keys, keys[4], 'U-Net 512^2', conditioning('unet_512'))


@pytest.mark.parametrize('mode', [2, 3])
def test_xception_1024_gray_eval_against_oracle(dev, mode):
    """configs[4]'s tile through Model.test in the default arithmetic (1e-3) and in precision mode 3 -- one fp16 plane per tensor: every operand
    carries 2^-12 of relative rounding per layer, a statistical error, so the bound is the one of tests/test_mode3_gpu.py scaled to the sample:
    over 23 M logits the largest deviation is 4.2e-2 (measured; 3e-2 holds for the 96^2 fixture's 0.1 M), the rms 7.0e-3; bounds: max
    6e-2, rms 1.5e-2, argmax identical wherever the oracle's margin exceeds 0.12, overall agreement > 97 % (measured 97.5 %: a third of this
    random-weight network's pixels are near-ties)."""
    from pylc_amd.lib import lib, check
    prev = lib.pylc_get_conv_precision()
    check(lib.pylc_set_conv_precision(mode))
    try:
        model, cfg, w, x, y = _setup('xception_1024', dev)
        if mode == 2:
            _check_eval(model, cfg, w, x, LOGIT_TOL, 'Xception 1024^2 f16x3')
        else:
            _check_eval(model, cfg, w, x, 6e-2, 'Xception 1024^2 mode 3', min_decided=0.25, min_agree=0.97, rms_tol=1.5e-2)
    finally:
        check(lib.pylc_set_conv_precision(prev))
