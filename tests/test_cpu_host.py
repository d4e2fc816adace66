"""CPU suite (-m "not gpu"): the C-ABI library loads and exports every declared symbol; host-side logic of the
package (module trees / state_dict names, flat parameter arena, layout helpers, argument validation); and the
"fail loudly" rule -- no op silently falls back to a CPU path."""
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(os.path.dirname(__file__), 'golden')


def test_library_exports_every_header_symbol():
    import ctypes
    from pylc_amd import lib as L
    hdr = open(os.path.join(ROOT, 'include', 'pylc_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    # entry points inside `#ifdef PYLC_EXPERIMENTAL ... #endif` exist only in a library built with EXPERIMENTAL=1 (pylc_amd/csrc/Makefile)
    experimental = sorted(set(n for blk in re.findall(r'#ifdef PYLC_EXPERIMENTAL(.*?)#endif', hdr, flags=re.S)
                              for n in re.findall(r'\b(pylc_[a-z0-9_]+)\s*\(', blk)))
    assert set(experimental) == set(L.EXPERIMENTAL)
    core = re.sub(r'#ifdef PYLC_EXPERIMENTAL.*?#endif', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(pylc_[a-z0-9_]+)\s*\(', core)))
    assert len(declared) >= 40
    dll = ctypes.CDLL(L.LIB_PATH)
    assert bool(dll.pylc_experimental_build()) == L.HAS_EXPERIMENTAL
    for name in declared + (experimental if L.HAS_EXPERIMENTAL else []):
        assert hasattr(dll, name), 'libpylc_hip.so does not export %s' % name
        assert name in L.SIGNATURES, 'pylc_amd/lib.py has no binding for %s' % name
    if not L.HAS_EXPERIMENTAL:
        for name in experimental:
            assert not hasattr(dll, name), 'the product library exports the experimental %s' % name
    assert set(L.SIGNATURES) <= set(declared) | set(experimental)
    assert dll.pylc_abi_version() == L.ABI_VERSION


def test_wgrad_queue_policy_follows_the_precision_mode():
    """runtime.side_stream_on(): filter gradients run on the compute stream for f16x3 (one queue, DESIGN.md 5.2 i) and on the side stream
    for precision mode 3, unless the process forces one schedule; a pending split-K slab sum needs no GPU to be noted and forgotten."""
    import ctypes as C
    from pylc_amd.lib import lib, check, SlabSum
    from pylc_amd.runtime import runtime
    from pylc_amd import ops
    prev_mode, prev_side = lib.pylc_get_conv_precision(), runtime.wgrad_side_stream
    try:
        runtime.wgrad_side_stream = None
        check(lib.pylc_set_conv_precision(2))
        assert runtime.side_stream_on() is False
        check(lib.pylc_set_conv_precision(3))
        assert runtime.side_stream_on() is True
        runtime.wgrad_side_stream = False
        assert runtime.side_stream_on() is False
        check(lib.pylc_set_conv_precision(2))
        runtime.wgrad_side_stream = True
        assert runtime.side_stream_on() is True
    finally:
        runtime.wgrad_side_stream = prev_side
        check(lib.pylc_set_conv_precision(prev_mode))
    assert C.sizeof(SlabSum) == 40                     # include/pylc_hip.h PylcSlabSum
    assert ops.side_stream_if_any('cpu') is None
    ops.flush_slab_sums()                              # nothing pending: no launch, no GPU needed


def test_error_reporting_without_gpu():
    """Argument validation happens on the host before any launch: callable (and failing cleanly) without a GPU."""
    import ctypes as C
    from pylc_amd.lib import lib, ConvDesc
    d = ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad, d.dil = 1, 8, 8, 3, 8, 3, 3, 1, 1, 1
    d.OH, d.OW, d.x_pitch, d.y_pitch = 8, 8, 3, 8
    rc = lib.pylc_conv2d_fwd(C.byref(d), 16, 16, None, 16, None)
    assert rc != 0 and b'multiple of 4' in lib.pylc_last_error()
    d.Cin, d.x_pitch, d.OH = 4, 4, 7
    rc = lib.pylc_conv2d_fwd(C.byref(d), 16, 16, None, 16, None)
    assert rc != 0 and b'inconsistent' in lib.pylc_last_error()
    assert lib.pylc_set_conv_precision(7) != 0
    assert lib.pylc_conv2d_wgrad_workspace(C.byref(d)) == 0          # invalid descriptor -> 0, never a crash


def test_ops_refuse_cpu_tensors():
    from pylc_amd import ops
    from pylc_amd.lib import PylcError
    x = torch.zeros(1, 4, 8, 8).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    with pytest.raises((PylcError, RuntimeError)):
        ops.conv2d(x, w, None, 1, 1, 1)


@pytest.mark.parametrize('tag,kw', [('deeplab_resnet', dict(backbone='resnet', n_classes=9)),
                                    ('deeplab_xception', dict(backbone='xception', n_classes=11)),
                                    ('unet', dict(in_channels=3, n_classes=9, dropout=0.5))])
def test_module_state_dict_names_match_reference(tag, kw):
    import pylc_amd
    keys = json.load(open(os.path.join(HERE, tag + '.json')))['keys']
    net = pylc_amd.UNet(**kw) if tag == 'unet' else pylc_amd.DeepLab(**kw)
    assert [(k, list(v.shape)) for k, v in net.state_dict().items()] == [(k, list(s)) for k, s in keys]
    for name, p in net.named_parameters():
        if p.dim() == 4 and p.shape[1] > 1:
            assert p.permute(0, 2, 3, 1).is_contiguous(), '%s is not KRSC in memory' % name


def test_reference_constructor_signatures():
    """models/model.py:140-147 and :166-173 call the nets with these keywords."""
    import pylc_amd
    pylc_amd.UNet(in_channels=3, n_classes=9, up_mode='upsample', activ_func=torch.nn.ReLU(), normalizer=torch.nn.BatchNorm2d,
                  dropout=0.5)
    pylc_amd.DeepLab(activ_func=torch.nn.ReLU(), normalizer=torch.nn.BatchNorm2d, backbone='resnet', n_classes=9, in_channels=3,
                     pretrained=False)
    with pytest.raises(ValueError):
        pylc_amd.DeepLab(backbone='drn')


def test_flat_arena_rehomes_parameters():
    import pylc_amd
    from pylc_amd.optim import FlatArena
    net = pylc_amd.UNet(in_channels=3, n_classes=9, dropout=0.5)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    strides = {k: p.stride() for k, p in net.named_parameters()}
    arena = FlatArena(net)
    assert arena.numel % 4 == 0 and arena.numel >= sum(p.numel() for p in net.parameters())
    for k, p in net.named_parameters():
        assert torch.equal(p.detach(), before[k]) and p.stride() == strides[k]
        assert p.data_ptr() >= arena.p.data_ptr() and p.data_ptr() < arena.p.data_ptr() + 4 * arena.numel
        assert p.grad is p._pylc_grad and p.grad.shape == p.shape and p.grad.stride() == p.stride()
        assert (p.data_ptr() - arena.p.data_ptr()) % 16 == 0
    # loading a state dict writes straight into the arena
    net.load_state_dict({k: torch.full_like(v, 3) if v.is_floating_point() else v for k, v in before.items()})
    n_param = sum(p.numel() for p in net.parameters())
    assert float(arena.p.double().sum()) == 3.0 * n_param


def test_layout_helpers():
    from pylc_amd import ops
    from pylc_amd.lib import PylcError
    t = ops.empty_nhwc(2, 9, 5, 7, 'cpu', pitch=12)
    assert tuple(t.shape) == (2, 9, 5, 7) and ops.pitch_of(t) == 12
    assert ops.pitch_of(ops.empty_nhwc(3, 16, 1, 1, 'cpu')) == 16
    with pytest.raises(PylcError):
        ops.pitch_of(torch.zeros(2, 8, 4, 4))              # NCHW memory
    assert ops.pitch_of(ops.as_nhwc(torch.zeros(2, 8, 4, 4))) == 8
    assert ops.conv_out_size(512, 7, 2, 3, 1) == 256 and ops.conv_out_size(32, 3, 1, 18, 18) == 32


def test_meta_update_only_existing_keys():
    from pylc_amd.model import Meta
    m = Meta(arch='unet')
    m.update({'lr': 0.5, 'optim': 'sgd', 'not_a_field': 1})      # config.py:259-269: unknown keys are dropped
    assert m.lr == 0.5 and m.optim_type == 'adam' and not hasattr(m, 'not_a_field')
    with pytest.raises(AttributeError):
        Meta(bogus=1)


def test_multiloss_validation_matches_reference_contract():
    from pylc_amd.loss import MultiLoss
    crit = MultiLoss({'weighted': False, 'weights': None, 'ce': 0.5, 'dice': 0.5, 'focal': 0.5},
                     {'n_classes': 9, 'class_codes': None, 'class_labels': None})
    with pytest.raises(TypeError):
        crit(np.zeros((1, 9, 4, 4)), torch.zeros(1, 4, 4, dtype=torch.int64))
    with pytest.raises(ValueError):
        crit(torch.zeros(1, 9, 4, 4), torch.zeros(2, 4, 4, dtype=torch.int64))
    with pytest.raises(ValueError):
        crit(torch.zeros(1, 5, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64))
    with pytest.raises(ValueError):
        MultiLoss({'weighted': True, 'weights': [1, 2], 'ce': 1, 'dice': 0, 'focal': 0}, {'n_classes': 9})


def test_bucket_ranges_cover_arena():
    from pylc_amd.parallel import bucket_ranges
    r = bucket_ranges(100, 32)
    assert r == [(0, 32), (32, 64), (64, 96), (96, 100)]
    assert bucket_ranges(59341228)[-1][1] == 59341228


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """checkpoint.py:51-67 layout; `meta` pickled as config.Parameters without the reference installed; optimizer state in
    torch.optim.AdamW's layout.  (Cross-loading with the real reference: tests/golden/check_checkpoint_compat.py.)"""
    import sys
    from pylc_amd.model import Model, Meta
    from pylc_amd import checkpoint as ck
    m = Model(Meta(arch='unet', n_classes=9, lr=3e-4), 'cpu').build()
    m.optim.steps = 3
    m.optim.m.normal_()
    m.optim.v.uniform_()
    m.epoch, m.iter = 2, 57
    path = str(tmp_path / 'checkpoint.pth')
    ck.save(m, path)
    assert 'config' not in sys.modules
    raw = ck.load_reference_file(path)
    assert sorted(raw) == ['epoch', 'iter', 'meta', 'model', 'optim']
    assert (type(raw['meta']).__module__, type(raw['meta']).__name__) == ('config', 'Parameters') and raw['meta'].lr == 3e-4
    assert list(raw['model']) == list(m.net.state_dict())
    m2 = Model(ck.meta_from_reference(raw['meta']), 'cpu').build()
    ck.load_into(m2, path, resume=True)
    assert (m2.iter, m2.epoch, m2.optim.steps) == (57, 2, 3) and abs(m2.optim.lr - 3e-4) < 1e-12
    for (p, off), (p2, off2) in zip(zip(m.arena.params, m.arena.offsets), zip(m2.arena.params, m2.arena.offsets)):
        assert torch.equal(p, p2)
        assert torch.equal(torch.as_strided(m.optim.v, p.shape, p.stride(), off), torch.as_strided(m2.optim.v, p2.shape, p2.stride(), off2))
    opt = torch.optim.AdamW(list(m2.net.parameters()), lr=1e-4)
    opt.load_state_dict(raw['optim'])                              # a stock AdamW accepts the exported state
    best = str(tmp_path / 'best.pth')
    ck.save(m, best, best=True)
    assert sorted(ck.load_reference_file(best)) == ['meta', 'model', 'optim']


def test_reference_written_checkpoint_loads(tmp_path):
    """SURVEY.md section 8 row f2 on every box: tests/golden/ref_checkpoint_tiny.pth / ref_model_tiny.pth were written by the REFERENCE's
    own Checkpoint.save (make_checkpoint_fixture.py: the reference's stem modules as a stand-in net, its AdamW stepped once, its pickled
    config.Parameters -- with numpy-typed fields, which pickle protocol 2 routes through _codecs.encode).  The tolerant, allowlisted
    reader must take both files without the reference installed, and the optimiser state must land in the flat arena."""
    import json
    import os
    import sys
    from torch import nn
    from pylc_amd import checkpoint as ck, layers, optim
    from tests import _data as D
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    want = json.load(open(os.path.join(here, 'ref_checkpoint_tiny.json')))
    data = ck.load_reference_file(os.path.join(here, 'ref_checkpoint_tiny.pth'))
    assert 'config' not in sys.modules and 'models' not in sys.modules            # nothing of the reference was imported to read it
    assert sorted(data) == ['epoch', 'iter', 'meta', 'model', 'optim'] and (data['epoch'], data['iter']) == (want['epoch'], want['iter'])
    assert [[k, list(v.shape)] for k, v in data['model'].items()] == want['keys']
    meta = ck.meta_from_reference(data['meta'])
    for k, v in want['meta'].items():
        assert getattr(meta, k) == v, k
    assert isinstance(data['meta'].px_mean, np.ndarray) and float(data['meta'].m2) == 0.125      # numpy-typed meta fields survive
    best = ck.load_reference_file(os.path.join(here, 'ref_model_tiny.pth'))
    assert sorted(best) == ['meta', 'model', 'optim']                               # checkpoint.py:61-66: no epoch / iter in the model file

    class Stem(nn.Module):              # the same two layers on pylc_amd's modules, under the reference's key names
        def __init__(self):
            super().__init__()
            self.backbone = nn.Module()
            self.backbone.conv1 = layers.Conv2d(3, 64, 7, 2, 3)
            self.backbone.bn1 = layers.BatchNorm2d(64)
    net = Stem()
    net.load_state_dict(data['model'])
    for k, v in net.state_dict().items():
        if v.is_floating_point():
            assert np.allclose(D.digest(v), want['digest'][k], rtol=1e-12, atol=0), k
    arena = optim.FlatArena(net)
    opt = optim.FlatAdamW(arena)
    ck.adamw_state_from_torch(opt, data['optim'])
    assert opt.steps == 1 and abs(opt.lr - want['lr']) < 1e-15
    for i, (p, off) in enumerate(zip(arena.params, arena.offsets)):
        assert np.allclose(D.digest(torch.as_strided(opt.m, p.shape, p.stride(), off)), want['exp_avg_digest'][str(i)], rtol=1e-12, atol=0)
        assert np.allclose(D.digest(torch.as_strided(opt.v, p.shape, p.stride(), off)), want['exp_avg_sq_digest'][str(i)], rtol=1e-12, atol=0)
    # and a stock torch.optim.AdamW over the same parameters accepts the reference's state as is
    torch.optim.AdamW(list(net.parameters()), lr=1e-4).load_state_dict(data['optim'])


def test_loss_log_persists_like_running_loss(tmp_path):
    """RunningLoss.save / .load (models/modules/loss.py:253-268, 296-305): losses.pth holds {"train", "valid", "test", "best_dice", "lr"};
    a resumed run takes up train / valid / test / best_dice, a fresh run deletes the file."""
    from pylc_amd.model import LossLog
    a = LossLog()
    a.train += [(0, 2.1, 0.9, 0.5), (20, 1.7, 0.8, 0.4)]
    a.valid += [(3, 1.9, 0.85, 0.45)]
    a.lr += [(0, 1e-4)]
    a.best_dice = 0.85
    path = str(tmp_path / 'losses.pth')
    a.save(path)
    raw = torch.load(path, weights_only=True)
    assert sorted(raw) == ['best_dice', 'lr', 'test', 'train', 'valid'] and raw['train'][1] == (20, 1.7, 0.8, 0.4)
    b = LossLog()
    assert b.load(path, resume=True) and b.train == a.train and b.valid == a.valid and b.best_dice == 0.85 and b.lr == []
    c = LossLog()
    assert not c.load(path, resume=False) and not os.path.exists(path) and c.train == []
    assert not c.load(path, resume=True)


def test_reference_written_loss_log_resumes(tmp_path):
    """the reference's RunningLoss stores validation averages and best_dice as NUMPY scalars (Model.eval appends
    `.cpu().numpy()` values, models/model.py:360-363, loss.py:284-304), which torch.load(weights_only=True) refuses.
    tests/golden/ref_losses_tiny.pth was written by the reference's RunningLoss.save (make_checkpoint_fixture.py); LossLog.load must take
    it through the allow-listed unpickler, coerce every entry to python floats, and `load_into(..., resume=True)` must not raise for a
    reference directory that holds a losses.pth.  An unreadable file raises (the reference's behaviour) unless PYLC_RESTART_LOSS_LOG=1."""
    import json
    import shutil
    import warnings
    from pylc_amd.model import LossLog
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    want = json.load(open(os.path.join(here, 'ref_checkpoint_tiny.json')))['losses']
    assert want['valid_types'][1:] == ['float32'] * 3           # the fixture really holds numpy scalars
    with pytest.raises(Exception):
        torch.load(os.path.join(here, 'ref_losses_tiny.pth'), weights_only=True)      # why the plain loader is not enough
    log = LossLog()
    assert log.load(os.path.join(here, 'ref_losses_tiny.pth'), resume=True)
    assert [list(r) for r in log.train] == want['train'] and [list(r) for r in log.valid] == want['valid'] and log.best_dice == want['best_dice']
    assert all(type(row[0]) is int and all(type(v) is float for v in row[1:]) for row in log.train + log.valid) and type(log.best_dice) is float
    out = str(tmp_path / 'again.pth')
    log.save(out)                                               # and what we write back stays loadable by the strict loader
    assert torch.load(out, weights_only=True)['best_dice'] == want['best_dice']
    # a torn / foreign file fails loudly, like the reference (torch.load raises at loss.py:259): silently restarting would reset best_dice and
    # let the first validation overwrite the best-model file ...
    bad = str(tmp_path / 'losses.pth')
    open(bad, 'wb').write(b'not a zip')
    fresh = LossLog()
    with pytest.raises(RuntimeError, match='unreadable'):
        fresh.load(bad, resume=True)
    assert fresh.train == [] and fresh.best_dice == 1.0
    # ... unless the caller opts in to restarting the log
    os.environ['PYLC_RESTART_LOSS_LOG'] = '1'
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            assert fresh.load(bad, resume=True) is False and fresh.train == [] and fresh.best_dice == 1.0
        assert any('loss log restarts' in str(x.message) for x in w)
    finally:
        del os.environ['PYLC_RESTART_LOSS_LOG']
    # checkpoint.load_into(resume=True) next to a reference-written losses.pth (the tiny stem checkpoint needs a matching net)
    from torch import nn
    from pylc_amd import checkpoint as ck, layers, optim

    class Stem(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = nn.Module()
            self.backbone.conv1 = layers.Conv2d(3, 64, 7, 2, 3)
            self.backbone.bn1 = layers.BatchNorm2d(64)

    class Holder:
        pass
    d = tmp_path / 'run'
    d.mkdir()
    shutil.copy(os.path.join(here, 'ref_checkpoint_tiny.pth'), str(d / 'checkpoint.pth'))
    shutil.copy(os.path.join(here, 'ref_losses_tiny.pth'), str(d / 'losses.pth'))
    m = Holder()
    m.net = Stem()
    m.optim = optim.FlatAdamW(optim.FlatArena(m.net))
    m.loss = LossLog()
    ck.load_into(m, str(d / 'checkpoint.pth'), resume=True)
    assert (m.epoch, m.iter) == (3, 41) and m.loss.best_dice == want['best_dice'] and len(m.loss.valid) == 1


def test_model_file_with_numpy_meta_loads(tmp_path):
    """torch.save's default pickle protocol 2 serialises numpy scalars / arrays through _codecs.encode, which the allowlisted
    unpickler has to resolve -- a meta holding np.float64 / np.float32 / an ndarray (profile.py's statistics) must load."""
    from pylc_amd import checkpoint as ck
    path = str(tmp_path / 'm.pth')
    meta = {'arch': 'unet', 'm2': np.float64(0.5), 'jsd': np.float32(0.25), 'px_mean': np.asarray([1.0, 2.0, 3.0]), 'probs': np.arange(4, dtype=np.float32)}
    torch.save({'model': {'w': torch.ones(2)}, 'optim': {}, 'meta': meta}, path)
    got = ck.load_reference_file(path)['meta']
    assert float(got['m2']) == 0.5 and float(got['jsd']) == 0.25 and got['px_mean'].tolist() == [1.0, 2.0, 3.0] and got['probs'].dtype == np.float32
    # the allowlist still refuses everything else
    import pickle

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('true',))
    torch.save({'model': {}, 'optim': {}, 'meta': Evil()}, path)
    with pytest.raises(pickle.UnpicklingError):
        ck.load_reference_file(path)


def test_custom_ops_are_registered():
    """torch.ops.pylc_hip.* exist with schemas and fake (meta) implementations (no GPU needed to trace through them)."""
    import torch
    import pylc_amd  # noqa: F401
    from pylc_amd import torch_ops
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in torch_ops.REGISTERED:
        assert hasattr(torch.ops.pylc_hip, name), name
    with FakeTensorMode():
        x, w = torch.empty(2, 64, 16, 20), torch.empty(128, 64, 3, 3)
        y = torch.ops.pylc_hip.conv2d(x, w, None, 2, 1, 1)
        assert tuple(y.shape) == (2, 128, 8, 10)
        out = torch.ops.pylc_hip.batch_norm_act(y, torch.empty(128), torch.empty(128), torch.empty(128), torch.empty(128), None, True, True, 1e-5, 0.1)
        assert tuple(out[0].shape) == (2, 128, 8, 10) and tuple(out[1].shape) == (512,)
        assert tuple(torch.ops.pylc_hip.bilinear(y, 32, 40).shape) == (2, 128, 32, 40)
        assert tuple(torch.ops.pylc_hip.max_pool2d(y, 3, 2, 1)[0].shape) == (2, 128, 4, 5)
        losses, stats = torch.ops.pylc_hip.multiloss(torch.empty(2, 9, 8, 8), torch.empty(2, 8, 8, dtype=torch.int64), None, 0.5, 0.5, 0.5)
        assert tuple(losses.shape) == (4,) and tuple(stats.shape) == (30,)


def test_pretrained_backbone_load(tmp_path):
    """DeepLab(pretrained=True) / load_pretrained_backbone (resnet.py:149-158): entries whose keys the backbone has are copied, others (fc.*)
    ignored; a missing file raises like the reference's torch.load."""
    import torch
    from pylc_amd import DeepLab
    net = DeepLab(backbone='resnet', n_classes=9)
    own = net.backbone.state_dict()
    fake = {'conv1.weight': torch.full_like(own['conv1.weight'], 0.25), 'layer1.0.bn1.running_var': torch.full_like(own['layer1.0.bn1.running_var'], 3.0),
            'fc.weight': torch.zeros(1000, 2048), 'fc.bias': torch.zeros(1000)}
    path = tmp_path / 'resnet101-5d3b4d8f.pth'
    torch.save(fake, str(path))
    picked = net.load_pretrained_backbone(str(path))
    assert picked == ['conv1.weight', 'layer1.0.bn1.running_var']
    sd = net.backbone.state_dict()
    assert float(sd['conv1.weight'].min()) == 0.25 and float(sd['layer1.0.bn1.running_var'].max()) == 3.0
    assert net.backbone.conv1.weight.permute(0, 2, 3, 1).is_contiguous()          # KRSC memory survives the load
    import pytest
    with pytest.raises(FileNotFoundError):
        DeepLab(backbone='resnet', n_classes=9, pretrained=True)                   # ./data/models/resnet101-5d3b4d8f.pth is not here
    with pytest.raises(ValueError):
        DeepLab(backbone='xception', n_classes=9).load_pretrained_backbone(str(path))


def test_unet_upconv_state_dict_keys():
    from pylc_amd import UNet
    n = UNet(in_channels=3, n_classes=9, up_mode='upconv', dropout=0.5)
    sd = n.state_dict()
    assert tuple(sd['decoder.0.up.weight'].shape) == (1024, 512, 2, 2) and tuple(sd['decoder.0.up.bias'].shape) == (512,)      # nn.ConvTranspose2d(1024, 512, 2, 2)
    assert 'decoder.0.up.1.weight' not in sd
    assert 'decoder.0.up.1.weight' in UNet(in_channels=3, n_classes=9, up_mode='upsample', dropout=0.5).state_dict()
    import pytest
    with pytest.raises(ValueError):
        UNet(in_channels=3, n_classes=9, up_mode='nearest')


def test_arena_refreshes_ranges_only_when_a_parameter_changed():
    """FlatArena.refresh_if_changed (Model.eval / Model.test call it per batch): nothing happens while no parameter was written through
    torch; an in-place update of a parameter (its version counter moves) triggers one refresh, i.e. one new `generation` for everything
    derived from the weights (prepared filter planes, folded inference filters, cached BatchNorm coefficients)."""
    import torch
    from pylc_amd import UNet
    from pylc_amd.optim import FlatArena
    net = UNet(in_channels=3, n_classes=4, depth=2, wf=2)
    arena = FlatArena(net)
    g0 = arena.generation
    for _ in range(3):
        arena.refresh_if_changed()
    assert arena.generation == g0
    with torch.no_grad():
        next(net.parameters()).mul_(0.5)
    arena.refresh_if_changed()
    assert arena.generation == g0 + 1
    arena.refresh_if_changed()
    assert arena.generation == g0 + 1
    arena.refresh_ranges()                 # what the optimisers and load_state_dict do themselves
    assert arena.generation == g0 + 2
    arena.refresh_if_changed()
    assert arena.generation == g0 + 2


def test_arena_data_writes_need_invalidate_or_the_forced_refresh():
    """`p.data.copy_()` / `p.data.mul_()` do NOT bump `p._version`, so refresh_if_changed() alone misses them.  The two remedies:
    `arena.invalidate()`, and the forced refresh that Model.eval / Model.test make on the first batch after a training step."""
    import torch
    from pylc_amd import UNet
    from pylc_amd.optim import FlatArena
    net = UNet(in_channels=3, n_classes=4, depth=2, wf=2)
    arena = FlatArena(net)
    g0 = arena.generation
    next(net.parameters()).data.mul_(2.0)             # invisible to the version counters
    arena.refresh_if_changed()
    assert arena.generation == g0                      # (documented blind spot)
    arena.invalidate()
    arena.refresh_if_changed()
    assert arena.generation == g0 + 1
    arena.refresh_if_changed()
    assert arena.generation == g0 + 1
    arena.refresh_if_changed(force=True)               # what Model._refresh_for_inference does after a training step
    assert arena.generation == g0 + 2


def test_native_comm_falls_back_with_a_logged_reason(monkeypatch, capsys):
    """PYLC_COMM=native (parallel.try_native_comm): if the C ABI's RCCL communicator cannot be created the run continues on the
    torch.distributed path and says why; a half-created pair of communicators is torn down."""
    from pylc_amd import parallel
    from pylc_amd.runtime import runtime

    made, destroyed = [], []

    def init(raw, rank, world):
        if made:                          # the first communicator came up, the second did not
            raise RuntimeError('pylc_comm: librccl.so.1 not loadable: test')
        made.append(object())
        return made[-1]
    monkeypatch.setattr(parallel, '_comm_available', lambda: None)
    monkeypatch.setattr(parallel, '_comm_unique_id', lambda: bytes(range(1, 129)))
    monkeypatch.setattr(parallel, '_comm_init', init)
    monkeypatch.setattr(parallel, '_comm_destroy', destroyed.append)
    monkeypatch.setattr(parallel, '_comm_first_message', lambda handles, world: None)
    assert parallel.try_native_comm(0, 1) is False
    assert runtime.comm is None and runtime.grad_comm is None and destroyed == made and len(made) == 1
    err = capsys.readouterr().err
    assert 'PYLC_COMM=native unavailable' in err and 'not loadable' in err and 'torch.distributed' in err
    # ... and the probe alone failing (no RCCL in the process): nothing is created, the reason is the loader's
    def no_rccl():
        raise RuntimeError('pylc_comm: librccl.so.1 not loadable: probe')
    monkeypatch.setattr(parallel, '_comm_available', no_rccl)
    assert parallel.try_native_comm(0, 1) is False and len(made) == 1
    assert 'RCCL probe' in capsys.readouterr().err
    # a message the C ABI does not take goes through torch.distributed instead of raising
    import torch
    assert runtime.native_takes(torch.zeros(4)) and runtime.native_takes(torch.zeros(4, dtype=torch.float64))
    assert not runtime.native_takes(torch.zeros(4, dtype=torch.int64)) and not runtime.native_takes(torch.zeros(4, 2)[:, 0])
