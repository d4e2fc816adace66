"""Checkpoint / model-file compatibility with the reference (models/modules/checkpoint.py:51-67, models/model.py:78-121,
222-236).

Reference files are `torch.save`d dicts  {"epoch", "iter", "model": state_dict, "optim": AdamW.state_dict(), "meta":
config.Parameters instance}  (the best-model file omits epoch/iter).  Two obstacles:
  * `meta` is a pickled `config.Parameters`, so loading needs that class importable -- here a tolerant unpickler maps it
    (and anything else from the reference's modules) onto a plain attribute bag;
  * files written here must unpickle inside the reference, so `meta` is pickled under the class path `config.Parameters`
    without the reference being installed (default object reduction: __new__ + __dict__ update, no __init__ call).
Optimizer state is exchanged in torch.optim.AdamW's state_dict layout (per-parameter step / exp_avg / exp_avg_sq)."""
import pickle
import sys
import types

import torch

from .model import Meta


class _Bag:
    """Stand-in for reference classes that are not importable here (config.Parameters)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


# Globals a PyLC model file may legitimately reference.  Model files are downloaded / published artefacts (README.md:90-103),
# so the unpickler resolves ONLY these; anything else raises instead of importing (and executing) an arbitrary callable.
_ALLOWED_GLOBALS = {
    ('collections', 'OrderedDict'),
    ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_tensor'), ('torch._utils', '_rebuild_parameter'),
    ('torch._utils', '_rebuild_parameter_with_state'), ('torch._tensor', '_rebuild_from_type_v2'),
    ('torch', 'Size'), ('torch', 'device'), ('torch', 'dtype'), ('torch', 'Tensor'), ('torch.nn.parameter', 'Parameter'),
    ('torch.serialization', '_get_layout'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'), ('numpy', 'dtype'),
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy', 'ndarray'),
    ('_codecs', 'encode'),      # how pickle protocol 2 (torch.save's default) carries the bytes of numpy scalars / arrays
    ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'complex'), ('builtins', 'bytearray'),
}
_ALLOWED_TORCH_STORAGE = ('FloatStorage', 'DoubleStorage', 'HalfStorage', 'BFloat16Storage', 'LongStorage', 'IntStorage',
                          'ShortStorage', 'CharStorage', 'ByteStorage', 'BoolStorage', 'UntypedStorage')
_BAGGED_PACKAGES = ('config', 'utils', 'models', 'db')      # the reference's own modules: mapped onto attribute bags, never imported


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split('.')[0] in _BAGGED_PACKAGES:
            return type(name, (_Bag,), {'__module__': module})
        if (module, name) in _ALLOWED_GLOBALS or (module in ('torch', 'torch.storage') and name in _ALLOWED_TORCH_STORAGE):
            return super().find_class(module, name)
        raise pickle.UnpicklingError('model file references %s.%s, which is not on the allowlist of a PyLC model file' % (module, name))


_tolerant_pickle = types.ModuleType('pylc_tolerant_pickle')
_tolerant_pickle.Unpickler = _TolerantUnpickler
_tolerant_pickle.load = lambda f, **kw: _TolerantUnpickler(f, **kw).load()
_tolerant_pickle.__name__ = 'pickle'


def load_reference_file(path, map_location='cpu'):
    """Read a reference checkpoint / model file (also the published Zenodo models) without the reference installed."""
    data = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_tolerant_pickle)
    if not isinstance(data, dict) or 'model' not in data:
        raise ValueError('%s is not a PyLC model file (no "model" entry)' % path)
    return data


def load_losses_file(path):
    """Read a `losses.pth` (RunningLoss.save, models/modules/loss.py:296-305) written by this package OR by the reference -- whose
    entries are numpy scalars -- through the allow-listed unpickler.  Returns the raw dict."""
    data = torch.load(path, map_location='cpu', weights_only=False, pickle_module=_tolerant_pickle)
    if not isinstance(data, dict) or 'train' not in data or 'best_dice' not in data:
        raise ValueError('%s is not a PyLC loss log (no "train" / "best_dice" entries)' % path)
    return data


def meta_from_reference(ref_meta, base=None):
    """Copy the hot-path fields of a (bagged) config.Parameters into a Meta (config.py:259-269 semantics)."""
    meta = base if base is not None else Meta()
    meta.update(vars(ref_meta) if not isinstance(ref_meta, dict) else ref_meta)
    return meta


# ---- optimizer state <-> torch.optim.AdamW.state_dict() ------------------------------------------------------------------
def adamw_state_to_torch(optim):
    a = optim.arena
    state = {}
    for i, (p, off) in enumerate(zip(a.params, a.offsets)):
        if optim.steps == 0:
            continue
        state[i] = {'step': torch.tensor(float(optim.steps)),
                    'exp_avg': torch.as_strided(optim.m, p.shape, p.stride(), off).detach().clone().contiguous().cpu(),
                    'exp_avg_sq': torch.as_strided(optim.v, p.shape, p.stride(), off).detach().clone().contiguous().cpu()}
    group = {'lr': optim.lr, 'betas': tuple(optim.betas), 'eps': optim.eps, 'weight_decay': optim.wd, 'amsgrad': False,
             'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
             'decoupled_weight_decay': True, 'params': list(range(len(a.params)))}
    return {'state': state, 'param_groups': [group]}


def adamw_state_from_torch(optim, sd):
    a = optim.arena
    steps = 0
    for i, (p, off) in enumerate(zip(a.params, a.offsets)):
        st = sd['state'].get(i)
        if st is None:
            continue
        torch.as_strided(optim.m, p.shape, p.stride(), off).copy_(st['exp_avg'])
        torch.as_strided(optim.v, p.shape, p.stride(), off).copy_(st['exp_avg_sq'])
        steps = max(steps, int(float(st['step'])))
    optim.steps = steps
    if sd.get('param_groups'):
        optim.set_lr(sd['param_groups'][0]['lr'])


# ---- save / load of a pylc_amd Model in the reference's format ---------------------------------------------------------
def _reference_meta_object(meta):
    cls = type('Parameters', (), {'__module__': 'config'})
    obj = cls()
    obj.__dict__.update({k: v for k, v in vars(meta).items()})
    return obj, cls


def save(model, path, best=False):
    """checkpoint.py:51-67: checkpoint.pth (with epoch/iter) or the best-model file (without)."""
    meta_obj, cls = _reference_meta_object(model.meta)
    payload = {'model': {k: v.detach().cpu().contiguous() for k, v in model.net.state_dict().items()},
               'optim': adamw_state_to_torch(model.optim), 'meta': meta_obj}
    if not best:
        payload = {'epoch': model.epoch, 'iter': model.iter, **payload}
    # pickle resolves classes by module path at dump time: expose `config.Parameters` for the duration of the save
    had = sys.modules.get('config')
    fake = types.ModuleType('config')
    fake.Parameters = cls
    sys.modules['config'] = fake
    try:
        torch.save(payload, path)
    finally:
        if had is not None:
            sys.modules['config'] = had
        else:
            del sys.modules['config']


def load_into(model, path, resume=False):
    """Model.load (model.py:78-121) / Model.resume (model.py:222-236) for an already-built Model of matching architecture."""
    data = load_reference_file(path, map_location='cpu')
    model.net.load_state_dict(data['model'])
    if resume:
        model.epoch = data.get('epoch', 0)
        model.iter = data.get('iter', 0)
        if data.get('optim'):
            adamw_state_from_torch(model.optim, data['optim'])
        import os
        model.loss.load(os.path.join(os.path.dirname(os.path.abspath(path)), 'losses.pth'), resume=True)      # loss.py:253-268: the loss log resumes too
    return data
