"""Model wrapper of the HIP path: the counterpart of the reference's models/model.py:29-492 for the training /
validation / inference step (`Model.train(x, y)`, `.eval(x, y)`, `.test(x)`), i.e. the three hot-path entry
points called from train.py:118, train.py:149 and test.py:82.

Differences that are deliberate and MI355X-driven (results are the same):
  * input normalisation (model.py:416-445), the grayscale x3 stack (:310-311) and NCHW->NHWC packing are ONE
    kernel on the device instead of CPU tensor ops before the H2D copy;
  * the three per-step `.item()` host syncs (model.py:319) are replaced by a device-side loss log that is
    only read at `report` intervals;
  * clip_grad_norm_ + AdamW run over one flat parameter arena (pylc_amd/optim.py).
The dead random flip (model.py:296-298: `random.randint(0, 1)` is always 0) is not reproduced.
"""
import numpy as np
import torch

from . import ops
from .loss import MultiLoss
from .nets import DeepLab, UNet
from .optim import FlatArena, FlatAdamW, FlatSGD, StepLR
from .runtime import runtime


class Meta:
    """The hot-path subset of config.py:85-248 `Parameters`, as explicit attributes (no global singleton)."""

    def __init__(self, **kw):
        self.arch = 'deeplab'              # config.py:215
        self.backbone = 'resnet'           # config.py:217
        self.ch = 3
        self.n_classes = 9                 # schema_a
        self.class_codes = None
        self.class_labels = None
        self.px_mean = [132.47, 144.47, 149.45]     # config.py:171 (px_rgb_mean) -- DB metadata overrides
        self.px_std = [24.85, 22.04, 18.77]         # config.py:172
        self.px_grayscale_mean = 142.01             # config.py:173
        self.px_grayscale_std = 23.66               # config.py:174
        self.normalize_default = False
        self.weights = None                # class weights from the dataset profile (utils/profile.py:129-130)
        self.weighted = False              # config.py:200
        self.dice_weight = 0.5             # config.py:201-203
        self.ce_weight = 0.5
        self.focal_weight = 0.5
        self.lr = 1e-4                     # config.py:193
        self.weight_decay = 5e-5           # config.py:205
        self.momentum = 0.9
        self.gamma = 0.9                   # config.py:196
        self.optim_type = 'adam'           # always 'adam' in the reference (flags ignored)
        self.clip_norm = 0.5               # model.py:326
        self.dropout = 0.5                 # config.py:191
        self.up_mode = 'upsample'          # config.py:231
        self.pad_size = 94                 # (512 - 324) // 2, config.py:230; the valid-conv shrink is 188 px at any size
        self.report = 20                   # config.py:240
        self.pretrained = False
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError('unknown Meta field %r' % k)
            setattr(self, k, v)

    def update(self, other):
        """Copy only keys that already exist (config.py:259-269 semantics)."""
        src = other if isinstance(other, dict) else vars(other)
        for k, v in src.items():
            if hasattr(self, k):
                setattr(self, k, v)
        return self


class LossLog:
    """Device-side replacement of RunningLoss.intv (models/modules/loss.py:218-305): per-step (ce, dice, focal)
    triples stay on the GPU; `flush()` does the one D2H copy per report interval."""

    def __init__(self):
        self._pending = []
        self.intv = []
        self.train, self.valid, self.test, self.lr = [], [], [], []
        self.avg_dice, self.best_dice, self.is_best = 1.0, 1.0, False

    # ---- losses.pth (RunningLoss.save / .load, loss.py:253-268, 296-305): the file the reference writes next to its checkpoints at every
    #      log() and save(), and re-reads on resume ------------------------------------------------------------------------------------
    def save(self, path):
        """torch.save of {"train", "valid", "test", "best_dice", "lr"} -- lists of (iteration, ce, dice, focal) tuples, as loss.py:296-305."""
        self.flush()
        tmp = path + '.tmp'
        torch.save({'train': list(self.train), 'valid': list(self.valid), 'test': list(self.test), 'best_dice': float(self.best_dice),
                    'lr': list(self.lr)}, tmp)
        import os
        os.replace(tmp, path)

    def load(self, path, resume=True):
        """RunningLoss.load (loss.py:253-268): with resume, take up train / valid / test / best_dice from the file (the reference does not
        restore `lr`); without, an existing file is deleted and tracking restarts.  Returns True if a file was resumed from."""
        import os
        import pickle
        if not os.path.exists(path):
            return False
        if not resume:
            os.remove(path)
            return False
        # The reference's RunningLoss stores validation averages and best_dice as NUMPY scalars (Model.eval appends `.cpu().numpy()`
        # values, models/model.py:360-363, loss.py:284-304), which torch.load(weights_only=True) refuses: read the file through the
        # checkpoint module's allow-listed unpickler (numpy scalar / dtype / _reconstruct are on its list) and coerce to python floats.
        from .checkpoint import load_losses_file
        try:
            res = load_losses_file(path)
            rows = {k: [(int(row[0]),) + tuple(float(v) for v in row[1:]) for row in res.get(k, [])] for k in ('train', 'valid', 'test')}
            best = float(res['best_dice'])
        except (pickle.UnpicklingError, EOFError, ValueError, KeyError, TypeError, IndexError, RuntimeError) as e:      # (torch raises RuntimeError for a torn zip)
            # A damaged or foreign-format log.  The reference fails here (torch.load raises, loss.py:259); restarting the log silently would
            # reset best_dice to 1.0 and let the first validation overwrite the best-model file with a possibly worse model.  So: fail like the
            # reference unless the caller opted in (PYLC_RESTART_LOSS_LOG=1), and never swallow anything but read / format errors -- an
            # allow-list miss of the tolerant unpickler is an UnpicklingError whose text names the class, a format-compat regression signal.
            if os.environ.get('PYLC_RESTART_LOSS_LOG') != '1':
                raise RuntimeError('losses.pth at %s is unreadable (%s: %s); fix or remove it, or set PYLC_RESTART_LOSS_LOG=1 to restart the '
                                   'loss log (best-Dice tracking restarts with it)' % (path, type(e).__name__, e)) from e
            import warnings
            warnings.warn('losses.pth at %s is unreadable (%s: %s); the loss log restarts (PYLC_RESTART_LOSS_LOG=1)' % (path, type(e).__name__, e))
            return False
        self.train, self.valid, self.test = rows['train'], rows['valid'], rows['test']
        self.best_dice = best
        return True

    def push(self, triple):
        self._pending.append(triple)

    def flush(self):
        if self._pending:
            vals = torch.stack(self._pending).cpu().tolist()
            self.intv += [tuple(v) for v in vals]
            self._pending = []
        return self.intv

    def log(self, it, training):
        """RunningLoss.log (loss.py:270-293): interval average; a validation entry also updates best-Dice tracking."""
        self.flush()
        if self.intv:
            avg = tuple(np.mean(np.asarray(self.intv, np.float64), axis=0).tolist())
            if training:
                self.train.append((it,) + avg)
            else:
                self.valid.append((it,) + avg)
                self.avg_dice = avg[1]
                self.is_best = self.avg_dice < self.best_dice
                if self.is_best:
                    self.best_dice = self.avg_dice
        self.intv = []


class Model:
    def __init__(self, meta=None, device=None):
        self.meta = meta if meta is not None else Meta()
        self.device = torch.device(device if device is not None else 'cuda:0')
        self.net = self.crit = self.optim = self.sched = self.arena = None
        self.loss = LossLog()
        self.iter = 0
        self.epoch = 0
        self._bucketer = None
        self._ranges_checked = False        # eval / test: False until the first batch after a training step has re-measured the weight ranges

    def update_meta(self, params):
        self.meta.update(params)
        return self

    # ---- construction (model.py:123-220) -------------------------------------------------------------------
    def build(self):
        m = self.meta
        if m.arch == 'unet':
            self.net = UNet(in_channels=m.ch, n_classes=m.n_classes, up_mode=m.up_mode, dropout=m.dropout)
        elif m.arch == 'deeplab':
            self.net = DeepLab(backbone=m.backbone, n_classes=m.n_classes, in_channels=m.ch, pretrained=False)
        else:
            raise ValueError('Model {} not available.'.format(m.arch))
        self.net = self.net.to(self.device)
        self.crit = MultiLoss(
            loss_weights={'weighted': m.weighted, 'weights': m.weights, 'ce': m.ce_weight, 'dice': m.dice_weight,
                          'focal': m.focal_weight},
            schema={'n_classes': m.n_classes, 'class_codes': m.class_codes, 'class_labels': m.class_labels}).to(self.device)
        self.init_optim()
        return self

    def init_optim(self):
        """model.py:238-280.  Call again after load_state_dict() into self.net is NOT needed: the arena aliases
        the parameters, so loading a state dict writes straight into it."""
        m = self.meta
        self.arena = FlatArena(self.net)
        if m.optim_type == 'adam':
            self.optim = FlatAdamW(self.arena, lr=m.lr, weight_decay=m.weight_decay, clip=m.clip_norm)
        elif m.optim_type == 'sgd':
            self.optim = FlatSGD(self.arena, lr=m.lr, momentum=m.momentum, clip=m.clip_norm)
        else:
            raise ValueError('Optimizer is not defined.')
        self.sched = StepLR(self.optim, m.gamma)

    # ---- input handling (model.py:301-311, 416-445) ------------------------------------------------------------
    def _stats(self, default=False):
        m = self.meta
        if m.ch == 1:
            if default:        # model.py:428-430: (img - px_grayscale_mean) / px_grayscale_std, WITHOUT the division by 255 (reference quirk)
                return [float(m.px_grayscale_mean)] * 3, [float(m.px_grayscale_std)] * 3, 1.0
            mean = float(np.mean(np.asarray(m.px_mean, np.float32)))
            std = float(np.mean(np.asarray(m.px_std, np.float32)))
            return [mean] * 3, [std] * 3, 255.0
        if default:
            return [132.47, 144.47, 149.45], [24.85, 22.04, 18.77], 255.0
        return list(m.px_mean), list(m.px_std), 255.0

    def pack_input(self, x, default=False):
        """raw [B,ch,H,W] 0..255 (host or device) -> normalised NHWC4 device tensor."""
        if x.dim() != 4 or x.shape[1] != self.meta.ch:
            raise ValueError('expected [B,%d,H,W] tiles, got %s' % (self.meta.ch, tuple(x.shape)))
        if x.dtype != torch.uint8:
            x = x.to(dtype=torch.float32)
        x = x.to(self.device, non_blocking=True)
        mean, std, denom = self._stats(default)
        return ops.image_pack(x, mean, std, denom)

    def crop_target(self, y):
        if self.meta.arch == 'unet':
            p = self.meta.pad_size
            y = y[:, p:y.shape[1] - p, p:y.shape[2] - p]          # model.py:306-307 at any tile size
        return y.contiguous()

    # ---- the three hot-path entry points ---------------------------------------------------------------------
    def train(self, x, y):
        """One optimisation step (model.py:282-336)."""
        self.net.train()       # (parameter ranges / prepared filters are current: refreshed by every optimiser step and state load)
        self._ranges_checked = False
        x4 = self.pack_input(x)
        y = self.crop_target(y.to(self.device, non_blocking=True).long())
        y_hat = self.net(x4)
        loss = self.crit(y_hat, y)
        self.loss.push(torch.stack((self.crit.ce, self.crit.dsc, self.crit.fl)))
        self.optim.zero_grad()
        if runtime.sync_group is not None:
            from .parallel import GradBucketer, assert_equal_shards
            if getattr(self, '_dp_batch', None) is None:
                # n_global = n_local * world everywhere (SyncBN, loss head).  Checked by a collective ONCE, at every rank's first step (the
                # same step on all ranks, so the collective sequences stay aligned); afterwards every step's loss exchange carries the
                # tile counts (ops.MultiLossFn) and the reduced pair is compared at the report interval (self.log) -- a shorter final
                # batch on SOME ranks then raises there instead of desynchronising the ranks with a collective only they enter.
                assert_equal_shards(x.shape[0], runtime.sync_group)
                self._dp_batch = x.shape[0]
            if self._bucketer is None:
                self._bucketer = GradBucketer(self.arena, runtime.grad_group)
            self._bucketer.reset()
            runtime.grad_ready = self._bucketer.ready
        ops.reset_slab_sums()                     # (sums noted by a backward pass that an exception cut short)
        loss.backward()
        ops.sync_side_streams()                   # side-stream wgrads (precision mode 3) joined, pending split-K slab sums launched
        self.arena.clear_undelivered()            # a parameter without a gradient this step must not keep last step's (before the exchange)
        if runtime.sync_group is not None:
            runtime.grad_ready = None
            self._bucketer.finish()               # gradient SUM all-reduce, overlapped with the backward above
        self.optim.step()
        if self.iter % self.meta.report == 0:
            self.log()
        self.loss.lr += [(self.iter, self.optim.lr)]
        self.iter += 1
        return loss.detach()

    def eval(self, x, y):
        """Validation step (model.py:338-365): eval-mode forward, the three losses, returns [y_hat]."""
        self.net.eval()
        self._refresh_for_inference()
        x4 = self.pack_input(x)
        y = self.crop_target(y.to(self.device, non_blocking=True).long())
        with torch.no_grad():
            y_hat = self.net(x4)
            self.loss.push(self.crit.all_losses(y_hat, y)[1:4])
        return [y_hat]

    def test(self, x):
        """Inference forward (model.py:367-382)."""
        self._refresh_for_inference()
        x4 = self.pack_input(x, default=self.meta.normalize_default)
        with torch.no_grad():
            return [self.net(x4)]

    def _refresh_for_inference(self):
        """Weight ranges / prepared filter planes for an eval-mode forward: re-measured unconditionally on the first batch after a training
        step or after construction (a `p.data` write -- EMA, weight surgery -- does not bump the version counters refresh_if_changed
        compares), then only when a counter moved or `arena.invalidate()` was called."""
        self.arena.refresh_if_changed(force=not self._ranges_checked)
        self._ranges_checked = True

    def log(self):
        if runtime.sync_group is not None:
            ops.check_equal_shards()
        self.loss.log(self.iter, self.net.training)

    def get_lr(self):
        return self.optim.lr

    def model_id(self):
        """gen_id (model.py:482-492): pylc_<arch>_ch<channels>_<schema>."""
        return 'pylc_%s_ch%d_schema_%s' % (self.meta.arch, self.meta.ch, 'a' if self.meta.n_classes == 9 else 'b')

    def save(self, save_dir):
        """Model.save (model.py:389-392 -> checkpoint.py:51-67): checkpoint.pth always, <id>.pth on a new best validation Dice.
        Data parallel: a COLLECTIVE over runtime.sync_group -- every rank of the job must call it (rank 0 writes, the others wait at the
        barrier below); a rank-0-only call would deadlock there."""
        import os
        from . import checkpoint, parallel
        # data parallel: replicas are identical, rank 0 writes (the reference is single-process); every file is written to a
        # temporary name and renamed into place, so a reader or a crash never sees a torn zip; the barrier keeps the other
        # ranks from racing ahead into a resume / read of the file
        rank0 = runtime.sync_group is None or parallel.rank() == 0
        if rank0:
            d = os.path.join(save_dir, self.model_id())
            os.makedirs(d, exist_ok=True)
            targets = [(os.path.join(d, 'checkpoint.pth'), False)]
            if self.loss.is_best:
                targets.append((os.path.join(d, self.model_id() + '.pth'), True))
            for path, best in targets:
                tmp = path + '.tmp.%d' % os.getpid()
                checkpoint.save(self, tmp, best=best)
                os.replace(tmp, path)
            self.loss.save(os.path.join(d, 'losses.pth'))          # model.py:389-392: Model.save also writes the loss log
        if runtime.sync_group is not None:
            parallel.barrier(runtime.sync_group)
