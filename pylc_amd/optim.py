"""Flat-arena optimiser: every parameter (and its gradient, and both Adam moments) lives in ONE contiguous
fp32 buffer, so global-norm clipping + AdamW is three kernel launches and the data-parallel gradient
exchange is a handful of large RCCL all-reduces instead of 341 small ones.

Replaces torch.optim.AdamW / SGD + clip_grad_norm_ as used at models/model.py:238-254,322-328 and
StepLR at models/model.py:256-263 / train.py:91."""
import os
import weakref

import torch

from . import lib as L
from .lib import lib, check, ptr, stream
from .runtime import runtime


def _dense_numel(p):
    return p.numel()


class FlatArena:
    """Re-homes the parameters of `module` into one buffer (keeping every tensor's shape and strides, e.g. the
    KRSC memory of conv weights) and gives each a same-layout gradient view registered as `p._pylc_grad`,
    which the backward kernels write directly (pylc_amd/ops/_core.py:_deliver_grad)."""

    ALIGN = 4       # floats (16 B): every tensor starts on a vector boundary

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError('module has no trainable parameters')
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = total
        self.params = params
        self.offsets = offs
        self.p = torch.zeros(total, device=dev)
        self.g = torch.zeros(total, device=dev)
        for p, o in zip(params, offs):
            dst = torch.as_strided(self.p, p.shape, p.stride(), o)
            dst.copy_(p.data)
            p.data = dst
            p._pylc_grad = torch.as_strided(self.g, p.shape, p.stride(), o)
            p.grad = p._pylc_grad
            p._pylc_arena = weakref.ref(self)
        # per-parameter max magnitude (float bits), the filter range of the f16x3 conv arithmetic: one launch refreshes all
        self.amax = torch.zeros(len(params), dtype=torch.int32, device=dev)
        self._segments = torch.tensor(offs + [total], dtype=torch.int64, device=dev)
        for i, p in enumerate(params):
            p._pylc_wamax = self.amax[i:i + 1]
        # prepared conv filters for the f16x3 arithmetic: two fp16 planes per filter in the forward (KRSC) and the dgrad
        # (CRSK) layout, rebuilt from the fp32 master weights by ONE launch whenever the ranges are refreshed
        entries, halves, tiles = [], 0, 0
        self._planes_il = 0              # 1: the last prepare launch wrote the filter planes chunk-interleaved where eligible (refresh_ranges)
        self._prep_entries = []
        for i, (p, o) in enumerate(zip(params, offs)):
            if p.dim() == 4 and p.shape[1] % 4 == 0 and p.shape[1] > 1 and p.permute(0, 2, 3, 1).is_contiguous():
                k, c, r, s_ = p.shape
                kp = (k + 3) & ~3
                e = L.WPrepEntry(o, halves, halves + 2 * k * r * s_ * c, tiles, k, r * s_, c, i)
                tiles += r * s_ * ((k + 31) // 32) * ((c + 31) // 32)
                halves += 2 * k * r * s_ * c + 2 * c * r * s_ * kp
                halves = (halves + 7) & ~7          # 16-byte aligned plane sets
                entries.append((p, e))
        self.planes = torch.zeros(max(halves, 8), dtype=torch.float16, device=dev)
        self._n_prep, self._prep_tiles = len(entries), tiles
        if entries and self.p.is_cuda:
            import ctypes as C
            arr = (L.WPrepEntry * len(entries))(*[e for _, e in entries])
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            self._prep_table = raw.to(dev)
            self._prep_entries = entries
            self._publish_planes(0)
        self._index = {id(p): i for i, p in enumerate(params)}
        self._delivered = set()          # parameters whose gradient a backward kernel wrote since the last zero_grad()
        self._stale = False              # invalidate(): values changed through a path the version counters do not see
        self.generation = 0              # bumped whenever parameter VALUES may have changed (refresh_ranges): derived buffers key on it
        me = weakref.ref(self)           # no module -> arena strong reference: a dropped model frees its memory by refcount
        module.register_load_state_dict_post_hook(lambda *_: me() is not None and me().refresh_ranges())
        self.refresh_ranges()

    def _publish_planes(self, il):
        """Give every prepared filter its planes object: an IMMUTABLE (forward planes, dgrad planes, format) tuple whose format is what the last
        prepare launch WROTE -- bit 0 / bit 1: the forward / dgrad planes are chunk-interleaved (channel count % 32 == 0 and il) -- so that the
        layout travels with the planes and cannot disagree with them (the format used to be a cell shared by all parameters and
        flipped at refresh time).  Replaced whenever a prepare launch changes the layout (a precision-mode switch)."""
        self._planes_il = il
        for p, e in self._prep_entries:
            k, c, r, s_ = p.shape
            fmt = ((1 if c % 32 == 0 else 0) | (2 if ((k + 3) & ~3) % 32 == 0 else 0)) if il else 0
            p._pylc_planes = (self.planes[e.fwd_offset:e.fwd_offset + 2 * k * r * s_ * c],
                              self.planes[e.t_offset:e.t_offset + 2 * c * r * s_ * ((k + 3) & ~3)], fmt)

    def refresh_if_changed(self, force=False):
        """refresh_ranges() unless nothing can have moved the parameters since the last refresh: everything in this package that writes them
        (the flat optimisers, load_state_dict through the hook above, parallel.broadcast_parameters) refreshes by itself, and an in-place
        torch operation ON THE PARAMETER (`p.mul_()`, `p.copy_()` under no_grad) bumps its version counter, which is what is compared here.
        NOT detected: writes through `p.data` (`p.data.copy_()`, EMA / weight surgery idioms) and raw-pointer writers -- `.data` is a
        fresh alias whose writes do not bump `p._version`.  After such a write call `invalidate()` (or `refresh_ranges()`): stale ranges
        mean stale fp16 scales and prepared filter planes, and weights that outgrew the old range would overflow the f16 split.
        Model.eval / Model.test call this per batch -- an inference loop then neither re-measures 59 M weights nor invalidates what was
        derived from them (generation) -- and FORCE it on the first batch after a training step or a mode change, so a `.data` write made
        between training and validation is picked up without the caller knowing about this cache."""
        if force or self._stale or sum(p._version for p in self.params) != self._version_sum:
            self.refresh_ranges()

    def invalidate(self):
        """Declare the parameter values changed behind this arena's back (a `.data` write, a raw-pointer writer): the next
        refresh_if_changed() -- i.e. the next Model.eval / Model.test batch -- re-measures the ranges and rebuilds the filter planes."""
        self._stale = True

    def refresh_ranges(self, ranges_current=False):
        """Recompute every parameter's max magnitude and rebuild the prepared filter planes from it.  Call after anything
        that changes parameter values (the optimiser steps here do; Model refreshes at the start of each step as well).
        ranges_current: self.amax was just written by the kernel that changed the parameters (pylc_adamw_step_ranges)."""
        self.generation += 1
        self._stale = False
        self._version_sum = sum(p._version for p in self.params)
        if self.p.is_cuda:
            L.init()
            if not ranges_current:
                check(lib.pylc_amax_segments(ptr(self.p), ptr(self._segments), len(self.params), ptr(self.amax), stream()))
            if self._n_prep:
                # f16x3 (precision mode 2): both planes of a K-step chunk in one cache line (include/pylc_hip.h PylcConvDesc.w_planes_fmt); the
                # one-plane kernels of mode 3 read plane 0 only and want it contiguous.  The kernels read either layout (the flag travels in
                # the conv descriptor), so a mode switch without a refresh stays correct.  PYLC_NO_FILTER_INTERLEAVE=1: A/B knob.
                il = 1 if (lib.pylc_get_conv_precision() == 2 and not os.environ.get('PYLC_NO_FILTER_INTERLEAVE')) else 0
                check(lib.pylc_weight_prepare(ptr(self.p), ptr(self._prep_table), self._n_prep, self._prep_tiles, ptr(self.amax),
                                              ptr(self.planes), il, stream()))
                if il != self._planes_il:
                    self._publish_planes(il)

    def zero_grad(self):
        self.g.zero_()

    def mark_delivered(self, p):
        self._delivered.add(id(p))

    def begin_step(self):
        """Forget which gradients were delivered (called by the optimiser's zero_grad(), i.e. before each backward)."""
        self._delivered.clear()

    def clear_undelivered(self):
        """Backward kernels OVERWRITE arena gradients, so a parameter that received no gradient this step (a frozen parameter, an unused
        branch, an early-returning backward) would otherwise be stepped with last step's values.  Zero those slices -- so that the
        gradient norm and the data-parallel exchange see zeros -- and return their (offset, numel) segments: the optimisers leave such a
        parameter and its moments untouched, as torch.optim skips a parameter whose .grad is None (reported once)."""
        if not self._delivered or len(self._delivered) == len(self.params):
            return []         # all delivered, or delivery not tracked (gradients written by something else than pylc_amd.ops)
        missing = [i for i, p in enumerate(self.params) if id(p) not in self._delivered]
        for i in missing:
            p, o = self.params[i], self.offsets[i]
            self.g[o:o + p.numel()].zero_()
        if not getattr(self, '_warned_undelivered', False):
            import warnings
            warnings.warn('%d parameter tensor(s) received no gradient this step; they are skipped by the optimiser step' % len(missing))
            self._warned_undelivered = True
        return [(self.offsets[i], self.params[i].numel()) for i in missing]


class _FlatOptimizer:
    def __init__(self, arena, lr, clip):
        self.arena = arena
        self.lr = float(lr)
        self.clip = clip
        self.steps = 0
        dev = arena.p.device
        self.norm = torch.zeros(2, device=dev)           # [grad L2 norm, clip coefficient] of the last step
        self._ws = torch.empty(lib.pylc_sqnorm_workspace_floats(arena.numel), device=dev)
        self.param_groups = [{'lr': self.lr}]             # models/model.py:394-397 get_lr() reads this

    def zero_grad(self, set_to_none=False):
        """Gradients are overwritten by the backward kernels every step; nothing to clear -- only the record of which
        parameters have been delivered (step() zeroes the slices of those that were not, see FlatArena.clear_undelivered)."""
        self.arena.begin_step()

    def _clip(self):
        a = self.arena
        self._skipped = a.clear_undelivered()
        if self.clip is None:
            return None
        check(lib.pylc_grad_norm_clip(ptr(a.g), a.numel, float(self.clip), ptr(self.norm), ptr(self._ws), stream()))
        return self.norm

    def _stash(self, *buffers):
        """Copies of the segments of parameters WITHOUT a gradient this step (rare: frozen parameters, unused branches).  The flat
        kernels step the whole arena; _restore() puts these back, so such a parameter sees no weight decay and its moments no decay --
        torch.optim's behaviour for a parameter whose .grad is None."""
        return [(buf, o, buf[o:o + n].clone()) for o, n in self._skipped for buf in buffers]

    @staticmethod
    def _restore(stash):
        for buf, o, saved in stash:
            buf[o:o + saved.numel()].copy_(saved)

    def set_lr(self, lr):
        self.lr = float(lr)
        self.param_groups[0]['lr'] = self.lr


class FlatAdamW(_FlatOptimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt) preceded by
    torch.nn.utils.clip_grad_norm_(params, clip)."""

    def __init__(self, arena, lr=1e-4, weight_decay=5e-5, betas=(0.9, 0.999), eps=1e-8, clip=0.5):
        super().__init__(arena, lr, clip)
        self.wd, self.betas, self.eps = float(weight_decay), betas, float(eps)
        self.m = torch.zeros_like(arena.p)
        self.v = torch.zeros_like(arena.p)

    def step(self):
        L.init()
        a = self.arena
        coef = self._clip()
        self.steps += 1
        stash = self._stash(a.p, self.m, self.v)
        fused = not stash and runtime.adamw_ranges      # (restored segments would not match the ranges taken in the kernel)
        if fused:
            # the parameter ranges of the next step come out of the update pass itself
            check(lib.pylc_adamw_step_ranges(ptr(a.p), ptr(a.g), ptr(self.m), ptr(self.v), a.numel, ptr(coef), self.lr, self.betas[0],
                                             self.betas[1], self.eps, self.wd, self.steps, ptr(a._segments), len(a.params), ptr(a.amax), stream()))
        else:
            check(lib.pylc_adamw_step(ptr(a.p), ptr(a.g), ptr(self.m), ptr(self.v), a.numel, ptr(coef), self.lr,
                                      self.betas[0], self.betas[1], self.eps, self.wd, self.steps, stream()))
        self._restore(stash)
        a.refresh_ranges(ranges_current=fused)

    def state_dict(self):
        return {'kind': 'flat_adamw', 'steps': self.steps, 'lr': self.lr, 'm': self.m, 'v': self.v}

    def load_state_dict(self, sd):
        self.steps = int(sd['steps'])
        self.set_lr(sd['lr'])
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])


class FlatSGD(_FlatOptimizer):
    def __init__(self, arena, lr=1e-4, momentum=0.9, clip=0.5):
        super().__init__(arena, lr, clip)
        self.momentum = float(momentum)
        self.buf = torch.zeros_like(arena.p)

    def step(self):
        L.init()
        a = self.arena
        coef = self._clip()
        self.steps += 1
        stash = self._stash(a.p, self.buf)
        check(lib.pylc_sgd_step(ptr(a.p), ptr(a.g), ptr(self.buf), a.numel, ptr(coef), self.lr, self.momentum, self.steps, stream()))
        self._restore(stash)
        a.refresh_ranges()

    def state_dict(self):
        return {'kind': 'flat_sgd', 'steps': self.steps, 'lr': self.lr, 'buf': self.buf}

    def load_state_dict(self, sd):
        self.steps = int(sd['steps'])
        self.set_lr(sd['lr'])
        self.buf.copy_(sd['buf'])


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size=1, gamma): lr <- lr * gamma once per epoch (train.py:91)."""

    def __init__(self, optim, gamma=0.9):
        self.optim, self.gamma = optim, gamma

    def step(self):
        self.optim.set_lr(self.optim.lr * self.gamma)
