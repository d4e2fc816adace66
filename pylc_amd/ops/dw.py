"""Depthwise 3x3 convolution (xception.py:25-39 SeparableConv2d's first half, fixed_padding folded in)."""
from . import _core
from ._core import *      # noqa: F401,F403  (layout / planes / range / stream helpers, lib bindings, torch)
from .bn import materialize_deferred

def _dw_desc(x, stride, dil, xp, yp):
    b, c, h, w = x.shape
    d = DwDesc()
    d.B, d.H, d.W, d.C, d.stride, d.dil = b, h, w, c, stride, dil
    d.OH, d.OW = (h - 1) // stride + 1, (w - 1) // stride + 1
    d.x_pitch, d.y_pitch = xp, yp
    return d


class _wgrad_stream:
    """Context of a depthwise filter-gradient launch: on the wgrad SIDE stream, like the dense convs' (ops.conv Conv2dFn.backward), when the
    gradient goes straight into the optimiser's arena (nobody on the main stream reads it before sync_side_streams()) -- the filter gradient is
    off the critical chain, and in the Xception step (BASELINE configs[4]) the main queue is the busy one: 56 of 58 ms against 14 ms on the
    side queue (profiles/r04_c5_trace_streams.txt), 3.2 ms of it depthwise wgrad + combine.  The side stream first waits for the main
    stream's position (dy and, with a deferred BatchNorm, its coefficients were produced there); the tensors the kernels read are kept alive
    until the streams are joined.  PYLC_RUNTIME=dw_wgrad_side=0 keeps it on the compute stream (A/B knob)."""

    def __init__(self, device, tgt, *reads):
        self.side = None
        if tgt is not None and _runtime.side_stream_on() and _runtime.dw_wgrad_side:
            self.side = _side_stream(device)
            ev = torch.cuda.Event()
            ev.record()
            self.side.wait_event(ev)
            _keep_for_side(device, *reads)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.side is not None:
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.side is not None:
            return self.ctx.__exit__(*a)
        return False


class DwConv3x3Fn(torch.autograd.Function):
    """Depthwise 3x3 (xception.py:29-31 with fixed_padding folded in).  Always returns a tuple (y, statistics partials or None, range bound
    or None): precision mode 3 with half activations runs the stride-1 / dilation-1 shapes on ONE-PLANE fp16 tensors (pylc_dwconv3x3_*_h:
    x and y at 2 bytes per element), y then being a planes tensor scaled with the returned bound."""

    @staticmethod
    def forward(ctx, x, w, stride, dil, res_link=None, want_stats=False, bn_src=None):
        L.init()
        ctx.set_materialize_grads(False)
        if tuple(w.shape) != (x.shape[1], 1, 3, 3) or not w.is_contiguous():
            raise L.PylcError('depthwise weight must be contiguous [C,1,3,3]')
        ctx.res_link = res_link if (res_link is not None and ctx.needs_input_grad[0]) else None
        if ctx.res_link is not None:
            res_link.pending += 1           # one more backward node that adds its part of x's gradient into the shared buffer
        ctx.bn_src = bn_src if ctx.needs_input_grad[0] else None
        b, c, h, wd = x.shape
        x_bound = None
        # the input may be a BatchNorm output whose apply pass was deferred to this conv (bn_act(defer=True)): x then aliases the BatchNorm's
        # INPUT and the kernels apply scale / shift / ReLU to their LDS patch (pylc_dwconv3x3_*_h_bn)
        defer = getattr(x, '_pylc_defer', None)
        if defer is not None and defer[4] != x._version:
            raise L.PylcError('depthwise conv: the deferred BatchNorm output was modified in place')
        if defer is not None:
            dh = _dw_desc(x, stride, dil, c, c)
            if not (nplanes() == 1 and any(ctx.needs_input_grad) and lib.pylc_dwconv3x3_bn_ok(C.byref(dh))):
                x, defer = materialize_deferred(x), None
        if defer is not None:
            x_bound = defer[3]
        elif is_planes(x) and nplanes() == 1 and half_acts() and _runtime.half_dw and any(ctx.needs_input_grad):
            dh = _dw_desc(x, stride, dil, c, c)
            if lib.pylc_dwconv3x3_half_ok(C.byref(dh)):
                x_bound = planes_amax(x)
        if x_bound is None:
            x = as_nhwc(x)
        d = _dw_desc(x, stride, dil, c if x_bound is not None else pitch_of(x), c)
        y = empty_nhwc(b, c, d.OH, d.OW, x.device)
        rows = (lib.pylc_dwconv3x3_fwd_h_stats_rows(C.byref(d)) if x_bound is not None else lib.pylc_dwconv3x3_fwd_stats_rows(C.byref(d))) if want_stats else 0
        sums = y_bound = None
        if rows > 0:        # the statistics of the BatchNorm that follows come out of this pass (stride-1 / dilation-1 shapes)
            sums = torch.empty((rows, 2 * c), device=x.device, dtype=torch.float32)
        if defer is not None:
            coef, bn_relu, yin_bound = defer[0], defer[1], defer[2]
            y_bound = amax_slot(x.device)
            check(lib.pylc_dwconv3x3_fwd_h_bn(C.byref(d), ptr(x), ptr(yin_bound), ptr(coef[2 * c:3 * c]), ptr(coef[3 * c:]), int(bn_relu), ptr(x_bound),
                                              ptr(w), ptr(weight_amax(w)), ptr(y), ptr(y_bound), ptr(sums), stream()))
        elif x_bound is not None:
            y_bound = amax_slot(x.device)
            check(lib.pylc_dwconv3x3_fwd_h(C.byref(d), ptr(x), ptr(x_bound), ptr(w), ptr(weight_amax(w)), ptr(y), ptr(y_bound), ptr(sums), stream()))
        elif rows > 0:
            check(lib.pylc_dwconv3x3_fwd_stats(C.byref(d), ptr(x), ptr(w), ptr(y), ptr(sums), stream()))
        else:
            check(lib.pylc_dwconv3x3_fwd(C.byref(d), ptr(x), ptr(w), ptr(y), stream()))
        ctx.save_for_backward(x, x_bound, *((defer[0], defer[2]) if defer is not None else (None, None)))
        ctx.bn_relu = defer[1] if defer is not None else None
        ctx.x_half = x_bound is not None
        ctx.w_param, ctx.geom = w, (stride, dil)
        aux = tuple(t for t in (sums, y_bound) if t is not None)
        if aux:
            ctx.mark_non_differentiable(*aux)
        return y, sums, y_bound

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            return (None,) * 7
        x, x_bound, bn_coef, yin_bound = ctx.saved_tensors
        w = ctx.w_param
        stride, dil = ctx.geom
        st = stream()
        link = ctx.res_link
        dx = dw = None
        half = ctx.x_half and is_planes(dy) and nplanes() == 1
        if bn_coef is not None and not half:       # (deferred BatchNorm input and an fp32 gradient: form x after all)
            x._pylc_defer = (bn_coef, ctx.bn_relu, yin_bound, x_bound, x._version)
            x, bn_coef = materialize_deferred(x), None
        if ctx.x_half and not half:          # the gradient arrived in fp32: run the fp32 kernels on an fp32 copy of x
            x = from_planes(mark_planes(x, x_bound))
        if half:
            c = x.shape[1]
            d = _dw_desc(x, stride, dil, c, c)
            dy_bound = planes_amax(dy)
            wa = weight_amax(w)
            if ctx.needs_input_grad[0]:
                # a ReLU'd residual gradient parked on the link as (dout, mask) is added by the dgrad kernel itself while it writes dx
                masked = None
                if (link is not None and link.masked is not None and link.buf is None and lib.pylc_dwconv3x3_dgrad_h_add_ok(C.byref(d))
                        and tuple(link.masked[0].shape) == tuple(x.shape) and pitch_of(link.masked[0]) == c and not is_planes(link.masked[0])):
                    masked, link.masked = link.masked, None
                sink = _link_sink(link) if masked is None else None
                bn_node = ctx.bn_src
                # dx as one fp16 plane only when it is the whole gradient of a BatchNorm output read by this conv alone; the gradient of a
                # block input (gradient link) stays fp32 and accumulates in fp32
                dx_half = link is None and bn_node is not None and getattr(bn_node, 'sole', False)
                if sink is not None and (tuple(sink.shape) != tuple(x.shape) or pitch_of(sink) != c or is_planes(sink)):
                    raise L.PylcError('depthwise dgrad: the parked gradient does not have the shape / format of the input')
                dx = sink if sink is not None else empty_nhwc(*x.shape, device=x.device)
                dx_bound = amax_slot(x.device) if dx_half else None
                if masked is not None:
                    check(lib.pylc_dwconv3x3_dgrad_h_add(C.byref(d), ptr(dy), ptr(dy_bound), ptr(w), ptr(wa), ptr(dx), ptr(masked[0]), ptr(masked[1]), st))
                else:
                    check(lib.pylc_dwconv3x3_dgrad_h(C.byref(d), ptr(dy), ptr(dy_bound), ptr(w), ptr(wa), ptr(dx), ptr(dx_bound),
                                                     1 if sink is not None else 0, None, st))
                if dx_half:
                    mark_planes(dx, dx_bound)
                if link is not None:
                    link.pending -= 1
                    if link.pending > 0:
                        link.buf, dx = dx, None
                    else:
                        link.buf = None
            if ctx.needs_input_grad[1]:
                tgt = _grad_target(w)
                with _wgrad_stream(x.device, tgt, x, x_bound, dy, dy_bound, bn_coef, yin_bound, w, wa):
                    nbytes = lib.pylc_dwconv3x3_wgrad_workspace(C.byref(d))
                    ws = _ws(nbytes, x.device)
                    dw = tgt if tgt is not None else torch.empty_like(w)
                    if bn_coef is not None:
                        check(lib.pylc_dwconv3x3_wgrad_h_bn(C.byref(d), ptr(x), ptr(yin_bound), ptr(bn_coef[2 * c:3 * c]), ptr(bn_coef[3 * c:]),
                                                            int(ctx.bn_relu), ptr(x_bound), ptr(dy), ptr(dy_bound), ptr(dw), ptr(ws), nbytes, stream()))
                    else:
                        check(lib.pylc_dwconv3x3_wgrad_h(C.byref(d), ptr(x), ptr(x_bound), ptr(dy), ptr(dy_bound), ptr(dw), ptr(ws), nbytes, stream()))
                dw = _deliver_grad(w, dw)
            return dx, dw, None, None, None, None, None
        dy = as_nhwc(dy)
        d = _dw_desc(x, stride, dil, pitch_of(x), pitch_of(dy))
        if ctx.needs_input_grad[0]:
            sink = _link_sink(link)
            if sink is not None and (tuple(sink.shape) != tuple(x.shape) or pitch_of(sink) != x.shape[1]):
                raise L.PylcError('depthwise dgrad: the parked gradient does not have the shape of the input')
            dx = sink if sink is not None else empty_nhwc(*x.shape, device=x.device)
            d.x_pitch = x.shape[1]
            check(lib.pylc_dwconv3x3_dgrad_acc(C.byref(d), ptr(dy), ptr(w), ptr(dx), 1 if sink is not None else 0, st))
            d.x_pitch = pitch_of(x)
            if link is not None:
                link.pending -= 1
                if link.pending > 0:        # other consumers of x follow: they accumulate into the same buffer
                    link.buf, dx = dx, None
                else:
                    link.buf = None
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(w)
            with _wgrad_stream(x.device, tgt, x, dy, w):
                nbytes = lib.pylc_dwconv3x3_wgrad_workspace(C.byref(d))
                ws = _ws(nbytes, x.device)
                dw = tgt if tgt is not None else torch.empty_like(w)
                check(lib.pylc_dwconv3x3_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(ws), nbytes, stream()))
            dw = _deliver_grad(w, dw)
        return dx, dw, None, None, None, None, None


def dwconv3x3_eval_half(x, w, stride=1, dil=1):
    """Inference, precision mode 3: depthwise 3x3 on a ONE-PLANE fp16 tensor (pylc_dwconv3x3_fwd_h_eval) -> one-plane tensor scaled with
    9 max|w| x (true max|x|), which is also its tag (the true maximum of a depthwise output is not taken: the bound is within 2^3 of it).
    Returns None when the geometry has no half kernel or x is not / cannot be held as a plane."""
    L.init()
    if not (half_dw() and nplanes() == 1):
        return None
    b, c, h, wd = x.shape
    d = _dw_desc(x, stride, dil, c, c)
    if not (lib.pylc_dwconv3x3_half_ok(C.byref(d)) and planes_ok(c, b * h * wd) and planes_ok(c, b * d.OH * d.OW)):
        return None
    if not is_planes(x):
        x = as_nhwc(x)
        true = amax_of(x)
        x = to_planes(x, true)
    else:
        true = amax_of(x)
    y = empty_nhwc(b, c, d.OH, d.OW, x.device)
    bound = amax_slot(x.device)
    check(lib.pylc_dwconv3x3_fwd_h_eval(C.byref(d), ptr(x), ptr(planes_amax(x)), ptr(true), ptr(w), ptr(weight_amax(w)), ptr(y), ptr(bound), stream()))
    return mark_planes(y, bound)


def dwconv3x3(x, w, stride=1, dil=1, res_link=None, want_stats=False):
    """want_stats: the output feeds a training-mode BatchNorm -- attach the statistics partials the forward pass can emit (as ops.conv2d)."""
    src = x.grad_fn if (torch.is_grad_enabled() and x.requires_grad) else None
    bn_src = src if hasattr(src, 'bn_emit_ok') else None
    y, sums, y_bound = DwConv3x3Fn.apply(x, w, stride, dil, res_link, bool(want_stats and torch.is_grad_enabled()), bn_src)
    if y_bound is not None:
        mark_planes(y, y_bound)          # one fp16 plane: the BatchNorm that follows reads it as such
        y._pylc_dy_pl = True             # ... and may hand its dy back the same way
    if sums is not None:
        y._pylc_sums = sums
    return y


__all__ = [n for n in dir() if not n.startswith('__')]      # everything, underscore helpers included: the package re-exports it (pylc_amd/ops/__init__.py)
