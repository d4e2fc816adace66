"""BatchNorm (+ residual + ReLU + dropout) operators, SyncBN message driver (models/sync_batchnorm/batchnorm.py:48-125 is the math)."""
from . import _core
from ._core import *      # noqa: F401,F403  (layout / planes / range / stream helpers, lib bindings, torch)


def _bn_extra(**kw):
    ex = L.BnExtra()
    ex.nplanes = nplanes()
    for k, v in kw.items():
        setattr(ex, k, v)
    return ex


class BnActFn(torch.autograd.Function):
    """out = [dropout](act(BN(y) (+ residual))).  Training: batch statistics (all-reduced over `group` when given --
    the SyncBN exchange of models/sync_batchnorm/batchnorm.py:48-125 as one RCCL all-reduce of
    [sum, sumsq, count]); eval: running statistics.

    out_planes: write the output as fp16 planes (ops.is_planes) for a conv that copies its operand tiles (conv_pl.hip); the scale
    comes from a range BOUND that the statistics give before the apply pass runs (pylc_bn_finalize*_ex).  The backward hands dy back as
    planes when the conv that produced y asked for it (y._pylc_dy_pl).  drop = (p, seed): the nn.Dropout that follows the activation
    in the reference (aspp.py:86, decoder.py:33,37), fused into both passes."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, residual, relu, training, eps, momentum, group, clamp_eps, pre_sums=None,
                want_amax=False, res_link=None, out_planes=False, drop=None, dy_planes=False, into=None, sole=False, defer=False):
        return _drive_collectives([BnActFn._forward(ctx, y, gamma, beta, running_mean, running_var, residual, relu, training, eps, momentum, group,
                                                    clamp_eps, pre_sums, want_amax, res_link, out_planes, drop, dy_planes, into, sole, defer)], group)[0]

    @staticmethod
    def _forward(ctx, y, gamma, beta, running_mean, running_var, residual, relu, training, eps, momentum, group, clamp_eps, pre_sums=None,
                 want_amax=False, res_link=None, out_planes=False, drop=None, dy_planes=False, into=None, sole=False, defer=False):
        """Generator: yields the tensor of each collective (the SyncBN moments) instead of all-reducing it, so that the BatchNorms of parallel
        branches can share one message (GroupBnActFn); `ctx` is the autograd context or a _MemberCtx stand-in."""
        L.init()
        ctx.set_materialize_grads(False)
        ctx.res_link = res_link
        # precision mode 3 with half activations: y may arrive as ONE fp16 plane (written by the conv / depthwise epilogue); it is read as such
        y_bound = planes_amax(y) if (is_planes(y) and nplanes() == 1 and half_acts() and training) else None
        if y_bound is None:
            y = as_nhwc(y)
        b, c, h, w = y.shape
        m = b * h * w
        dev = y.device
        st = stream()
        yp = c if y_bound is not None else pitch_of(y)
        refine_y = y if (_runtime.bn_refine and y_bound is None) else None        # the second-pass variance refinement reads an fp32 y
        coef = torch.empty(4 * c, device=dev)            # mean | invstd | scale | shift
        mean, invstd, scale, shift = coef[:c], coef[c:2 * c], coef[2 * c:3 * c], coef[3 * c:]
        n_global = float(m)
        if training and m == 1 and group is None:
            # torch.nn.BatchNorm2d's behaviour (the ASPP image-pool branch normalises over the batch only: B must be > 1)
            raise ValueError('Expected more than 1 value per channel when training, got input size %s' % (tuple(y.shape),))
        drop_p, drop_seed = drop if (drop is not None and training) else (0.0, 0)
        out_planes = bool(out_planes and training and planes_ok(c, m) and m >= _core.PLANES_MIN_PIXELS)
        res = res_pl = res_amax = None
        if residual is not None:
            if is_planes(residual) and training:
                res_pl, res_amax = residual, planes_amax(residual)
            else:
                res = as_nhwc(residual)
                if out_planes:
                    res_amax = amax_of(res)
        bound = amax_slot(dev) if out_planes else None
        mul = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        if training:
            partial = pre_sums if (pre_sums is not None and pre_sums.dim() == 2 and pre_sums.shape[1] == 2 * c) else None
            kshift = getattr(partial, '_pylc_shift', None) if partial is not None else None
            if partial is not None and group is None:
                # statistics came out of the conv epilogue as per-tile partials: combine + coefficients in one launch
                check(lib.pylc_bn_finalize_from_partial_ex(ptr(partial), partial.shape[0], n_global, c, ptr(gamma), ptr(beta), eps, momentum,
                                                           int(clamp_eps), ptr(running_mean), ptr(running_var), ptr(mean), ptr(invstd),
                                                           ptr(scale), ptr(shift), ptr(res_amax), mul, ptr(bound),
                                                           ptr(refine_y), yp, m, ptr(kshift), st))
            else:
                sums = torch.empty(2 * c, device=dev)                     # [sum | sumsq]
                if partial is not None:
                    check(lib.pylc_bn_stats_from_partial(ptr(partial), partial.shape[0], c, ptr(sums), st))
                else:
                    if y_bound is not None:          # no statistics came with the half tensor: take them from an fp32 copy (rare)
                        y, y_bound = from_planes(mark_planes(y, y_bound)), None
                    ws = torch.empty(lib.pylc_bn_workspace_floats(m, c), device=dev)
                    check(lib.pylc_bn_stats(ptr(y), m, c, yp, ptr(sums), ptr(ws), st))
                if group is not None:
                    # SyncBN: this rank's moments in fp64 [sum | sumsq | count], ONE all-reduce, coefficients from the global moments
                    moments = torch.empty(2 * c + 1, device=dev, dtype=torch.float64)
                    check(lib.pylc_bn_local_moments(ptr(sums), float(m), c, ptr(refine_y), yp, m, ptr(kshift),
                                                    ptr(moments), st))
                    yield moments                                         # all-reduced (SUM) by the driver, alone or with other layers' moments
                    n_global = float(m) * dist.get_world_size(group)      # equal shards (checked by parallel.init_from_env / DataParallel setup)
                    check(lib.pylc_bn_finalize_moments(ptr(moments), n_global, c, ptr(gamma), ptr(beta), eps, momentum, int(clamp_eps),
                                                       ptr(running_mean), ptr(running_var), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                                       ptr(res_amax), mul, ptr(bound), ptr(kshift), st))
                else:
                    check(lib.pylc_bn_finalize_ex(ptr(sums), n_global, c, ptr(gamma), ptr(beta), eps, momentum, int(clamp_eps),
                                                  ptr(running_mean), ptr(running_var), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                                  ptr(res_amax), mul, ptr(bound), ptr(refine_y), yp, m, ptr(kshift), st))
        else:
            check(lib.pylc_bn_eval_coeffs_full(ptr(running_mean), ptr(running_var), ptr(gamma), ptr(beta), eps, c,
                                               ptr(scale), ptr(shift), ptr(mean), ptr(invstd), st))
        if defer and training and y_bound is not None and out_planes and residual is None and drop_p == 0 and into is None:
            # Deferred apply (precision mode 3, half activations): the ONLY consumer is a depthwise conv that applies scale / shift / ReLU to
            # its LDS patch (pylc_dwconv3x3_*_h_bn), so no pass runs and no output is written here -- the result aliases y and travels with
            # the coefficients (ops.bn_act: `_pylc_defer`).  The backward is the ordinary one (ReLU mask recomputed from y).
            out = torch.as_strided(y, y.shape, y.stride())
            ctx.save_for_backward(y, None, coef, bound, None, y_bound)
            ctx.cfg = (relu, training, group, n_global, False)
            ctx.clamp = (bool(clamp_eps), float(eps))
            ctx.bn_emit_ok = False
            ctx.sole = True
            ctx.y_shape = (b, c, h, w)
            ctx.pre_sums = None
            ctx.g_param, ctx.b_param = gamma, beta
            ctx.want_amax = want_amax
            ctx.out_pl = True
            ctx.drop = (0.0, 0)
            ctx.dy_pl = bool(dy_planes and planes_ok(c, m) and yp == c)
            ctx.mark_non_differentiable(bound, coef)
            return out, bound, coef
        op_ = c
        if into is not None:               # write into channels [c0, c0 + c) of a caller-owned concat buffer: into = ([buffer], c0)
            buf, c0 = into[0][0], into[1]
            op_ = pitch_of(buf)
            if out_planes or tuple(buf.shape[2:]) != (h, w) or buf.shape[0] != b or c0 % 4 or c0 + c > buf.shape[1]:
                raise L.PylcError('bn_act into=: slice [%d, %d) does not fit the %s buffer' % (c0, c0 + c, tuple(buf.shape)))
            out = buf[:, c0:c0 + c]
        else:
            out = empty_nhwc(b, c, h, w, dev)
        amax = amax_slot(dev) if (want_amax and not out_planes) else None
        # a ReLU behind a residual add: its mask cannot be recomputed from y, so this pass leaves one bit per element for the backward
        # (bn.hip "1-bit ReLU masks") instead of the backward re-reading `out` twice
        mask = None
        if training and relu and residual is not None and c % 8 == 0 and drop_p == 0 and any(ctx.needs_input_grad) and not _runtime.no_relu_bits:
            mask = torch.empty(m * c // 8, dtype=torch.uint8, device=dev)
        tm = _bn_time('apply%s%s%s' % ('+res' if residual is not None else '', '+bits' if mask is not None else '', '+drop' if drop_p > 0 else ''), m, c,
                      # algorithmic bytes: 4 B per fp32 / two-plane element, 2 B per one-plane (precision mode 3) element, 1/8 B per mask bit
                      m * c * ((2 if y_bound is not None else 4) + (2 * nplanes() if out_planes else 4)
                               + (0 if residual is None else (2 * nplanes() if res_pl is not None else 4)) + (0.125 if mask is not None else 0)))
        tm.__enter__()
        if out_planes or res_pl is not None or drop_p > 0 or mask is not None or y_bound is not None:
            ex = _bn_extra(drop_p=drop_p, drop_seed=drop_seed)
            if mask is not None:
                ex.relu_mask = ptr(mask)
            if y_bound is not None:
                ex.y_half_bound = ptr(y_bound)
            if out_planes:
                ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), pstride(m, c), ptr(bound)
            if res_pl is not None:
                ex.res_planes, ex.res_plane_stride, ex.res_amax = ptr(res_pl), pstride(m, c), ptr(res_amax)
            check(lib.pylc_bn_apply_ex(ptr(y), yp, ptr(scale), ptr(shift), ptr(res), pitch_of(res) if res is not None else (c if res_pl is not None else 0),
                                       None if out_planes else ptr(out), op_, m, c, int(relu), ptr(amax), C.byref(ex), st))
        else:
            check(lib.pylc_bn_apply(ptr(y), yp, ptr(scale), ptr(shift), ptr(res), pitch_of(res) if res is not None else 0,
                                    ptr(out), op_, m, c, int(relu), ptr(amax), st))
        tm.__exit__()
        # ReLU mask in backward: without a residual it is recomputed from y (y*scale + shift > 0, the forward's own
        # expression), so `out` is neither kept alive for it nor read again
        ctx.save_for_backward(y, out if (relu and residual is not None and mask is None) else None, coef, bound, mask, y_bound)
        ctx.cfg = (relu, training, group, n_global, residual is not None)
        ctx.clamp = (bool(clamp_eps), float(eps))
        # a conv dgrad that writes this output's complete gradient may take the backward sums in its epilogue (Conv2dFn.backward): possible
        # for a training-mode pass without fused dropout over a dense fp32 y; `sole` = the caller says the output has ONE consumer
        ctx.bn_emit_ok = bool(training and drop_p == 0 and into is None and yp == c and c % 8 == 0 and any(ctx.needs_input_grad) and y_bound is None)
        ctx.sole = bool(sole)
        ctx.y_shape = (b, c, h, w)
        ctx.pre_sums = None
        ctx.g_param, ctx.b_param = gamma, beta
        ctx.want_amax = want_amax
        ctx.out_pl = out_planes
        ctx.drop = (drop_p, drop_seed)
        ctx.dy_pl = bool(dy_planes and training and planes_ok(c, m) and yp == c)
        if out_planes:
            ctx.mark_non_differentiable(bound)
            return out, bound
        if want_amax:
            ctx.mark_non_differentiable(amax)
            return out, amax
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        return _drive_collectives([BnActFn._backward(ctx, dout)], ctx.cfg[2])[0]

    @staticmethod
    def _backward(ctx, dout):
        """Generator (as _forward): yields the [sum g xhat | sum g] message of a synchronised layer."""
        if dout is None:
            return (None,) * 21
        y, out, coef, out_bound, mask, y_bound = ctx.saved_tensors
        relu, training, group, n_global, has_res = ctx.cfg
        gamma, beta = ctx.g_param, ctx.b_param
        # precision mode 3 with half activations: dout may arrive as ONE fp16 plane (a dgrad's output); read as such unless a gradient link
        # is going to accumulate fp32 values into what this pass hands on
        a_bound = None
        if is_planes(dout) and nplanes() == 1 and half_acts() and training and not (ctx.res_link is not None and ctx.res_link.armed):
            a_bound = planes_amax(dout)
        else:
            dout = as_nhwc(dout)
        b, c, h, w = y.shape
        m = b * h * w
        dev = y.device
        st = stream()
        mean, invstd = coef[:c], coef[c:2 * c]
        scale, shift = (coef[2 * c:3 * c], coef[3 * c:]) if (relu and out is None and mask is None) else (None, None)
        # [dgamma | dbeta] go straight into the flat gradient arena when gamma/beta own adjacent slots there
        tg, tb = _grad_target(gamma), _grad_target(beta)
        direct = (tg is not None and tb is not None and tb.data_ptr() == tg.data_ptr() + 4 * c
                  and ctx.needs_input_grad[1] and ctx.needs_input_grad[2])
        sums = torch.as_strided(tg, (2 * c,), (1,)) if direct else torch.empty(2 * c, device=dev)
        ws = torch.empty(lib.pylc_bn_workspace_floats(m, c), device=dev)
        out_pl = ctx.out_pl and out is not None
        drop_p, drop_seed = ctx.drop
        dy_pl = ctx.dy_pl
        use_ex = out_pl or drop_p > 0 or dy_pl or mask is not None or a_bound is not None or y_bound is not None
        op = (c if out_pl else pitch_of(out)) if out is not None else 0
        dout_pitch = c if a_bound is not None else pitch_of(dout)
        y_pitch = c if y_bound is not None else pitch_of(y)
        ex = None
        dy_bound = None
        msrc = 0.125 if mask is not None else (4 if (relu and out is not None) else 0)         # bytes per element read for the ReLU mask
        pre = getattr(ctx, 'pre_sums', None)
        ctx.pre_sums = None
        e_in = (2 if a_bound is not None else 4) + (2 if y_bound is not None else 4)          # dout + y: one fp16 plane each in precision mode 3
        tm = _bn_time('bwd_sums(from dgrad)' if pre is not None else 'bwd_reduce(+sums)', m, c, m * c * (e_in + msrc) if pre is None else 0)
        tm.__enter__()
        if pre is not None:
            # the conv dgrad that produced `dout` took the per-tile sums in its epilogue: only the combine (and the dy bound) is left
            part, rows, g_amax = pre
            if use_ex:
                ex = _bn_extra(drop_p=drop_p, drop_seed=drop_seed)
                if mask is not None:
                    ex.relu_mask = ptr(mask)
                ex.y_half_bound, ex.dout_half_bound = ptr(y_bound), ptr(a_bound)
                if out_pl:
                    ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), pstride(m, c), ptr(out_bound)
                if dy_pl:
                    dy_bound = amax_slot(dev)
                    ex.g_amax = ptr(g_amax)
            local_bound = dy_pl and not (training and group is not None)
            check(lib.pylc_bn_bwd_sums_from_partial(ptr(part), rows, c, ptr(sums), ptr(gamma), ptr(invstd), n_global, ptr(g_amax),
                                                    ptr(dy_bound) if local_bound else None, st))
        elif use_ex:
            ex = _bn_extra(drop_p=drop_p, drop_seed=drop_seed)
            if mask is not None:
                ex.relu_mask = ptr(mask)
            ex.y_half_bound, ex.dout_half_bound = ptr(y_bound), ptr(a_bound)
            if out_pl:
                ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), pstride(m, c), ptr(out_bound)
            if dy_pl:
                g_amax, dy_bound = amax_slot(dev), amax_slot(dev)
                ex.g_amax = ptr(g_amax)
            local_bound = dy_pl and not (training and group is not None)
            check(lib.pylc_bn_bwd_reduce_ex(ptr(dout), dout_pitch, None if out_pl else ptr(out), op, ptr(y), y_pitch, ptr(mean), ptr(invstd),
                                            m, c, int(relu), ptr(sums), ptr(ws), ptr(scale), ptr(shift), ptr(gamma), n_global, C.byref(ex),
                                            ptr(dy_bound) if local_bound else None, st))
        else:
            check(lib.pylc_bn_bwd_reduce(ptr(dout), pitch_of(dout), ptr(out), op, ptr(y), pitch_of(y), ptr(mean), ptr(invstd),
                                         m, c, int(relu), ptr(sums), ptr(ws), ptr(scale), ptr(shift), st))
        tm.__exit__()
        local_sums = sums
        if training and group is not None:
            sums = local_sums.clone()          # parameter grads stay local (the gradient all-reduce sums them later)
            yield sums
            if dy_pl:
                check(lib.pylc_bn_bwd_bound(ptr(sums), ptr(gamma), ptr(invstd), n_global, c, ptr(g_amax), ptr(dy_bound), st))
        clamp_eps, eps = getattr(ctx, 'clamp', (False, 1e-5))
        if training and clamp_eps:
            # batchnorm.py:125 inv_std = clamp(var, eps)^-1/2: where the clamp is active inv_std no longer depends on the batch, so autograd
            # sends nothing through the variance there -- dy loses its xhat * sum(g xhat) / n term on those channels (dgamma keeps the sum).
            # The finalize kernels store exactly (float)(1 / sqrt((double)eps)) for a clamped channel.
            thr = torch.tensor(eps, dtype=torch.float32, device=dev).double().rsqrt().float()
            sums = torch.cat((sums[:c] * (invstd < thr), sums[c:]))
        if not training:
            sums_apply = torch.zeros(2 * c, device=dev)   # running statistics are constants: dy = gamma*invstd*g
            if dy_pl:
                check(lib.pylc_bn_bwd_bound(ptr(sums_apply), ptr(gamma), ptr(invstd), n_global, c, ptr(g_amax), ptr(dy_bound), st))
        else:
            sums_apply = sums
        dy = empty_nhwc(b, c, h, w, dev)
        want_res = has_res and ctx.needs_input_grad[5]
        # without a ReLU (and without dropout) the residual's gradient IS dout: hand the tensor on instead of having the kernel write a
        # copy (unless a conv is going to accumulate its dgrad into the buffer, which must then be ours)
        res_is_dout = want_res and not relu and drop_p == 0 and not (ctx.res_link is not None and ctx.res_link.armed)
        # with the 1-bit mask and a gradient link, the residual's gradient relu'(dout) is not written out at all: the (dout, mask) pair is
        # parked on the link and the conv dgrad that consumes it forms the masked gradient in its epilogue (pylc_conv2d_dgrad_add)
        lk = ctx.res_link
        park_masked = (want_res and relu and mask is not None and drop_p == 0 and lk is not None and lk.armed and lk.buf is None
                       and lk.masked is None and a_bound is None and pitch_of(dout) == c and _runtime.fuse_res_grad)
        g_out = empty_nhwc(b, c, h, w, dev) if (want_res and not res_is_dout and not park_masked) else None
        amax_dy = amax_slot(dev) if (ctx.want_amax and not dy_pl) else None
        tm = _bn_time('bwd_apply%s' % ('+gres' if g_out is not None else ''), m, c,
                      m * c * (e_in + (2 * nplanes() if dy_pl else 4) + msrc + (4 if g_out is not None else 0)))
        tm.__enter__()
        if use_ex:
            if dy_pl:
                ex.dy_planes, ex.dy_plane_stride, ex.dy_bound = ptr(dy), pstride(m, c), ptr(dy_bound)
            check(lib.pylc_bn_bwd_apply_ex(ptr(dout), dout_pitch, None if out_pl else ptr(out), op, ptr(y), y_pitch, ptr(mean), ptr(invstd),
                                           ptr(gamma), ptr(sums_apply), n_global, m, c, int(relu), None if dy_pl else ptr(dy), c,
                                           ptr(g_out), c if g_out is not None else 0, ptr(amax_dy), ptr(scale), ptr(shift), C.byref(ex), st))
        else:
            check(lib.pylc_bn_bwd_apply(ptr(dout), pitch_of(dout), ptr(out), op, ptr(y), pitch_of(y), ptr(mean), ptr(invstd),
                                        ptr(gamma), ptr(sums_apply), n_global, m, c, int(relu), ptr(dy), c,
                                        ptr(g_out), c if g_out is not None else 0, ptr(amax_dy), ptr(scale), ptr(shift), st))
        tm.__exit__()
        if dy_pl:
            mark_planes(dy, dy_bound)   # the conv backward that receives dy reads it as planes (autograd hands the tensor on unchanged:
                                        # y has ONE consumer, this BatchNorm)
        elif amax_dy is not None:
            tag_amax(dy, amax_dy)       # the conv backward that receives dy reuses it (when autograd hands the tensor on unchanged)
        if training:
            dy._pylc_zero_colsum = True   # batch statistics: dy sums to zero over the rows of every channel (Conv2dFn.backward: bias gradient)
        dgamma = dbeta = None
        if direct:
            _deliver_grad(gamma, tg)
            _deliver_grad(beta, tb)
        else:
            if ctx.needs_input_grad[1]:
                if tg is not None:
                    tg.copy_(local_sums[:c])
                    dgamma = _deliver_grad(gamma, tg)
                else:
                    dgamma = local_sums[:c].clone()
            if ctx.needs_input_grad[2]:
                if tb is not None:
                    tb.copy_(local_sums[c:])
                    dbeta = _deliver_grad(beta, tb)
                else:
                    dbeta = local_sums[c:].clone()
        if g_out is not None and a_bound is not None:
            mark_planes(g_out, a_bound)      # the residual's gradient leaves in dout's format and scale
        if res_is_dout:
            g_out = dout
        if park_masked:
            lk.masked = (dout, mask)
        link = ctx.res_link
        if g_out is not None and link is not None and link.armed and link.buf is None and tuple(g_out.shape) == tuple(y.shape):
            link.buf = g_out         # the first conv's dgrad accumulates into it and returns it as x's whole gradient
            g_out = None
        return (dy, dgamma, dbeta, None, None, g_out) + (None,) * 15


def _drive_collectives(gens, group):
    """Run BatchNorm generators (BnActFn._forward / _backward) in lockstep: whatever they yield in one round is all-reduced as ONE message
    (a lone generator: its own tensor, no copy).  Returns their return values."""
    results = [None] * len(gens)
    live = list(range(len(gens)))
    while live:
        msgs = []
        for i in list(live):
            try:
                msgs.append(next(gens[i]))
            except StopIteration as e:
                results[i] = e.value
                live.remove(i)
        if len(msgs) == 1:
            _runtime.sync_all_reduce(msgs[0], group)
        elif msgs:
            flat = torch.cat([t.reshape(-1) for t in msgs])        # one dtype per round: fp64 moments (forward), fp32 sums (backward)
            _runtime.sync_all_reduce(flat, group)
            o = 0
            for t in msgs:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()
    return results


class _MemberCtx:
    """What BnActFn._forward / _backward use of an autograd context, for one BatchNorm inside a GroupBnActFn node."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()
        self.non_differentiable = ()

    def set_materialize_grads(self, value):
        pass

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        self.non_differentiable = tensors


class GroupBnActFn(torch.autograd.Function):
    """Several BatchNorm(+act) layers over PARALLEL branches (the ASPP's five, aspp.py:73-86) as one autograd node, so that under SyncBN their
    statistics travel in one all-reduce per direction instead of one per layer: the members run BnActFn's own code (same kernels, same order
    per layer) with the collectives of a round concatenated.  apply(group, n, nargs, *member_args) -> the members' outputs, flattened."""

    @staticmethod
    def forward(ctx, group, n, nargs, *flat):
        ctx.set_materialize_grads(False)
        members = [_MemberCtx(ctx.needs_input_grad[3 + i * nargs:3 + (i + 1) * nargs]) for i in range(n)]
        results = _drive_collectives([BnActFn._forward(m, *flat[i * nargs:(i + 1) * nargs]) for i, m in enumerate(members)], group)
        saved, outs, nondiff = [], [], []
        ctx.layout = []
        for m, r in zip(members, results):
            r = r if isinstance(r, tuple) else (r,)
            ctx.layout.append((len(saved), len(m.saved_tensors), len(r)))
            saved.extend(m.saved_tensors)
            outs.extend(r)
            nondiff.extend(m.non_differentiable)
        ctx.save_for_backward(*saved)
        ctx.members, ctx.group = members, group
        if nondiff:
            ctx.mark_non_differentiable(*nondiff)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        gens, k = [], 0
        for m, (o, ns, nout) in zip(ctx.members, ctx.layout):
            m.saved_tensors = saved[o:o + ns]
            gens.append(BnActFn._backward(m, grads[k]))
            k += nout
        results = _drive_collectives(gens, ctx.group)
        return (None, None, None) + tuple(g for r in results for g in r)


def bn_act_group(specs, group):
    """bn_act for the BatchNorms of parallel branches, as one node (GroupBnActFn).  specs: one dict per layer with the keyword arguments of
    bn_act (y, gamma, beta, running_mean, running_var + options); returns the outputs in order."""
    flat, marks = [], []
    for sp in specs:
        sp = dict(sp)
        y, training, into = sp['y'], sp.get('training', True), sp.get('into')
        pre = getattr(y, '_pylc_sums', None) if training else None
        dy_pl = bool(getattr(y, '_pylc_dy_pl', False)) and not _runtime.no_planes and _runtime.planes_dy
        out_planes = bool(sp.get('out_planes', False)) and ranges_needed() and not _runtime.no_planes
        drop = sp.get('drop')
        if drop is not None and not (training and _runtime.dropout_enabled and drop[0] > 0):
            drop = None
        ranged = ranges_needed()
        flat += [y, sp['gamma'], sp['beta'], sp['running_mean'], sp['running_var'], sp.get('residual'), sp.get('relu', True), training,
                 sp.get('eps', 1e-5), sp.get('momentum', 0.1), group, sp.get('clamp_eps', False), pre, ranged, sp.get('res_link'),
                 (out_planes and into is None) if ranged else False, drop, dy_pl if ranged else False, into, sp.get('sole', False), False]
        marks.append((ranged, is_planes_candidate(out_planes and into is None, training, y) if ranged else False))
    outs = list(GroupBnActFn.apply(group, len(specs), 21, *flat))
    res = []
    for ranged, as_planes in marks:
        out = outs.pop(0)
        if ranged:
            tagv = outs.pop(0)
            if as_planes:
                mark_planes(out, tagv)
            else:
                tag_amax(out, tagv)
        res.append(out)
    return res


def bn_act(y, gamma, beta, running_mean, running_var, residual=None, relu=True, training=True, eps=1e-5, momentum=0.1,
           group=None, clamp_eps=False, res_link=None, out_planes=False, drop=None, into=None, sole=False, defer=False):
    pre = getattr(y, '_pylc_sums', None) if training else None
    dy_pl = bool(getattr(y, '_pylc_dy_pl', False)) and not _runtime.no_planes and _runtime.planes_dy
    out_planes = bool(out_planes) and ranges_needed() and not _runtime.no_planes
    if drop is not None and not (training and _runtime.dropout_enabled and drop[0] > 0):
        drop = None
    if ranges_needed():
        defer = bool(defer and _runtime.defer_bn_apply and training and out_planes and into is None and residual is None and drop is None
                     and nplanes() == 1 and half_dw() and is_planes(y))
        res = BnActFn.apply(y, gamma, beta, running_mean, running_var, residual, relu, training, eps, momentum, group, clamp_eps, pre,
                            True, res_link, out_planes and into is None, drop, dy_pl, into, sole, defer)
        if len(res) == 3:
            # deferred apply: `out` aliases y (the BatchNorm's INPUT, one fp16 plane); what a consumer needs to form the output travels here.
            # NOT marked as planes: only DwConv3x3Fn understands it (anything else goes through ops.materialize_deferred)
            out, bound, coef = res
            out._pylc_defer = (coef, bool(relu), planes_amax(y), bound, out._version)
            return out
        out, tagv = res
        if is_planes_candidate(out_planes and into is None, training, y):
            mark_planes(out, tagv)
        else:
            tag_amax(out, tagv)
        return out
    return BnActFn.apply(y, gamma, beta, running_mean, running_var, residual, relu, training, eps, momentum, group, clamp_eps, pre,
                         False, res_link, False, drop, False, into, sole, False)


def materialize_deferred(x):
    """The fp16-plane output of a BatchNorm whose apply pass was deferred (bn_act(defer=True)), for a consumer that cannot apply it itself:
    the pass pylc_bn_apply_ex would have made (no autograd: callers are inside a Function's forward)."""
    coef, relu, y_bound, bound, _ = x._pylc_defer
    b, c, h, w = x.shape
    m = b * h * w
    out = empty_nhwc(b, c, h, w, x.device)
    ex = _bn_extra()
    ex.y_half_bound = ptr(y_bound)
    ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), pstride(m, c), ptr(bound)
    check(lib.pylc_bn_apply_ex(ptr(x), c, ptr(coef[2 * c:3 * c]), ptr(coef[3 * c:]), None, 0, None, c, m, c, int(relu), None, C.byref(ex), stream()))
    return mark_planes(out, bound)


def is_planes_candidate(out_planes, training, y):
    """Mirror of BnActFn.forward's decision whether the output was written as planes."""
    b, c, h, w = y.shape
    return bool(out_planes and training and planes_ok(c, b * h * w) and b * h * w >= _core.PLANES_MIN_PIXELS)


__all__ = [n for n in dir() if not n.startswith('__')]      # everything, underscore helpers included: the package re-exports it (pylc_amd/ops/__init__.py)
