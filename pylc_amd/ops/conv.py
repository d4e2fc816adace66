"""Dense convolution operators: Conv2dFn (forward / dgrad / wgrad on the implicit-GEMM kernels), the 2x2 transposed conv of the U-Net, and the
fused inference conv + BatchNorm (fp32 and fp16-plane tensors).  Replaces nn.Conv2d / nn.ConvTranspose2d as used in models/backbone/*.py,
models/modules/aspp.py, models/decoder.py, models/architectures/unet.py."""
from . import _core
from ._core import *      # noqa: F401,F403  (layout / planes / range / stream helpers, lib bindings, torch)


def _padded_stem_filter(w, cin):
    """The thin-input stem filter ([Cout, 3, k, k]) zero-padded to the 4-channel input pack, KRSC.  Built once per optimiser step, not per
    forward: cached on the parameter, keyed by its version counter and -- the flat-arena optimisers update parameters through raw
    pointers, which does not bump it -- the arena's generation."""
    arena = getattr(w, '_pylc_arena', None)
    arena = arena() if arena is not None else None
    key = (w._version, arena.generation if arena is not None else -1, cin)
    cache = getattr(w, '_pylc_w4', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    cout, cin_w, r, s = w.shape
    w_k = torch.zeros((cout, r, s, cin), device=w.device, dtype=torch.float32)
    w_k[..., :cin_w] = w.detach().permute(0, 2, 3, 1)
    w._pylc_w4 = (key, w_k)
    return w_k


class Conv2dFn(torch.autograd.Function):
    """y = conv2d(x, w) + bias on the MFMA implicit-GEMM kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, want_stats=False, x_amax=None, w_amax=None, res_link=None, out=None, convert=False, bn_src=None):
        L.init()
        ctx.set_materialize_grads(False)      # the auxiliary outputs (statistics, ranges) carry no gradient: no zero fills
        ctx.res_link = res_link if (res_link is not None and ctx.needs_input_grad[0]) else None
        if ctx.res_link is not None:
            res_link.pending += 1
        ctx.bn_src = bn_src if ctx.needs_input_grad[0] else None      # the BatchNorm node that produced x (ops.conv2d): see backward
        cout, cin_w, r, s = w.shape
        b_, _, h_, w_ = x.shape
        # (a bias is added by the planes kernels' epilogue like any other; its gradient is a column sum over the dy planes,
        # pylc_planes_colsum)
        takes = conv_takes_planes(w, b_ * h_ * w_, 0) and w_amax is not None and out is None
        if takes and not is_planes(x) and convert:
            # (convert: training graphs only -- grad mode as seen by ops.conv2d; inference keeps fp32 operands and the kernels the
            # fused conv + BatchNorm epilogue path runs)
            # one pass; the forward AND the wgrad then copy their operand tiles instead of splitting them.  A tensor read by several
            # convs (projection blocks, the ASPP input) is converted once: the planes copy rides on the tensor object
            cache = getattr(x, '_pylc_plcache', None)
            if cache is not None and cache[1] == x._version:
                x = cache[0]
            else:
                src = x
                x = to_planes(x, x_amax)
                src._pylc_plcache = (x, src._version)
        x_pl = takes and is_planes(x)
        if not x_pl:
            x = as_nhwc(x)
            if _runtime.debug_planes:
                print('[pylc] conv fwd on fp32 operands: x %s w %s takes=%s grad=%s' % (tuple(x.shape), tuple(w.shape), takes, torch.is_grad_enabled()), flush=True)
        cin = x.shape[1]
        xp = pitch_of(x)
        w_k = w
        if cin_w % 4 != 0:
            # thin-input stem (Cin=3): zero-pad the KRSC rows to 4 channels; x must already be the 4-channel pack
            if cin != _r4(cin_w):
                raise L.PylcError('conv expects the %d-channel packed input for a %d-channel filter' % (_r4(cin_w), cin_w))
            w_k = _padded_stem_filter(w, cin)
        elif cin != cin_w:
            raise L.PylcError('conv: input has %d channels, filter expects %d' % (cin, cin_w))
        elif not (w.permute(0, 2, 3, 1).is_contiguous()):
            raise L.PylcError('conv weight must have KRSC (channels_last) memory')
        b, _, h, wd = x.shape
        oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
        if out is not None:
            # write into channels [0, cout) of a caller-owned NHWC buffer (a concat target): out = [buffer]
            buf = out[0]
            yp = pitch_of(buf)
            if tuple(buf.shape[2:]) != (oh, ow) or buf.shape[0] != b or buf.shape[1] < cout or cout % 4:
                raise L.PylcError('conv out= buffer %s does not fit a [%d,%d,%d,%d] result' % (tuple(buf.shape), b, cout, oh, ow))
            y = buf[:, :cout]
        else:
            yp = _r4(cout)
            y = empty_nhwc(b, cout, oh, ow, x.device, yp)
        d = _conv_desc(x, cin, cout, r, s, stride, pad, dil, xp, yp)
        if x_pl:
            d.x_fmt = 1
            x_amax = planes_amax(x)          # the bound the producer scaled the planes with
        d.x_amax, d.w_amax = ptr(x_amax), ptr(w_amax)
        ctx.ranges = (x_amax, w_amax)
        ctx.x_pl = x_pl
        # precision mode 3: y leaves as one fp16 plane when a BatchNorm is going to read it (want_stats) -- half the bytes of the store-bound
        # epilogue and of the three BatchNorm passes over y
        y_bound = None
        if want_stats and x_pl and convert and half_acts() and bias is None and out is None and cout % 8 == 0 and yp == cout and planes_ok(cout, b * oh * ow):
            y_bound = amax_slot(x.device)
            d.out_fmt, d.out_bound = 1, ptr(y_bound)
        planes = getattr(w, '_pylc_planes', None) if (w_amax is not None and w_k is w) else None
        if planes is not None:
            d.w_planes, d.w_planes_fmt = ptr(planes[0]), filter_planes_fmt(planes)
        ev = None
        if _core._timer is not None and (x_pl if _core._timer.planes else _is_dominant_tile(b * oh * ow, yp, cin, r * s)):
            ev = _core._timer.bracket(2.0 * b * oh * ow * cout * r * s * cin, 1, 'fwd%dx%d' % (r, s),
                                4.0 * (b * h * wd * cin + cout * r * s * cin + b * oh * ow * cout))
            ev[0].record()
        sums = None
        if want_stats:
            if cout % 4:
                raise L.PylcError('fused BatchNorm statistics need Cout % 4 == 0')
            part = torch.empty(lib.pylc_conv2d_fwd_stats_floats(C.byref(d)), device=x.device)
            rows = C.c_int(0)
            check(lib.pylc_conv2d_fwd_stats(C.byref(d), ptr(x), ptr(w_k), ptr(bias), ptr(y), ptr(part), C.byref(rows), stream()))
            sums = part[:rows.value * 2 * cout].view(rows.value, 2 * cout)      # per-tile partials; the BatchNorm combines them
        else:
            check(lib.pylc_conv2d_fwd(C.byref(d), ptr(x), ptr(w_k), ptr(bias), ptr(y), stream()))
        if ev is not None:
            ev[1].record()
        ctx.save_for_backward(x, w_k)
        # dy may come back as fp16 planes (BatchNorm backward writes them) when the backward kernels can take them
        ctx.dy_pl_ok = x_pl and cout % 8 == 0 and ow >= 16 and planes_ok(cout, b * oh * ow) and yp == cout
        ctx.geom = (stride, pad, dil, cin_w, bias is not None)
        ctx.w_param, ctx.b_param = w, bias
        if want_stats:
            if y_bound is not None:
                ctx.mark_non_differentiable(sums, y_bound)
                return y, sums, y_bound
            ctx.mark_non_differentiable(sums)
            return y, sums
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            return (None,) * 13
        x, w_k = ctx.saved_tensors
        flush_deferred_wgrad(x.device)
        stride, pad, dil, cin_w, has_bias = ctx.geom
        w, bias = ctx.w_param, ctx.b_param
        x_pl = ctx.x_pl
        if x_pl:
            mark_planes(x, ctx.ranges[0])       # saved tensors come back as new Python objects: restore the marker
        dy_pl = is_planes(dy)
        if x_pl and _runtime.planes_fwd_only:          # debug: planes in the forward pass only
            x, x_pl = from_planes(x), False
        if dy_pl and not x_pl:
            dy, dy_pl = from_planes(dy), False
        elif x_pl and not dy_pl:
            dy = as_nhwc(dy)
            if dy.shape[1] % 8 == 0 and pitch_of(dy) == dy.shape[1] and planes_ok(dy.shape[1], dy.shape[0] * dy.shape[2] * dy.shape[3]) \
                    and dy.shape[3] >= 16:
                dy, dy_pl = to_planes(dy), True           # one pass; dgrad and wgrad then both read planes
            else:
                x, x_pl = from_planes(x), False
        if not dy_pl:
            dy = as_nhwc(dy)
            if _runtime.debug_planes:
                print('[pylc] conv bwd on fp32 operands: x %s w %s x_pl=%s dy_pl_ok=%s' % (tuple(x.shape), tuple(w.shape), ctx.x_pl, ctx.dy_pl_ok), flush=True)
        cout, _, r, s = w.shape
        cin = x.shape[1]
        yp = pitch_of(dy)
        if yp < _r4(cout):      # a grad produced outside our kernels: re-pitch so vector loads stay in bounds
            t = zeros_nhwc(dy.shape[0], cout, dy.shape[2], dy.shape[3], dy.device, _r4(cout))
            t.copy_(dy)
            dy, yp = t, _r4(cout)
        d = _conv_desc(x, cin, cout, r, s, stride, pad, dil, pitch_of(x), yp)
        d.x_fmt, d.dy_fmt = int(x_pl), int(dy_pl)
        x_amax, w_amax = ctx.ranges
        dy_amax = None
        if ranges_needed():
            if x_amax is None or w_amax is None:
                raise L.PylcError('conv backward in f16x3 mode, but the forward ran without operand ranges')
            dy_amax = planes_amax(dy) if dy_pl else amax_of(dy)
            d.x_amax, d.w_amax, d.dy_amax = ptr(x_amax), ptr(w_amax), ptr(dy_amax)
            planes = getattr(w, '_pylc_planes', None) if w_k is w else None
            if planes is not None:
                d.w_planes_t, d.w_planes_fmt = ptr(planes[1]), filter_planes_fmt(planes)
        st = stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            kp = _r4(cout)
            link = ctx.res_link
            masked = None
            if (link is not None and link.masked is not None and link.buf is None and dy_pl and stride == 1 and cin % 8 == 0
                    and tuple(link.masked[0].shape) == tuple(x.shape)):
                masked, link.masked = link.masked, None       # relu'(dout) is formed in this dgrad's epilogue: no buffer to accumulate into
            sink = _link_sink(link)
            if sink is not None:            # part of x's gradient is already in `sink`: dgrad adds to it (no autograd add pass)
                dx = sink
            else:
                dx = empty_nhwc(*x.shape, device=x.device)
            d.x_pitch = cin
            # precision mode 3: dx leaves as one fp16 plane when it is the whole gradient of a BatchNorm output with this conv as its only
            # consumer (nothing will be added to it: autograd would add the raw bytes)
            dx_bound = None
            bn_node = ctx.bn_src
            if (half_acts() and dy_pl and stride == 1 and sink is None and masked is None and link is None and cin % 8 == 0 and bn_node is not None
                    and getattr(bn_node, 'sole', False) and not _runtime.fuse_bn_sums and planes_ok(cin, x.shape[0] * x.shape[2] * x.shape[3])):
                dx_bound = amax_slot(x.device)
                d.out_fmt, d.out_bound = 1, ptr(dx_bound)
            wt = None
            if lib.pylc_conv2d_dgrad_needs_f32_weights(C.byref(d)):      # else the prepared planes are all the kernel reads
                wt = torch.empty((cin, r * s, kp), device=x.device, dtype=torch.float32)
                check(lib.pylc_weight_transpose(ptr(w_k), ptr(wt), cout, r * s, cin, st))
            ev = None
            n_launch = 1 if stride == 1 else min(r, 2) * min(s, 2)     # one launch per non-empty output parity class
            if _core._timer is not None and (dy_pl if _core._timer.planes else _is_dominant_tile(x.shape[0] * x.shape[2] * x.shape[3] // n_launch, cin, kp, 2)):
                ev = _core._timer.bracket(2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * r * s * cin, n_launch, 'dgrad%dx%d' % (r, s),
                                    4.0 * (dy.numel() + cout * r * s * cin + x.numel()))
                ev[0].record()
            # x is a BatchNorm's output and this dgrad writes its COMPLETE gradient (sole consumer, or the last consumer of the gradient
            # link): the sums that BatchNorm's backward starts with are taken in this epilogue (pylc_conv2d_dgrad_bn) and handed to its
            # node, which then skips its read pass over (dout, y)
            bn = ctx.bn_src
            emit = (bn is not None and dy_pl and stride == 1 and cin % 8 == 0 and _runtime.fuse_bn_sums and getattr(bn, 'pre_sums', None) is None
                    and getattr(bn, 'bn_emit_ok', False) and tuple(bn.y_shape) == tuple(x.shape)
                    and ((link is None and bn.sole) or (link is not None and link.pending == 1)))
            if emit:
                L._need_experimental('PYLC_RUNTIME=fuse_bn_sums=1')
                y_bn, _, coef, _, bmask, _ = bn.saved_tensors
                cb = x.shape[1]
                relu_bn = bn.cfg[0]
                bb = L.BnBack()
                bb.y, bb.mean, bb.invstd = ptr(y_bn), ptr(coef[:cb]), ptr(coef[cb:2 * cb])
                if relu_bn and bmask is None:
                    bb.scale, bb.shift = ptr(coef[2 * cb:3 * cb]), ptr(coef[3 * cb:])
                bb.relu_mask, bb.relu = ptr(bmask) if relu_bn else None, int(relu_bn)
                gmx = amax_slot(x.device)
                bb.g_amax = ptr(gmx)
                part = torch.empty(lib.pylc_conv2d_dgrad_bn_floats(C.byref(d)), device=x.device)
                rows = C.c_int(0)
                check(lib.pylc_conv2d_dgrad_bn(C.byref(d), ptr(dy), ptr(wt), ptr(dx), 1 if (sink is not None and masked is None) else 0,
                                               ptr(masked[0]) if masked is not None else None, ptr(masked[1]) if masked is not None else None,
                                               C.byref(bb), ptr(part), C.byref(rows), st))
                bn.pre_sums = (part, rows.value, gmx)
            elif masked is not None:
                check(lib.pylc_conv2d_dgrad_add(C.byref(d), ptr(dy), ptr(wt), ptr(dx), 0, ptr(masked[0]), ptr(masked[1]), st))
            else:
                check(lib.pylc_conv2d_dgrad(C.byref(d), ptr(dy), ptr(wt), ptr(dx), 1 if sink is not None else 0, st))
            if ev is not None:
                ev[1].record()
            d.x_pitch = pitch_of(x)
            d.out_fmt, d.out_bound = 0, None
            if dx_bound is not None:
                mark_planes(dx, dx_bound)
            if link is not None:
                link.pending -= 1
                if link.pending > 0:        # other consumers of x follow: they accumulate into the same buffer
                    link.buf, dx = dx, None
                else:
                    link.buf = None
        if ctx.needs_input_grad[1]:
            # wgrad is off the critical chain (only the optimiser needs it), so it runs on a side stream: the matrix-bound
            # wgrad kernels then overlap the HBM-bound BatchNorm-backward kernels of the layers that follow on the main stream
            side = _side_stream(x.device) if _runtime.side_stream_on() else None
            tgt = _grad_target(w)

            def launch_wgrad(side=side, tgt=tgt, x=x, dy=dy, w=w, w_k=w_k, d=d, dy_amax=dy_amax, x_amax=x_amax):   # bound now: it may run later
                if side is not None:
                    ev = torch.cuda.Event()      # recorded AFTER the dgrad launch: wgrad starts when the dgrad is done (letting it
                    ev.record()                  # start next to the dgrad was measured 5 % slower, making every dgrad wait for the
                                                 # previous wgrad 2 % slower)
                    side.wait_event(ev)
                    _keep_for_side(x.device, x, dy, w_k, dy_amax, x_amax)
                with torch.cuda.stream(side) if side is not None else _nullcontext():
                    sst = stream()
                    nbytes = lib.pylc_conv2d_wgrad_workspace(C.byref(d))
                    ws = _ws(nbytes, x.device)
                    if cin_w % 4 == 0 and tgt is not None and side is None and _runtime.batch_slab_sums and nbytes > 0:
                        # one queue: the split-K slabs stay in this weight's OWN workspace and every layer's slab sum runs in one launch
                        # before the gradients are read (ops.flush_slab_sums, from sync_side_streams / the gradient bucketer)
                        own = getattr(w, '_pylc_slab_ws', None)
                        if own is not None and own.is_cuda and _core.slab_sum_pending(own.device, own):
                            # the same weight again in this pass: sum its first slabs before they are overwritten -- or, when the second plan
                            # needs a larger block, before the old block goes back to the allocator with a noted sum still pointing at it
                            _core.flush_slab_sums(own.device)
                        if own is None or own.numel() * 4 < nbytes or own.device != x.device:
                            own = w._pylc_slab_ws = torch.empty(nbytes // 4 + 1, device=x.device)
                        dwl = tgt
                        pend = L.SlabSum()
                        check(lib.pylc_conv2d_wgrad_slabs(C.byref(d), ptr(x), ptr(dy), ptr(dwl), ptr(own), nbytes, C.byref(pend), sst))
                        if pend.splits > 0:
                            _core.add_slab_sum(x.device, pend)
                    elif cin_w % 4 == 0:
                        dwl = tgt if tgt is not None else torch.empty((cout, r, s, cin), device=x.device).permute(0, 3, 1, 2)
                        check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dwl), None, ptr(ws), nbytes, sst))
                    else:
                        dw4 = torch.empty((cout, r, s, cin), device=x.device)
                        check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw4), None, ptr(ws), nbytes, sst))
                        dwl = tgt if tgt is not None else torch.empty((cout, r, s, cin_w), device=x.device).permute(0, 3, 1, 2)
                        dwl.copy_(dw4[..., :cin_w].permute(0, 3, 1, 2))
                if side is not None and tgt is None:
                    torch.cuda.current_stream().wait_stream(side)      # the returned tensor is consumed by autograd on the main stream
                return _deliver_grad(w, dwl)

            hold = getattr(w, '_pylc_wgrad_hold', 0) if _runtime.wgrad_hold else 0
            if hold and side is not None and tgt is not None:
                # a layer-specific launch ORDER (set by the network, e.g. nets/deeplabv3p.py Decoder): this conv's wgrad fills every CU for
                # milliseconds; started now it would sit in front of the short kernels that follow on the main queue (the 256->48 dgrad of
                # decoder.py:27 waited 2 ms behind the 304->256 wgrad).  Held until `hold` further conv backwards have begun.
                _defer_wgrad(x.device, launch_wgrad, hold)
                dw = None
            elif _runtime.defer_wgrad_1x1 and r * s == 1 and side is not None and tgt is not None:
                # an HBM-heavy 1x1 wgrad started now would run beside the (HBM-bound) BatchNorm backward that follows on the main
                # stream; held back until the NEXT conv backward begins, it runs beside that conv's matrix-bound dgrad instead
                _defer_wgrad(x.device, launch_wgrad)
                dw = None
            else:
                dw = launch_wgrad()
        if has_bias and ctx.needs_input_grad[2] and getattr(dy, '_pylc_zero_colsum', False) and _runtime.skip_zero_bias_grad:
            # dy was written by the backward of a TRAINING-mode BatchNorm that reads this conv's output directly (unet.py:112-118):
            # dy = k (g - mean g - xhat mean(g xhat)) sums to ZERO over the rows of every channel (sum xhat = 0), i.e. the loss does not
            # depend on a bias that the batch mean removes again.  What a column sum over dy -- or autograd in the reference -- returns
            # here is the rounding noise of that cancellation (tests/golden: these keys are the fixtures' `zero_grad_keys`), so the pass
            # over dy is skipped and the gradient is the exact value.  (SyncBN: the sum over ALL ranks' rows is zero, and the gradient
            # all-reduce adds the ranks' bias gradients.)
            tgt = _grad_target(bias)
            if tgt is not None:
                tgt.zero_()
                db = _deliver_grad(bias, tgt)
            else:
                db = torch.zeros_like(bias)
        elif has_bias and ctx.needs_input_grad[2]:
            m = dy.shape[0] * dy.shape[2] * dy.shape[3]
            cp = _r4(cout)
            sums = torch.empty(2 * cp, device=x.device)
            if dy_pl:       # per-channel sums straight from the planes (one read pass, as the fp32 form)
                ws = torch.empty(lib.pylc_planes_colsum_workspace_floats(cout), device=x.device)
                check(lib.pylc_planes_colsum(ptr(dy), cout, pstride(m, cout), nplanes(), ptr(planes_amax(dy)), m, cout, ptr(sums), ptr(ws), st))
            else:
                ws = torch.empty(lib.pylc_bn_workspace_floats(m, cp), device=x.device)
                check(lib.pylc_bn_stats(ptr(dy), m, cp, yp, ptr(sums), ptr(ws), st))
            tgt = _grad_target(bias)
            if tgt is not None:
                tgt.copy_(sums[:cout])
                db = _deliver_grad(bias, tgt)
            else:
                db = sums[:cout].clone()
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, dil=1, want_stats=False, res_link=None, out=None):
    """want_stats: also produce the per-channel (sum, sum of squares) of y in the conv epilogue and attach them to the
    returned tensor as `_pylc_sums` for the BatchNorm that consumes it (ops.bn_act picks them up).
    out: [buffer] -- write the result into the leading channels of that NHWC buffer (see crop_concat)."""
    xa = wa = None
    if ranges_needed():
        L.init()
        xa, wa = amax_of(x), weight_amax(w)
    src = x.grad_fn if (torch.is_grad_enabled() and x.requires_grad) else None
    bn_src = src if hasattr(src, 'bn_emit_ok') else None      # x is the output of a BatchNorm (BnActFn node: carries bn_emit_ok / sole)
    if want_stats:
        res = Conv2dFn.apply(x, w, bias, stride, pad, dil, True, xa, wa, res_link, out, torch.is_grad_enabled(), bn_src)
        y, sums = res[0], res[1]
        if len(res) == 3:
            mark_planes(y, res[2])       # one fp16 plane (precision mode 3 with half activations): the BatchNorm reads it as such
        y._pylc_sums = sums
        if bias is not None:
            sums._pylc_shift = bias.detach()      # the epilogue takes the statistics of (y - bias): the finalize adds it back to the mean
    else:
        y = Conv2dFn.apply(x, w, bias, stride, pad, dil, False, xa, wa, res_link, out, torch.is_grad_enabled(), bn_src)
    fn = y.grad_fn
    if fn is not None and getattr(fn, 'dy_pl_ok', False):
        y._pylc_dy_pl = True          # the BatchNorm that consumes y (its ONLY consumer, layers.conv_bn) may hand dy back as fp16 planes
    return y


class ConvTranspose2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(cin, cout, 2, stride=2) on the conv kernels: forward = the data gradient of the 2x2 / stride-2 conv whose
    weight is the same tensor ([cin, cout, 2, 2] read as [Cout', Cin', kh, kw]); backward: dx = that conv's forward on dy, dw = its
    weight gradient (x in the role of dy).  Bias added / reduced by the BatchNorm row-slab kernels."""

    @staticmethod
    def forward(ctx, x, w, bias):
        L.init()
        x = as_nhwc(x)
        b, cin, h, wd = x.shape
        if tuple(w.shape[:1] + w.shape[2:]) != (cin, 2, 2) or not w.permute(0, 2, 3, 1).is_contiguous() or cin % 4 or w.shape[1] % 4:
            raise L.PylcError('conv_transpose2x2: weight must be [Cin, Cout, 2, 2] with KRSC memory, channels multiples of 4')
        cout = w.shape[1]
        # the "forward conv" this is the dgrad of: input [B, cout, 2h, 2w] -> output [B, cin, h, w], 2x2 stride 2
        y = empty_nhwc(b, cout, 2 * h, 2 * wd, x.device)
        d = _conv_desc(y, cout, cin, 2, 2, 2, 0, 1, cout, pitch_of(x))
        xa = wa = None
        if ranges_needed():
            xa, wa = amax_of(x), weight_amax(w)
            d.dy_amax, d.w_amax = ptr(xa), ptr(wa)
        wt = torch.empty((cout, 4, cin), device=x.device, dtype=torch.float32)
        st = stream()
        check(lib.pylc_weight_transpose(ptr(w), ptr(wt), cin, 4, cout, st))
        check(lib.pylc_conv2d_dgrad(C.byref(d), ptr(x), ptr(wt), ptr(y), 0, st))
        if bias is not None:
            one = torch.ones(cout, device=x.device)
            m = b * 4 * h * wd
            check(lib.pylc_bn_apply(ptr(y), cout, ptr(one), ptr(bias), None, 0, ptr(y), cout, m, cout, 0, None, st))       # y*1 + bias, in place
        ctx.save_for_backward(x, w)
        ctx.ranges = (xa, wa)
        ctx.has_bias = bias is not None
        ctx.b_param = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = as_nhwc(dy)
        b, cin, h, wd = x.shape
        cout = w.shape[1]
        st = stream()
        d = _conv_desc(dy, cout, cin, 2, 2, 2, 0, 1, pitch_of(dy), cin)
        xa, wa = ctx.ranges
        dya = None
        if ranges_needed():
            dya = amax_of(dy)
            d.x_amax, d.w_amax, d.dy_amax = ptr(dya), ptr(wa), ptr(xa)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(b, cin, h, wd, x.device)
            check(lib.pylc_conv2d_fwd(C.byref(d), ptr(dy), ptr(w), None, ptr(dx), st))
        if ctx.needs_input_grad[1]:
            d.y_pitch = pitch_of(x)
            nbytes = lib.pylc_conv2d_wgrad_workspace(C.byref(d))
            ws = _ws(nbytes, x.device)
            tgt = _grad_target(w)
            dw = tgt if tgt is not None else torch.empty((cin, 2, 2, cout), device=x.device).permute(0, 3, 1, 2)
            check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(dy), ptr(x), ptr(dw), None, ptr(ws), nbytes, st))
            dw = _deliver_grad(w, dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            m = dy.shape[0] * dy.shape[2] * dy.shape[3]
            sums = torch.empty(2 * cout, device=x.device)
            wsb = torch.empty(lib.pylc_bn_workspace_floats(m, cout), device=x.device)
            check(lib.pylc_bn_stats(ptr(dy), m, cout, pitch_of(dy), ptr(sums), ptr(wsb), st))
            tgt = _grad_target(ctx.b_param)
            if tgt is not None:
                tgt.copy_(sums[:cout])
                db = _deliver_grad(ctx.b_param, tgt)
            else:
                db = sums[:cout].clone()
        return dx, dw, db


def conv_transpose2x2(x, w, bias=None):
    return ConvTranspose2x2Fn.apply(x, w, bias)


eval_plane_convs = [0]      # diagnostics: fused inference convs that ran on fp16-plane tensors (conv_bn_act_eval_planes)


def conv_bn_act_eval_planes(x, w, bias, stride, pad, dil, coef, coef_ranges, residual, relu, out_planes, into=None):
    """conv_bn_act_eval on fp16-PLANE tensors (pylc_conv2d_fwd_bnact_ex): x is (or is converted to) a planes tensor, the residual is read
    as planes or fp32, the result leaves as planes (out_planes: every consumer reads that format) or fp32.  A planes tensor of the
    inference path carries two device scalars -- the bound it was SCALED with (the `_pylc_pl` marker, what a consumer needs to undo the
    scale) and its TRUE maximum (the `_pylc_amax` tag, max-accumulated by the producing epilogue, what the consumer's own bound starts
    from): an eval-mode net has no batch statistics to re-anchor the bounds, and a bound derived from the previous bound would grow by
    ~2^8 per layer.  Returns None when the launch is not eligible (the caller takes the fp32 form)."""
    cout, cin_w, r, s = w.shape
    planes = getattr(w, '_pylc_planes', None)
    if not is_planes(x):
        x = as_nhwc(x)
    b, cin, h, wd = x.shape
    oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
    if (planes is None or cin != cin_w or not conv_takes_planes(w, b * h * wd, b * oh * ow) or cout % 8 or not planes_ok(cout, b * oh * ow)
            or not w.permute(0, 2, 3, 1).is_contiguous()):
        return None
    if residual is not None and not is_planes(residual):
        residual = as_nhwc(residual)
        if tuple(residual.shape) != (b, cout, oh, ow) or pitch_of(residual) != cout:
            return None
    if not is_planes(x):
        x_true = amax_of(x)                       # (a valid tag, or one read pass: the stem / a pooled tensor)
        x = to_planes(x, x_true)
    else:
        x_true = amax_of(x)                       # the producer's true maximum if it left one, else the scale bound itself
    yp = cout
    if into is not None:                  # fp32 result into channels [c0, c0 + Cout) of a caller-owned concat buffer (aspp.py:80, decoder.py:47)
        buf, c0 = into[0][0], into[1]
        if tuple(buf.shape[2:]) != (oh, ow) or buf.shape[0] != b or c0 % 4 or c0 + cout > buf.shape[1] or residual is not None:
            raise L.PylcError('conv_bn_act_eval into=: slice [%d, %d) does not fit the %s buffer' % (c0, c0 + cout, tuple(buf.shape)))
        y, yp, out_planes = buf[:, c0:c0 + cout], pitch_of(buf), False
    else:
        y = empty_nhwc(b, cout, oh, ow, x.device)
    d = _conv_desc(x, cin, cout, r, s, stride, pad, dil, cin, yp)
    d.x_fmt = 1
    w_amax = weight_amax(w)
    d.x_amax, d.w_amax, d.w_planes = ptr(planes_amax(x)), ptr(w_amax), ptr(planes[0])
    d.w_planes_fmt = filter_planes_fmt(planes)
    true_amax = amax_slot(x.device)
    bound = None
    if out_planes:
        bound = amax_slot(x.device)
        d.out_fmt, d.out_bound = nplanes(), ptr(bound)
    ep = L.FwdEp()
    ep.scale, ep.shift = ptr(coef[:cout]), ptr(coef[cout:])
    ep.scale_amax, ep.shift_amax = ptr(coef_ranges[0:1]), ptr(coef_ranges[1:2])
    ep.x_true_amax, ep.relu, ep.amax_out = ptr(x_true), int(relu), ptr(true_amax)
    keep = [x_true, w_amax, coef_ranges]
    if residual is not None:
        ep.residual = ptr(residual)
        if is_planes(residual):
            ep.res_fmt, ep.res_scale_bound = nplanes(), ptr(planes_amax(residual))
        res_true = amax_of(residual)
        ep.res_amax = ptr(res_true)
        keep.append(res_true)
    ev = None
    if _core._timer is not None and _core._timer.inference and cout > 64:
        ev = _core._timer.bracket(2.0 * b * oh * ow * cout * r * s * cin, 1, 'fwd_bnact%dx%d' % (r, s), 4.0 * (b * h * wd * cin + w.numel() + b * oh * ow * cout))
        ev[0].record()
    check(lib.pylc_conv2d_fwd_bnact_ex(C.byref(d), ptr(x), ptr(w), ptr(bias), C.byref(ep), ptr(y), stream()))
    if ev is not None:
        ev[1].record()
    eval_plane_convs[0] += 1
    if out_planes:
        mark_planes(y, bound)
    tag_amax(y, true_amax)                        # (after mark_planes, which tags with the bound)
    return y


def conv_bn_act_eval(x, w, bias, stride, pad, dil, running_mean, running_var, gamma, beta, eps, residual=None, relu=False, into=None, coef=None,
                     coef_ranges=None, out_planes=False):
    """Inference only (no autograd): act(BN_eval(conv(x)) (+ residual)) with the BatchNorm coefficients, the residual add and
    the ReLU applied in the conv epilogue -- bit-identical to conv2d followed by bn_act(training=False), one pass less.
    into = ([buffer], c0): write the result into channels [c0, c0 + Cout) of that NHWC concat buffer (aspp.py:80, decoder.py:47).
    coef_ranges (int32[2]: float bits of max|scale|, max|shift|) switches the fp16-plane form on where the launch is eligible
    (conv_bn_act_eval_planes); out_planes: the caller states that every consumer of the result reads planes."""
    L.init()
    if (coef is not None and coef_ranges is not None and ranges_needed() and _runtime.eval_planes and not _runtime.no_planes
            and (into is None or is_planes(x))):
        y = conv_bn_act_eval_planes(x, w, bias, stride, pad, dil, coef, coef_ranges, residual, relu, out_planes, into)
        if y is not None:
            return y
    x = as_nhwc(x)
    if residual is not None:
        residual = as_nhwc(residual)
    cout, cin_w, r, s = w.shape
    cin = x.shape[1]
    w_k = w
    if cin_w % 4 != 0 and cin == _r4(cin_w):
        w_k = _padded_stem_filter(w, cin)          # the thin-input stem (resnet.py:72, unet.py:111 first block): x is the 4-channel pack
    elif cin != cin_w or cin_w % 4 or not w.permute(0, 2, 3, 1).is_contiguous():
        raise L.PylcError('conv_bn_act_eval: needs a KRSC filter with Cin % 4 == 0 matching the input')
    b, _, h, wd = x.shape
    oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
    if into is not None:
        buf, c0 = into[0][0], into[1]
        if tuple(buf.shape[2:]) != (oh, ow) or buf.shape[0] != b or c0 % 4 or cout % 4 or c0 + cout > buf.shape[1] or residual is not None:
            raise L.PylcError('conv_bn_act_eval into=: slice [%d, %d) does not fit the %s buffer' % (c0, c0 + cout, tuple(buf.shape)))
        y, yp = buf[:, c0:c0 + cout], pitch_of(buf)
    else:
        yp = _r4(cout)
        y = empty_nhwc(b, cout, oh, ow, x.device, yp)
    d = _conv_desc(x, cin, cout, r, s, stride, pad, dil, pitch_of(x), yp)
    amax = None
    keep = None
    if ranges_needed():
        keep = (amax_of(x), weight_amax(w))
        d.x_amax, d.w_amax = ptr(keep[0]), ptr(keep[1])
        planes = getattr(w, '_pylc_planes', None)
        if planes is not None and w_k is w:
            d.w_planes, d.w_planes_fmt = ptr(planes[0]), filter_planes_fmt(planes)
        amax = amax_slot(x.device)
    st = stream()
    if coef is None:          # (coef: [scale | shift] a caller computed once for this set of weights, layers.BatchNorm2d.eval_coeffs)
        coef = torch.empty(2 * cout, device=x.device)
        check(lib.pylc_bn_eval_coeffs(ptr(running_mean), ptr(running_var), ptr(gamma), ptr(beta), eps, cout, ptr(coef[:cout]), ptr(coef[cout:]), st))
    res = None
    if residual is not None:
        res = as_nhwc(residual)
        if tuple(res.shape) != tuple(y.shape) or pitch_of(res) != yp:
            raise L.PylcError('conv_bn_act_eval: the residual must have the output\'s shape and pitch')
    ev = None
    if _core._timer is not None and _core._timer.inference and cout > 64 and cin % 8 == 0:
        ev = _core._timer.bracket(2.0 * b * oh * ow * cout * r * s * cin, 1, 'fwd_bnact%dx%d' % (r, s), 4.0 * (b * h * wd * cin + w.numel() + b * oh * ow * cout))
        ev[0].record()
    check(lib.pylc_conv2d_fwd_bnact(C.byref(d), ptr(x), ptr(w_k), ptr(bias), ptr(coef[:cout]), ptr(coef[cout:]), ptr(res), int(relu), ptr(y),
                                    ptr(amax), st))
    if ev is not None:
        ev[1].record()
    if amax is not None:
        tag_amax(y, amax)
    return y


__all__ = [n for n in dir() if not n.startswith('__')]      # everything, underscore helpers included: the package re-exports it (pylc_amd/ops/__init__.py)
