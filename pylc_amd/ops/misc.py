"""ReLU, dropout, pooling, concat, bilinear resize, global average pool, image pack, MultiLoss."""
from . import _core
from ._core import *      # noqa: F401,F403  (layout / planes / range / stream helpers, lib bindings, torch)


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L.init()
        x = as_nhwc(x)
        b, c, h, w = x.shape
        out = empty_nhwc(b, c, h, w, x.device)
        check(lib.pylc_relu_fwd(ptr(x), pitch_of(x), ptr(out), c, b * h * w, c, stream()))
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = as_nhwc(dout)
        b, c, h, w = out.shape
        dx = empty_nhwc(b, c, h, w, out.device)
        check(lib.pylc_relu_bwd(ptr(dout), pitch_of(dout), ptr(out), c, ptr(dx), c, b * h * w, c, stream()))
        return dx


def relu(x):
    return ReluFn.apply(x)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        L.init()
        x = as_nhwc(x)
        b, c, h, w = x.shape
        out = empty_nhwc(b, c, h, w, x.device)
        check(lib.pylc_dropout(ptr(x), pitch_of(x), ptr(out), c, b * h * w, c, p, seed, stream()))
        ctx.cfg = (p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, seed = ctx.cfg
        dout = as_nhwc(dout)
        b, c, h, w = dout.shape
        dx = empty_nhwc(b, c, h, w, dout.device)
        check(lib.pylc_dropout(ptr(dout), pitch_of(dout), ptr(dx), c, b * h * w, c, p, seed, stream()))
        return dx, None, None


def dropout(x, p, seed):
    out = DropoutFn.apply(x, p, seed)
    if ranges_needed() and 0.0 <= p < 1.0:
        binades = 0
        while (1 << binades) * (1.0 - p) < 1.0:      # 1 / (1 - p) <= 2^binades
            binades += 1
        inherit_amax(out, x, binades)
    return out


# ----------------------------------------------------------------------------------------------
# pooling / resize
# ----------------------------------------------------------------------------------------------
class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad, link=None, planes_bound=None):
        L.init()
        x = as_nhwc(x)
        if pitch_of(x) != x.shape[1]:
            x = x.contiguous(memory_format=torch.channels_last)
        b, c, h, w = x.shape
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        y = empty_nhwc(b, c, oh, ow, x.device)
        need_idx = ctx.needs_input_grad[0]
        idx = torch.empty((b, oh, ow, c), device=x.device, dtype=torch.uint8) if need_idx else None
        if planes_bound is not None:       # the pooled tensor as fp16 planes (its only reader is a conv that takes them)
            check(lib.pylc_maxpool_fwd_planes(ptr(x), ptr(y), pstride(b * oh * ow, c), nplanes(), ptr(planes_bound), ptr(idx), b, h, w, c, k, stride, pad,
                                              oh, ow, stream()))
        else:
            check(lib.pylc_maxpool_fwd(ptr(x), ptr(y), ptr(idx), b, h, w, c, k, stride, pad, oh, ow, stream()))
        ctx.save_for_backward(idx)
        ctx.cfg = (b, c, h, w, k, stride, pad, oh, ow)
        ctx.link = link if need_idx else None
        if ctx.link is not None:
            link.pool_armed = True
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        b, c, h, w, k, stride, pad, oh, ow = ctx.cfg
        dy = as_nhwc(dy)
        if pitch_of(dy) != c:
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = empty_nhwc(b, c, h, w, dy.device)
        link = ctx.link
        crop = None
        if link is not None:
            link.pool_armed = False
            crop, link.crop = link.crop, None
        if crop is not None:        # the skip connection's gradient (centre crop) is summed in the same pass
            g, c0, h0, w0 = crop
            check(lib.pylc_maxpool_bwd_add(ptr(dy), ptr(idx), ptr(dx), b, h, w, c, k, stride, pad, oh, ow, g.data_ptr() + 4 * c0, pitch_of(g),
                                           h0, w0, g.shape[2], g.shape[3], stream()))
        else:
            check(lib.pylc_maxpool_bwd(ptr(dy), ptr(idx), ptr(dx), b, h, w, c, k, stride, pad, oh, ow, stream()))
        return dx, None, None, None, None, None


def maxpool(x, k, stride, pad=0, link=None, out_planes=False):
    """out_planes: the pooled tensor has ONE reader, a conv that takes fp16 planes (the U-Net's next block): written as planes directly."""
    b, c, h, w = x.shape
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    if (out_planes and (torch.is_grad_enabled() or _runtime.eval_planes) and ranges_needed() and not _runtime.no_planes and not is_planes(x) and c % 8 == 0
            and planes_ok(c, b * oh * ow) and b * oh * ow >= _core.PLANES_MIN_PIXELS and _runtime.pool_planes):
        bound = amax_of(x)                 # the maxima are bounded by the input's range
        return mark_planes(MaxPoolFn.apply(x, k, stride, pad, link, bound), bound)
    y = MaxPoolFn.apply(x, k, stride, pad, link)
    return inherit_amax(y, x) if ranges_needed() else y


class CropConcatFn(torch.autograd.Function):
    """U-Net's `torch.cat([up, center_crop(bridge)], 1)` (unet.py:145-152) without the copies that can be avoided: `up` was
    written into the leading channels of the concat buffer by its conv (conv2d(out=[buffer])); the crop of `bridge` is copied
    behind it.  Backward: up's gradient is a channel slice (a view) of the buffer's gradient; the bridge's gradient is
    non-zero only inside the crop window, so when a max-pool also reads `bridge` (always, in the U-Net) it is parked on the
    shared link and summed by the pool's backward kernel -- no zero-padded tensor, no autograd add."""

    @staticmethod
    def forward(ctx, up, bridge, holder, link):
        L.init()
        buf = holder[0]
        bridge = as_nhwc(bridge)
        b, cu, th, tw = up.shape
        cb, hh, ww = bridge.shape[1:]
        if (tuple(buf.shape) != (b, cu + cb, th, tw) or pitch_of(buf) != cu + cb or up.data_ptr() != buf.data_ptr() or cu % 4 or cb % 4
                or th > hh or tw > ww):
            raise L.PylcError('crop_concat: `up` must be the leading channels of the [B, C_up + C_bridge, h, w] buffer')
        h0, w0 = (hh - th) // 2, (ww - tw) // 2
        check(lib.pylc_crop_copy(ptr(bridge), pitch_of(bridge), hh, ww, h0, w0, buf.data_ptr() + 4 * cu, cu + cb, b, th, tw, cb, stream()))
        ctx.geom = (cu, cb, hh, ww, h0, w0)
        ctx.link = link if ctx.needs_input_grad[1] else None
        return buf

    @staticmethod
    def backward(ctx, dy):
        cu, cb, hh, ww, h0, w0 = ctx.geom
        dy = as_nhwc(dy)
        d_up = dy[:, :cu] if ctx.needs_input_grad[0] else None
        d_bridge = None
        if ctx.needs_input_grad[1]:
            link = ctx.link
            if link is not None and link.pool_armed and link.crop is None:
                link.crop = (dy, cu, h0, w0)
            else:
                d_bridge = zeros_nhwc(dy.shape[0], cb, hh, ww, dy.device)
                d_bridge[:, :, h0:h0 + dy.shape[2], w0:w0 + dy.shape[3]] = dy[:, cu:]
        return d_up, d_bridge, None, None


def crop_concat(up, bridge, holder, link=None):
    return CropConcatFn.apply(up, bridge, holder, link)


class UpCatPlanesFn(torch.autograd.Function):
    """U-Net up path, `torch.cat([upsample_x2(z), center_crop(bridge)], 1)` (unet.py:135-152), written in ONE pass as the fp16-plane tensor the
    block's first conv reads (pylc_upsample2_crop_concat_planes): no fp32 concat buffer, no range pass, no conversion.  Backward: z's gradient
    is the bilinear backward of the first C1 channels of the concat gradient, the bridge's gradient is handled as in CropConcatFn (summed by the
    max-pool backward through the shared link)."""

    @staticmethod
    def forward(ctx, z, bridge, bound, link):
        L.init()
        z, bridge = as_nhwc(z), as_nhwc(bridge)
        b, c1, h, w = z.shape
        c2, hh, ww = bridge.shape[1:]
        oh, ow = 2 * h, 2 * w
        out = empty_nhwc(b, c1 + c2, oh, ow, z.device)
        check(lib.pylc_upsample2_crop_concat_planes(ptr(z), pitch_of(z), b, h, w, c1, ptr(bridge), pitch_of(bridge), hh, ww, c2, ptr(out),
                                                    pstride(b * oh * ow, c1 + c2), nplanes(), ptr(bound), stream()))
        ctx.geom = (b, c1, h, w, c2, hh, ww, (hh - oh) // 2, (ww - ow) // 2)
        ctx.link = link if ctx.needs_input_grad[1] else None
        return out

    @staticmethod
    def backward(ctx, dy):
        b, c1, h, w, c2, hh, ww, h0, w0 = ctx.geom
        dy = as_nhwc(dy)
        oh, ow = 2 * h, 2 * w
        dz = d_bridge = None
        if ctx.needs_input_grad[0]:
            dz = empty_nhwc(b, c1, h, w, dy.device)
            tmp = torch.empty(lib.pylc_bilinear_bwd_workspace(b, w, c1, oh) // 4, device=dy.device, dtype=torch.float32)
            amax = torch.empty(1, dtype=torch.int32, device=dy.device) if ranges_needed() and _runtime.fused_grad_ranges else None
            check(lib.pylc_bilinear_bwd_separable(ptr(dy), pitch_of(dy), ptr(dz), c1, b, h, w, c1, oh, ow, ptr(tmp), ptr(amax), stream()))
            if amax is not None:
                tag_amax(dz, amax)          # the 1x1 conv's backward reads dz: its range comes from the pass that wrote it
        if ctx.needs_input_grad[1]:
            link = ctx.link
            if link is not None and link.pool_armed and link.crop is None:
                link.crop = (dy, c1, h0, w0)
            else:
                d_bridge = zeros_nhwc(b, c2, hh, ww, dy.device)
                d_bridge[:, :, h0:h0 + oh, w0:w0 + ow] = dy[:, c1:]
        return dz, d_bridge, None, None


def upsample2_crop_concat(z, bridge, link=None):
    """cat([upsample_x2_bilinear(z), center_crop(bridge)], 1) as one fp16-plane tensor (training graphs and plane-tensor inference with ranged
    arithmetic, channel counts that are multiples of 8); None when that form does not apply -- the caller then uses bilinear(into=) +
    crop_concat."""
    c1, c2 = z.shape[1], bridge.shape[1]
    pixels = z.shape[0] * 4 * z.shape[2] * z.shape[3]
    if not ((torch.is_grad_enabled() or _runtime.eval_planes) and ranges_needed() and not _runtime.no_planes and c1 % 8 == 0 and c2 % 8 == 0 and planes_ok(c1 + c2, pixels)
            and pixels >= _core.PLANES_MIN_PIXELS and not is_planes(z) and _runtime.upcat_planes):
        return None
    bound = torch.maximum(amax_of(z), amax_of(bridge))          # float bit patterns of non-negative values: integer order = float order
    out = UpCatPlanesFn.apply(z, bridge, bound, link)
    return mark_planes(out, bound)


class ConcatSlicesFn(torch.autograd.Function):
    """torch.cat(parts, 1) (aspp.py:80, decoder.py:47) without the copy: every part was WRITTEN into its channel slice of one NHWC
    buffer by the kernel that produced it (bn_act / bilinear `into=`); the "concat" is the buffer.  Backward: each part's gradient is a
    channel-slice view of the buffer's gradient."""

    @staticmethod
    def forward(ctx, holder, *parts):
        buf = holder[0]
        c0 = 0
        offs = []
        for p in parts:
            if p.data_ptr() != buf.data_ptr() + 4 * c0 or tuple(p.shape[2:]) != tuple(buf.shape[2:]) or pitch_of(p) != pitch_of(buf):
                raise L.PylcError('concat_slices: part at channel %d is not that slice of the buffer' % c0)
            offs.append((c0, p.shape[1]))
            c0 += p.shape[1]
        if c0 != buf.shape[1]:
            raise L.PylcError('concat_slices: the parts cover %d of %d channels' % (c0, buf.shape[1]))
        ctx.offs = offs
        return buf

    @staticmethod
    def backward(ctx, dy):
        dy = as_nhwc(dy)
        return (None,) + tuple(dy[:, c0:c0 + c] for c0, c in ctx.offs)


def concat_slices(holder, parts):
    out = ConcatSlicesFn.apply(holder, *parts)
    if ranges_needed():
        tags = [getattr(t, '_pylc_amax', None) for t in parts]
        if all(tg is not None and tg[1] == t._version for tg, t in zip(tags, parts)):
            a = tags[0][0]
            for tg in tags[1:]:
                a = torch.maximum(a, tg[0])
            tag_amax(out, a)
    return out


class BilinearFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=True) to an explicit output size."""

    @staticmethod
    def forward(ctx, x, oh, ow, into=None):
        L.init()
        x = as_nhwc(x)
        b, c, h, w = x.shape
        cp = pitch_of(x)
        cc = _r4(c)
        if cc > cp:
            raise L.PylcError('bilinear: channel count %d needs a pitch >= %d' % (c, cc))
        if into is not None:               # channels [c0, c0 + c) of a concat buffer: into = ([buffer], c0)
            buf, c0 = into[0][0], into[1]
            if tuple(buf.shape[2:]) != (oh, ow) or buf.shape[0] != b or c0 % 4 or c % 4 or c0 + c > buf.shape[1]:
                raise L.PylcError('bilinear into=: slice [%d, %d) does not fit the %s buffer' % (c0, c0 + c, tuple(buf.shape)))
            y, yp = buf[:, c0:c0 + c], pitch_of(buf)
        else:
            y, yp = empty_nhwc(b, c, oh, ow, x.device, cc), cc
        check(lib.pylc_bilinear_fwd(ptr(x), cp, ptr(y), yp, b, h, w, cc, oh, ow, stream()))
        ctx.cfg = (b, c, h, w, oh, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, c, h, w, oh, ow = ctx.cfg
        dy = as_nhwc(dy)
        cc = _r4(c)
        if pitch_of(dy) < cc:
            t = zeros_nhwc(b, c, oh, ow, dy.device, cc)
            t.copy_(dy)
            dy = t
        dx = empty_nhwc(b, c, h, w, dy.device, cc)
        if oh >= 2 * h and ow >= 2 * w:      # up-sampling: one axis at a time (10 + 10 instead of 100 candidate taps per element at x4)
            tmp = torch.empty(lib.pylc_bilinear_bwd_workspace(b, w, cc, oh) // 4, device=dy.device, dtype=torch.float32)
            amax = torch.empty(1, dtype=torch.int32, device=dy.device) if ranges_needed() and _runtime.fused_grad_ranges else None
            check(lib.pylc_bilinear_bwd_separable(ptr(dy), pitch_of(dy), ptr(dx), cc, b, h, w, cc, oh, ow, ptr(tmp), ptr(amax), stream()))
            if amax is not None:
                tag_amax(dx, amax)
        else:
            check(lib.pylc_bilinear_bwd(ptr(dy), pitch_of(dy), ptr(dx), cc, b, h, w, cc, oh, ow, stream()))
        return dx, None, None, None


def bilinear(x, oh, ow, into=None):
    y = BilinearFn.apply(x, oh, ow, into)
    return inherit_amax(y, x) if ranges_needed() else y       # interpolation weights are a convex combination


class GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res_link=None):
        L.init()
        b, c, h, w = x.shape
        y = empty_nhwc(b, c, 1, 1, x.device)
        if is_planes(x) and c % 8 == 0 and _runtime.gap_planes:
            # the backbone's last BatchNorm left fp16 planes for the atrous convs: pool them as they are (same bits as converting first)
            check(lib.pylc_gap_fwd_planes(ptr(x), pstride(b * h * w, c), nplanes(), ptr(planes_amax(x)), ptr(y), b, h * w, c, stream()))
        else:
            x = as_nhwc(x)
            if pitch_of(x) != x.shape[1]:
                x = x.contiguous(memory_format=torch.channels_last)
            check(lib.pylc_gap_fwd(ptr(x), ptr(y), b, h * w, c, stream()))
        ctx.cfg = (b, c, h, w)
        ctx.res_link = res_link if (res_link is not None and ctx.needs_input_grad[0]) else None
        if ctx.res_link is not None:
            res_link.pending += 1           # one more backward node that adds its part of x's gradient into the shared buffer
        return y

    @staticmethod
    def backward(ctx, dy):
        b, c, h, w = ctx.cfg
        dy = dy.reshape(b, c).contiguous()
        link = ctx.res_link
        sink = _link_sink(link)
        dx = sink if sink is not None else empty_nhwc(b, c, h, w, dy.device)
        check(lib.pylc_gap_bwd_acc(ptr(dy), ptr(dx), b, h * w, c, 1 if sink is not None else 0, stream()))
        if link is not None:
            link.pending -= 1
            if link.pending > 0:            # the convs that read x follow: their dgrads accumulate into the same buffer
                link.buf, dx = dx, None
            else:
                link.buf = None
        return dx, None


def global_avg_pool(x, res_link=None):
    return GapFn.apply(x, res_link)


# ----------------------------------------------------------------------------------------------
# image ingest
# ----------------------------------------------------------------------------------------------
def image_pack(img, mean3, std3, denom=255.0):
    """Raw [B,1|3,H,W] 0..255 tiles -> normalised NHWC4 network input (Model.normalize_image + x3 stack): ((x - mean) / std) / denom."""
    L.init()
    b, c, h, w = img.shape
    img = img.contiguous()
    out = empty_nhwc(b, 4, h, w, img.device)
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    s = (C.c_float * 3)(*[float(v) for v in std3])
    u8 = img.dtype == torch.uint8          # tiles as stored in the database: normalise straight from bytes
    src = img if u8 else img.float()
    check(lib.pylc_image_pack_denom(ptr(src), int(u8), b, c, h, w, m, s, float(denom), ptr(out), stream()))
    return out


def pack_nchw(x, pitch):
    """Already-normalised NCHW fp32 -> NHWC with `pitch` channels (extra channels zero)."""
    L.init()
    b, c, h, w = x.shape
    x = x.contiguous()
    out = zeros_nhwc(b, pitch, h, w, x.device)
    check(lib.pylc_nchw_to_nhwc(ptr(x), ptr(out), pitch, b, h, w, c, stream()))
    return out


# ----------------------------------------------------------------------------------------------
# MultiLoss
# ----------------------------------------------------------------------------------------------
_shard_pairs = {}


def _shard_pair(b, dev):
    """[b, b^2] as a cached device tensor (exact in fp32 for any realistic per-rank tile count)."""
    key = (int(b), str(dev))
    if key not in _shard_pairs:
        _shard_pairs[key] = torch.tensor([float(b), float(b) * float(b)], device=dev)
    return _shard_pairs[key]


SHARD_PAIRS_MAX = 64        # pending pairs folded into one device scalar beyond this many (runs that never call Model.log)


def note_shard_pair(pair, world):
    """Remember the reduced [sum b, sum b^2] pair of ONE data-parallel loss exchange until the host next looks (check_equal_shards).  Every
    step's pair is kept -- a short final batch on some ranks that is followed by full-size steps before the next Model.log() must still be
    reported (the pair used to be overwritten, so only the last step before a log was ever checked).  No host synchronisation
    here: the pairs are views of the steps' statistics messages; past SHARD_PAIRS_MAX of them they are folded on the device into the
    largest violation seen."""
    pend = _runtime.shard_check
    if pend is None:
        pend = _runtime.shard_check = {'pairs': [], 'world': world, 'worst': None}
    pend['pairs'].append(pair)
    pend['world'] = world
    if len(pend['pairs']) >= SHARD_PAIRS_MAX:
        _fold_shard_pairs(pend)


def _shard_violation(pairs, world):
    """max over steps of |world * sum b^2 - (sum b)^2| (zero iff every rank held the same count), with the offending (sum b, sum b^2)"""
    p = torch.stack([q.double() for q in pairs])
    viol = (p[:, 1] * world - p[:, 0] * p[:, 0]).abs()
    k = torch.argmax(viol)
    return torch.stack([viol[k], p[k, 0], p[k, 1]])


def _fold_shard_pairs(pend):
    v = _shard_violation(pend['pairs'], pend['world'])
    pend['pairs'] = []
    pend['worst'] = v if pend['worst'] is None else torch.where(v[0] > pend['worst'][0], v, pend['worst'])      # (no host read)


def check_equal_shards():
    """Raise if ANY data-parallel loss exchange since the last call saw different tile counts on different ranks (the pairs that rode on
    them, see MultiLossFn.forward / note_shard_pair).  One tiny D2H copy: called where the host reads the loss log anyway (Model.log)."""
    pend, _runtime.shard_check = _runtime.shard_check, None
    if pend is None:
        return
    if isinstance(pend, tuple):          # (pair, world): the one-pair form (kept for callers that set it directly)
        pend = {'pairs': [pend[0]], 'world': pend[1], 'worst': None}
    world = pend['world']
    cands = [] if pend['worst'] is None else [pend['worst']]
    if pend['pairs']:
        cands.append(_shard_violation(pend['pairs'], world))
    if not cands:
        return
    viol, sb, sb2 = max((tuple(float(x) for x in c.cpu().tolist()) for c in cands), key=lambda t: t[0])
    if viol > 0.5:
        raise RuntimeError('data-parallel ranks hold different batch sizes (sum b = %g, sum b^2 = %g over %d ranks in one of the steps since the '
                           'last check): SyncBN and the loss head need equal shards -- use a drop_last loader' % (sb, sb2, world))


class MultiLossFn(torch.autograd.Function):
    """Returns a [4] tensor (total, ce, dice, focal); only total carries gradient."""

    @staticmethod
    def forward(ctx, logits, target, class_weights, w_ce, w_dice, w_focal, group):
        L.init()
        logits = as_nhwc(logits)
        b, c, h, w = logits.shape
        n = b * h * w
        target = target.contiguous()
        if target.dtype != torch.int64 or tuple(target.shape) != (b, h, w):
            raise L.PylcError('target must be int64 [B,H,W] matching the logits')
        dev = logits.device
        st = stream()
        k = 3 + 3 * c
        # data parallel: two more floats ride on the statistics message -- this rank's tile count b and b^2 -- so that unequal shards are
        # DETECTED without a collective of their own (sum b^2 * world == (sum b)^2 iff all equal; Model.train checks the reduced pair at its
        # report interval).  A separate all_gather triggered by a rank-local condition would desynchronise the ranks' collective sequences
        # in exactly the case it is meant to catch.
        stats = torch.empty(k + (2 if group is not None else 0), device=dev)
        ws = torch.empty(lib.pylc_multiloss_workspace_floats(n, c), device=dev)
        check(lib.pylc_multiloss_stats(ptr(logits), pitch_of(logits), ptr(target), n, c, ptr(class_weights), ptr(stats), ptr(ws), st))
        n_global = float(n)
        if group is not None:
            stats[k:].copy_(_shard_pair(b, dev), non_blocking=True)
            _runtime.sync_all_reduce(stats, group)   # Dice / weighted CE are not shard-decomposable
            n_global = float(n) * dist.get_world_size(group)
            note_shard_pair(stats[k:], dist.get_world_size(group))
        losses = torch.empty(4, device=dev)
        check(lib.pylc_multiloss_finalize(ptr(stats), n_global, c, w_ce, w_dice, w_focal, ptr(losses), st))
        ctx.save_for_backward(logits, target, stats, class_weights)
        ctx.cfg = (n_global, w_ce, w_dice, w_focal, group)
        return losses

    @staticmethod
    def backward(ctx, dlosses):
        logits, target, stats, cw = ctx.saved_tensors
        n_global, w_ce, w_dice, w_focal, group = ctx.cfg
        b, c, h, w = logits.shape
        n = b * h * w
        # the loss is already the GLOBAL loss; each rank back-propagates its own pixels' share and the
        # gradient all-reduce SUMS the shares (pylc_amd/parallel.py)
        gs = dlosses[0:1].contiguous().float()
        cp = _r4(c)
        dl = empty_nhwc(b, c, h, w, logits.device, cp)
        amax = torch.empty(1, dtype=torch.int32, device=logits.device) if ranges_needed() and _runtime.fused_grad_ranges else None
        check(lib.pylc_multiloss_bwd(ptr(logits), pitch_of(logits), ptr(target), n, c, ptr(cw), ptr(stats), n_global,
                                     w_ce, w_dice, w_focal, ptr(gs), ptr(dl), cp, ptr(amax), stream()))
        if amax is not None:
            tag_amax(dl, amax)              # (read by a conv backward directly when the net has no logits up-sampling: the U-Net)
        return dl, None, None, None, None, None, None


def multiloss(logits, target, class_weights, w_ce, w_dice, w_focal, group=None):
    return MultiLossFn.apply(logits, target, class_weights, w_ce, w_dice, w_focal, group)


__all__ = [n for n in dir() if not n.startswith('__')]      # everything, underscore helpers included: the package re-exports it (pylc_amd/ops/__init__.py)
