"""torch.autograd bindings of the HIP kernels (host-side plumbing only: shapes, buffers, streams).

Tensor convention: every activation is a 4-D tensor of logical shape [B, C, H, W] whose MEMORY is
NHWC (``channels_last``), possibly a channel slice of a wider buffer (pitch > C).  Conv weights are
logical [Cout, Cin, kh, kw] with KRSC memory.  Nothing here computes on the CPU or through ATen math
kernels; if libpylc_hip.so is missing, importing ``pylc_amd.lib`` already failed.
"""
import ctypes as C

import os
import time
import torch
import torch.distributed as dist

from . import lib as L
from .lib import lib, check, ptr, stream, ConvDesc, DwDesc
from .runtime import runtime as _runtime


# ----------------------------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------------------------
def empty_nhwc(b, c, h, w, device, pitch=None, dtype=torch.float32):
    pitch = c if pitch is None else pitch
    t = torch.empty((b, h, w, pitch), device=device, dtype=dtype).permute(0, 3, 1, 2)
    return t if pitch == c else t[:, :c]


def zeros_nhwc(b, c, h, w, device, pitch=None):
    pitch = c if pitch is None else pitch
    t = torch.zeros((b, h, w, pitch), device=device, dtype=torch.float32).permute(0, 3, 1, 2)
    return t if pitch == c else t[:, :c]


def pitch_of(t):
    """Channel pitch (floats between pixels) of an NHWC-memory tensor; raises if the layout is anything else."""
    b, c, h, w = t.shape
    if w > 1:
        p = t.stride(3)
    elif h > 1:
        p = t.stride(2)
    elif b > 1:
        p = t.stride(0)
    else:
        p = c
    ok = (c == 1 or t.stride(1) == 1) and (h == 1 or t.stride(2) == w * p) and (b == 1 or t.stride(0) == h * w * p) and p >= c
    if not ok:
        raise L.PylcError('tensor is not NHWC-in-memory: shape %s strides %s' % (tuple(t.shape), t.stride()))
    return p


# ----------------------------------------------------------------------------------------------
# fp16-plane tensors (include/pylc_hip.h "fp16 planes")
# ----------------------------------------------------------------------------------------------
# A planes tensor is carried through autograd as an ordinary float32 tensor of the logical [B, C, H, W] shape (NHWC memory, pitch ==
# C) whose BYTES hold the two fp16 planes [2][B*H*W][C] -- the same 4 bytes per element, so no second allocation, and autograd sees
# the dtype and shape it expects.  The marker attribute names the range bound the planes were scaled with.  Only kernels that know
# the format may touch the bytes: every other op goes through as_nhwc(), which converts back to fp32 (one pass, counted).
plane_conversions = [0, 0]      # [planes -> fp32 conversion passes, elements]: diagnostics (0 on the hot path)
planes_marked = [0]             # tensors produced (or re-marked in a backward) in the fp16-plane format: diagnostics


def is_planes(t):
    tag = getattr(t, '_pylc_pl', None)
    return tag is not None and tag[1] == t._version


def planes_amax(t):
    return t._pylc_pl[0]


def nplanes():
    """2 for the f16x3 arithmetic, 1 for plain fp16 operands (precision mode 3)."""
    return 1 if lib.pylc_get_conv_precision() == 3 else 2


def half_acts():
    """Precision mode 3 with ONE-PLANE fp16 tensors end to end (2 bytes per element): conv and depthwise outputs y, and the gradients the
    dgrads hand back, leave their kernels as a single fp16 plane (a planes tensor with nplanes() == 1) wherever the consumer reads that
    format -- BatchNorm (y, dout), the depthwise kernels (x, dy), the conv kernels (as before)."""
    return lib.pylc_get_conv_precision() == 3 and _runtime.half_acts


def half_dw():
    """half_acts() and the depthwise kernels take part (runtime.half_dw): a BatchNorm whose output a depthwise conv reads writes one fp16 plane."""
    return half_acts() and _runtime.half_dw


def planes_ok(c, pixels):
    """Can an activation of `c` channels x `pixels` pixels be kept as fp16 planes (16-byte rows per 8 channels, one plane below 2 GiB)?"""
    return lib.pylc_get_conv_precision() >= 2 and c % 8 == 0 and pixels * c * 2 < (1 << 31)


# Below this many input pixels (64 row tiles of 128: a quarter of the chip) a conv is launch-bound and gains nothing from the planes
# kernels; it keeps fp32 operands and the round-1 kernels.  This also keeps the 96x96 DeepLab golden fixtures on the kernels they
# were tuned against: their BatchNorms average over as few as 72 pixels and amplify a last-bit change of ONE early conv output
# into percents of the upstream gradients (tools/mode_sensitivity.py: f16x3 vs bf16x6, both fp32-grade, move the Xception fixture's
# backbone gradients by 4e-3 elementwise; swapping the first three convs for their bit-compatible planes kernels -- which differ from
# the small-grid fp32 kernels only in the MFMA shape -- by 2e-2).  The planes kernels themselves are pinned bit for bit against the
# fp32-operand kernels in tests/test_planes_gpu.py, and the full-size network tests run them.
PLANES_MIN_PIXELS = int(os.environ.get('PYLC_PLANES_MIN_PIXELS', '8192'))


def conv_takes_planes(w, pixels_in, pixels_out):
    """Will conv2d() run this filter on the fp16-plane kernels (conv_pl.hip / wgrad_pl.hip)?  Needs the prepared filter planes (flat
    arena) and channel counts the 16-byte plane rows allow.  Narrow convs (<= 64 output channels) take them too: on a 128-wide tile
    half the MFMAs multiply zeros, but those layers are bound by bytes and by the per-tile prologue / epilogue, which two blocks per
    CU overlap (measured: the 256x128 one-block kernel ran the K = 48 / 64 dgrads of layer1 and the decoder at 6-90 TFLOP/s)."""
    cout, cin, r, s_ = w.shape
    only = os.environ.get('PYLC_PLANES_ONLY')          # debug: "cin:cout:k,cin:cout:k,..." with * wildcards -- planes for these filters only
    if only and not any(all(p == '*' or int(p) == v for p, v in zip(pat.split(':'), (cin, cout, r))) for pat in only.split(',')):
        return False
    return (lib.pylc_get_conv_precision() >= 2 and getattr(w, '_pylc_planes', None) is not None and cin % 8 == 0 and cout % 4 == 0
            and pixels_in >= PLANES_MIN_PIXELS and planes_ok(cin, pixels_in) and not _runtime.no_planes)


def mark_planes(t, amax):
    planes_marked[0] += 1
    t._pylc_pl = (amax, t._version)
    tag_amax(t, amax)
    return t


def to_planes(x, amax=None):
    """fp32 NHWC tensor -> planes tensor (one pass); `amax`: device int32[1] range bound (default: the tensor's own range)."""
    L.init()
    x = as_nhwc(x)
    b, c, h, w = x.shape
    if not planes_ok(c, b * h * w):
        raise L.PylcError('to_planes: %d channels x %d pixels cannot be held as fp16 planes' % (c, b * h * w))
    if amax is None:
        amax = amax_of(x)
    out = empty_nhwc(b, c, h, w, x.device)
    m = b * h * w
    check(lib.pylc_to_planes(ptr(x), pitch_of(x), ptr(out), c, m * c, m, c, ptr(amax), nplanes(), stream()))
    return mark_planes(out, amax)


def from_planes(t):
    """planes tensor -> fp32 NHWC tensor (one pass)."""
    L.init()
    b, c, h, w = t.shape
    m = b * h * w
    out = empty_nhwc(b, c, h, w, t.device)
    check(lib.pylc_from_planes(ptr(t), c, m * c, ptr(out), c, m, c, ptr(planes_amax(t)), nplanes(), stream()))
    plane_conversions[0] += 1
    plane_conversions[1] += t.numel()
    tag_amax(out, planes_amax(t))
    return out


class FromPlanesFn(torch.autograd.Function):
    """planes -> fp32 inside a training graph (the gradient passes through unchanged: the producer's backward takes fp32)."""

    @staticmethod
    def forward(ctx, t, amax):
        if not is_planes(t):          # (should autograd hand the function a fresh alias of the tensor: restore the marker)
            t._pylc_pl = (amax, t._version)
        return from_planes(t)

    @staticmethod
    def backward(ctx, dy):
        return dy, None


def export_activation(t):
    """What a module hands to a caller that does not know the fp16-plane format (a public module boundary: ResNet101.forward
    without keep_planes): a planes tensor is a float32-TYPED tensor whose bytes are fp16 planes, so any foreign op -- a torch
    function, a forward hook, feature extraction -- would compute on reinterpreted bytes without an error.  Converts (one pass,
    differentiable); fp32 tensors pass through."""
    if not is_planes(t):
        return t
    if torch.is_grad_enabled() and t.requires_grad:
        return FromPlanesFn.apply(t, planes_amax(t))
    return from_planes(t)


def as_nhwc(t):
    """Return `t` as an fp32 tensor with NHWC memory (copying through torch only if an upstream op handed us another layout;
    converting if it is an fp16-plane tensor)."""
    if is_planes(t):
        return from_planes(t)
    defer = getattr(t, '_pylc_defer', None)
    if defer is not None and defer[4] == t._version:
        # a BatchNorm output whose apply pass was left to its (depthwise) consumer: anything else that reads it gets the applied values
        # (gradients do not flow through this copy -- the tensor has ONE designated consumer; this serves hooks and debugging)
        return from_planes(materialize_deferred(t))
    if t.dtype != torch.float32:
        t = t.float()
    try:
        pitch_of(t)
        return t
    except L.PylcError:
        out = empty_nhwc(*t.shape, device=t.device)
        out.copy_(t)
        return out


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 4) // 4 + 1, device=device, dtype=torch.float32)


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _r4(c):
    return (c + 3) & ~3


# ----------------------------------------------------------------------------------------------
# live kernel timing (bench.py's roofline leg)
# ----------------------------------------------------------------------------------------------
class KernelTimer:
    """HIP-event timing of the dominant kernel's launches on the stream they are launched on.

    Only launches that dispatch to the 256x128-tile kernel (pylc_amd/csrc/conv_igemm.hip dispatch_gg_p: gather_gemm_pp_kernel
    for the f16x3 arithmetic, gather_gemm_kernel<256,128,64,64,false,1> for bf16x6; stored Cout > 64, not the thin-input
    mode, >= 192 tiles of 256x128, reduction channels % 8 == 0) are bracketed; FLOPs are algorithmic fp32 FLOPs (2*M*N*K, every tap counted).  The kernel executes 3 (f16x3) or
    6 (bf16x6) 16-bit MFMA FLOPs per algorithmic FLOP, so its roofline is the dense 16-bit MFMA peak / 3 (or / 6)."""

    TERMS = {1: 6, 2: 3, 3: 1}

    def __init__(self, inference=False):
        self.records = []          # (start_event, end_event, flops, launches, kind)
        self.alg_bytes = 0.0       # algorithmic operand bytes (input + weights + output, each touched once)
        self.mode = lib.pylc_get_conv_precision()
        self.planes = self.mode >= 2 and not _runtime.no_planes
        self.inference = bool(inference)      # eval-mode nets: the fused conv + BatchNorm(+ residual + ReLU) launches (conv_bn_act_eval) are bracketed
        # the fp16-plane gather-GEMM (conv_pl.hip) in its two tile heights is what the conv forward / dgrad launches run when the
        # activations travel as planes; '*' = both tile heights (rocprof lists them as two rows) and the 3x3 halo variant: one kernel
        # family (conv_pl.hip), the same loop body, dispatched by shape
        self.KERNEL = (('gg_pl_kernel<%d,*,EP> + gg_plh_kernel<%d,EP> (pylc_conv2d_fwd_bnact_ex: plane tensors, conv + eval BatchNorm + residual + ReLU '
                        'in the epilogue)' % (((3 if self.mode == 2 else 1),) * 2) if (_runtime.eval_planes and self.planes) else
                        'gather_gemm_pp_kernel<..., %s> (pylc_conv2d_fwd_bnact: conv + eval BatchNorm + residual + ReLU in the epilogue)'
                        % ('ONE-plane fp16' if self.mode == 3 else 'f16x3')) if self.inference else
                       'gg_pl_kernel<%d,*> + gg_plh_kernel<%d>' % (((3 if self.mode == 2 else 1),) * 2) if self.planes else
                       'gather_gemm_pp_kernel<false,true,true,true,true,false>' if self.mode == 2 else
                       'gather_gemm_kernel<256,128,64,64,false,%d>' % self.mode)

    def bracket(self, flops, launches, kind, nbytes=0.0):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((a, b, flops, launches, kind))
        self.alg_bytes += nbytes
        return a, b

    def roofline(self, peak_16bit_tflops=2500.0):
        torch.cuda.synchronize()
        tot_ms = sum(a.elapsed_time(b) for a, b, _, _, _ in self.records)
        flops = sum(r[2] for r in self.records)
        launches = sum(r[3] for r in self.records)
        by = {}
        for a, b, f, n, kind in self.records:
            e = by.setdefault(kind, [0.0, 0.0, 0])
            e[0] += a.elapsed_time(b); e[1] += f; e[2] += n
        ach = flops / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        terms = self.TERMS.get(self.mode, 6)
        peak = peak_16bit_tflops / terms
        arith = ('3-term scaled fp16 split ("f16x3": a0b0 + 2^-11 (a1b0 + a0b1), cross terms in their own fp32 accumulator) '
                 'on v_mfma_f32_16x16x32_f16' if self.mode == 2 else
                 'plain fp16 operands (scaled per tensor), fp32 accumulation, on v_mfma_f32_16x16x32_f16' if self.mode == 3 else
                 '6-term bf16 split ("bf16x6") on v_mfma_f32_32x32x16_bf16')
        return {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                'kernel': self.KERNEL, 'launches': launches, 'avg_launch_ms': tot_ms / max(launches, 1),
                'kernel_time_ms_total': tot_ms, 'algorithmic_bytes_per_launch': self.alg_bytes / max(launches, 1),
                'note': ('achieved = algorithmic FLOP/s; arithmetic = %s: 16-bit operand accuracy (2^-11 per operand, fp32 accumulation), NOT '
                         'fp32-grade -- judged by the statistical parity bar (losses, argmax agreement, mIoU); peak = dense 16-bit MFMA peak '
                         '(2500 TFLOP/s)' % arith) if self.mode == 3 else
                        ('achieved = algorithmic fp32 FLOP/s; arithmetic = %s with fp32-grade accuracy (measured error vs fp64 '
                         'no larger than the fp32 FMA chain\'s), so peak = dense 16-bit MFMA peak (2500 TFLOP/s) / %d; executed '
                         'MFMA rate = %d x achieved; the exact-fp32 matrix pipe peaks at 157.3 TFLOP/s' % (arith, terms, terms)),
                'mfma_executed_tflops': terms * ach,
                'by_kind': {k: {'ms': v[0], 'tflops': v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 else 0.0, 'launches': v[2]}
                            for k, v in by.items()}}


_timer = None
bn_timing = None       # diagnostics (tools/bn_table.py): a list that receives (kind, M, C, bytes, start_event, end_event) per BatchNorm pass


def _bn_time(kind, m, c, nbytes):
    """Context manager bracketing one BatchNorm pass with HIP events when tools/bn_table.py has switched the table on."""
    if bn_timing is None:
        return _nullcontext()
    return _BnTimed(kind, m, c, nbytes)


class _BnTimed:
    def __init__(self, kind, m, c, nbytes):
        self.rec = [kind, m, c, nbytes, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]

    def __enter__(self):
        self.rec[4].record()

    def __exit__(self, *a):
        self.rec[5].record()
        bn_timing.append(tuple(self.rec))
        return False


def set_kernel_timer(t):
    global _timer
    _timer = t


def _is_dominant_tile(m, n_store, cin, taps):
    """Mirror of dispatch_gg_p in conv_igemm.hip: does this launch run the 256x128 8-wave split-arithmetic kernel?"""
    if lib.pylc_get_conv_precision() == 0 or n_store <= 64 or (cin == 4 and taps > 1):
        return False
    if lib.pylc_get_conv_precision() == 2 and cin % 8:
        return False
    return ((m + 255) // 256) * ((n_store + 127) // 128) >= 192


# ----------------------------------------------------------------------------------------------
# dense convolution
# ----------------------------------------------------------------------------------------------
def _conv_desc(x, cin, cout, r, s, stride, pad, dil, x_pitch, y_pitch):
    b, _, h, w = x.shape
    d = ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.R, d.S = b, h, w, cin, cout, r, s
    d.stride, d.pad, d.dil = stride, pad, dil
    d.OH, d.OW = conv_out_size(h, r, stride, pad, dil), conv_out_size(w, s, stride, pad, dil)
    d.x_pitch, d.y_pitch = x_pitch, y_pitch
    return d


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_side_streams = {}


def _runs_concurrently(cand, device):
    """True if work on `cand` executes while the current stream is busy, i.e. the two HIP streams sit on different
    hardware queues.  HIP multiplexes streams onto a few hardware queues in creation order, so a fresh stream can land on
    the compute stream's queue -- observed once RCCL had created its streams -- and would then serialise behind it."""
    with torch.cuda.stream(cand):                      # first use of a stream can take milliseconds (queue creation):
        torch.zeros(1, device=device)                  # keep that out of the timed part
    cand.synchronize()
    try:
        torch.cuda._sleep(40_000_000)                  # tens of ms of busy-wait on the current stream
    except (AttributeError, RuntimeError):
        return True                                    # cannot probe: take the stream as it is
    with torch.cuda.stream(cand):
        torch.zeros(1, device=device)
        ev = torch.cuda.Event()
        ev.record()
    ok = False
    t0 = time.perf_counter()
    while not ok and time.perf_counter() - t0 < 0.010:
        time.sleep(0.001)
        ok = ev.query()
    torch.cuda.synchronize(device)
    if os.environ.get('PYLC_DEBUG_STREAMS'):
        print('[pylc] side-stream candidate %s: %s' % (cand, 'concurrent' if ok else 'serialised behind the compute stream'), flush=True)
    return ok


_deferred_wgrad = {}     # device index -> [[conv backwards still to pass, closure that launches the held-back wgrad], ...] in launch order


def _defer_wgrad(device, fn, hold=1):
    """Hold a wgrad launch back until `hold` more conv backwards have STARTED on this device (each conv backward calls
    flush_deferred_wgrad first), or until sync_side_streams().  hold = 1: the wgrad starts beside the next conv's dgrad."""
    _deferred_wgrad.setdefault(torch.device(device).index, []).append([int(hold), fn])


def flush_deferred_wgrad(device, everything=False):
    held = _deferred_wgrad.get(torch.device(device).index)
    if not held:
        return
    keep = []
    for item in held:
        item[0] -= 1
        if everything or item[0] <= 0:
            item[1]()
        else:
            keep.append(item)
    held[:] = keep


def cu_masked_stream(device, n_cus, from_top=False):
    """A HIP stream whose kernels only occupy `n_cus` of the 256 compute units (pylc_stream_create_cu_mask), as a torch stream object.
    The HIP stream lives as long as the process (side streams are created once per device)."""
    L.init()
    h = C.c_void_p()
    with torch.cuda.device(device):
        check(lib.pylc_stream_create_cu_mask(int(n_cus), int(bool(from_top)), C.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=device)


def _side_stream(device):
    key = torch.device(device).index
    if key not in _side_streams:
        cands = []
        for _ in range(8):
            # runtime.wgrad_cus (PYLC_WGRAD_CUS): confine the wgrad stream to that many compute units, so that the HBM-bound passes of
            # the main stream keep the rest to themselves
            st = cu_masked_stream(device, _runtime.wgrad_cus) if _runtime.wgrad_cus else torch.cuda.Stream(device=device)
            cands.append(st)                          # keep the rejected ones alive so the next candidate is a new stream
            if _runs_concurrently(st, device):
                break
        _side_streams[key] = cands[-1]
    return _side_streams[key]


def side_stream_if_any(device):
    """The wgrad side stream of `device` if one has been created (None on CPU / before the first backward)."""
    if not torch.device(device).type == 'cuda':
        return None
    return _side_streams.get(torch.device(device).index)


_side_keep = {}      # device index -> tensors of the main stream that kernels queued on the side stream still read


def _keep_for_side(device, *tensors):
    """Keep main-stream tensors alive while side-stream kernels read them; sync_side_streams() lets go of them once the main
    stream has been told to wait for the side stream, so their blocks return to the allocator in stream order.
    (Tensor.record_stream would do, but it makes the blocks reusable only when the side stream's events have COMPLETED: with
    the host a step ahead of the GPU nothing of the previous step is reusable yet and every step calls hipMalloc for its conv
    inputs and gradients again -- measured: 281 hipMalloc calls inside bench.py's 10 timed steps, 74-95 GB reserved for models
    that peak at 15-43 GB.)"""
    keep = _side_keep.setdefault(torch.device(device).index, [])
    keep.extend(t for t in tensors if t is not None)
    if len(keep) > 8192:                 # a caller that never synchronises: do it for them rather than grow without bound
        sync_side_streams()


def sync_side_streams():
    """Make the current stream wait for everything queued on the wgrad side stream (before the optimiser / a gradient
    all-reduce reads the arena), then release the tensors kept alive for it."""
    for idx in list(_deferred_wgrad):
        flush_deferred_wgrad(torch.device('cuda', idx), everything=True)
    for st in _side_streams.values():
        torch.cuda.current_stream().wait_stream(st)
    for keep in _side_keep.values():
        keep.clear()


def _grad_target(param):
    """Arena-backed gradient view for `param` if the optimiser registered one, else None."""
    return getattr(param, '_pylc_grad', None)


def _deliver_grad(param, g):
    """Hand a parameter gradient back.  With an arena view registered the kernel already wrote into it:
    publish it as .grad and tell autograd there is nothing to accumulate."""
    if _grad_target(param) is not None:
        if param.grad is None or param.grad.data_ptr() != g.data_ptr():
            param.grad = g
        arena = getattr(param, '_pylc_arena', None)
        if arena is not None:
            a = arena()
            if a is not None:
                a.mark_delivered(param)
        if _runtime.grad_ready is not None:
            _runtime.grad_ready(param)        # data-parallel: may trigger this bucket's asynchronous all-reduce
        return None
    return g


amax_passes = [0, 0]      # [stand-alone range passes, elements read]: diagnostics for the producer-side fusion

_amax_pools = {}          # device index -> [zeroed int32 pool, next free slot]


def amax_slot(device):
    """A zero-initialised device scalar for a kernel that max-accumulates a tensor's range into it (pylc_bn_apply,
    pylc_bn_bwd_apply).  Slots are views of a pool that is zeroed once per 4096 slots instead of one memset per use."""
    key = torch.device(device).index
    pool = _amax_pools.get(key)
    if pool is None or pool[1] >= pool[0].numel():
        pool = [torch.zeros(4096, dtype=torch.int32, device=device), 0]      # the old pool lives on while tags reference it
        _amax_pools[key] = pool
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


def ranges_needed():
    """True when the conv kernels run the f16x3 arithmetic (precision mode 2), which scales every operand by a power of
    two taken from its max magnitude."""
    return lib.pylc_get_conv_precision() >= 2


def tag_amax(t, amax):
    """Attach a device scalar holding (the float bits of) an upper bound of max|t| to `t`; trusted only while the
    tensor's version counter is unchanged (in-place autograd accumulation bumps it)."""
    t._pylc_amax = (amax, t._version)


def inherit_amax(out, src, binades=0):
    """Tag `out` with an upper bound of its range derived from the tag of the tensor it was computed from, instead of a read
    pass over `out`: max-pooling and (align_corners) bilinear interpolation never exceed max|src|; dropout scales by
    1 / (1 - p) <= 2^binades.  The conv kernels only need an upper bound within a few binades."""
    tag = getattr(src, '_pylc_amax', None)
    if tag is None or tag[1] != src._version:
        return out
    a = tag[0]
    if binades:
        a = a + (binades << 23)            # float bits of 2^binades * amax (one-element int32 tensor)
    tag_amax(out, a)
    return out


def cat_channels(parts):
    """torch.cat along the channels (plumbing) that carries the parts' ranges along: max|cat| = max of the parts' maxima
    (non-negative float bit patterns order like integers)."""
    out = torch.cat(parts, 1)
    if ranges_needed():
        tags = [getattr(t, '_pylc_amax', None) for t in parts]
        if all(tg is not None and tg[1] == t._version for tg, t in zip(tags, parts)):
            a = tags[0][0]
            for tg in tags[1:]:
                a = torch.maximum(a, tg[0])
            tag_amax(out, a)
    return out


def amax_of(t):
    """Device int32[1] with the float bits of max|t| for an NHWC activation / gradient: the producer's tag when one is
    attached and still valid, else one read pass over the tensor."""
    tag = getattr(t, '_pylc_amax', None)
    if tag is not None and tag[1] == t._version:
        return tag[0]
    t = as_nhwc(t)
    b, c, h, w = t.shape
    out = torch.empty(1, dtype=torch.int32, device=t.device)
    check(lib.pylc_amax(ptr(t), b * h * w, c, pitch_of(t), ptr(out), stream()))
    amax_passes[0] += 1
    amax_passes[1] += t.numel()
    tag_amax(t, out)
    return out


def bound_conv_output(y, x, w, bias=None):
    """Tags the output y of a conv (no BatchNorm behind it) with the range BOUND Cin R S max|w| max|x| + max|bias| (pylc_range_product) when
    x's range is known without a pass (fp16 planes or a valid tag) and the filter's is in the arena table: the U-Net's 1x1 up convs, whose
    output only the interpolation + concat kernel reads.  The bound is loose by ~log2(sqrt(Cin)) + 3 binades, well inside the 2^29 the split
    arithmetic tolerates (include/pylc_hip.h, precision mode 2).  Returns y."""
    if not (ranges_needed() and _runtime.fused_grad_ranges) or is_planes(y):
        return y
    if is_planes(x):
        xa = planes_amax(x)
    else:
        tag = getattr(x, '_pylc_amax', None)
        if tag is None or tag[1] != x._version:
            return y
        xa = tag[0]
    wa, ba = getattr(w, '_pylc_wamax', None), (getattr(bias, '_pylc_wamax', None) if bias is not None else None)
    if wa is None or (bias is not None and ba is None):
        return y
    bound = torch.empty(1, dtype=torch.int32, device=y.device)
    check(lib.pylc_range_product(ptr(xa), ptr(wa), float(w.shape[1] * w.shape[2] * w.shape[3]), ptr(ba), ptr(bound), stream()))
    tag_amax(y, bound)
    return y


def weight_amax(w):
    """Range of a conv filter: the flat arena's per-parameter table when the parameter lives in one (refreshed by
    FlatArena.refresh_ranges), else computed here."""
    tab = getattr(w, '_pylc_wamax', None)
    if tab is not None:
        return tab
    out = torch.empty(1, dtype=torch.int32, device=w.device)
    check(lib.pylc_amax(ptr(w), 1, w.numel(), w.numel(), ptr(out), stream()))
    return out


class ResidualLink:
    """Couples the backward nodes that each produce a part of ONE tensor's gradient, so that the parts are summed by the
    kernels that compute them instead of by autograd (an add is a 12 B/element pass):
      * identity-residual block: BatchNorm(+residual x) parks its residual gradient here, the block's first conv (input x)
        accumulates its dgrad into that buffer in the epilogue (`accumulate`);
      * a tensor read by several convs (projection blocks: conv1 + downsample; the low-level features: layer2 + decoder;
        the ASPP branches): the first dgrad to run writes a fresh buffer, the others accumulate into it.
    Every conv armed in the forward counts in `pending`; the backward that brings it to zero returns the buffer as the whole
    gradient of x, the earlier ones return None (autograd adds whatever other consumers of x deliver).  All consumers must
    take part in the backward pass -- true for the networks of this package, where every branch reaches the loss."""
    __slots__ = ('pending', 'buf', 'pool_armed', 'crop', 'masked')

    def __init__(self):
        self.pending = 0
        self.buf = None
        self.pool_armed = False     # U-Net skips: a max-pool reads x and will add the parked crop gradient in its backward
        self.crop = None            # (dy of the concat buffer, channel offset, h0, w0) parked by CropConcatFn.backward
        self.masked = None          # (dout, 1-bit ReLU mask) parked by BnActFn.backward INSTEAD of a written-out residual gradient: the
                                    # conv dgrad that consumes the link forms relu'(dout) in its own epilogue (pylc_conv2d_dgrad_add)

    @property
    def armed(self):
        return self.pending > 0


def _link_sink(link):
    """The buffer a backward node accumulates its part of x's gradient into (None: nothing parked yet).  A parked (dout, mask) pair is
    written out first (one pass) -- the path of consumers that cannot form it in their own epilogue."""
    if link is None:
        return None
    if link.masked is not None:
        dout, mask = link.masked
        link.masked = None
        b, c, h, w = dout.shape
        g = empty_nhwc(b, c, h, w, dout.device)
        check(lib.pylc_relu_bwd_bits(ptr(dout), ptr(mask), ptr(g), b * h * w, c, stream()))
        if link.buf is not None:
            raise L.PylcError('gradient link holds both a buffer and a masked residual gradient')
        link.buf = g
    return link.buf


def grad_link(x):
    """The link shared by all consumers of tensor x (kept on the tensor object); None when x needs no gradient."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return None
    link = getattr(x, '_pylc_link', None)
    if link is None:
        link = ResidualLink()
        x._pylc_link = link
    return link


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
            if os.environ.get('PYLC_DEBUG_PLANES'):
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
            d.w_planes = ptr(planes[0])
        ev = None
        if _timer is not None and (x_pl if _timer.planes else _is_dominant_tile(b * oh * ow, yp, cin, r * s)):
            ev = _timer.bracket(2.0 * b * oh * ow * cout * r * s * cin, 1, 'fwd%dx%d' % (r, s),
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
        if x_pl and os.environ.get('PYLC_PLANES_FWD_ONLY'):          # debug: planes in the forward pass only
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
            if os.environ.get('PYLC_DEBUG_PLANES'):
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
                d.w_planes_t = ptr(planes[1])
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
            if _timer is not None and (dy_pl if _timer.planes else _is_dominant_tile(x.shape[0] * x.shape[2] * x.shape[3] // n_launch, cin, kp, 2)):
                ev = _timer.bracket(2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * cout * r * s * cin, n_launch, 'dgrad%dx%d' % (r, s),
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
            side = _side_stream(x.device) if _runtime.wgrad_side_stream else None
            # PYLC_WGRAD_1X1_MAIN (A/B knob): 1x1 wgrads move as many bytes per FLOP as the BatchNorm passes they would run beside; 1 keeps
            # all of them on the compute stream, 2 only those of maps with at most 32768 pixels (layer3 / layer4 / ASPP)
            if side is not None and r * s == 1 and _runtime.wgrad_1x1_main and (_runtime.wgrad_1x1_main == 1 or x.shape[0] * x.shape[2] * x.shape[3] <= 32768):
                side = None
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
                    if cin_w % 4 == 0:
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
                check(lib.pylc_planes_colsum(ptr(dy), cout, m * cout, nplanes(), ptr(planes_amax(dy)), m, cout, ptr(sums), ptr(ws), st))
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
    if _timer is not None and _timer.inference and cout > 64:
        ev = _timer.bracket(2.0 * b * oh * ow * cout * r * s * cin, 1, 'fwd_bnact%dx%d' % (r, s), 4.0 * (b * h * wd * cin + w.numel() + b * oh * ow * cout))
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
    if cin != cin_w or cin_w % 4 or not w.permute(0, 2, 3, 1).is_contiguous():
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
        if planes is not None:
            d.w_planes = ptr(planes[0])
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
    if _timer is not None and _timer.inference and cout > 64 and cin % 8 == 0:
        ev = _timer.bracket(2.0 * b * oh * ow * cout * r * s * cin, 1, 'fwd_bnact%dx%d' % (r, s), 4.0 * (b * h * wd * cin + w.numel() + b * oh * ow * cout))
        ev[0].record()
    check(lib.pylc_conv2d_fwd_bnact(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(coef[:cout]), ptr(coef[cout:]), ptr(res), int(relu), ptr(y),
                                    ptr(amax), st))
    if ev is not None:
        ev[1].record()
    if amax is not None:
        tag_amax(y, amax)
    return y


# ----------------------------------------------------------------------------------------------
# depthwise 3x3 (Xception separable convs)
# ----------------------------------------------------------------------------------------------
def _dw_desc(x, stride, dil, xp, yp):
    b, c, h, w = x.shape
    d = DwDesc()
    d.B, d.H, d.W, d.C, d.stride, d.dil = b, h, w, c, stride, dil
    d.OH, d.OW = (h - 1) // stride + 1, (w - 1) // stride + 1
    d.x_pitch, d.y_pitch = xp, yp
    return d


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
                nbytes = lib.pylc_dwconv3x3_wgrad_workspace(C.byref(d))
                ws = _ws(nbytes, x.device)
                tgt = _grad_target(w)
                dw = tgt if tgt is not None else torch.empty_like(w)
                if bn_coef is not None:
                    check(lib.pylc_dwconv3x3_wgrad_h_bn(C.byref(d), ptr(x), ptr(yin_bound), ptr(bn_coef[2 * c:3 * c]), ptr(bn_coef[3 * c:]),
                                                        int(ctx.bn_relu), ptr(x_bound), ptr(dy), ptr(dy_bound), ptr(dw), ptr(ws), nbytes, st))
                else:
                    check(lib.pylc_dwconv3x3_wgrad_h(C.byref(d), ptr(x), ptr(x_bound), ptr(dy), ptr(dy_bound), ptr(dw), ptr(ws), nbytes, st))
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
            nbytes = lib.pylc_dwconv3x3_wgrad_workspace(C.byref(d))
            ws = _ws(nbytes, x.device)
            tgt = _grad_target(w)
            dw = tgt if tgt is not None else torch.empty_like(w)
            check(lib.pylc_dwconv3x3_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(ws), nbytes, st))
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


# ----------------------------------------------------------------------------------------------
# BatchNorm (+ ReLU, + residual), optionally synchronised across a process group
# ----------------------------------------------------------------------------------------------
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
        out_planes = bool(out_planes and training and planes_ok(c, m) and m >= PLANES_MIN_PIXELS)
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
                      m * c * (4 + 4 + (4 if residual is not None else 0) + (0.125 if mask is not None else 0)))
        tm.__enter__()
        if out_planes or res_pl is not None or drop_p > 0 or mask is not None or y_bound is not None:
            ex = _bn_extra(drop_p=drop_p, drop_seed=drop_seed)
            if mask is not None:
                ex.relu_mask = ptr(mask)
            if y_bound is not None:
                ex.y_half_bound = ptr(y_bound)
            if out_planes:
                ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), m * c, ptr(bound)
            if res_pl is not None:
                ex.res_planes, ex.res_plane_stride, ex.res_amax = ptr(res_pl), m * c, ptr(res_amax)
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
        tm = _bn_time('bwd_sums(from dgrad)' if pre is not None else 'bwd_reduce(+sums)', m, c, m * c * (8 + msrc) if pre is None else 0)
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
                    ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), m * c, ptr(out_bound)
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
                ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), m * c, ptr(out_bound)
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
        tm = _bn_time('bwd_apply%s' % ('+gres' if g_out is not None else ''), m, c, m * c * (12 + msrc + (4 if g_out is not None else 0)))
        tm.__enter__()
        if use_ex:
            if dy_pl:
                ex.dy_planes, ex.dy_plane_stride, ex.dy_bound = ptr(dy), m * c, ptr(dy_bound)
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
        dy_pl = bool(getattr(y, '_pylc_dy_pl', False)) and not _runtime.no_planes and not os.environ.get('PYLC_NO_PLANES_DY')
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
    dy_pl = bool(getattr(y, '_pylc_dy_pl', False)) and not _runtime.no_planes and not os.environ.get('PYLC_NO_PLANES_DY')
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
    ex.out_planes, ex.out_plane_stride, ex.out_bound = ptr(out), m * c, ptr(bound)
    check(lib.pylc_bn_apply_ex(ptr(x), c, ptr(coef[2 * c:3 * c]), ptr(coef[3 * c:]), None, 0, None, c, m, c, int(relu), None, C.byref(ex), stream()))
    return mark_planes(out, bound)


def is_planes_candidate(out_planes, training, y):
    """Mirror of BnActFn.forward's decision whether the output was written as planes."""
    b, c, h, w = y.shape
    return bool(out_planes and training and planes_ok(c, b * h * w) and b * h * w >= PLANES_MIN_PIXELS)


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
            check(lib.pylc_maxpool_fwd_planes(ptr(x), ptr(y), b * oh * ow * c, nplanes(), ptr(planes_bound), ptr(idx), b, h, w, c, k, stride, pad,
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
    if (out_planes and torch.is_grad_enabled() and ranges_needed() and not _runtime.no_planes and not is_planes(x) and c % 8 == 0
            and planes_ok(c, b * oh * ow) and b * oh * ow >= PLANES_MIN_PIXELS and not os.environ.get('PYLC_NO_POOL_PLANES')):
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
                                                    b * oh * ow * (c1 + c2), nplanes(), ptr(bound), stream()))
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
    """cat([upsample_x2_bilinear(z), center_crop(bridge)], 1) as one fp16-plane tensor (training graphs with ranged arithmetic, channel counts
    that are multiples of 8); None when that form does not apply -- the caller then uses bilinear(into=) + crop_concat."""
    c1, c2 = z.shape[1], bridge.shape[1]
    pixels = z.shape[0] * 4 * z.shape[2] * z.shape[3]
    if not (torch.is_grad_enabled() and ranges_needed() and not _runtime.no_planes and c1 % 8 == 0 and c2 % 8 == 0 and planes_ok(c1 + c2, pixels)
            and pixels >= PLANES_MIN_PIXELS and not is_planes(z) and _runtime.upcat_planes):
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
            check(lib.pylc_gap_fwd_planes(ptr(x), b * h * w * c, nplanes(), ptr(planes_amax(x)), ptr(y), b, h * w, c, stream()))
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


def check_equal_shards():
    """Raise if the last data-parallel loss exchange saw different tile counts on different ranks (the pair that rode on it, see
    MultiLossFn.forward).  One tiny D2H copy: called where the host reads the loss log anyway (Model.log)."""
    chk, _runtime.shard_check = _runtime.shard_check, None
    if chk is None:
        return
    pair, world = chk
    sb, sb2 = (float(v) for v in pair.cpu().tolist())
    if abs(sb2 * world - sb * sb) > 0.5:
        raise RuntimeError('data-parallel ranks hold different batch sizes (sum b = %g, sum b^2 = %g over %d ranks): SyncBN and the loss '
                           'head need equal shards -- use a drop_last loader' % (sb, sb2, world))


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
            _runtime.shard_check = (stats[k:], dist.get_world_size(group))
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
