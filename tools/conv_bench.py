"""Micro-benchmark of the conv kernels on the DeepLab/R101 layer shapes (GPU box): TFLOP/s per shape for
fwd / dgrad / wgrad, timed with HIP events on the launch stream, median of interleaved rounds."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, lib as L
from pylc_amd.lib import lib, check, ptr, stream

SHAPES = [  # name, cin, cout, k, stride, pad, dil, B, H
    ('l3 3x3 256->256 @32', 256, 256, 3, 1, 1, 1, 32, 32),
    ('l3 1x1 1024->256 @32', 1024, 256, 1, 1, 0, 1, 32, 32),
    ('l3 1x1 256->1024 @32', 256, 1024, 1, 1, 0, 1, 32, 32),
    ('dec 3x3 304->256 @128', 304, 256, 3, 1, 1, 1, 32, 128),
    ('dec 3x3 256->256 @128', 256, 256, 3, 1, 1, 1, 32, 128),
    ('aspp 3x3 d12 2048->256 @32', 2048, 256, 3, 1, 12, 12, 32, 32),
    ('l4 3x3 d2 512->512 @32', 512, 512, 3, 1, 2, 2, 32, 32),
    ('l2 3x3 128->128 @64', 128, 128, 3, 1, 1, 1, 32, 64),
    ('l1 3x3 64->64 @128', 64, 64, 3, 1, 1, 1, 32, 128),
    ('l1 1x1 64->256 @128', 64, 256, 1, 1, 0, 1, 32, 128),
    ('l1 1x1 256->64 @128', 256, 64, 1, 1, 0, 1, 32, 128),
    ('l2 3x3 s2 128->128 @128', 128, 128, 3, 2, 1, 1, 32, 128),
    ('l4 1x1 512->2048 @32', 512, 2048, 1, 1, 0, 1, 32, 32),
]
which = [a for a in sys.argv[1:] if a in ('fwd', 'dgrad', 'wgrad')] or ['fwd', 'dgrad', 'wgrad']
L.init()
mode = int(os.environ.get('PYLC_MODE', '2'))
check(lib.pylc_set_conv_precision(mode))
big = int(os.environ.get('PYLC_BIG', '2'))
lib.pylc_debug_set_big_tile(big)
lib.pylc_debug_pp_flags(int(os.environ.get('PP_FLAGS', '0')))
print('conv precision mode', mode, 'big tile', big)
dev = torch.device('cuda:0')
rounds = 7
for name, cin, cout, k, stride, pad, dil, b, h in SHAPES:
    x = torch.randn(b, h, h, cin, device=dev).permute(0, 3, 1, 2)
    w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).permute(0, 3, 1, 2)
    d = ops._conv_desc(x, cin, cout, k, k, stride, pad, dil, cin, (cout + 3) & ~3)
    y = ops.empty_nhwc(b, cout, d.OH, d.OW, dev)
    dy = torch.randn(b, d.OH, d.OW, cout, device=dev).permute(0, 3, 1, 2)
    wt = torch.empty((cin, k * k, (cout + 3) & ~3), device=dev)
    check(lib.pylc_weight_transpose(ptr(w), ptr(wt), cout, k * k, cin, stream()))
    if mode == 2:                                 # operand ranges of the f16x3 arithmetic (kept alive in `rng`)
        rng = (ops.amax_of(x), ops.weight_amax(w), ops.amax_of(dy))
        d.x_amax, d.w_amax, d.dy_amax = (ptr(t) for t in rng)
    if mode == 2 and os.environ.get('PYLC_PLANES'):          # prepared filter planes (what FlatArena does once per step)
        kp = (cout + 3) & ~3
        e = L.WPrepEntry(0, 0, 2 * cout * k * k * cin, 0, cout, k * k, cin, 0)
        tab = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).clone().to(dev)
        planes = torch.zeros(2 * cout * k * k * cin + 2 * cin * k * k * kp, dtype=torch.float16, device=dev)
        tiles = k * k * ((cout + 31) // 32) * ((cin + 31) // 32)
        check(lib.pylc_weight_prepare(ptr(w), ptr(tab), 1, tiles, ptr(rng[1]), ptr(planes), stream()))
        d.w_planes = planes.data_ptr()
        d.w_planes_t = planes.data_ptr() + 2 * (2 * cout * k * k * cin)
    dx = ops.empty_nhwc(b, cin, h, h, dev)
    dw = torch.empty((cout, k, k, cin), device=dev)
    nbytes = lib.pylc_conv2d_wgrad_workspace(C.byref(d))
    ws = torch.empty(max(nbytes, 4) // 4 + 1, device=dev)
    flops = 2.0 * b * d.OH * d.OW * cout * k * k * cin
    fns = {
        'fwd': lambda: check(lib.pylc_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, ptr(y), stream())),
        'dgrad': lambda: check(lib.pylc_conv2d_dgrad(C.byref(d), ptr(dy), ptr(wt), ptr(dx), int(os.environ.get('PYLC_ACC', '0')), stream())),
        'wgrad': lambda: check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw), None, ptr(ws), nbytes, stream())),
    }
    for kind in which:
        fns[kind](); fns[kind]()
    torch.cuda.synchronize()
    times = {kind: [] for kind in which}
    for _ in range(rounds):
        for kind in which:
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fns[kind](); fns[kind](); fns[kind](); e.record()
            torch.cuda.synchronize()
            times[kind].append(a.elapsed_time(e) / 3)
    out = '%-28s %6.1f GF ' % (name, flops / 1e9)
    for kind in which:
        t = sorted(times[kind])[len(times[kind]) // 2]
        out += ' %s %7.1f us %6.1f TF |' % (kind, t * 1e3, flops / (t * 1e-3) / 1e12)
    print(out, flush=True)
