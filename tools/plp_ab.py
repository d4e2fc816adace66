"""A/B of the persistent 1x1 kernel (conv_pl.hip gg_plp_kernel) against the per-tile kernel it replaces (pylc_debug_pp_flags bit 19 = 524288
switches it off): per-shape time of the forward (+ BatchNorm statistics), the plain dgrad and the dgrad that adds a ReLU-masked residual
gradient; bit-identity of every output (y, the statistics partials, dx).
    usage: python tools/plp_ab.py [reps] [mode]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd import lib as L
from pylc_amd.lib import lib, check, ptr, stream

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L.init()
check(lib.pylc_set_conv_precision(mode))
OFF = int(os.environ.get('PLP_OFF', '524288'))      # flags of side A (default: persistent kernel off; 8388608: the wide 128 x 256 tile off)
VARIANT = int(os.environ.get('PLP_VARIANT', '0'))       # extra pylc_debug_pp_flags bits for the persistent side (20: one tile per block, 21: no soft waits, 22: 768 blocks)
SHAPES = [  # B, H, Cin, Cout  (the forward conv; its dgrad maps Cout -> Cin)
    (32, 32, 256, 1024), (32, 32, 1024, 256), (32, 64, 128, 512), (32, 64, 512, 128), (32, 128, 64, 256), (32, 128, 256, 64),
    (32, 32, 512, 2048), (32, 32, 2048, 512), (32, 32, 1024, 2048), (32, 32, 2048, 256), (32, 32, 1280, 256), (32, 128, 256, 128), (32, 64, 512, 256),
    (8, 64, 728, 728), (30, 36, 256, 1024),
]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if os.environ.get('PLP_SHAPES'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['PLP_SHAPES'].split(',')]
tot = {}
for (B, H, cin, cout) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, 1, 1, 0, 1, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev)
    x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
    xp = ops.to_planes(x)
    dy = ops.empty_nhwc(B, cout, H, H, dev)
    dy.copy_(torch.randn(B, cout, H, H, device=dev))
    dyp = ops.to_planes(dy)
    res = ops.empty_nhwc(B, cin, H, H, dev)
    res.copy_(torch.randn(B, cin, H, H, device=dev))
    mask = torch.randint(0, 256, (B * H * H * cin // 8,), dtype=torch.uint8, device=dev)
    fl = 2.0 * B * H * H * cout * cin
    d = ops._conv_desc(x, cin, cout, 1, 1, 1, 0, 1, cin, cout)
    d.x_fmt, d.dy_fmt = 1, 1
    d.x_amax, d.w_amax, d.dy_amax = ptr(ops.planes_amax(xp)), ptr(ops.weight_amax(conv.weight)), ptr(ops.planes_amax(dyp))
    d.w_planes_t = ptr(conv.weight._pylc_planes[1])
    d.w_planes_fmt = ops.filter_planes_fmt(conv.weight._pylc_planes)
    assert not lib.pylc_conv2d_dgrad_needs_f32_weights(C.byref(d))
    dx = ops.empty_nhwc(B, cin, H, H, dev)

    def fwd():
        return ops.conv2d(xp, conv.weight, None, 1, 0, 1, want_stats=True)

    def dgrad():
        check(lib.pylc_conv2d_dgrad(C.byref(d), ptr(dyp), None, ptr(dx), 0, stream()))

    def dgrad_add():
        check(lib.pylc_conv2d_dgrad_add(C.byref(d), ptr(dyp), None, ptr(dx), 0, ptr(res), ptr(mask), stream()))

    line = '%-22s' % str((B, H, cin, cout))
    with torch.no_grad():
        for name, fn in (('fwd', fwd), ('dgrad', dgrad), ('dgrad+res', dgrad_add)):
            out = {}
            for off in (OFF, 0, OFF, 0):
                lib.pylc_debug_pp_flags(off if off else VARIANT)
                dx.fill_(float('nan'))
                r = fn()
                torch.cuda.synchronize()
                got = [t.clone() for t in r[:2]] if name == 'fwd' else [dx.clone()]
                t = timeit(fn)
                out.setdefault(off, []).append((t, got))
            lib.pylc_debug_pp_flags(0)
            ta, tb = min(v[0] for v in out[OFF]), min(v[0] for v in out[0])
            same = all(torch.equal(u, w) for u, w in zip(out[OFF][0][1], out[0][0][1])) and not any(torch.isnan(u).any() for u in out[0][0][1])
            tot[name] = [tot.get(name, [0, 0])[0] + ta, tot.get(name, [0, 0])[1] + tb]
            line += ' | %s %.0f -> %.0f us (%.0f -> %.0f TF/s)%s' % (name, ta * 1e3, tb * 1e3, fl / ta / 1e9, fl / tb / 1e9, '' if same else ' DIFFERENT')
    print(line, flush=True)
print('sum over shapes (ms): ' + ' | '.join('%s %.3f -> %.3f' % (k, v[0], v[1]) for k, v in tot.items()))
