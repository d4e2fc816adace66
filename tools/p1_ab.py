"""A/B of the persistent 1x1 kernel (conv_p1.hip) against the per-tile kernel (pylc_debug_p1(0)): per-shape time, bit-identity of the output and
of the BatchNorm statistics partials.   usage: python tools/p1_ab.py [reps] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib, check

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
check(lib.pylc_set_conv_precision(mode))
SHAPES = [  # B, H, Cin, Cout
    (32, 32, 256, 1024), (32, 32, 1024, 256), (32, 64, 128, 512), (32, 64, 512, 128), (32, 128, 64, 256), (32, 128, 256, 64),
    (32, 32, 512, 2048), (32, 32, 2048, 512), (32, 32, 1024, 2048), (32, 32, 2048, 256), (32, 128, 256, 48), (8, 64, 728, 728),
    (8, 256, 128, 128), (3, 50, 72, 200),
]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (B, H, cin, cout) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, 1, 1, 0, 1, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev)
    x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
    xp = ops.to_planes(x)
    fl = 2.0 * B * H * H * cout * cin
    res = {}
    with torch.no_grad():
        for name, on in (('per-tile', 0), ('persistent', 1), ('per-tile2', 0), ('persistent2', 1)):
            lib.pylc_debug_p1(on)
            y = ops.conv2d(xp, conv.weight, None, 1, 0, 1, want_stats=True)
            torch.cuda.synchronize()
            t = timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, 0, 1, want_stats=True))
            res[name] = (t, y.clone(), y._pylc_sums.clone())
    same = torch.equal(res['per-tile'][1], res['persistent'][1]) and torch.equal(res['per-tile'][2], res['persistent'][2])
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.double())
    err = ((res['persistent'][1].double() - ref).abs().max() / ref.abs().max()).item()
    print('%-24s' % str((B, H, cin, cout)), ' | '.join('%s %.0f us %.0f TF/s' % (n, 1e3 * t, fl / t / 1e9) for n, (t, _, _) in res.items()),
          '| identical' if same else '| DIFFERENT', '| err vs fp64 %.1e' % err, flush=True)
lib.pylc_debug_p1(1)
