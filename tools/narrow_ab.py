"""Per-shape time of the <= 64-output-channel launches: the default (gg_plhn_kernel -- four-wave halo kernel -- for 3x3 / unit steps on large grids),
32 x 64 wave tiles (gg_pl_kernel<.., NARROW>; debug flag 16777216 = no gg_plhn_kernel) against the wide 2-column wave grid
(debug flag 65536) and, for 3x3 shapes the halo kernel takes, that kernel (flag 131072); the narrow form also with the tile height forced
to 128 / 256 rows (flags 2048 / 8192).   usage: python tools/narrow_ab.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 16777216      # no gg_plhn_kernel
W = 134217728     # ... for wide launches (the two `wide` columns force the wide wave grid on these <= 64-channel shapes)
SHAPES = [  # B, H, W, Cin, Cout, k, pad      (forward launches; a dgrad with Cin <= 64 is the mirrored forward shape)
    (32, 128, 128, 256, 64, 1, 0), (32, 128, 128, 64, 64, 3, 1), (32, 128, 128, 256, 48, 1, 0),
    (16, 510, 510, 64, 64, 3, 0), (16, 256, 256, 128, 64, 3, 0), (16, 324, 324, 64, 64, 3, 0), (16, 252, 252, 128, 64, 1, 0),
    (8, 512, 512, 64, 64, 1, 0), (8, 512, 512, 32, 64, 3, 1),
]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 0.1:      # clocks: the first launches after a host-side pause run slow (the first column of a row used to read 3-10 % low)
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (B, H, W, cin, cout, k, pad) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, 1, pad, 1, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, W, dev)
    x.copy_(torch.randn(B, cin, H, W, device=dev) * 3)
    xp = ops.to_planes(x)
    oh, ow = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    fl = 2.0 * B * oh * ow * cout * k * k * cin
    res = []
    with torch.no_grad():
        for name, flags in (('default', 0), ('narrow', N), ('narrow/128 rows', N | 2048), ('narrow/256 rows', N | 8192), ('wide', 65536 | W), ('halo/wide', 131072 | W)):
            lib.pylc_debug_pp_flags(flags)
            t = timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True))
            res.append('%s %.0f us %.0f TF' % (name, 1e3 * t, fl / t / 1e9))
        if os.environ.get('NARROW_STAGGER'):      # start delay of each CU's second block in the four-wave halo kernel (2048-cycle units; default 3)
            for stg in (0, 1, 2, 5, 8):
                lib.pylc_debug_pp_flags(0)
                lib.pylc_debug_stagger(stg)
                t = timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True))
                res.append('stagger %d %.0f us' % (stg, 1e3 * t))
            lib.pylc_debug_stagger(-1)
    lib.pylc_debug_pp_flags(0)
    print('%-40s' % str((B, H, W, cin, cout, k, pad)), ' | '.join(res), flush=True)
