"""Per-shape time of the 3x3 / unit-step launches the halo kernels take: the default (gg_plhn_kernel -- four waves on 256 pixels x 64 channels, two blocks per
CU, 64-wide column tiles -- for every launch of at least 512 such blocks) against gg_plh_kernel (eight waves, 256 pixels x 128 channels, one block per CU:
debug flag 134217728 for the wide launches) and against the per-tap kernel (flag 16384).  Forward conv + statistics; outputs compared bit for bit.   usage: python tools/halo_ab.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [  # B, H, W, Cin, Cout, pad
    (32, 32, 32, 256, 256, 1), (32, 128, 128, 304, 256, 1), (32, 128, 128, 256, 256, 1), (32, 64, 64, 128, 128, 1), (32, 128, 128, 64, 64, 1),
    (16, 252, 252, 128, 128, 0), (16, 123, 123, 256, 256, 0), (16, 168, 168, 256, 128, 0), (16, 48, 48, 1024, 512, 0), (8, 64, 64, 728, 728, 1),
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


for (B, H, W, cin, cout, pad) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, 3, 1, pad, 1, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, W, dev)
    x.copy_(torch.randn(B, cin, H, W, device=dev) * 3)
    xp = ops.to_planes(x)
    oh, ow = H + 2 * pad - 2, W + 2 * pad - 2
    fl = 2.0 * B * oh * ow * cout * 9 * cin
    res, ys = [], []
    with torch.no_grad():
        for name, flags in (('default', 0), ('gg_plh_kernel', 134217728), ('per tap', 16384)):
            lib.pylc_debug_pp_flags(flags)
            ys.append(ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True).clone())
            t = timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True))
            res.append('%s %.0f us %.0f TF' % (name, 1e3 * t, fl / t / 1e9))
    lib.pylc_debug_pp_flags(0)
    same = all(torch.equal(ys[0], y) for y in ys[1:])
    print('%-36s' % str((B, H, W, cin, cout, pad)), ' | '.join(res), '| identical' if same else '| DIFFERENT', flush=True)
