"""What the conv epilogue's parts cost on the short-K class: forward with / without the fused BatchNorm statistics, per shape (isolated).
Usage: python tools/epi_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SHAPES = [(32, 32, 256, 1024, 1, 0, 1), (32, 32, 1024, 256, 1, 0, 1), (32, 128, 64, 256, 1, 0, 1), (32, 64, 128, 512, 1, 0, 1),
          (32, 32, 256, 256, 3, 1, 1), (32, 32, 512, 2048, 1, 0, 1)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (B, H, cin, cout, k, pad, dil) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, 1, pad, dil, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev)
    x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
    xp = ops.to_planes(x)
    out = []
    with torch.no_grad():
        for flags in (0, 8):
            lib.pylc_debug_pp_flags(flags)
            for st in (True, False):
                out.append(min(timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, pad, dil, want_stats=st)) for _ in range(2)))
    lib.pylc_debug_pp_flags(0)
    print('B%d %3d^2 %4d->%4d k%d | lean: stats %6.1f us, no stats %6.1f us | general: stats %6.1f us, no stats %6.1f us' % (B, H, cin, cout, k, *out), flush=True)
