"""Timing probe for the chunk-interleaved plane layout: the conv_pl kernels address both operands as if the two planes of a
K-step row shared one 128-byte line (pylc_debug_pp_flags 512; RESULTS ARE GARBAGE, only the time counts) against today's addressing.
Usage: python tools/il_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SHAPES = [  # B, H, Cin, Cout, k, stride, pad, dil   (the R101 step's classes; a dgrad is the same GEMM with the channel roles swapped)
    (32, 32, 256, 1024, 1, 1, 0, 1),
    (32, 32, 1024, 256, 1, 1, 0, 1),
    (32, 32, 256, 256, 3, 1, 1, 1),
    (32, 128, 256, 256, 3, 1, 1, 1),
    (32, 128, 64, 256, 1, 1, 0, 1),
    (32, 128, 256, 64, 1, 1, 0, 1),
    (32, 64, 128, 512, 1, 1, 0, 1),
    (32, 64, 512, 128, 1, 1, 0, 1),
    (32, 64, 128, 128, 3, 1, 1, 1),
    (32, 32, 512, 2048, 1, 1, 0, 1),
    (32, 32, 2048, 512, 1, 1, 0, 1),
    (32, 32, 2048, 256, 3, 1, 12, 12),
    (32, 32, 512, 512, 3, 1, 2, 2),
]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = [0.0, 0.0]
for (B, H, cin, cout, k, st, pad, dil) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev)
    x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
    xp = ops.to_planes(x)
    ts = []
    with torch.no_grad():
        for flags in (0, 512 | 4, 4, 512, 0, 512 | 4, 4, 512):
            lib.pylc_debug_pp_flags(flags)
            ts.append(timeit(lambda: ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True)))
    lib.pylc_debug_pp_flags(0)
    t0, t1, tb, ta = min(ts[0], ts[4]), min(ts[1], ts[5]), min(ts[2], ts[6]), min(ts[3], ts[7])
    fl = 2.0 * B * H * H * cout * k * k * cin
    tot[0] += t0; tot[1] += t1
    print('B%d %3dx%-3d %4d->%4d k%d d%-2d | separate planes %7.1f us %4.0f TF/s | interleaved addressing %7.1f us %4.0f TF/s | %+5.1f %% | filters only %+5.1f %% | pixels only %+5.1f %%'
          % (B, H, H, cin, cout, k, dil, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9, 100 * (t0 / t1 - 1), 100 * (t0 / tb - 1), 100 * (t0 / ta - 1)), flush=True)
print('sum %.1f -> %.1f us (%+.1f %%)' % (tot[0] * 1e3, tot[1] * 1e3, 100 * (tot[0] / tot[1] - 1)))
