"""A/B of the wave-pair de-phasing in conv_pl.hip's 256x128 kernel (debug flag 32768 = all waves request first, the lockstep schedule):
per-shape time of the 256-row kernel (halo kernel off), both schedules, and bit-identity of the results.
usage: python tools/pl_dephase_ab.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [  # B, H, Cin, Cout, k, stride, pad, dil
    (32, 128, 256, 256, 3, 1, 1, 1),
    (32, 64, 128, 128, 3, 1, 1, 1),
    (32, 32, 256, 256, 3, 1, 1, 1),
    (32, 32, 2048, 256, 3, 1, 12, 12),
    (32, 32, 1024, 256, 1, 1, 0, 1),
    (32, 32, 256, 1024, 1, 1, 0, 1),
    (32, 128, 64, 256, 1, 1, 0, 1),
    (32, 64, 512, 128, 1, 1, 0, 1),
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


for (B, H, cin, cout, k, st, pad, dil) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev)
    x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
    xp = ops.to_planes(x)
    fl = 2.0 * B * (H // st) ** 2 * cout * k * k * cin
    res = {}
    with torch.no_grad():
        for name, flags in (('lockstep', 1024 | 8192 | 16384 | 32768), ('dephased', 1024 | 8192 | 16384), ('lockstep2', 1024 | 8192 | 16384 | 32768),
                            ('dephased2', 1024 | 8192 | 16384), ('pl128', 1024 | 2048 | 16384), ('default', 0)):
            lib.pylc_debug_pp_flags(flags)
            y = ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True)
            t = timeit(lambda: ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True))
            res[name] = (t, y.clone())
    same = torch.equal(res['lockstep'][1], res['dephased'][1])
    print('%-34s' % str((B, H, cin, cout, k, st, pad, dil)), ' | '.join('%s %.0f us %.0f TF/s' % (n, 1e3 * t, fl / t / 1e9) for n, (t, _) in res.items()),
          '| identical' if same else '| DIFFERENT', flush=True)
lib.pylc_debug_pp_flags(0)
