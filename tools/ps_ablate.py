"""Where the specialised-wave 1x1 kernel (conv_ps.hip) spends its time: the same launch with the DMA (flag 64), the MFMAs (128) and / or the
stores (256) removed (results are garbage, timings informative).   usage: python tools/ps_ablate.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd import lib as L
from pylc_amd.lib import lib, check

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L.init()
check(lib.pylc_set_conv_precision(2))


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (B, H, cin, cout) in [(32, 32, 256, 1024), (32, 32, 1024, 256), (32, 128, 64, 256), (32, 32, 512, 2048)]:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, 1, 1, 0, 1, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev)
    x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
    xp = ops.to_planes(x)
    line = '%-22s' % str((B, H, cin, cout))
    with torch.no_grad():
        lib.pylc_debug_ps(0)
        line += ' per-tile %.0f us |' % (1e3 * timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, 0, 1, want_stats=True)))
        lib.pylc_debug_ps(1)
        for name, fl in (('full', 0), ('no-store', 256), ('no-DMA', 64), ('no-MFMA', 128), ('no-DMA no-MFMA', 192), ('no-DMA no-store', 320),
                         ('no-MFMA no-store', 384), ('nothing', 448)):
            lib.pylc_debug_pp_flags(fl)
            line += ' %s %.0f' % (name, 1e3 * timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, 0, 1, want_stats=True)))
        lib.pylc_debug_pp_flags(0)
        lib.pylc_debug_ps(0)
    print(line, flush=True)
