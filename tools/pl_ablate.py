"""Timing-only ablations of conv_pl.hip's kernel (unspread schedule): full / no DMA in the loop / no MFMA phase.
usage: pl_ablate.py B H cin cout k pad"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['PYLC_DEBUG_FLAGS'] = '1024'
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib
B, H, cin, cout, k, pad = [int(v) for v in sys.argv[1:7]]
dev = torch.device('cuda:0')
torch.manual_seed(1)
conv = layers.Conv2d(cin, cout, k, 1, pad, 1, bn=True).to(dev)
arena = optim.FlatArena(conv)
x = ops.empty_nhwc(B, cin, H, H, dev); x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
xp = ops.to_planes(x)
def timeit(reps=20):
    fn = lambda: ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True)
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
fl = 2.0 * B * H * H * cout * k * k * cin
with torch.no_grad():
    for tile, tf in (('128', 2048), ('256', 8192)):
        res = []
        for name, extra in (('full', 0), ('no-DMA', 64), ('no-MFMA', 128), ('neither', 192)):
            lib.pylc_debug_pp_flags(1024 | tf | extra)
            t = timeit()
            res.append('%s %.0f us (%.0f TF/s-equiv)' % (name, t, fl / t / 1e6))
        print('tile', tile, '|', ' | '.join(res), flush=True)
