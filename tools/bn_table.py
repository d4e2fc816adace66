"""Per-shape table of the BatchNorm passes INSIDE one training step (GPU box): HIP-event time and achieved HBM rate per (pass, rows, channels),
summed over the layers of that shape.  bytes = algorithmic (4 B per fp32 / plane element touched, 1/8 B per mask bit).

    python tools/bn_table.py [c3|c2|c5]          PYLC_SERIAL=1: wgrad on the main stream (un-overlapped kernel times)
"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd.model import Model, Meta
from pylc_amd import ops, runtime
from pylc_amd.lib import lib, check

cfg = (sys.argv[1:] or ['c3'])[0]
dev = torch.device('cuda:0')
meta, b, ch, hw, ncls, prec = {'c3': (Meta(report=10**9), 32, 3, 512, 9, 2),
                               'c2': (Meta(arch='unet', ce_weight=1.0, dice_weight=0.0, focal_weight=0.0, report=10**9), 16, 3, 512, 9, 2),
                               'c5': (Meta(backbone='xception', ch=1, n_classes=11, report=10**9), 8, 1, 1024, 11, 3)}[cfg]
from pylc_amd import lib as L
L.init()
if os.environ.get('PYLC_SERIAL'):          # wgrad on the compute stream (un-overlapped kernel times); the product's default for f16x3 since round 5
    runtime.wgrad_side_stream = False
elif os.environ.get('PYLC_SIDE'):          # wgrad on the side stream (the schedule of rounds 1-5)
    runtime.wgrad_side_stream = True
check(lib.pylc_set_conv_precision(prec))
model = Model(meta, dev).build()
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (b, ch, hw, hw)).astype(np.float32)).to(dev)
y = torch.from_numpy(np.random.RandomState(2).randint(0, ncls, (b, hw, hw)).astype(np.int64)).to(dev)
for _ in range(6):
    model.train(x, y)
torch.cuda.synchronize()
ops.bn_timing = []
n = 3
for _ in range(n):
    model.train(x, y)
torch.cuda.synchronize()
rows = collections.OrderedDict()
for kind, m, c, nbytes, a, e in ops.bn_timing:
    r = rows.setdefault((kind, m, c), [0, 0.0, 0.0])
    r[0] += 1; r[1] += a.elapsed_time(e); r[2] += nbytes
ops.bn_timing = None
tot = collections.defaultdict(lambda: [0.0, 0.0])
print('%-22s %9s %6s %7s %10s %9s %8s' % ('pass', 'rows', 'C', 'layers', 'ms/step', 'us/launch', 'TB/s'))
for (kind, m, c), (cnt, ms, nb) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print('%-22s %9d %6d %7d %10.3f %9.1f %8.2f' % (kind, m, c, cnt // n, ms / n, 1e3 * ms / cnt, nb / ms / 1e9))
    tot[kind][0] += ms / n; tot[kind][1] += nb / n
print()
for kind, (ms, nb) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print('%-22s %10.3f ms/step %8.2f GB/step %8.2f TB/s' % (kind, ms, nb / 1e9, nb / ms / 1e9))
print('all BatchNorm passes: %.2f ms/step, %.1f GB/step' % (sum(v[0] for v in tot.values()), sum(v[1] for v in tot.values()) / 1e9))
