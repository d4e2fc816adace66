"""Per-layer-shape conv timing inside the real training step (GPU box): wraps the three conv entry points with HIP events
for a few steps of the bench workload and prints, per distinct shape, calls/step, ms/step and algorithmic TFLOP/s."""
import sys, os, collections, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd import lib as L
from pylc_amd.model import Model, Meta

recs = []
def wrap(name):
    fn = getattr(L.lib, name)
    def timed(d, *a):
        dd = d._obj
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(); rc = fn(d, *a); a1.record()
        key = (name.replace('pylc_conv2d_', '').replace('fwd_stats', 'fwd').replace('dgrad_add', 'dgrad').replace('wgrad_slabs', 'wgrad'), dd.Cin, dd.Cout, dd.R, dd.stride, dd.dil, dd.H, dd.OH)
        flops = 2.0 * dd.B * dd.OH * dd.OW * dd.Cout * dd.R * dd.S * dd.Cin
        recs.append((key, a0, a1, flops))
        return rc
    return timed
import pylc_amd.ops as ops
class Shim:
    def __getattr__(self, n):
        if n in ('pylc_conv2d_fwd', 'pylc_conv2d_fwd_stats', 'pylc_conv2d_dgrad', 'pylc_conv2d_dgrad_add', 'pylc_conv2d_wgrad', 'pylc_conv2d_wgrad_slabs'):
            return wrap(n)
        return getattr(L.lib, n)
for _m in (ops._core, ops.conv, ops.bn, ops.dw, ops.misc):
    _m.lib = Shim()
from pylc_amd.runtime import runtime
if os.environ.get('PYLC_SERIAL'):          # wgrad on the compute stream (un-overlapped kernel times); the product's default for f16x3 since round 5
    runtime.wgrad_side_stream = False
elif os.environ.get('PYLC_SIDE'):          # wgrad on the side stream (the schedule of rounds 1-5)
    runtime.wgrad_side_stream = True
dev = torch.device('cuda:0')
if os.environ.get('PYLC_TABLE_CFG') == 'c2':        # BASELINE configs[1]: U-Net, bs 16, CE only
    model = Model(Meta(arch='unet', ce_weight=1.0, dice_weight=0.0, focal_weight=0.0, report=10**9), dev).build()
    bs = 16
else:
    model = Model(Meta(report=10**9), dev).build()
    bs = 32
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (bs, 3, 512, 512)).astype(np.float32)).to(dev)
y = torch.from_numpy(np.random.RandomState(2).randint(0, 9, (bs, 512, 512)).astype(np.int64)).to(dev)
for _ in range(2): model.train(x, y)
recs.clear()
STEPS = 3
for _ in range(STEPS): model.train(x, y)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for key, a0, a1, fl in recs:
    e = agg[key]; e[0] += 1; e[1] += a0.elapsed_time(a1); e[2] += fl
tot = sum(v[1] for v in agg.values()) / STEPS
print('total conv ms/step %.1f' % tot)
for kind in ('fwd', 'dgrad', 'wgrad'):
    print('   %s %.1f ms/step' % (kind, sum(v[1] for k, v in agg.items() if k[0] == kind) / STEPS))
print('%-7s %5s %5s k s d  %4s->%-4s %6s %8s %8s' % ('kind', 'cin', 'cout', 'H', 'OH', 'n/step', 'ms/step', 'TF/s'))
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, cin, cout, r, s, d, h, oh = key
    print('%-7s %5d %5d %d %d %-2d %4d->%-4d %6.1f %8.2f %8.1f' % (kind, cin, cout, r, s, d, h, oh, v[0] / STEPS, v[1] / STEPS, v[2] / (v[1] * 1e-3) / 1e12))
