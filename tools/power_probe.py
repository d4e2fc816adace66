"""Board power and shader clock while one conv shape runs back to back (is the MFMA kernel power / clock limited?)."""
import sys, os, time, subprocess, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, lib as L
from pylc_amd.lib import lib, check, ptr, stream
L.init()
dev = torch.device('cuda:0')
cin, cout, k, pad, b, h = 256, 256, 3, 1, 32, 128
x = torch.randn(b, h, h, cin, device=dev).permute(0, 3, 1, 2)
w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).permute(0, 3, 1, 2)
d = ops._conv_desc(x, cin, cout, k, k, 1, pad, 1, cin, cout)
rng = (ops.amax_of(x), ops.weight_amax(w))
d.x_amax, d.w_amax = ptr(rng[0]), ptr(rng[1])
y = ops.empty_nhwc(b, cout, d.OH, d.OW, dev)
samples = []
stop = False
def sampler():
    while not stop:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showmaxpower'], capture_output=True, text=True).stdout
        samples.append(' | '.join(l.split(':', 1)[-1].strip() for l in out.splitlines() if any(s in l for s in ('sclk', 'Power (W)', 'Package Power'))))
th = threading.Thread(target=sampler); th.start()
for mode, name in ((2, 'f16x3'), (1, 'bf16x6'), (0, 'f32 mfma')):
    check(lib.pylc_set_conv_precision(mode))
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(20): check(lib.pylc_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, ptr(y), stream()))
        torch.cuda.synchronize(); n += 20
    dt = (time.perf_counter() - t0) / n
    print('%-9s %.0f us/launch %.0f TFLOP/s algorithmic | last samples: %s' % (name, dt * 1e6, 2.0 * b * h * h * cout * 9 * cin / dt / 1e12, ' || '.join(samples[-2:])), flush=True)
stop = True; th.join()
