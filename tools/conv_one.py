"""Run ONE conv shape a few times (for rocprofv3 --pmc passes on the GPU box).
usage: conv_one.py <fwd|dgrad|wgrad> cin cout k stride pad dil B H [reps]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, lib as L
from pylc_amd.lib import lib, check, ptr, stream
kind = sys.argv[1]
cin, cout, k, stride, pad, dil, b, h = [int(v) for v in sys.argv[2:10]]
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 3
L.init()
mode = int(os.environ.get('PYLC_MODE', '2'))
check(lib.pylc_set_conv_precision(mode))
lib.pylc_debug_set_big_tile(int(os.environ.get('PYLC_BIG', '2')))
dev = torch.device('cuda:0')
x = torch.randn(b, h, h, cin, device=dev).permute(0, 3, 1, 2)
w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).permute(0, 3, 1, 2)
d = ops._conv_desc(x, cin, cout, k, k, stride, pad, dil, cin, (cout + 3) & ~3)
y = ops.empty_nhwc(b, cout, d.OH, d.OW, dev)
dy = torch.randn(b, d.OH, d.OW, cout, device=dev).permute(0, 3, 1, 2)
wt = torch.empty((cin, k * k, (cout + 3) & ~3), device=dev)
check(lib.pylc_weight_transpose(ptr(w), ptr(wt), cout, k * k, cin, stream()))
if mode == 2:
    rng = (ops.amax_of(x), ops.weight_amax(w), ops.amax_of(dy))
    d.x_amax, d.w_amax, d.dy_amax = (ptr(t) for t in rng)
if mode == 2 and os.environ.get('PYLC_PLANES'):
    kp = (cout + 3) & ~3
    e = L.WPrepEntry(0, 0, 2 * cout * k * k * cin, 0, cout, k * k, cin, 0)
    tab = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).clone().to(dev)
    planes = torch.zeros(2 * cout * k * k * cin + 2 * cin * k * k * kp, dtype=torch.float16, device=dev)
    check(lib.pylc_weight_prepare(ptr(w), ptr(tab), 1, k * k * ((cout + 31) // 32) * ((cin + 31) // 32), ptr(rng[1]), ptr(planes), stream()))
    d.w_planes = planes.data_ptr()
    d.w_planes_t = planes.data_ptr() + 2 * (2 * cout * k * k * cin)
dx = ops.empty_nhwc(b, cin, h, h, dev)
dw = torch.empty((cout, k, k, cin), device=dev)
nbytes = lib.pylc_conv2d_wgrad_workspace(C.byref(d))
ws = torch.empty(max(nbytes, 4) // 4 + 1, device=dev)
for _ in range(reps):
    if kind == 'fwd':
        check(lib.pylc_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, ptr(y), stream()))
    elif kind == 'dgrad':
        check(lib.pylc_conv2d_dgrad(C.byref(d), ptr(dy), ptr(wt), ptr(dx), 0, stream()))
    else:
        check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw), None, ptr(ws), nbytes, stream()))
torch.cuda.synchronize()
