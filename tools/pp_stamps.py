"""Per-segment timeline of the ping-pong conv kernel (block 0, waves 0 and 4): s_memtime stamps at every segment boundary."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, lib as L
from pylc_amd.lib import lib, check, ptr, stream
L.init()
check(lib.pylc_set_conv_precision(2))
lib.pylc_debug_set_big_tile(2)
lib.pylc_debug_pp_flags(int(os.environ.get('PP_FLAGS', '0')))
dev = torch.device('cuda:0')
cin, cout, k, pad, b, h = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (256, 256, 3, 1, 32, 128)
x = torch.randn(b, h, h, cin, device=dev).permute(0, 3, 1, 2)
w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).permute(0, 3, 1, 2)
d = ops._conv_desc(x, cin, cout, k, k, 1, pad, 1, cin, cout)
rng = (ops.amax_of(x), ops.weight_amax(w))
d.x_amax, d.w_amax = ptr(rng[0]), ptr(rng[1])
if os.environ.get('PYLC_PLANES'):
    e = L.WPrepEntry(0, 0, 2 * cout * k * k * cin, 0, cout, k * k, cin, 0)
    tab = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).clone().to(dev)
    planes = torch.zeros(2 * cout * k * k * cin + 2 * cin * k * k * ((cout + 3) & ~3), dtype=torch.float16, device=dev)
    check(lib.pylc_weight_prepare(ptr(w), ptr(tab), 1, k * k * ((cout + 31) // 32) * ((cin + 31) // 32), ptr(rng[1]), ptr(planes), stream()))
    d.w_planes = planes.data_ptr()
y = ops.empty_nhwc(b, cout, d.OH, d.OW, dev)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
for _ in range(2):
    check(lib.pylc_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, ptr(y), stream()))
lib.pylc_debug_pp_stamps.argtypes = [C.c_void_p]
lib.pylc_debug_pp_flags(int(os.environ.get('PP_FLAGS', '0')))
lib.pylc_debug_pp_stamps(buf.data_ptr())
check(lib.pylc_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, ptr(y), stream()))
torch.cuda.synchronize()
lib.pylc_debug_pp_stamps(None)
t = buf.cpu().view(2, 256)
names = {0: ['compute', 'store+load', 'bar', 'compute', 'store+load', 'bar'],
         1: ['store+load', 'compute', 'bar', 'store+load', 'compute', 'bar']}      # three-stage loop: one barrier per step
if int(os.environ.get('PP_FLAGS', '0')) & 16:
    names = {0: ['compute0', 'bar', 'vmwait', 'itemA0', 'itemA1', 'itemB+load', 'bar', 'compute1', 'bar', 'vmwait', 'itemA0', 'itemA1', 'itemB+load', 'bar'],
             1: ['vmwait', 'itemA0', 'itemA1', 'itemB+load', 'bar', 'compute0', 'bar', 'vmwait', 'itemA0', 'itemA1', 'itemB+load', 'bar', 'compute1', 'bar']}
for g in (0, 1):
    ph = [int(v) for v in t[g][248:253]]
    pro = [int(v) for v in t[g][253:256]]
    if all(ph) and all(pro):
        print('group', g, 'prologue [ticks]: launch -> geometry / tap mask done', pro[0] - ph[0], '| first operand round trip', pro[1] - pro[0],
              '| split + LDS stores', pro[2] - pro[1], '| next loads issued + barrier', ph[1] - pro[2])
    if all(ph):        # whole-tile phases of block 0 (ticks = shader cycles)
        print('group', g, 'tile phases [ticks]: prologue (geometry, first loads, first LDS stores)', ph[1] - ph[0], '| main loop', ph[2] - ph[1],
              '| fold', ph[3] - ph[2], '| epilogue stores + statistics', ph[4] - ph[3], '| total', ph[4] - ph[0])
    ts = [int(v) for v in t[g][:248] if v != 0]
    base = ts[0]
    print('group', g, 'stamps', len(ts), 'clock units: s_memtime ticks')
    # average segment durations over pairs 2.. (skip warm-up)
    k = len(names[g])
    n = (len(ts) - 1) // k
    if n < 3:
        continue
    acc = [0] * k
    for p in range(1, n):
        for j in range(k):
            acc[j] += ts[p * k + j + 1] - ts[p * k + j]
    print('  avg ticks per segment over %d pairs:' % (n - 1), {names[g][j] + str(j): round(acc[j] / max(n - 1, 1)) for j in range(k)})
    print('  pair period', round((ts[(n - 1) * k] - ts[k]) / max(n - 2, 1)))
