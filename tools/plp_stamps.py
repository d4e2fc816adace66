"""Timeline of the persistent 1x1 kernel (conv_pl.hip gg_plp_kernel, STAMPS build): s_memtime stamps of waves 0 and 3 of one block.
usage: plp_stamps.py B H cin cout [block] [fwd|dgrad]      (fwd: with BatchNorm statistics, EPK 1; dgrad: plain, EPK 0)"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, H, cin, cout = [int(v) for v in sys.argv[1:5]]
block = int(sys.argv[5]) if len(sys.argv) > 5 else 0
kind = sys.argv[6] if len(sys.argv) > 6 else 'fwd'
os.environ['PYLC_DEBUG_FLAGS'] = str((block << 16) | int(os.environ.get('PLP_VARIANT', '0')))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib, check, ptr, stream
dev = torch.device('cuda:0')
torch.manual_seed(1)
conv = layers.Conv2d(cin, cout, 1, 1, 0, 1, bn=True).to(dev)
arena = optim.FlatArena(conv)
x = ops.empty_nhwc(B, cin, H, H, dev); x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
xp = ops.to_planes(x)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
run = lambda: ops.conv2d(xp, conv.weight, None, 1, 0, 1, want_stats=(kind == 'fwd'))
with torch.no_grad():
    for _ in range(3):
        run()
    lib.pylc_debug_pp_stamps.argtypes = [C.c_void_p]
    lib.pylc_debug_pp_stamps(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.pylc_debug_pp_stamps(None)
KS = 32
S = (cin + KS - 1) // KS
t = buf.cpu().view(2, 256)
for gi in (0, 1):
    ts = [int(v) for v in t[gi] if v != 0]
    print('wave', 0 if gi == 0 else 3, ': stamps', len(ts))
    if len(ts) < 10:
        continue
    i = 1
    print('  kernel start -> first step top: %d' % (ts[1] - ts[0]))
    tile = 0
    while i + 4 * S + 5 <= len(ts):
        steps = []
        for s in range(S):
            a, b, c, d = ts[i:i + 4]
            nxt = ts[i + 4]
            steps.append((b - a, c - b, d - c, nxt - d))
            i += 4
        b0, b1, e1, e2, e3 = ts[i:i + 5]
        i += 5
        print('  tile %d: steps (vm wait | barrier | DMA issue | reads + MFMA): %s' % (tile, ' '.join('%d|%d|%d|%d' % st for st in steps)))
        print('          loop %d | boundary barrier %d | loads / statistics %d | DMA issue %d | stores %d | tile total %d' %
              (sum(sum(st) for st in steps), b1 - b0, e1 - b1, e2 - e1, e3 - e2, e3 - (ts[i - 5 - 4 * S])))
        tile += 1
    print('  total %d' % (ts[-1] - ts[0]))
