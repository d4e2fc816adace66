"""Per-K-step timeline of conv_pl.hip's kernel (one block, first and last wave): s_memtime stamps at the phase boundaries.
usage: pl_stamps.py <pl128|pl256> B H cin cout k pad [block]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kind = sys.argv[1]
B, H, cin, cout, k, pad = [int(v) for v in sys.argv[2:8]]
block = int(sys.argv[8]) if len(sys.argv) > 8 else 0
flags = (1024 | (2048 if kind == 'pl128' else 8192)) | (block << 16)
os.environ['PYLC_DEBUG_FLAGS'] = str(flags)
import ctypes as C
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib
dev = torch.device('cuda:0')
torch.manual_seed(1)
conv = layers.Conv2d(cin, cout, k, 1, pad, 1, bn=True).to(dev)
arena = optim.FlatArena(conv)
x = ops.empty_nhwc(B, cin, H, H, dev); x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
xp = ops.to_planes(x)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True)
    lib.pylc_debug_pp_stamps.argtypes = [C.c_void_p]
    lib.pylc_debug_pp_stamps(buf.data_ptr())
    ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True)
    torch.cuda.synchronize()
    lib.pylc_debug_pp_stamps(None)
t = buf.cpu().view(2, 256)
for g in (0, 1):
    ts = [int(v) for v in t[g] if v != 0]
    if len(ts) < 8:
        print('wave group', g, 'no stamps'); continue
    print('wave', 'first' if g == 0 else 'last', ': stamps', len(ts))
    print('  launch -> geometry/tapmask done -> loop entry:', ts[1] - ts[0])
    body = ts[2:-2]                      # triples: top, after barrier, after issue   (compute ends at the next top)
    n = (len(body)) // 3
    segs = [[body[3 * i + 1] - body[3 * i], body[3 * i + 2] - body[3 * i + 1], (body[3 * i + 3] if 3 * i + 3 < len(body) else ts[-2]) - body[3 * i + 2]] for i in range(n)]
    for i, sg in enumerate(segs[:6]):
        print('  step %d: wait+barrier %d | DMA issue %d | frag reads + MFMA %d' % (i, *sg))
    mid = segs[4:-2] if n > 8 else segs
    if mid:
        avg = [sum(s[j] for s in mid) / len(mid) for j in range(3)]
        print('  steady (steps 4..%d) avg: wait+barrier %.0f | DMA issue %.0f | frag reads + MFMA %.0f | period %.0f' % (n - 3, *avg, sum(avg)))
    print('  loop end -> epilogue done:', ts[-1] - ts[-2], ' total', ts[-1] - ts[0])
