"""conv_pl.hip (fp16-plane operands, LDS-DMA, 128x128 tiles, two blocks per CU) against conv_igemm.hip's 256x128 ping-pong kernel
(fp32 operands split in the kernel): results must be BIT-IDENTICAL (same pieces, same MFMA order), speed is reported per shape.
Usage: python tools/conv_pl_check.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PYLC_DEBUG_FLAGS', '1024')       # force the 256x128 ping-pong kernel for every N > 64 shape (same 16x16x32 MFMAs)
import torch
from pylc_amd import ops, layers, optim
from pylc_amd.lib import lib

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [  # B, H, Cin, Cout, k, stride, pad, dil
    (32, 64, 128, 128, 3, 1, 1, 1),
    (16, 64, 512, 512, 3, 1, 1, 1),
    (4, 48, 64, 64, 3, 1, 0, 1),
    (32, 32, 256, 1024, 1, 1, 0, 1),
    (32, 32, 1024, 256, 1, 1, 0, 1),
    (32, 32, 256, 256, 3, 1, 1, 1),
    (32, 32, 2048, 256, 3, 1, 12, 12),
    (32, 128, 256, 256, 3, 1, 1, 1),
    (8, 128, 304, 256, 3, 1, 1, 1),
    (32, 64, 128, 512, 1, 1, 0, 1),
    (32, 64, 128, 128, 3, 1, 1, 1),
    (8, 64, 256, 256, 3, 2, 1, 1),
    (3, 30, 72, 200, 3, 1, 1, 1),
    (2, 17, 64, 136, 3, 1, 2, 2),
]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


bad = 0
for (B, H, cin, cout, k, st, pad, dil) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, st, pad, dil, bn=True).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev)
    x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
    xp = ops.to_planes(x)
    chk = ops.from_planes(xp)
    rt = (chk - x).abs().max().item() / x.abs().max().item()
    with torch.no_grad():
        y1 = ops.conv2d(x, conv.weight, None, st, pad, dil, want_stats=True)
        y2 = ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True)
    s1 = y1._pylc_sums.double().sum(0); s2 = y2._pylc_sums.double().sum(0)
    stat_err = ((s1 - s2).abs().max() / s1.abs().max()).item()
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), None, st, pad, dil)
    err = ((y2.double() - ref).abs().max() / ref.abs().max()).item()
    with torch.no_grad():
        t1 = timeit(lambda: ops.conv2d(x, conv.weight, None, st, pad, dil, want_stats=True))
        lib.pylc_debug_pp_flags(1024 | 2048 | 16384)
        y2 = ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True)
        t2 = timeit(lambda: ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True))
        lib.pylc_debug_pp_flags(1024 | 8192 | 16384)
        y3 = ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True)
        same = (torch.equal(y1, y2), torch.equal(y1, y3))
        s3 = y3._pylc_sums.double().sum(0)
        stat_err = max(stat_err, ((s1 - s3).abs().max() / s1.abs().max()).item())
        t3 = timeit(lambda: ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True))
        lib.pylc_debug_pp_flags(1024)
        y4 = ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True)
        same = same + (torch.equal(y1, y4), (y1 - y4).abs().max().item())
        s4 = y4._pylc_sums.double().sum(0)
        stat_err = max(stat_err, ((s1 - s4).abs().max() / s1.abs().max()).item())
        t4 = timeit(lambda: ops.conv2d(xp, conv.weight, None, st, pad, dil, want_stats=True))
    oh = y1.shape[2]
    fl = 2.0 * B * oh * oh * cout * k * k * cin
    # dgrad through the same kernels: dx = conv_transpose(dy)
    dy = ops.empty_nhwc(B, cout, oh, oh, dev); dy.copy_(torch.randn(B, cout, oh, oh, device=dev))
    print('fwd  B%d %dx%d %4d->%4d k%d s%d d%-2d  bit-identical %s  stats rel %.1e  vs fp64 %.1e  planes roundtrip %.1e | pp %.1f us %.0f TF/s | pl128 %.1f us %.0f TF/s | pl256 %.1f us %.0f | auto (halo kernel where it applies) %.1f us %.0f'
          % (B, H, H, cin, cout, k, st, dil, same, stat_err, err, rt, t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9,
             t4 * 1e3, fl / t4 / 1e9), flush=True)
    bad += (not all(same[:3])) or stat_err > 1e-5 or err > 1e-5
print('FAILED' if bad else 'ALL OK')
sys.exit(1 if bad else 0)
