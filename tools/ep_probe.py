"""Fused inference conv (pylc_conv2d_fwd_bnact_ex on plane tensors) per shape, isolated: the lean epilogue, the general epilogue of the same
build (pylc_debug_pp_flags 8) and the TRAINING forward of the same conv (statistics epilogue, fp32 / one-plane output) for scale.
usage: python tools/ep_probe.py [precision mode 2|3] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim, runtime
from pylc_amd import lib as L
from pylc_amd.lib import lib, check
dev = torch.device('cuda:0')
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
L.init()
check(lib.pylc_set_conv_precision(mode))
ops.PLANES_MIN_PIXELS = 0
runtime.eval_planes = True
# B, H, cin, cout, k, pad, residual
SHAPES = {2: [(32, 32, 256, 1024, 1, 0, True), (32, 32, 1024, 256, 1, 0, False), (32, 32, 256, 256, 3, 1, False), (32, 128, 64, 256, 1, 0, False), (32, 64, 128, 512, 1, 0, True)],
          3: [(8, 64, 728, 728, 1, 0, False), (8, 64, 728, 728, 1, 0, True), (8, 256, 128, 128, 1, 0, False), (8, 128, 256, 256, 1, 0, False), (8, 64, 1024, 1536, 1, 0, False), (8, 256, 256, 256, 3, 1, False)]}[mode]
def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (B, H, cin, cout, k, pad, with_res) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, 1, pad, 1, bn=True).to(dev)
    bn = layers.BatchNorm2d(cout).to(dev)
    arena = optim.FlatArena(torch.nn.ModuleList([conv, bn]))
    x = ops.empty_nhwc(B, cin, H, H, dev); x.copy_(torch.randn(B, cin, H, H, device=dev))
    res = None
    if with_res:
        res = ops.empty_nhwc(B, cout, H, H, dev); res.copy_(torch.randn(B, cout, H, H, device=dev))
    out = {}
    with torch.no_grad():
        xp = ops.to_planes(x)
        rp = ops.to_planes(res) if with_res else None
        conv.train(); bn.train()
        out['train fwd'] = timeit(lambda: ops.conv2d(xp, conv.weight, None, 1, pad, 1, want_stats=True))
        conv.eval(); bn.eval()
        for name, flags in (('general', 8), ('lean', 0)):
            lib.pylc_debug_pp_flags(flags)
            out[name] = timeit(lambda: layers.conv_bn(conv, bn, xp, residual=rp, relu=True, out_planes=True))
        lib.pylc_debug_pp_flags(0)
    gf = 2.0 * B * H * H * cin * cout * k * k / 1e3
    print('B%d %3d^2 %4d->%4d k%d %s | %s' % (B, H, cin, cout, k, 'res' if with_res else '   ', ' | '.join('%s %6.1f us %5.0f TF' % (n, t, gf / t / 1e3) for n, t in out.items())), flush=True)
    del arena
