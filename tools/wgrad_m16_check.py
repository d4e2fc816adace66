"""wgrad_pl.hip: the 16x16x32-MFMA form (pylc_debug_wgrad_m16) against the 32x32x16 form on the same operands -- largest difference relative
to the filter gradient's scale, and the time of both (GPU box).  Precision mode from PYLC_MODE (2 = f16x3, 3 = one plane)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, layers, optim
from pylc_amd import lib as L
from pylc_amd.lib import lib, check, ptr, stream

SHAPES = [  # B, H, W, Cin, Cout, k, pad, dil
    (32, 32, 32, 256, 256, 3, 1, 1), (32, 128, 128, 256, 256, 3, 1, 1), (32, 128, 128, 304, 256, 3, 1, 1), (32, 32, 32, 512, 512, 3, 2, 2),
    (32, 32, 32, 2048, 256, 3, 12, 12), (32, 32, 32, 1024, 256, 1, 0, 1), (32, 32, 32, 256, 1024, 1, 0, 1), (32, 32, 32, 2048, 512, 1, 0, 1),
    (2, 20, 44, 304, 256, 3, 1, 1), (3, 17, 23, 136, 200, 3, 1, 1), (2, 32, 32, 256, 256, 1, 0, 1),
]
mode = int(os.environ.get('PYLC_MODE', '2'))
dev = torch.device('cuda:0')
L.init()
check(lib.pylc_set_conv_precision(mode))
reps = 6
worst = 0.0
for (B, H, W, cin, cout, k, pad, dil) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, 1, pad, dil).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, W, dev); x.copy_(torch.randn(B, cin, H, W, device=dev).relu_())
    dy = ops.empty_nhwc(B, cout, H, W, dev); dy.copy_(torch.randn(B, cout, H, W, device=dev))
    xp, dyp = ops.to_planes(x), ops.to_planes(dy)
    d = ops._conv_desc(x, cin, cout, k, k, 1, pad, dil, cin, cout)
    d.x_fmt, d.dy_fmt = 1, 1
    d.x_amax, d.w_amax, d.dy_amax = ptr(ops.planes_amax(xp)), ptr(ops.weight_amax(conv.weight)), ptr(ops.planes_amax(dyp))
    nbytes = lib.pylc_conv2d_wgrad_workspace(C.byref(d))
    ws = torch.empty(max(nbytes, 4) // 4 + 1, device=dev)
    out, times = [], []
    for m16, dma in ((0, 0), (2, 0), (2, 1)):
        lib.pylc_debug_wgrad_m16(m16)
        lib.pylc_debug_wgrad_dma(dma)
        dw = torch.zeros((cout, k, k, cin), device=dev)
        check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(xp), ptr(dyp), ptr(dw), None, ptr(ws), nbytes, stream()))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(xp), ptr(dyp), ptr(dw), None, ptr(ws), nbytes, stream()))
        b.record(); torch.cuda.synchronize()
        out.append(dw); times.append(a.elapsed_time(b) / reps)
    lib.pylc_debug_wgrad_m16(0)
    lib.pylc_debug_wgrad_dma(0)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), stride=1, padding=pad, dilation=dil) if B * H * W <= 4096 else None
    scale = out[0].abs().max().item()
    diff = (out[0] - out[1]).abs().max().item() / scale
    worst = max(worst, diff)
    fl = 2.0 * B * H * W * cout * cin * k * k
    same = torch.equal(out[1], out[2])
    worst = max(worst, 0.0 if same else 1.0)
    line = '%-44s 32x32x16 %8.1f us %6.1f TF/s | 16x16x32 %8.1f us %6.1f TF/s | + LDS-DMA %8.1f us %6.1f TF/s (%s) | max |diff| / max|dw| %.2e' % (
        str((B, H, W, cin, cout, k, dil)), 1e3 * times[0], fl / times[0] / 1e9, 1e3 * times[1], fl / times[1] / 1e9,
        1e3 * times[2], fl / times[2] / 1e9, 'bit-identical' if same else 'DIFFERENT', diff)
    if ref is not None:
        r = ref.permute(0, 2, 3, 1)
        line += ' | vs fp64: %.2e / %.2e' % ((out[0].double() - r).abs().max().item() / scale, (out[1].double() - r).abs().max().item() / scale)
    print(line, flush=True)
print('worst relative difference %.2e' % worst)
assert worst < (1e-5 if mode == 2 else 1e-5), 'the two MFMA forms disagree'
