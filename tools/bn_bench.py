"""HBM throughput of the BatchNorm kernels in isolation (GPU box): GB/s per kernel on ResNet-sized activations."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, lib as L
from pylc_amd.lib import lib, check, ptr, stream
L.init()
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (b, c, h) in ((32, 256, 128), (32, 64, 128), (32, 1024, 32), (32, 256, 32), (32, 64, 256)):
    m = b * h * h
    y = torch.randn(b, h, h, c, device=dev).permute(0, 3, 1, 2)
    res = torch.randn(b, h, h, c, device=dev).permute(0, 3, 1, 2)
    dout = torch.randn(b, h, h, c, device=dev).permute(0, 3, 1, 2)
    out = ops.empty_nhwc(b, c, h, h, dev); dy = ops.empty_nhwc(b, c, h, h, dev); g_out = ops.empty_nhwc(b, c, h, h, dev)
    coef = torch.rand(4 * c, device=dev) + 0.5
    mean, invstd, scale, shift = coef[:c], coef[c:2*c], coef[2*c:3*c], coef[3*c:]
    gamma = torch.ones(c, device=dev)
    sums = torch.empty(2 * c + 1, device=dev)
    ws = torch.empty(lib.pylc_bn_workspace_floats(m, c), device=dev)
    amax = torch.zeros(1, dtype=torch.int32, device=dev)
    gb = m * c * 4 / 1e9
    t = timeit(lambda: check(lib.pylc_bn_apply(ptr(y), c, ptr(scale), ptr(shift), None, 0, ptr(out), c, m, c, 1, ptr(amax), stream())))
    line = '[%d,%d,%d,%d] %.2f GB/tensor | apply %.0f us %.2f TB/s' % (b, c, h, h, gb, t * 1e3, 2 * gb / t)
    t = timeit(lambda: check(lib.pylc_bn_apply(ptr(y), c, ptr(scale), ptr(shift), ptr(res), c, ptr(out), c, m, c, 1, ptr(amax), stream())))
    line += ' | apply+res %.0f us %.2f TB/s' % (t * 1e3, 3 * gb / t)
    t = timeit(lambda: check(lib.pylc_bn_stats(ptr(y), m, c, c, ptr(sums), ptr(ws), stream())))
    line += ' | stats %.0f us %.2f TB/s' % (t * 1e3, gb / t)
    t = timeit(lambda: check(lib.pylc_bn_bwd_reduce(ptr(dout), c, None, 0, ptr(y), c, ptr(mean), ptr(invstd), m, c, 1, ptr(sums), ptr(ws), ptr(scale), ptr(shift), stream())))
    line += ' | bwd_reduce %.0f us %.2f TB/s' % (t * 1e3, 2 * gb / t)
    t = timeit(lambda: check(lib.pylc_bn_bwd_apply(ptr(dout), c, None, 0, ptr(y), c, ptr(mean), ptr(invstd), ptr(gamma), ptr(sums), float(m), m, c, 1, ptr(dy), c, None, 0, ptr(amax), ptr(scale), ptr(shift), stream())))
    line += ' | bwd_apply %.0f us %.2f TB/s' % (t * 1e3, 3 * gb / t)
    t = timeit(lambda: check(lib.pylc_bn_bwd_apply(ptr(dout), c, ptr(out), c, ptr(y), c, ptr(mean), ptr(invstd), ptr(gamma), ptr(sums), float(m), m, c, 1, ptr(dy), c, ptr(g_out), c, ptr(amax), None, None, stream())))
    line += ' | bwd_apply(res) %.0f us %.2f TB/s' % (t * 1e3, 5 * gb / t)
    t = timeit(lambda: torch.add(y, res, out=out))
    line += ' | torch add %.0f us %.2f TB/s' % (t * 1e3, 3 * gb / t)
    print(line, flush=True)
