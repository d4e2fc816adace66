"""Probe of the f16x3 conv mode on the GPU: fp16 subnormal handling of the MFMA and error vs fp64."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops, lib as L
from pylc_amd.lib import lib, check, ptr, stream
L.init()
dev = torch.device('cuda:0')
def conv(x, w, mode, k=1, pad=0):
    check(lib.pylc_set_conv_precision(mode))
    b, cin, h, _ = x.shape; cout = w.shape[0]
    d = ops._conv_desc(x, cin, cout, k, k, 1, pad, 1, cin, (cout + 3) & ~3)
    y = ops.empty_nhwc(b, cout, d.OH, d.OW, dev)
    rng = (ops.amax_of(x), ops.weight_amax(w))
    d.x_amax, d.w_amax = ptr(rng[0]), ptr(rng[1])
    check(lib.pylc_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, ptr(y), stream()))
    torch.cuda.synchronize()
    return y
# 1. subnormal operands
x = torch.full((1, 16, 16, 32), 2.0 ** -20, device=dev).permute(0, 3, 1, 2)
w = torch.ones(64, 1, 1, 32, device=dev).permute(0, 3, 1, 2)
y = conv(x, w, 2)
print('subnormal A (2^-20) x 1, K=32: got', y.flatten()[0].item(), 'expected', 32 * 2.0 ** -20)
x = torch.ones((1, 16, 16, 32), device=dev).permute(0, 3, 1, 2)
w = torch.full((64, 1, 1, 32), 2.0 ** -22, device=dev).permute(0, 3, 1, 2)
y = conv(x, w, 2)
print('1 x subnormal B (2^-22): got', y.flatten()[0].item(), 'expected', 32 * 2.0 ** -22)
# 2. accuracy vs fp64
torch.manual_seed(0)
x = torch.relu(torch.randn(4, 64, 64, 256, device=dev)).permute(0, 3, 1, 2)
w = (torch.randn(256, 3, 3, 256, device=dev) * 0.02).permute(0, 3, 1, 2)
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
for mode in (0, 1, 2):
    y = conv(x, w, mode, 3, 1)
    print('mode', mode, 'max rel err vs fp64', ((y.double() - ref).abs().max() / ref.abs().max()).item(), 'l2', ((y.double() - ref).norm() / ref.norm()).item())
