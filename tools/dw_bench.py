"""Depthwise 3x3 kernels at the Aligned-Xception shapes of configuration 5 (1024^2 tiles, batch 8): achieved GB/s
against the algorithmic bytes (fwd/dgrad: one read + one write; wgrad: two reads)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pylc_amd import ops
dev = torch.device('cuda:0')
SHAPES = [(64, 512, 1), (128, 512, 1), (128, 512, 2), (128, 256, 1), (256, 256, 1), (256, 256, 2), (256, 128, 1), (728, 128, 1),
          (728, 128, 2), (728, 64, 1), (1024, 64, 1), (1536, 64, 1)]
B = int(os.environ.get('B', '8'))
for c, hw, stride in SHAPES:
    dil = 2 if c == 1536 else 1
    x = ops.empty_nhwc(B, c, hw, hw, dev).normal_().requires_grad_(True)
    w = torch.randn(c, 1, 3, 3, device=dev, requires_grad=True)
    y = ops.dwconv3x3(x, w, stride, dil)
    dy = torch.randn_like(y)
    def timed(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    tf = timed(lambda: ops.dwconv3x3(x, w, stride, dil))
    def bwd():
        x.grad = None; w.grad = None
        y.backward(dy, retain_graph=True)
    tb = timed(bwd)
    nb_f = 4.0 * (x.numel() + y.numel())
    nb_b = 4.0 * (2 * y.numel() + 2 * x.numel())          # dgrad: dy -> dx; wgrad: x, dy
    print('C=%4d %4d^2 s%d d%d  fwd %7.1f us %5.2f TB/s | dgrad+wgrad %7.1f us %5.2f TB/s' % (c, hw, stride, dil, tf * 1e6, nb_f / tf / 1e12,
                                                                                            tb * 1e6, nb_b / tb / 1e12), flush=True)
