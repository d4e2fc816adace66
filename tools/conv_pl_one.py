"""Run ONE conv shape through one kernel variant a few times (for rocprofv3 --pmc passes).
usage: conv_pl_one.py <pp|pl128|pl256|pl> B H cin cout k stride pad dil [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kind = sys.argv[1]
os.environ['PYLC_DEBUG_FLAGS'] = str({'pp': 1024, 'pl128': 1024 | 2048, 'pl256': 1024 | 8192, 'pl': 1024}[kind])
import torch
from pylc_amd import ops, layers, optim
B, H, cin, cout, k, st, pad, dil = [int(v) for v in sys.argv[2:10]]
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 10
dev = torch.device('cuda:0')
torch.manual_seed(1)
conv = layers.Conv2d(cin, cout, k, st, pad, dil, bn=True).to(dev)
arena = optim.FlatArena(conv)
x = ops.empty_nhwc(B, cin, H, H, dev)
x.copy_(torch.randn(B, cin, H, H, device=dev) * 3)
xin = x if kind == 'pp' else ops.to_planes(x)
with torch.no_grad():
    for _ in range(reps):
        y = ops.conv2d(xin, conv.weight, None, st, pad, dil, want_stats=True)
torch.cuda.synchronize()
