"""How far do two fp32-grade conv arithmetics (f16x3 vs bf16x6, both on fp32 operands) move one training step's gradients on a
golden fixture?  (Conditioning of the fixture: the yardstick for planes-vs-fp32 differences.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_nets_gpu import load_golden, make_model
import pylc_amd
from pylc_amd import ops
tag = sys.argv[1] if len(sys.argv) > 1 else 'deeplab_xception'
dev = torch.device('cuda:0')
meta_g, arr = load_golden(tag)
pylc_amd.runtime.no_planes = True
grads = {}
for mode in (2, 1):
    pylc_amd.lib.lib.pylc_set_conv_precision(mode)
    model, cfg, w, x, y = make_model(meta_g, dev)
    model.net.train()
    loss = model.crit(model.net(model.pack_input(x)), model.crop_target(y.to(dev).long()))
    loss.backward()
    ops.sync_side_streams(); torch.cuda.synchronize()
    grads[mode] = {k: p.grad.detach().clone() for k, p in model.net.named_parameters()}
    print('mode', mode, 'loss %.9f' % float(loss.detach()))
big = sorted(((float(grads[2][k].double().norm()), k) for k in grads[2]), reverse=True)[:8]
for n, k in big:
    a, b = grads[2][k].double(), grads[1][k].double()
    print('%-44s %10.5f %10.5f  rel L2 of difference %.2e' % (k, n, float(b.norm()), float((a - b).norm() / a.norm())))
