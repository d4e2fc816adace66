"""Gradients of one training step with fp16-plane activations vs fp32 activations (same process, same weights): per-parameter
relative L2 difference.  usage: planes_vs_fp32.py <deeplab_resnet|deeplab_xception|unet>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_nets_gpu import load_golden, make_model
import pylc_amd
tag = sys.argv[1] if len(sys.argv) > 1 else 'deeplab_xception'
dev = torch.device('cuda:0')
meta_g, arr = load_golden(tag)
grads = {}
for mode in ('fp32', 'planes', 'fp32_other_tiles'):
    pylc_amd.runtime.no_planes = mode != 'planes'
    pylc_amd.lib.lib.pylc_debug_set_big_tile(0 if mode == 'fp32_other_tiles' else 2)
    model, cfg, w, x, y = make_model(meta_g, dev)
    model.net.train()
    x4 = model.pack_input(x)
    yy = model.crit(model.net(x4), model.crop_target(y.to(dev).long()))
    yy.backward()
    from pylc_amd import ops
    ops.sync_side_streams()
    torch.cuda.synchronize()
    grads[mode] = {k: p.grad.detach().clone() for k, p in model.net.named_parameters()}
    print(mode, 'loss', float(yy))
import json
gcond = meta_g.get('grad_conditioning_step0', {})
for other in ('planes', 'fp32_other_tiles'):
  worst = []
  for k in grads['fp32']:
    a, b = grads['fp32'][k].double(), grads[other][k].double()
    rel = float((a - b).norm() / (a.norm() + 1e-30))
    worst.append((rel, k, float(a.norm())))
  gmax = max(n for _, _, n in worst)
  worst = [t for t in worst if t[2] > 1e-4 * gmax]
  worst.sort(reverse=True)
  print('== fp32 (256x128 tiles) vs', other)
  for rel, k, n in worst[:int(os.environ.get('TOPN', '6'))]:
    print('%.3e  %-50s |g| %.3e   fixture ref-vs-ref noise of its L2 %.1e' % (rel, k, n, gcond.get(k, float('nan'))))
gd = meta_g['grad_digest_step0']
import math
tot = {m: math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads[m].values())) for m in grads}
print('pre-clip grad norm', tot, 'reference', meta_g['train_steps'][0]['grad_norm_preclip'])
print('%-44s %12s %12s %12s   (L2 of the clipped gradient; fixture noise)' % ('key', 'fp32', 'planes', 'reference'))
rows = []
for k in grads['fp32']:
    ref_l2 = gd[k][2]
    if ref_l2 < 1e-3 * max(v[2] for v in gd.values()):
        continue
    a = float(grads['fp32'][k].double().norm()) * min(1.0, 0.5 / tot['fp32'])
    b = float(grads['planes'][k].double().norm()) * min(1.0, 0.5 / tot['planes'])
    rows.append((abs(b - ref_l2) / ref_l2, k, a, b, ref_l2))
rows.sort(reverse=True)
for r, k, a, b, ref_l2 in rows[:12]:
    print('%-44s %12.6f %12.6f %12.6f   fp32 %+.2e planes %+.2e noise %.1e' % (k, a, b, ref_l2, a / ref_l2 - 1, b / ref_l2 - 1, gcond.get(k, float('nan'))))
print('largest gradients: key, |g| fp32, |g| planes, relative L2 of the difference')
big = sorted(((float(grads['fp32'][k].double().norm()), k) for k in grads['fp32']), reverse=True)[:14]
for n, k in big:
    a, b = grads['fp32'][k].double(), grads['planes'][k].double()
    print('%-44s %10.5f %10.5f  %.2e' % (k, n, float(b.norm()), float((a - b).norm() / a.norm())))
