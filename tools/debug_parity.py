"""Debug helper (GPU box): per-block activations and per-parameter gradients of the HIP path vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from oracle import step as ostep
from oracle import loss as oloss
from pylc_amd.model import Model, Meta
from pylc_amd import runtime
from tests import _data as D

arch, backbone, ch, ncls, b, hw = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
training = len(sys.argv) > 7 and sys.argv[7] == 'train'
runtime.dropout_enabled = False
dev = torch.device('cuda:0')
cfg = ostep.StepConfig(arch, backbone, ncls, ch, dropout=False)
spec = oracle.state_spec(arch, backbone, ncls, 3 if arch == 'deeplab' else ch)
x = D.tiles(100, b, ch, hw, hw)
y = D.blob_masks(101, b, hw, hw, ncls, cell=8)
w = ostep.calibrate_bn(oracle.formula_state(spec, salt=1), cfg, x.clone())
model = Model(Meta(arch=arch, backbone=backbone, ch=ch, n_classes=ncls), dev).build()
model.net.load_state_dict(w)
model.net.train(training)

taps_mine = {}
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        taps_mine[name] = o.detach().float().cpu()
    return f
for name, mod in model.net.named_modules():
    if name.count('.') <= 2 and name:
        mod.register_forward_hook(hook(name))

sd = {k: v.clone() for k, v in w.items()}
opt = ostep.make_optimizer(sd, cfg)
taps = {}
xin, yy = ostep._prep(cfg, x.clone(), y.clone())
ref_logits = ostep.forward(sd, cfg, xin, training, taps)
x4 = model.pack_input(x)
logits = model.net(x4)
print('logits max|diff| %.3g  (|ref| max %.3g)' % ((logits.detach().float().cpu() - ref_logits.detach()).abs().max().item(), ref_logits.abs().max().item()))
for k, t in taps.items():
    if k in taps_mine:
        m = taps_mine[k]
        t = t.detach()
        if m.shape == t.shape:
            print('  tap %-32s max|diff| %.3g  rel %.3g' % (k, (m - t).abs().max().item(), (m - t).abs().max().item() / (t.abs().max().item() + 1e-30)))
if training:
    tot, ce, dsc, fl = oloss.multiloss(ref_logits, yy)
    tot.backward()
    yd = model.crop_target(y.to(dev))
    loss = model.crit(logits, yd)
    loss.backward()
    print('loss mine %.7f ref %.7f' % (loss.item(), tot.item()))
    rows = []
    for k, p in model.net.named_parameters():
        g, r = p.grad.float().cpu(), sd[k].grad
        rows.append((((g - r).abs().max() / (r.abs().max() + 1e-12)).item(), k, r.abs().max().item(), g.abs().max().item()))
    rows.sort(reverse=True)
    for rel, k, rm, gm in rows[:25]:
        print('  grad %-48s rel %.3g  |ref|max %.3g |mine|max %.3g' % (k, rel, rm, gm))
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.net.parameters())).item()
    rn = torch.sqrt(sum((sd[k].grad.double() ** 2).sum() for k, _ in model.net.named_parameters())).item()
    print('grad norm mine %.6f ref %.6f' % (gn, rn))
