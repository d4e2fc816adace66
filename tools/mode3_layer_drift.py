"""Where does a mode-3 (one fp16 plane) training-mode forward drift from the mode-2 (f16x3) one?  Relative L2 difference of every
top-level block's output.   usage: python tools/mode3_layer_drift.py [--backbone resnet] [--batch 8] [--tile 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backbone', default='resnet')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--tile', type=int, default=256)
    ap.add_argument('--classes', type=int, default=11)
    ap.add_argument('--init', default='kaiming')
    a = ap.parse_args()
    import oracle
    import pylc_amd
    from pylc_amd import ops, runtime
    from pylc_amd.model import Model, Meta
    from tests import _data as D
    dev = torch.device('cuda:0')
    runtime.dropout_enabled = False
    ops.PLANES_MIN_PIXELS = 0
    ch = 3 if a.backbone == 'resnet' else 1
    spec = oracle.state_spec('deeplab', a.backbone, a.classes, 3)
    w0 = oracle.init_state(spec, seed=12) if a.init == 'kaiming' else oracle.formula_state(spec, salt=4)
    x3, y = D.learnable_tiles(2000, a.batch, a.tile, a.classes)
    x = x3[:, :ch].contiguous()
    acts = {}
    for mode in (2, 3):
        pylc_amd.lib.lib.pylc_set_conv_precision(mode)
        model = Model(Meta(backbone=a.backbone, ch=ch, n_classes=a.classes), dev).build()
        model.net.load_state_dict(w0)
        model.net.train()
        rec = {}

        def hook(name):
            def f(mod, inp, out):
                t = out[0] if isinstance(out, (tuple, list)) else out
                rec[name] = ops.as_nhwc(t).detach().float().clone()
            return f
        hs = []
        for name, mod in model.net.named_modules():
            if name.count('.') <= 2 and name and not list(mod.children()) == [] and ('layer' in name or 'block' in name or name in ('aspp', 'decoder', 'backbone')):
                hs.append(mod.register_forward_hook(hook(name)))
            elif name in ('backbone.conv1', 'backbone.bn1', 'backbone.maxpool', 'backbone.conv2', 'backbone.bn2'):
                hs.append(mod.register_forward_hook(hook(name)))
        out = model.net(model.pack_input(x))
        rec['logits'] = out.detach().float().clone()
        acts[mode] = rec
        for h in hs:
            h.remove()
        del model
    for k in acts[2]:
        p, q = acts[2][k].double(), acts[3][k].double()
        print('%-32s shape %-22s rel L2 diff %.3e   max|diff|/max %.3e' % (k, tuple(p.shape), float((p - q).norm() / (p.norm() + 1e-30)),
                                                                      float((p - q).abs().max() / (p.abs().max() + 1e-30))))


if __name__ == '__main__':
    main()
