"""One training step's loss and gradients in precision mode 3 (one fp16 plane, 16-bit MFMA operands) against mode 2 (f16x3, fp32-grade) on the
HIP path itself, at a configuration of choice -- how much of a mode-3 deviation is operand rounding, how much the problem's conditioning.
usage: python tools/mode3_vs_mode2.py [--backbone xception] [--batch 4] [--tile 256] [--ch 1] [--init kaiming|formula]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backbone', default='xception')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--tile', type=int, default=256)
    ap.add_argument('--ch', type=int, default=1)
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
    spec = oracle.state_spec('deeplab', a.backbone, a.classes, 3)
    w0 = oracle.init_state(spec, seed=12) if a.init == 'kaiming' else oracle.formula_state(spec, salt=4)
    x3, y = D.learnable_tiles(2000, a.batch, a.tile, a.classes)
    x = x3[:, :a.ch].contiguous()
    out = {}
    for mode in (2, 3):
        pylc_amd.lib.lib.pylc_set_conv_precision(mode)
        model = Model(Meta(backbone=a.backbone, ch=a.ch, n_classes=a.classes), dev).build()
        model.net.load_state_dict(w0)
        model.net.train()
        logits = model.net(model.pack_input(x))
        loss = model.crit(logits, model.crop_target(y.to(dev).long()))
        loss.backward()
        ops.sync_side_streams()
        torch.cuda.synchronize()
        out[mode] = (float(loss.detach()), logits.detach().float().clone(), {k: p.grad.detach().double().clone() for k, p in model.net.named_parameters()})
        print('mode %d: loss %.7f' % (mode, out[mode][0]))
        del model
    l2, l3 = out[2][1], out[3][1]
    print('logits max|diff| %.3g (|logits| max %.3g); argmax agreement %.5f' % ((l2 - l3).abs().max().item(), l2.abs().max().item(),
                                                                              (l2.argmax(1) == l3.argmax(1)).float().mean().item()))
    g2, g3 = out[2][2], out[3][2]
    n2 = sum(float(g.pow(2).sum()) for g in g2.values()) ** 0.5
    n3 = sum(float(g.pow(2).sum()) for g in g3.values()) ** 0.5
    dot = sum(float((g2[k] * g3[k]).sum()) for k in g2)
    print('gradient norm mode 2 %.5f mode 3 %.5f ; cosine over all parameters %.6f' % (n2, n3, dot / (n2 * n3)))
    cos = sorted((float((g2[k] * g3[k]).sum() / (g2[k].norm() * g3[k].norm() + 1e-300)), k) for k in g2 if g2[k].numel() >= 64 and float(g2[k].norm()) > 1e-6 * n2)
    print('per-tensor cosine: min %.5f (%s), 5th percentile %.5f, median %.5f' % (cos[0][0], cos[0][1], cos[len(cos) // 20][0], cos[len(cos) // 2][0]))


if __name__ == '__main__':
    main()
