#!/usr/bin/env python3
"""Container-only check (needs /root/reference): checkpoint files cross-load between the reference and pylc_amd.
  1. a checkpoint written by the reference's Checkpoint.save loads through pylc_amd.checkpoint (tolerant unpickler,
     optimizer state import);
  2. a file written by pylc_amd.checkpoint.save loads through the reference's Model.load and reproduces its logits."""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402
from make_golden import enter_reference, build_reference_model  # noqa: E402


def main():
    enter_reference()
    import oracle
    from oracle import step as ostep
    from tests import _data as D
    from pylc_amd import checkpoint as ck
    from pylc_amd.model import Model, Meta
    cw = D.class_weights(9)
    ref = build_reference_model('unet', 'resnet', 9, 3, ostep.PX_RGB_MEAN, ostep.PX_RGB_STD, cw, False, tile=256)
    x, y = D.tiles(1, 1, 3, 256, 256), D.blob_masks(2, 1, 256, 256, 9)
    ref.net.train()
    ref.train(x.clone(), y.clone())                       # one step so that the AdamW state exists
    ref.epoch, ref.iter = 3, 41
    ref.save()
    path = ref.checkpoint.checkpoint_file
    # torch >= 2.6 needs weights_only=False for the reference's own file
    data = ck.load_reference_file(path)
    assert data['epoch'] == 3 and data['iter'] == 41 and data['meta'].arch == 'unet'
    mine = Model(ck.meta_from_reference(data['meta'], Meta()), 'cpu').build()
    ck.load_into(mine, path, resume=True)
    for (k, v), (k2, v2) in zip(mine.net.state_dict().items(), ref.net.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k
    st = ref.optim.state_dict()['state']
    for i, p in enumerate(mine.arena.params):
        off = mine.arena.offsets[i]
        assert torch.equal(torch.as_strided(mine.optim.m, p.shape, p.stride(), off), st[i]['exp_avg'])
    assert mine.optim.steps == 1 and mine.iter == 41
    print('1. reference checkpoint -> pylc_amd: OK (%d tensors, AdamW state imported)' % len(st))

    out = os.path.join(tempfile.mkdtemp(), 'pylc_unet_ch3_schema_a.pth')
    mine.meta.id = 'pylc_unet_ch3_schema_a'
    ck.save(mine, out, best=True)
    from models.model import Model as RefModel
    torch_load = torch.load
    torch.load = lambda *a, **k: torch_load(*a, **{**k, 'weights_only': False})      # reference predates torch 2.6
    try:
        r2 = RefModel()
        from torch import nn

        class BN(nn.BatchNorm2d):
            @classmethod
            def evaluate(cls, c):
                return cls(c)
        r2.normalizers['batch'] = BN
        r2.load(out)
    finally:
        torch.load = torch_load
    r2.net.eval(); ref.net.eval()
    with torch.no_grad():
        a, b = r2.net(torch.rand(1, 3, 220, 220)), None
    for (k, v), (k2, v2) in zip(r2.net.state_dict().items(), ref.net.state_dict().items()):
        assert torch.equal(v, v2), k
    print('2. pylc_amd model file -> reference Model.load: OK (meta unpickled as %s.%s)' % (type(r2.meta).__module__, type(r2.meta).__name__))


if __name__ == '__main__':
    main()
