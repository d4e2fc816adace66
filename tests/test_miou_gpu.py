"""mIoU parity after N training steps (BASELINE.json: "mIoU within +-0.1 of reference after N steps").

Both paths start from the same reference-law initialisation, train N steps of Model.train on the same learnable
synthetic tiles WITH dropout live (so the trajectories differ by their RNG streams -- the check is statistical) and
are scored with the reference's metric (weighted Jaccard incl. the Evaluator coverage quirk) on held-out tiles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_STEPS, B, HW, NCLS, LR = 40, 4, 64, 9, 1e-3


def test_miou_parity_after_training(dev):
    import oracle
    from oracle import step as ostep
    from pylc_amd.model import Model, Meta
    from pylc_amd import runtime
    from tests import _data as D
    runtime.dropout_enabled = True
    runtime.manual_seed(7)
    torch.manual_seed(7)
    spec = oracle.state_spec('deeplab', 'resnet', NCLS, 3)
    w0 = oracle.init_state(spec, seed=11)
    batches = [D.learnable_tiles(1000 + i, B, HW, NCLS) for i in range(N_STEPS)]
    xv, yv = D.learnable_tiles(5000, 8, HW, NCLS)

    # --- HIP path ---
    model = Model(Meta(lr=LR), dev).build()
    model.net.load_state_dict(w0)
    for x, y in batches:
        model.train(x, y)
    model.net.eval()
    pred = model.test(xv)[0].argmax(1).cpu().numpy()
    miou_hip = oracle.weighted_jaccard(yv.numpy(), pred, NCLS)
    # batch-statistics forward (dropout off): after only N steps the running statistics lag behind the weights, so the
    # reference's eval-mode score is still near chance in BOTH paths; this second score shows that learning happened
    runtime.dropout_enabled = False
    model.net.train()
    with torch.no_grad():
        pred_b = model.net(model.pack_input(xv)).argmax(1).cpu().numpy()
    miou_hip_b = oracle.weighted_jaccard(yv.numpy(), pred_b, NCLS)
    last_hip = [float(model.crit.ce), float(model.crit.dsc), float(model.crit.fl)]

    # --- CPU oracle (reference semantics) ---
    cfg = ostep.StepConfig('deeplab', 'resnet', NCLS, 3, lr=LR, dropout=True)
    sd = {k: v.clone() for k, v in w0.items()}
    opt = ostep.make_optimizer(sd, cfg)
    for x, y in batches:
        o = ostep.train_step(sd, opt, cfg, x, y)
    pred_o = ostep.test_step(sd, cfg, xv).argmax(1).numpy()
    miou_ref = oracle.weighted_jaccard(yv.numpy(), pred_o, NCLS)
    cfg.dropout = False
    xin, _ = ostep._prep(cfg, xv)
    with torch.no_grad():
        pred_ob = ostep.forward(sd, cfg, xin, True).argmax(1).numpy()
    miou_ref_b = oracle.weighted_jaccard(yv.numpy(), pred_ob, NCLS)
    print('mIoU after %d steps: eval-mode HIP %.4f / oracle %.4f ; batch-stat HIP %.4f / oracle %.4f | last losses HIP %s oracle %s'
          % (N_STEPS, miou_hip, miou_ref, miou_hip_b, miou_ref_b, last_hip, list(o[:3])))
    assert abs(miou_hip - miou_ref) <= 0.1
    assert abs(miou_hip_b - miou_ref_b) <= 0.1
    assert miou_hip_b > 0.3 and miou_ref_b > 0.3      # both actually learned (chance is ~0.06)
    runtime.dropout_enabled = True
