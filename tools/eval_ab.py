"""Inference throughput with fp16-plane tensors between the kernels (runtime.eval_planes, pylc_conv2d_fwd_bnact_ex) against fp32 tensors
(PYLC_RUNTIME=eval_planes=0, the round-3 path): Model.test on a resident batch of the BASELINE shapes.   usage: python tools/eval_ab.py [names...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd.model import Model, Meta
from pylc_amd import ops, runtime
from pylc_amd import lib as L
from pylc_amd.lib import lib, check
dev = torch.device('cuda:0')
CFG = {
    'r101_512_bs32': (Meta(report=10**9), 32, 3, 512, 2),
    'unet_512_bs16': (Meta(arch='unet', report=10**9), 16, 3, 512, 2),
    'xception_1024_gray_bs8_f16x3': (Meta(backbone='xception', ch=1, n_classes=11, report=10**9), 8, 1, 1024, 2),
    'xception_1024_gray_bs8_mode3': (Meta(backbone='xception', ch=1, n_classes=11, report=10**9), 8, 1, 1024, 3),
}
L.init()
for name in (sys.argv[1:] or list(CFG)):
    meta, b, ch, hw, prec = CFG[name]
    check(lib.pylc_set_conv_precision(prec))
    model = Model(meta, dev).build()
    model.net.eval()
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (b, ch, hw, hw)).astype(np.float32)).to(dev)
    res = {}
    with torch.no_grad():
        for rep in range(2):
            for on in ((False, True) if 'PYLC_EVAL_ONLY' not in os.environ else (os.environ['PYLC_EVAL_ONLY'] == '1',) * 2):
                runtime.eval_planes = on
                for _ in range(3): out = model.test(x)[0]
                ops.eval_plane_convs[0] = 0; ops.plane_conversions[:] = [0, 0]; ops.amax_passes[:] = [0, 0]
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n = 20
                for _ in range(n): out = model.test(x)[0]
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
                res.setdefault(on, []).append((dt, ops.eval_plane_convs[0] // n, ops.plane_conversions[0] // n, ops.amax_passes[0] // n, out.float().clone()))
    res.setdefault(True, res.get(False)); res.setdefault(False, res.get(True))
    d = (res[True][0][4] - res[False][0][4]).abs().max().item()
    agree = (res[True][0][4].argmax(1) == res[False][0][4].argmax(1)).float().mean().item()
    t0, t1 = min(r[0] for r in res[False]), min(r[0] for r in res[True])
    print('%-30s fp32 tensors %7.1f ms %7.1f tiles/s | plane tensors %7.1f ms %7.1f tiles/s (%d plane convs, %d conversions, %d range passes) | '
          'max |logit diff| %.2e, argmax agreement %.5f' % (name, 1e3 * t0, b / t0, 1e3 * t1, b / t1, res[True][0][1], res[True][0][2], res[True][0][3], d, agree), flush=True)
    del model, x
    import gc; gc.collect(); torch.cuda.empty_cache()
