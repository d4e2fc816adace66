"""Run a few resident-batch inference passes (Model.test) of one configuration, for rocprofv3 --kernel-trace --stats.
usage: python tools/profile_eval.py [r101|unet|xception] [passes]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd.model import Model, Meta
dev = torch.device('cuda:0')
CFG = {
    'r101': (Meta(report=10**9), 32, 3, 512),
    'unet': (Meta(arch='unet', report=10**9), 16, 3, 512),
    'xception': (Meta(backbone='xception', ch=1, n_classes=11, report=10**9), 8, 1, 1024),
}
meta, b, ch, hw = CFG[sys.argv[1] if len(sys.argv) > 1 else 'r101']
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = Model(meta, dev).build()
model.net.eval()
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (b, ch, hw, hw)).astype(np.float32)).to(dev)
with torch.no_grad():
    for _ in range(n): out = model.test(x)[0]
torch.cuda.synchronize()
