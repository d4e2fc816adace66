"""Run a few training steps of one net_bench configuration (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd.model import Model, Meta
dev = torch.device('cuda:0')
CFG = {
    'unet': (Meta(arch='unet', ce_weight=1.0, dice_weight=0.0, focal_weight=0.0, report=10**9), 16, 3, 512, 9),
    'xception': (Meta(backbone='xception', ch=1, n_classes=11, report=10**9), 8, 1, 1024, 11),
}
meta, b, ch, hw, ncls = CFG[sys.argv[1]]
model = Model(meta, dev).build()
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (b, ch, hw, hw)).astype(np.float32)).to(dev)
y = torch.from_numpy(np.random.RandomState(2).randint(0, ncls, (b, hw, hw)).astype(np.int64)).to(dev)
for _ in range(5): model.train(x, y)
torch.cuda.synchronize()
