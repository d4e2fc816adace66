import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['PYLC_RUNTIME'] = 'debug_planes=1'
import numpy as np, torch
from pylc_amd.model import Model, Meta
dev = torch.device('cuda:0')
m = Model(Meta(report=10**9), dev).build()
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (2, 3, 256, 256)).astype(np.float32)).to(dev)
y = torch.from_numpy(np.random.RandomState(2).randint(0, 9, (2, 256, 256)).astype(np.int64)).to(dev)
m.train(x, y)
torch.cuda.synchronize()
from pylc_amd import ops
print('conversions', ops.plane_conversions, 'amax passes', ops.amax_passes)
