"""Throughput over time within one process (does the box throttle under sustained load?)."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd.model import Model, Meta
dev = torch.device('cuda:0')
model = Model(Meta(report=10**9), dev).build()
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (32, 3, 512, 512)).astype(np.float32)).to(dev)
y = torch.from_numpy(np.random.RandomState(2).randint(0, 9, (32, 512, 512)).astype(np.int64)).to(dev)
for _ in range(3): model.train(x, y)
torch.cuda.synchronize()
def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.split(':', 2)[-1].strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'Average Graphics Package Power', 'Current Socket Graphics Package Power', 'junction'))]
        return ' | '.join(keep[:4])
    except Exception as e:
        return str(e)
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t0 = time.perf_counter()
    for _ in range(10): model.train(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('steps %3d-%3d: %.1f tiles/s   %s' % (blk * 10, blk * 10 + 9, 320 / dt, smi() if blk % 3 == 2 else ''), flush=True)
