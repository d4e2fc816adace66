"""Host enqueue time vs GPU time per training step (is the Python side keeping the GPU queue full?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd import parallel
from pylc_amd.model import Model, Meta
rank, world = parallel.init_from_env()
dev = torch.device('cuda', torch.cuda.current_device())
model = Model(Meta(report=10**9), dev).build()
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (32, 3, 512, 512)).astype(np.float32)).to(dev)
y = torch.from_numpy(np.random.RandomState(2).randint(0, 9, (32, 512, 512)).astype(np.int64)).to(dev)
for _ in range(3): model.train(x, y)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(8):
    a = time.perf_counter(); model.train(x, y); host.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 8
print('world', world, 'forced PG' if os.environ.get('PYLC_FORCE_PG') else '', '| step %.1f ms | host enqueue per step: %s ms' % (tot * 1e3, ' '.join('%.1f' % (h * 1e3) for h in host)))
# host-only cost: same loop again but synchronising first each step so the GPU never back-pressures... (enqueue cost = time until train() returns)
host2 = []
for _ in range(4):
    torch.cuda.synchronize(); a = time.perf_counter(); model.train(x, y); host2.append(time.perf_counter() - a)
torch.cuda.synchronize()
print('   enqueue time with an empty queue: %s ms' % ' '.join('%.1f' % (h * 1e3) for h in host2))
if torch.distributed.is_initialized(): torch.distributed.destroy_process_group()
