"""PCIe-inclusive throughput of the headline configuration: every step's tiles and masks start in HOST memory (uint8, as the reference's
database stores them) and cross PCIe through data.TileFeeder (pinned double-buffered staging on a copy stream), against the bench's
HBM-resident batch.   usage: python tools/feed_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pylc_amd.model import Model, Meta
from pylc_amd.data import TileFeeder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device('cuda:0')
model = Model(Meta(report=10 ** 9), dev).build()
rs = np.random.RandomState(1)
host = [(rs.randint(0, 256, (32, 3, 512, 512)).astype(np.uint8), rs.randint(0, 9, (32, 512, 512)).astype(np.uint8)) for _ in range(4)]
x = torch.from_numpy(host[0][0]).to(dev)
y = torch.from_numpy(host[0][1]).to(dev).long()
for _ in range(8):
    model.train(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.train(x, y)
torch.cuda.synchronize()
res = (time.perf_counter() - t0) / steps
batches = [host[i % 4] for i in range(steps + 4)]
feeder = iter(TileFeeder(batches, dev))
for _ in range(4):
    xb, yb = next(feeder)
    model.train(xb, yb)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for xb, yb in feeder:
    model.train(xb, yb)
    n += 1
torch.cuda.synchronize()
fed = (time.perf_counter() - t0) / n
mb = (host[0][0].nbytes + host[0][1].nbytes) / 1e6
print('HBM-resident batch: %.2f ms/step  %.1f tiles/s | fed from host over PCIe (%.0f MB per step, uint8 tiles + uint8 masks): %.2f ms/step  %.1f tiles/s'
      % (res * 1e3, 32 / res, mb, fed * 1e3, 32 / fed))
