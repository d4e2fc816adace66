"""CPU oracle throughput (DeepLabV3+/R101 512x512 bs 2 train step) at several thread counts on the GPU box: the basis of bench.py cpu_baseline.cores."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import oracle
from oracle import step as ostep
for nt in (128, 64, 32, 16):
    torch.set_num_threads(nt)
    cfg = ostep.StepConfig('deeplab', 'resnet', 9, 3)
    sd = oracle.init_state(oracle.state_spec('deeplab', 'resnet', 9, 3), seed=0)
    opt = ostep.make_optimizer(sd, cfg)
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (2, 3, 512, 512)).astype(np.float32))
    y = torch.from_numpy(np.random.RandomState(2).randint(0, 9, (2, 512, 512)).astype(np.int64))
    ostep.train_step(sd, opt, cfg, x, y)
    ts = []
    for _ in range(2):
        t0 = time.time(); ostep.train_step(sd, opt, cfg, x, y); ts.append(time.time() - t0)
    print('threads', nt, 'tiles/s %.3f' % (2 / min(ts)), flush=True)
