"""HBM-side (L2-miss) traffic and time of the fp16-plane wgrad kernel per shape and rasterisation (pylc_debug_wgrad_flags): run under
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/wgrad_traffic.py <flags>
and read the counter CSV with tools/wgrad_traffic.py --parse <dir> (prints fetched MB per launch against the operand bytes).
Without the profiler it prints times only.   flags: 0 product order, 1 no XCD remap, 2 split fastest"""
import ctypes as C
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # B, H, Cin, Cout, k, pad
    (32, 32, 256, 256, 3, 1), (32, 128, 256, 256, 3, 1), (32, 128, 304, 256, 3, 1), (32, 64, 128, 128, 3, 1), (32, 128, 64, 64, 3, 1), (32, 32, 512, 512, 3, 1), (32, 32, 2048, 256, 3, 1), (32, 32, 1024, 256, 1, 0), (32, 32, 256, 1024, 1, 0), (32, 32, 2048, 512, 1, 0),
]

if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    rows = []
    for f in glob.glob(sys.argv[2] + '/*/*counter_collection.csv'):
        per = {}
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == 'FETCH_SIZE' and 'wgrad_pl_kernel' in r['Kernel_Name']:
                per[int(r['Dispatch_Id'])] = per.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
        rows = [per[k] for k in sorted(per)]
    reps = len(rows) // len(SHAPES)
    for i, (B, H, cin, cout, k, pad) in enumerate(SHAPES):
        mb = 2.0 * 1024 * sum(rows[i * reps:(i + 1) * reps]) / max(reps, 1) / 1e6          # FETCH_SIZE in KB, x2 (gfx950)
        alg = 4.0 * B * H * H * (cin + cout) / 1e6
        print('%-28s fetched %8.1f MB per launch, operands %7.1f MB: x %.2f' % (str((B, H, cin, cout, k)), mb, alg, mb / alg))
    sys.exit(0)

import torch
from pylc_amd import ops, layers, optim
from pylc_amd import lib as L
from pylc_amd.lib import lib, check, ptr, stream
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = 6
dev = torch.device('cuda:0')
L.init()
check(lib.pylc_set_conv_precision(2))
lib.pylc_debug_wgrad_flags(flags & 15)
lib.pylc_debug_wgrad_max_steps(flags >> 4)          # flags = raster | (max K-steps << 4)
for (B, H, cin, cout, k, pad) in SHAPES:
    torch.manual_seed(1)
    conv = layers.Conv2d(cin, cout, k, 1, pad, 1).to(dev)
    arena = optim.FlatArena(conv)
    x = ops.empty_nhwc(B, cin, H, H, dev); x.copy_(torch.randn(B, cin, H, H, device=dev))
    dy = ops.empty_nhwc(B, cout, H, H, dev); dy.copy_(torch.randn(B, cout, H, H, device=dev))
    xp, dyp = ops.to_planes(x), ops.to_planes(dy)
    d = ops._conv_desc(x, cin, cout, k, k, 1, pad, 1, cin, cout)
    d.x_fmt, d.dy_fmt = 1, 1
    d.x_amax, d.w_amax, d.dy_amax = ptr(ops.planes_amax(xp)), ptr(ops.weight_amax(conv.weight)), ptr(ops.planes_amax(dyp))
    nbytes = lib.pylc_conv2d_wgrad_workspace(C.byref(d))
    ws = torch.empty(max(nbytes, 4) // 4 + 1, device=dev)
    dw = torch.empty((cout, k, k, cin), device=dev)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        check(lib.pylc_conv2d_wgrad(C.byref(d), ptr(xp), ptr(dyp), ptr(dw), None, ptr(ws), nbytes, stream()))
    b.record(); torch.cuda.synchronize()
    fl = 2.0 * B * H * H * cout * cin * k * k
    t = a.elapsed_time(b) / reps
    print('flags %d %-28s %7.1f us per wgrad (+ split-K reduce) %6.1f TF/s' % (flags, str((B, H, cin, cout, k)), 1e3 * t, fl / t / 1e9), flush=True)
