#!/usr/bin/env python3
"""Headline benchmark: 512x512 tiles/sec, fwd+bwd(+clip+AdamW), DeepLabV3+/ResNet101, 9 classes, bs 32 per GPU
(BASELINE.json configs[2]; weak scaling over N GPUs), synthetic tiles resident in HBM.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` / `ms_per_step` come from EXACTLY K steps of the clean
product path; a second block of instrumented steps right after it feeds
  roofline     : the dominant kernel family = the fp16-plane gather-GEMM (conv_pl.hip: implicit-GEMM conv fwd + dgrad, f16x3 split
                 arithmetic); achieved = algorithmic fp32 FLOPs (2*M*N*K with all taps counted) / its launch time
                 measured live with HIP events on the launch stream; peak = dense 16-bit MFMA
                 peak 2500 TFLOP/s / 3 (three fp16 MFMA terms per fp32 product; MI355X_MICROARCH.md).
  roofline.hbm : the HBM-bound family = the BatchNorm passes (bn.hip): algorithmic bytes per step / HIP-event time vs 8 TB/s.
  cpu_baseline : the CPU oracle (a restatement pinned bit-exactly to the reference) timed on the host cores on a
                 bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA (MI355X_MICROARCH.md); the exact-fp32 matrix pipe peaks at 157.3
# algorithmic fwd+bwd work per tile, DeepLabV3+/ResNet101 @512^2, 9 classes
GFLOP_PER_TILE_ALL = 531.40
GFLOP_PER_TILE_3X3 = 366.0


# BASELINE.json configurations that fit this contract (one line per run; c3 is the headline the driver records)
CONFIGS = {
    'c2': dict(arch='unet', backbone='resnet', ch=3, batch=16, tile=512, classes=9, precision=2, losses=(1.0, 0.0, 0.0),
               metric='512x512 tiles/sec fwd+bwd (U-Net, 9-class, CE only)', gflop_tile=673.66,
               label='U-Net valid 3x3, 3-ch 512x512 tiles -> 324x324, 9 classes, CE only, bs=16/GPU (BASELINE.json configs[1])'),
    'c3': dict(arch='deeplab', backbone='resnet', ch=3, batch=32, tile=512, classes=9, precision=2, losses=(0.5, 0.5, 0.5),
               metric='512x512 tiles/sec fwd+bwd (DeepLabV3+/ResNet101, 9-class)', gflop_tile=531.40, label=None),
    'c4': dict(arch='deeplab', backbone='resnet', ch=3, batch=32, tile=512, classes=11, precision=2, losses=(0.5, 0.5, 0.5),
               metric='512x512 tiles/sec fwd+bwd (DeepLabV3+/ResNet101, 11-class schema_b)', gflop_tile=531.45,
               label='DeepLabV3+ ResNet101 OS16, 3-ch 512x512 tiles, 11 classes, bs=32/GPU (BASELINE.json configs[3]: the per-GPU shard of global bs 256)'),
    # configs[4]: Xception, grayscale 1024^2, bs 8 per GPU, 16-bit MFMA operands with fp32 accumulation and fp32 master weights (precision
    # mode 3: activations between BatchNorm and conv as ONE fp16 plane, 2 bytes per element)
    'c5': dict(arch='deeplab', backbone='xception', ch=1, batch=8, tile=1024, classes=11, precision=3, losses=(0.5, 0.5, 0.5),
               metric='1024x1024 tiles/sec fwd+bwd (DeepLabV3+/Xception, 1-ch, fp16 MFMA)', gflop_tile=1984.0,
               label='DeepLabV3+ Aligned-Xception OS16, 1-ch 1024x1024 tiles, 11 classes, bs=8/GPU, fp16 MFMA operands / fp32 accumulate (BASELINE.json configs[4], training leg)'),
}


def synth(rank, b, ch, hw, n_cls, dev):
    x = torch.from_numpy(np.random.RandomState(1234 + rank).randint(0, 256, (b, ch, hw, hw)).astype(np.float32)).to(dev)
    y = torch.from_numpy(np.random.RandomState(4321 + rank).randint(0, n_cls, (b, hw, hw)).astype(np.int64)).to(dev)
    return x, y


def _side_stream_on():
    from pylc_amd.runtime import runtime
    return runtime.side_stream_on()


def csrc_fingerprint():
    """sha256 over the kernel sources: ties a committed PMC profile to the build it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'pylc_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'pylc_amd', 'csrc', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def _family_matcher(kernel):
    """Predicate over rocprofv3 kernel names for a family label such as 'gg_pl_kernel<3,*> + gg_plh_kernel<3>': a member is any
    instantiation of one of the named templates whose FIRST template argument (NTERMS) is the label's -- matched on the name before the
    template list plus that argument, never on a substring of the whole argument list (round 4's substring match silently dropped
    'gg_plh_kernel<3, false, false>')."""
    import re
    pats = []
    for part in kernel.split('+'):
        m = re.match(r'\s*(\w+)\s*<\s*(\d+)', part)
        if m is None:
            raise ValueError('kernel family label %r: expected name<NTERMS...>' % part)
        pats.append(re.compile(r'(?<![\w])%s<%s[,>]' % (re.escape(m.group(1)), m.group(2))))
    return lambda name: any(p.search(''.join(name.split())) for p in pats)


def _pmc_family(pattern, kernel, per_launch_key, launches_per_step, need_key=None):
    """Launch-weighted mean of `per_launch_key` over the family's rows of the newest committed profile matching `pattern`, with its
    provenance.  Fails loudly if the family's launch count in the profile is not a whole number of steps of `launches_per_step` launches
    (i.e. if the name match and the live launch count of the KernelTimer disagree about what the family is)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    if not files:
        return None, None
    prof = json.load(open(files[-1]))
    member = _family_matcher(kernel)
    tot = n = 0.0
    rows = []
    for name, v in prof.items():
        if name.startswith('_') or (need_key is not None and need_key not in v) or not member(name):
            continue
        tot += v[per_launch_key] * v['launches']
        n += v['launches']
        rows.append(name)
    meta = prof.get('_meta', {})
    src = {'file': os.path.relpath(files[-1], ROOT), 'measured_on_csrc_sha256': meta.get('csrc_sha256'),
           'this_build_csrc_sha256': csrc_fingerprint(), 'launches_in_profile': int(n), 'rows': len(rows)}
    src['same_build'] = src['measured_on_csrc_sha256'] == src['this_build_csrc_sha256']
    if launches_per_step:
        steps = n / launches_per_step
        src['profile_steps'] = steps
        if n == 0 or abs(steps - round(steps)) > 1e-9:
            raise RuntimeError('%s: the family %r has %d launches in the profile, not a multiple of the %g launches per step the live timer '
                               'counted -- the name match and the timer disagree (rows: %s)' % (src['file'], kernel, n, launches_per_step, rows))
    return (tot / n if n else None), src


def pmc_traffic(kernel, launches_per_step=None):
    """HBM bytes per launch of the kernel family `kernel` from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc_bench.sh -> profiles/*_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE exact).  PMC collection
    needs the profiler, so it cannot be measured live here: the value comes with its provenance -- the profile file, the kernel-source
    fingerprint it was taken on and whether that is THIS build's.  (None, None) if no profile is committed."""
    return _pmc_family('*_pmc_traffic.json', kernel, 'hbm_bytes_per_launch', launches_per_step)


def pmc_mfma(kernel, launches_per_step=None):
    """Matrix-pipe busy fraction of the family (launch-weighted over its instantiations) from the committed SQ-counter pass of this same
    command (tools/pmc_mfma.sh -> profiles/*_pmc_mfma.json: SQ_VALU_MFMA_BUSY_CYCLES over kernel cycles x 256 CUs x 4 SIMDs), with the
    same provenance fields as pmc_traffic().  The counters serialise the kernels, so this is the kernels' ISOLATED matrix-pipe
    occupancy at the clock the board holds, not a share of the overlapped step.  (None, None) if no profile is committed."""
    return _pmc_family('*_pmc_mfma.json', kernel, 'mfma_busy_frac', launches_per_step, need_key='mfma_busy_frac')


def cpu_baseline(hw, n_cls, budget_s=25.0):
    """Reference CPU path (oracle) on this box's host cores: bs=2 steps of the same train step, bounded in time."""
    import oracle
    from oracle import step as ostep
    # the GPU box shows 128 host threads but a one-GPU job's CPU share is 16: at 128 threads oneDNN's workers mostly wait for each other
    # (measured on the box, bs 2: 0.31 tiles/s at 128 threads, 0.84 at 64, 1.43 at 32, 2.39 at 16 -- tools/cpu_threads.py)
    cores = min(torch.get_num_threads(), 16)
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    b = 2
    cfg = ostep.StepConfig('deeplab', 'resnet', n_cls, 3)
    sd = oracle.init_state(oracle.state_spec('deeplab', 'resnet', n_cls, 3), seed=0)
    opt = ostep.make_optimizer(sd, cfg)
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (b, 3, hw, hw)).astype(np.float32))
    y = torch.from_numpy(np.random.RandomState(2).randint(0, n_cls, (b, hw, hw)).astype(np.int64))
    t0 = time.time()
    ostep.train_step(sd, opt, cfg, x, y)          # warm-up (allocator, oneDNN primitive cache)
    warm = time.time() - t0
    times = []
    while sum(times) + warm < budget_s and len(times) < 5:
        t0 = time.time()
        ostep.train_step(sd, opt, cfg, x, y)
        times.append(time.time() - t0)
    if not times:
        times = [warm]
    med = float(np.median(times))
    torch.set_num_threads(prev_threads)
    try:
        import psutil
        phys, logical = psutil.cpu_count(logical=False), psutil.cpu_count(logical=True)
    except Exception:
        phys, logical = None, os.cpu_count()
    return {'value': b / med, 'unit': 'tiles/s', 'cores': cores, 'host_physical_cores': phys, 'host_logical_cpus': logical, 'kind': 'port',
            'sample': 'CPU oracle (restatement pinned bit-exactly to the reference), DeepLabV3+/R101 %dx%d bs=%d full '
                      'train step, %d timed step(s) after 1 warm-up, median' % (hw, hw, b, len(times))}


PEAK_HBM_TBS = 8.0                 # HBM3E (MI355X_MICROARCH.md)


def hbm_roofline(bn_records, n_steps):
    """SURVEY.md section 8d's HBM half: the BatchNorm passes (apply, backward reduce, backward apply -- the step's HBM-bound kernel
    family, bn.hip) measured live with HIP events on their launch stream (ops.bn_timing), per step.  bytes = algorithmic: every fp32 /
    fp16-plane element a pass has to touch once (4 B; 2 B for one-plane tensors in mode 3), 1/8 B per ReLU-mask bit."""
    by = {}
    tot_ms = tot_b = 0.0
    n = 0
    for kind, m, c, nbytes, a, e in bn_records:
        ms = a.elapsed_time(e)
        k = kind.split('(')[0].split('+')[0]
        v = by.setdefault(k, [0.0, 0.0, 0])
        v[0] += ms; v[1] += nbytes; v[2] += 1
        tot_ms += ms; tot_b += nbytes; n += 1
    ach = tot_b / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    return {'bound': 'hbm', 'kernel': 'bn_apply_kernel + bn_reduce_kernel + bn_bwd_apply_kernel (bn.hip)', 'achieved': ach, 'peak': PEAK_HBM_TBS,
            'unit': 'TB/s', 'frac': ach / PEAK_HBM_TBS, 'bytes_per_step': tot_b / n_steps, 'ms_per_step': tot_ms / n_steps,
            'launches_per_step': n / n_steps,
            'note': 'achieved = algorithmic bytes / HIP-event time of the passes INSIDE the step (config.wgrad_queue: on the compute stream nothing '
                    'runs beside them; a wgrad side stream shares the chip with the backward passes); per shape: profiles/*_bn_table_serial.txt',
            'by_pass': {k: {'ms_per_step': v[0] / n_steps, 'GB_per_step': v[1] / n_steps / 1e9,
                            'TBps': v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 else 0.0, 'launches_per_step': v[2] / n_steps} for k, v in by.items()}}


def inference_leg(args, cfg, model, rank, world, dev):
    """configs[4]'s second half: the reference's test driver (test.py:50-110) on one full-resolution image -- tiles cut and normalised on the
    device, Model.test semantics per batch of 8 tiles, overlap blend + argmax (utils/tools.py:209-319) -- through
    pylc_amd.inference.predict_image.  N > 1: replicas, the tile batches of the image dealt round-robin over the ranks, rank 0 gathers
    and stitches.  value = tiles of the image / wall time per image (whole job)."""
    import pylc_amd
    from pylc_amd import inference, parallel
    h, w, tile, stride, batch = 3072, 4096, args.tile, args.tile // 2, 8        # pylc_gpu.ipynb:1057-1063; test.py:63,69
    rs = np.random.RandomState(99)
    img = torch.from_numpy(rs.randint(0, 256, (cfg['ch'], h, w)).astype(np.float32)).to(dev)
    rows, cols = inference.tile_grid(h, w, tile, stride)
    group = pylc_amd.runtime.sync_group if world > 1 else None
    model.net.eval()
    for _ in range(max(args.warmup, 2)):
        inference.predict_image(model, img, tile, stride, batch, group=group)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mask = inference.predict_image(model, img, tile, stride, batch, group=group)
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from pylc_amd import ops as _ops
    _ops.eval_plane_convs[0] = 0
    inference.predict_image(model, img, tile, stride, batch, group=group)
    plane_convs = _ops.eval_plane_convs[0]
    # roofline leg: one more image with HIP events around the fused conv + BatchNorm launches (after the timed region)
    roof = None
    if not args.no_kernel_timing:
        from pylc_amd import ops
        timer = ops.KernelTimer(inference=True)
        ops.set_kernel_timer(timer)
        inference.predict_image(model, img, tile, stride, batch, group=group)
        ops.set_kernel_timer(None)
        roof = timer.roofline(PEAK_BF16_MFMA_TFLOPS)
        roof['note'] += '; measured on ONE extra image after the timed region (the timed images run without these HIP-event brackets)'
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    n_tiles = rows * cols
    prec = pylc_amd.lib.lib.pylc_get_conv_precision()
    line = {
        'metric': '1024x1024 tiles/sec sliding-window inference (DeepLabV3+/Xception, 1-ch, full-res 3072x4096 image)',
        'value': n_tiles * args.steps / dt, 'unit': 'tiles/s', 'n_gpus': world, 'steps': args.steps, 'warmup': max(args.warmup, 2),
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f16' if prec == 3 else 'f32', 'data': 'synthetic',
        'config': {'workload': 'DeepLabV3+ Aligned-Xception OS16, inference only: one 1-ch %dx%d image -> %d x %d = %d tiles of %d^2 at stride %d, '
                               'batches of %d, overlap blend + argmax on the device (BASELINE.json configs[4], inference leg; test.py:50-110)'
                               % (h, w, rows, cols, n_tiles, tile, stride, batch),
                   'parallelism': 'replicas x%d (tile batches round-robin, gather to rank 0)' % world,
                   'conv_arithmetic': {0: 'f32 MFMA', 1: 'bf16x6 split', 2: 'f16x3 split (fp32-grade)',
                                       3: 'fp16 operands (ONE plane per operand, rounded inside the fused conv kernel), fp32 accumulation, fp32 weights'}[prec]
                                      + inference_format_note(),
                   'plane_convs_per_image': plane_convs,
                   'mask_checksum': int(mask.to(torch.int64).sum().item()), 'pixels_per_s': h * w * args.steps / dt},
    }
    if roof is not None:
        line['roofline'] = roof
    print(json.dumps(line), flush=True)


def inference_format_note():
    """What travels between the kernels of the inference path, derived from the switches the run used (so the label cannot drift from the code)."""
    import pylc_amd
    from pylc_amd import ops
    if pylc_amd.runtime.eval_planes and not pylc_amd.runtime.no_planes and ops.eval_plane_convs[0] > 0:
        return ('; activations between the kernels: fp16 planes (%d B/element) written by the fused conv + eval-BatchNorm + residual + ReLU epilogues '
                '(pylc_conv2d_fwd_bnact_ex)%s' % (2 * ops.nplanes(), ', depthwise convs half -> half (pylc_dwconv3x3_fwd_h_eval)' if ops.nplanes() == 1 else ''))
    return '; activations between the kernels: fp32 (conv + eval BatchNorm + residual + ReLU fused in the conv epilogue, depthwise strips fp32)'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='tiles per GPU')
    ap.add_argument('--tile', type=int, default=512)
    ap.add_argument('--classes', type=int, default=9)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-dp-overhead', action='store_true', help='N=1: skip the extra steps that time the data-parallel code path at world size 1')
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS), help='BASELINE.json configuration (default c3 = configs[2], the headline)')
    ap.add_argument('--inference', action='store_true', help='with --config c5: the inference leg of configs[4] -- sliding-window prediction of a '
                    'synthetic full-resolution 3072x4096 image (1024^2 tiles, stride 512, batches of 8: test.py:50-110); a step = one image')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.config != 'c3':
        args.batch, args.tile, args.classes = cfg['batch'], cfg['tile'], cfg['classes']
        os.environ.setdefault('PYLC_CONV_PRECISION', str(cfg['precision']))

    import pylc_amd
    from pylc_amd import parallel, ops
    from pylc_amd.model import Model, Meta
    rank, world = parallel.init_from_env()
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    dev = torch.device('cuda', torch.cuda.current_device())
    torch.manual_seed(0)
    w_ce, w_dice, w_focal = cfg['losses']
    meta = Meta(arch=cfg['arch'], backbone=cfg['backbone'], ch=cfg['ch'], n_classes=args.classes, report=10 ** 9,
                ce_weight=w_ce, dice_weight=w_dice, focal_weight=w_focal)
    model = Model(meta, dev).build()
    if world > 1:
        parallel.broadcast_parameters(model.arena)
    if args.inference:
        return inference_leg(args, cfg, model, rank, world, dev)
    x, y = synth(rank, args.batch, cfg['ch'], args.tile, args.classes, dev)

    # Settle (untimed, before the W warm-up steps): a process started right after another GPU job may see that job's
    # memory teardown for a few seconds (measured: back-to-back launches lose up to 35 % for some runs, 3 s apart none).
    # Run steps until three consecutive ones agree within 2 % of the fastest seen, at most 40.
    # A step also only counts once the caching allocator is steady: tensors that the side-stream wgrad still reads are handed
    # back with record_stream, so their blocks are reusable late and the pool keeps growing (hipMalloc inside the step) for the
    # first handful of steps.
    best, streak = float('inf'), 0
    for _ in range(40):
        n_malloc = torch.cuda.memory_stats().get('num_device_alloc', 0)
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        model.train(x, y)
        torch.cuda.synchronize()
        d_s = time.perf_counter() - t_s
        best = min(best, d_s)
        steady = torch.cuda.memory_stats().get('num_device_alloc', 0) == n_malloc
        streak = streak + 1 if (d_s <= 1.02 * best and steady) else 0
        done = streak >= 3
        if world > 1:       # every rank must run the same number of steps (each step contains collectives): stop only when all agree
            flag = torch.tensor([0.0 if done else 1.0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            done = flag.item() == 0.0
        if done:
            break
    for _ in range(args.warmup):
        model.train(x, y)
    ops.amax_passes[:] = [0, 0]
    ops.plane_conversions[:] = [0, 0]
    ops.planes_marked[0] = 0
    # ---- the timed region: EXACTLY `steps` steps of the clean product path, no instrumentation (the live HIP-event brackets of the
    #      roofline leg cost 4-5 % of the step: they run in a second block below) ----------------------------------------------------
    mallocs = torch.cuda.memory_stats().get('num_device_alloc', 0)
    pylc_amd.runtime.collectives = 0
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train(x, y)
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mallocs = torch.cuda.memory_stats().get('num_device_alloc', 0) - mallocs
    clean_collectives = pylc_amd.runtime.collectives
    range_passes, conversions = ops.amax_passes[0] / args.steps, ops.plane_conversions[0] / args.steps
    planes_marked = ops.planes_marked[0] / args.steps
    # ---- the instrumented block (roofline): the same steps again with HIP events around every launch of the dominant conv family
    #      (ops.KernelTimer) and around every BatchNorm pass (ops.bn_timing), on the streams the kernels are launched on.  Its wall time is
    #      reported (roofline.instrumented_ms_per_step) but is NOT the headline. -------------------------------------------------------
    timer = None
    dt_instr = None
    if not args.no_kernel_timing:
        timer = ops.KernelTimer()
        ops.set_kernel_timer(timer)
        ops.bn_timing = []
        n_instr = max(2, min(args.steps, 10))
        parallel.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_instr):
            model.train(x, y)
        parallel.barrier()
        torch.cuda.synchronize()
        dt_instr = (time.perf_counter() - t1) / n_instr
        ops.set_kernel_timer(None)
        bn_records, ops.bn_timing = ops.bn_timing, None
    pylc_amd.runtime.collectives = clean_collectives
    n_buckets = len(model._bucketer.buckets) if model._bucketer is not None else 0
    collectives = pylc_amd.runtime.collectives / args.steps + n_buckets if world > 1 else 0.0
    # N = 1: what the data-parallel code path costs before any fabric is involved -- the same model, a one-rank RCCL group switched on
    # (SyncBN + loss collectives, their stream hops, the bucketed gradient all-reduce), a few more steps
    dp_overhead = dp_collectives = dp_comm = None
    if world == 1 and not args.no_dp_overhead and not torch.distributed.is_initialized():
        try:
            # RCCL prints its version banner to STDOUT when the group comes up: keep this process's stdout the one JSON line
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                parallel.init_single_rank_group()
                x_ = torch.zeros(1, device=dev)
                torch.distributed.all_reduce(x_)          # (the banner comes with the first collective)
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
            for _ in range(3):
                model.train(x, y)
            pylc_amd.runtime.collectives = 0
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_dp = max(3, args.steps // 2)
            for _ in range(n_dp):
                model.train(x, y)
            torch.cuda.synchronize()
            dp_overhead = (time.perf_counter() - t1) / n_dp / (dt / args.steps) - 1.0
            dp_collectives = pylc_amd.runtime.collectives / n_dp + len(model._bucketer.buckets)
            dp_comm = 'native' if pylc_amd.runtime.comm is not None else 'torch'
        except Exception as e:          # no RCCL on this box: report that rather than fail the benchmark
            dp_overhead = 'unavailable: %s' % str(e)[:80]
        pylc_amd.runtime.sync_group = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    losses = model.loss.flush()
    # which transport the collectives of the TIMED steps went through (read before the group is torn down)
    dist_backend = torch.distributed.get_backend() if (world > 1 and torch.distributed.is_initialized()) else None
    comm_kind = None if world == 1 else ('native' if pylc_amd.runtime.comm is not None else 'torch')
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    tiles = args.batch * world * args.steps
    value = tiles / dt
    out = {
        'metric': cfg['metric'],
        'value': value, 'unit': 'tiles/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f16' if pylc_amd.lib.lib.pylc_get_conv_precision() == 3 else 'f32', 'data': 'synthetic',
        'config': {'workload': cfg['label'] or ('DeepLabV3+ ResNet101 OS16, 3-ch %dx%d tiles, %d classes, CE+Dice+Focal, clip 0.5 + AdamW, '
                                                'bs=%d/GPU (BASELINE.json configs[2])' % (args.tile, args.tile, args.classes, args.batch)),
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                   'net_tflops_algorithmic': value * cfg['gflop_tile'] / 1e3 / world,
                   'conv_arithmetic': {0: 'f32 MFMA', 1: 'bf16x6 split', 2: 'f16x3 split (fp32-grade)',
                                       3: 'fp16 operands (one plane, 2 B/element), fp32 accumulation, fp32 master weights'}[pylc_amd.lib.lib.pylc_get_conv_precision()],
                   # data-parallel diagnostics (so that a scaling run is checkable): ranks on RCCL, collectives issued per step and rank
                   # (SyncBN forward + backward per BatchNorm layer, loss statistics, gradient buckets), and at N = 1 the cost of that
                   # code path at world size 1
                   # rccl_ranks = ranks whose collectives ran on RCCL (0 for a gloo rehearsal or a single process); dist_backend = the
                   # torch.distributed backend of the timed steps; comm = who issued the SyncBN / loss / bucket collectives ('torch':
                   # torch.distributed work objects; 'native': the C ABI's own RCCL communicator, pylc_comm_*)
                   'rccl_ranks': world if (dist_backend == 'nccl' or comm_kind == 'native') else 0,
                   'dist_backend': dist_backend, 'comm': comm_kind,
                   # where the conv filter gradients run (pylc_amd.runtime.side_stream_on: one queue for f16x3 since round 5, DESIGN.md 5.2 i)
                   'wgrad_queue': 'side stream' if _side_stream_on() else 'compute stream',
                   'sync_bn': bool(pylc_amd.runtime.sync_bn),
                   'collectives_per_step': collectives if world > 1 else dp_collectives,
                   'dp_codepath_overhead': dp_overhead,
                   # ... and who issued that leg's collectives: 'native' = the C ABI's RCCL communicators (the default since round 6), 'torch' = torch.distributed
                   'dp_codepath_comm': dp_comm,
                   'standalone_range_passes_per_step': range_passes,
                   'activation_format': ('fp32' if (pylc_amd.runtime.no_planes or planes_marked == 0) else
                                         'fp16 planes between BatchNorm and conv kernels (%d B/element)' % (2 * ops.nplanes())),
                   'plane_tensors_per_step': planes_marked,      # 0: this network's convs (bias) stay on the fp32-operand kernels
                   'planes_to_fp32_conversions_per_step': conversions,
                   'hipmalloc_calls_in_timed_region': mallocs,      # 0 in steady state (diagnostic: see DESIGN.md section 5.2, open observation)
                   'last_loss': [float(v) for v in losses[-1]] if losses else None},
    }
    if timer is not None:
        out['roofline'] = timer.roofline(PEAK_BF16_MFMA_TFLOPS)
        out['roofline']['instrumented_ms_per_step'] = 1e3 * dt_instr
        out['roofline']['note'] += ('; measured in a SECOND block of %d instrumented steps after the timed region (value / ms_per_step are the clean '
                                    'steps without these HIP-event brackets; instrumented_ms_per_step is this block)' % n_instr)
        out['roofline']['hbm'] = hbm_roofline(bn_records, n_instr)
        if args.config in ('c3', 'c4'):        # the committed PMC profiles are of the DeepLab/R101 workload: not attached to other networks' lines
            per_step = out['roofline']['launches'] / n_instr
            out['roofline']['launches_per_step'] = per_step
            out['roofline']['traffic'], out['roofline']['traffic_source'] = pmc_traffic(out['roofline']['kernel'], per_step)
            out['roofline']['mfma_busy_frac'], out['roofline']['mfma_busy_source'] = pmc_mfma(out['roofline']['kernel'], per_step)
        else:
            out['roofline']['traffic'] = None
    if world == 1 and not args.no_cpu_baseline and args.config == 'c3':
        out['cpu_baseline'] = cpu_baseline(args.tile, args.classes)
    print(json.dumps(out), flush=True)
    del model, x, y
    torch.cuda.empty_cache()          # hand the 27 GB back before exit: a following launch does not run into the teardown


if __name__ == '__main__':
    main()
