"""Tile-batch data parallelism: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

The reference has no working multi-GPU path (nn.DataParallel is commented out at models/model.py:186-188 and
the vendored models/sync_batchnorm is never constructed), so the contract here is "N ranks == one process at the
global batch" (SURVEY.md section 8e):
  * BatchNorm statistics and their backward sums are all-reduced per layer (pylc_amd/ops/bn.py BnActFn) -- the
    RCCL form of batchnorm.py:48-125's master/slave reduce+broadcast;
  * the loss head all-reduces its 3+3C partial sums before the non-linear Dice / weighted-CE finalisation;
  * gradients are SUMMED over ranks (the loss is already the global mean) in a few large buckets of the flat
    gradient arena.  xGMI is point-to-point (7 links/GPU): large buckets keep every link busy and amortise the
    per-collective latency; 59.3 M fp32 gradients = 237 MB = 4 buckets of 64 MB.
"""
import os

import torch
import torch.distributed as dist

from .runtime import runtime

BUCKET_FLOATS = 16 * 1024 * 1024        # 64 MB


def init_from_env(backend=None):
    """Join the job described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world == 1 and not os.environ.get('PYLC_FORCE_PG'):
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
        return 0, 1
    # PYLC_FORCE_PG=1: create the process group even for one rank so that the SyncBN / loss / gradient all-reduce code
    # paths run (and can be tested) on a single-GPU box
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        # PYLC_DIST_BACKEND=gloo: rehearse the multi-rank path with several ranks sharing ONE GPU (RCCL refuses two ranks on a device)
        backend = os.environ.get('PYLC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend)
    runtime.sync_group = dist.group.WORLD
    # One communicator for everything by default: torch.distributed documents concurrent use of several NCCL/RCCL process
    # groups as unsafe (collectives of different communicators may start in different orders on different ranks and wait
    # for each other).  On a single communicator the bucket all-reduces and the SyncBN messages execute in program order,
    # identical on every rank; a SyncBN message can queue behind a 64 MB bucket (<= ~0.5 ms each, 4 per step).
    # PYLC_SEPARATE_GRAD_COMM=1 opts into a dedicated communicator for the buckets.
    separate = backend == 'nccl' and os.environ.get('PYLC_SEPARATE_GRAD_COMM') == '1'
    runtime.grad_group = dist.new_group(backend=backend) if separate else dist.group.WORLD
    runtime.manual_seed(runtime.seed, rank)
    # Who issues the SyncBN / loss / bucket collectives.  Default since round 6: the C ABI's own RCCL communicators (PYLC_COMM=native made the
    # default) -- one ctypes call per message that enqueues ncclAllReduce on the compute stream, no work object, no stream hop.  Measured on the
    # one-rank RCCL leg of bench.py, four cells, three boxes (profiles/r06_dp_cells.txt): the data-parallel step costs +2.2..3.0 % over the
    # group-less step with native communicators against +4.9..5.9 % through torch.distributed on the one-queue schedule (side stream: +1.1..1.4 %
    # against +3.1..3.2 %, but on a 2.3 % slower base: one queue + native is the fastest data-parallel step in absolute terms).  The hand-shake
    # falls back to torch.distributed on every rank together if RCCL cannot be reached, a communicator cannot be created or its first message
    # comes back wrong (try_native_comm).  PYLC_COMM=torch keeps torch.distributed.
    if backend == 'nccl' and os.environ.get('PYLC_COMM', 'native') == 'native':
        try_native_comm(rank, world)
    return rank, world


def _side_device():
    """Where the side-channel tensors of the hand-shake live: the GPU under torch.distributed's RCCL backend, the host under gloo."""
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def _agree(ok, world):
    """MIN over all ranks of a local outcome (1 / 0): every decision of the hand-shake below is taken on a value ALL ranks hold."""
    if world <= 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_side_device())
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


# the three local steps of the hand-shake, each a thin call into the C ABI (tests replace them to make one rank fail)
def _comm_available():
    from .lib import lib, check
    check(lib.pylc_comm_available())


def _comm_unique_id():
    import ctypes as C
    from .lib import lib, check
    buf = (C.c_char * 128)()
    check(lib.pylc_comm_unique_id(buf))
    return bytes(buf.raw)


def _comm_init(raw, rank, world):
    import ctypes as C
    from .lib import lib, check
    handle = C.c_void_p()
    check(lib.pylc_comm_init(raw, rank, world, C.byref(handle)))
    return handle


def _comm_destroy(handle):
    from .lib import lib
    lib.pylc_comm_destroy(handle)


def n_native_comms():
    return 2 if os.environ.get('PYLC_SEPARATE_GRAD_COMM', '1') != '0' else 1


def try_native_comm(rank, world):
    """PYLC_COMM=native with a fall-back: if the C ABI's communicators cannot be brought up (no loadable RCCL, ncclGetUniqueId or
    ncclCommInitRank failing on any rank), the run continues on the torch.distributed path and every rank logs why.

    The hand-shake uses torch.distributed as its side channel, and the SEQUENCE of side-channel collectives is the same on every rank
    whatever happens locally (a rank that raised before its peers' broadcast went on to an all-reduce they had not reached --
    mismatched collectives on one process group hang or corrupt):
        1. every rank probes RCCL locally (pylc_comm_available: no collective)               -> MIN all-reduce;  any 0: all ranks stop here
        2. per communicator: rank 0 makes the id (zeros if that fails) -> broadcast, ALWAYS  -> each rank joins ncclCommInitRank unless the
           id is all zeros                                                                   -> MIN all-reduce of the outcomes; any 0: all
           ranks destroy what they hold and stop.
    Every early exit is taken on an all-reduced value, so all ranks leave at the same point."""
    import sys
    handles = []
    why = ''

    def give_up(stage):
        for h in handles:
            if h is not None:
                try:
                    _comm_destroy(h)
                except Exception:
                    pass
        runtime.comm = runtime.grad_comm = None
        print('[pylc_amd] rank %d: PYLC_COMM=native unavailable (%s: %s); collectives stay on torch.distributed'
              % (rank, stage, why or 'another rank failed'), file=sys.stderr, flush=True)
        return False

    ok = 1
    try:
        _comm_available()
    except Exception as e:          # PylcError (PYLC_ERR_UNSUPPORTED: RCCL not loadable) or a ctypes failure
        ok, why = 0, '%s: %s' % (type(e).__name__, e)
    if not _agree(ok, world):
        return give_up('RCCL probe')
    for i in range(n_native_comms()):
        ident = torch.zeros(128, dtype=torch.uint8, device=_side_device())
        ok = 1
        if rank == 0:
            try:
                ident.copy_(torch.frombuffer(bytearray(_comm_unique_id()), dtype=torch.uint8))
            except Exception as e:
                ok, why = 0, '%s: %s' % (type(e).__name__, e)
                ident.zero_()                               # the peers read "no id" and skip ncclCommInitRank
        if world > 1:
            dist.broadcast(ident, 0)
        handle = None
        if bool(ident.any()):
            try:
                handle = _comm_init(bytes(ident.cpu().numpy().tobytes()), rank, world)
            except Exception as e:
                ok, why = 0, '%s: %s' % (type(e).__name__, e)
        else:
            ok = 0
        handles.append(handle)
        if not _agree(ok, world):
            return give_up('communicator %d' % i)
    # 3. the first message: a SUM of ones over each communicator, on the stream the product uses, checked on the host.  A communicator that
    #    came up but cannot carry a message (or carries a wrong one) is found HERE, where falling back is still possible -- not in step 1
    ok = 1
    try:
        _comm_first_message(handles, world)
    except Exception as e:
        ok, why = 0, '%s: %s' % (type(e).__name__, e)
    if not _agree(ok, world):
        return give_up('first message')
    runtime.comm, runtime.grad_comm = handles[0], handles[-1]
    return True


def _comm_first_message(handles, world):
    """All-reduce [1, 2] (fp32) and [1] (fp64) over every communicator and check the sums: what SyncBN, the loss head and the buckets send."""
    dev = torch.device('cuda', torch.cuda.current_device())
    for h in {id(h): h for h in handles}.values():
        a = torch.tensor([1.0, 2.0], dtype=torch.float32, device=dev)
        b = torch.ones(1, dtype=torch.float64, device=dev)
        runtime.native_all_reduce(a, h)
        runtime.native_all_reduce(b, h)
        torch.cuda.synchronize()
        got = a.cpu().tolist() + b.cpu().tolist()
        if got != [float(world), 2.0 * world, float(world)]:
            raise RuntimeError('first all-reduce over the native communicator returned %s for world size %d' % (got, world))


def init_native_comm(rank, world):
    """PYLC_COMM=native: the SyncBN / loss / gradient collectives go through the C ABI's own RCCL communicators (include/pylc_hip.h
    pylc_comm_*) instead of torch.distributed's: a collective is then ONE ctypes call that enqueues ncclAllReduce on a stream of ours -- no
    Python work object, no hop to a communication stream and back.  TWO communicators (PYLC_SEPARATE_GRAD_COMM=0: one):
      * runtime.comm       SyncBN and loss messages, enqueued on the COMPUTE stream (they sit on the critical chain);
      * runtime.grad_comm  the 64 MB gradient buckets, enqueued on a dedicated communication stream (GradBucketer) behind events of both
                           producers of a bucket -- on one communicator a 2 KB SyncBN message would queue behind a 64 MB bucket that itself
                           waits for the wgrads (stream-ordered RCCL serialises a communicator's collectives in enqueue order).
    Every rank enqueues each communicator's collectives in the same program order; the two never wait for each other on the host, and the
    compute stream waits for the bucket stream only in GradBucketer.finish(), before the optimiser.  torch.distributed stays the side
    channel that carries the communicator ids from rank 0 (and the barrier / parameter broadcast at set-up).  Raises if the communicators
    cannot be created (try_native_comm: the same hand-shake, all ranks falling back together)."""
    if not try_native_comm(rank, world):
        raise RuntimeError('PYLC_COMM=native: the RCCL communicators could not be created on every rank')


def destroy_native_comm():
    from .lib import lib
    for h in {id(h): h for h in (runtime.comm, runtime.grad_comm) if h is not None}.values():
        try:
            lib.pylc_comm_destroy(h)
        except Exception:
            pass
    runtime.comm = runtime.grad_comm = None


class GradBucketer:
    """Overlaps the gradient exchange with the backward pass.

    The flat gradient arena is cut into buckets of whole parameters (>= BUCKET_FLOATS each, in arena = module order).
    Backward produces gradients in reverse module order (decoder -> ASPP -> layer4 -> ... -> stem), so buckets complete
    from the END of the arena; the moment the last gradient of a bucket has been written (pylc_amd/ops/conv.py calls
    `ready(param)` right after launching the kernel that writes it) its SUM all-reduce is enqueued asynchronously (same
    communicator as the SyncBN messages unless PYLC_SEPARATE_GRAD_COMM=1, see init_from_env).  torch's NCCL work objects
    order each collective after the kernels already enqueued on the compute stream."""

    def __init__(self, arena, group=None, bucket_floats=BUCKET_FLOATS):
        self.arena = arena
        self.group = group
        self.buckets = []            # [lo, hi, n_params]
        lo, n = 0, 0
        for p, off in zip(arena.params, arena.offsets):
            if n and off - lo >= bucket_floats:
                self.buckets.append([lo, off, n])
                lo, n = off, 0
            p._pylc_bucket = len(self.buckets)
            n += 1
        self.buckets.append([lo, arena.numel, n])
        self.pending = [b[2] for b in self.buckets]
        self.works = []
        # runtime.grad_overlap = False: exchange all buckets after the backward pass instead of as they complete (A/B knob)
        self.overlap = bool(runtime.grad_overlap)

    def reset(self):
        self.pending = [b[2] for b in self.buckets]
        self.works = []

    def ready(self, param):
        b = getattr(param, '_pylc_bucket', None)
        if b is None:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0 and self.overlap:
            self._launch(b)

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self.pending[b] = -1
        # The bucket holds conv gradients written on the wgrad side stream and BatchNorm gradients written on the compute stream; the
        # collective must be ordered after BOTH, and neither producer stream may wait for it.
        from . import ops
        if self.arena.g.is_cuda:
            ops.flush_slab_sums(self.arena.g.device)        # conv gradients whose split-K slabs are still to be summed (one queue)
        side = ops.side_stream_if_any(self.arena.g.device)
        native = runtime.grad_comm is not None and self.arena.g.is_cuda
        if native:
            # native communicator: ncclAllReduce is stream-ordered, so it goes on a communication stream of its own that first waits for
            # an event of each producer -- on the wgrad stream itself (round 4) every later wgrad queued behind a 64 MB exchange
            if runtime.grad_comm is runtime.comm:
                # ONE communicator (PYLC_SEPARATE_GRAD_COMM=0, an A/B knob): RCCL runs a communicator's collectives in host enqueue order,
                # so its buckets must sit on the stream its SyncBN / loss messages use -- the compute stream -- or a 2 KB message would
                # queue behind a 64 MB bucket on another stream with no ordering between the two
                if side is not None:
                    torch.cuda.current_stream().wait_stream(side)
                runtime.native_all_reduce(self.arena.g[lo:hi], runtime.grad_comm)
                return
            cs = self._comm_stream()
            ev = torch.cuda.Event()
            ev.record()                                     # compute stream: BatchNorm / bias gradients of the bucket
            cs.wait_event(ev)
            if side is not None:
                ev2 = torch.cuda.Event()
                ev2.record(side)                            # wgrad stream: the conv gradients of the bucket
                cs.wait_event(ev2)
            with torch.cuda.stream(cs):
                runtime.native_all_reduce(self.arena.g[lo:hi], runtime.grad_comm)
            self._native_pending = True
            return
        if side is None:
            self.works.append(dist.all_reduce(self.arena.g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        # torch.distributed: the work object orders the collective (on RCCL's own stream) after the stream it is issued from -- the side
        # stream, once that has been told to wait for the compute stream's current position
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            self.works.append(dist.all_reduce(self.arena.g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _comm_stream(self):
        cs = getattr(self, '_cs', None)
        if cs is None:
            cs = self._cs = torch.cuda.Stream(device=self.arena.g.device)
        return cs

    def finish(self):
        """Launch whatever did not complete on its own (parameters without a gradient this step) and wait for all."""
        for b, n in enumerate(self.pending):
            if n >= 0:
                self._launch(b)
        for w in self.works:
            w.wait()
        self.works = []
        if getattr(self, '_native_pending', False):
            torch.cuda.current_stream().wait_stream(self._cs)           # the optimiser reads the reduced arena on the compute stream
            self._native_pending = False


def bucket_ranges(numel, bucket=BUCKET_FLOATS):
    return [(o, min(numel, o + bucket)) for o in range(0, numel, bucket)]


def allreduce_gradients(arena, group=None):
    """SUM-all-reduce the flat gradient arena in large buckets; returns when the reduced values are visible to the
    current stream (torch's NCCL work objects order the collective against it)."""
    works = []
    for lo, hi in bucket_ranges(arena.numel):
        works.append(dist.all_reduce(arena.g[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()


def broadcast_parameters(arena, group=None, src=0):
    """Make every rank start from rank `src`'s parameters (replicas must be identical)."""
    for lo, hi in bucket_ranges(arena.numel):
        dist.broadcast(arena.p[lo:hi], src, group=group)
    arena.refresh_ranges()          # parameter ranges and prepared conv filters follow the new values


def assert_equal_shards(local_batch, group):
    """The SyncBN / loss-head exchanges take the global count as local count x world size (ops.BnActFn, ops.MultiLossFn): every rank
    must hold the same number of tiles (a drop_last loader).  Checked once, when the data-parallel step is first set up."""
    world = dist.get_world_size(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    mine = torch.tensor([int(local_batch)], dtype=torch.int64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    sizes = [int(t.item()) for t in every]
    if len(set(sizes)) != 1:
        raise RuntimeError('data-parallel ranks hold different batch sizes %s: SyncBN and the loss head need equal shards' % sizes)


def init_single_rank_group():
    """A one-rank RCCL process group in THIS process (bench.py: what the data-parallel code path -- SyncBN and loss collectives, stream
    hops, bucketed all-reduce -- costs at world size 1)."""
    if dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29537')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.cuda.current_device()
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', dev))
    runtime.sync_group = dist.group.WORLD
    runtime.grad_group = dist.group.WORLD
    if os.environ.get('PYLC_COMM', 'native') == 'native':
        try_native_comm(0, 1)


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def barrier(group=None):
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=group)
