"""Process-wide runtime switches of the HIP path (no reference counterpart: the reference keeps the
equivalent state in the `defaults` singleton, config.py:329, which SURVEY.md appendix D.12 says the
build must replace with explicit arguments)."""
import itertools
import os


class _Runtime:
    def side_stream_on(self):
        """Whether wgrad kernels go to the side stream (see wgrad_side_stream)."""
        if self.wgrad_side_stream is None:
            from .lib import lib
            return lib.pylc_get_conv_precision() == 3
        return bool(self.wgrad_side_stream)

    def __init__(self):
        self.sync_group = None        # torch.distributed group for SyncBN / loss statistics (None = single GPU)
        # SyncBN on (default): every BatchNorm all-reduces [sum, sumsq, n] forward and [sum g, sum g xhat] backward, so N ranks compute
        # exactly what one process computes at the global batch -- 2 collectives per BatchNorm layer per step, and they cannot be
        # batched across layers (layer k+1's statistics depend on layer k's output).  PYLC_SYNC_BN=0: per-GPU statistics (what
        # torch DDP does without SyncBatchNorm): only the loss statistics and the gradient buckets cross the fabric.
        self.sync_bn = os.environ.get('PYLC_SYNC_BN', '1') != '0'
        # SyncBN: the BatchNorms of parallel branches (the ASPP's five) share one all-reduce per direction (ops.GroupBnActFn)
        self.coalesce_sync_bn = True
        # BatchNorm statistics: re-measure ill-conditioned channels (mean^2 >> var) in a second pass (bn.hip kRefineRatio); PYLC_RUNTIME=bn_refine=0
        # keeps the plain sum / sum-of-squares variance (A/B knob)
        self.bn_refine = True
        # hold a 1x1 conv's wgrad back until the next conv backward starts (ops.Conv2dFn.backward): PYLC_RUNTIME=defer_wgrad_1x1=1 (A/B knob)
        self.defer_wgrad_1x1 = False
        # per-layer wgrad launch order (a weight's `_pylc_wgrad_hold`: start its wgrad only after that many further conv backwards have begun;
        # set by nets/deeplabv3p.py for the decoder's two 3x3 convs); PYLC_RUNTIME=wgrad_hold=0 launches every wgrad right after its dgrad (A/B knob)
        self.wgrad_hold = True
        self.grad_group = None        # separate RCCL communicator for the bucketed gradient all-reduce
        self.grad_ready = None        # callable(param) invoked when a parameter gradient has been enqueued (GradBucketer.ready)
        # conv wgrad kernels on a second HIP stream (they then share the chip with the BatchNorm-backward and dgrad kernels that follow on the
        # main stream).  None = by precision mode (`side_stream_on()`): OFF for f16x3 since the wgrad takes its tiles by LDS-DMA (round 5: the
        # kernel is 25 % faster ALONE and no faster beside other kernels -- one queue: R101 +0.9 %, U-Net +0.8 %; DESIGN.md 5.2 i), ON for
        # precision mode 3, whose main queue is short kernels that a second queue fills the gaps of (one queue: -4 %).
        # PYLC_NO_SIDE_STREAM=1 / PYLC_SIDE_STREAM=1 force it.
        self.wgrad_side_stream = False if os.environ.get('PYLC_NO_SIDE_STREAM') else (True if os.environ.get('PYLC_SIDE_STREAM') else None)
        # confine the wgrad side stream to this many compute units (0 = all 256): PYLC_RUNTIME=wgrad_cus=<n>, a multiple of 8 (A/B knob)
        # (experimental: needs a library built with EXPERIMENTAL=1 -- pylc_stream_create_cu_mask)
        self.wgrad_cus = 0
        # PYLC_RUNTIME=no_relu_bits=1: BatchNorms behind a residual add re-read `out` for their ReLU mask in the backward instead of the 1-bit mask
        # their forward leaves (A/B knob)
        self.no_relu_bits = False
        # PYLC_RUNTIME=fuse_res_grad=0: BatchNorms behind a residual add write the residual's gradient out (and the block's first conv dgrad
        # accumulates into it) instead of parking (dout, mask) for that dgrad's epilogue (A/B knob)
        self.fuse_res_grad = True
        # the bias of a conv whose output a training-mode BatchNorm reads has an exactly zero gradient: no column-sum pass over dy for it
        self.skip_zero_bias_grad = True
        # precision mode 3: a BatchNorm whose only consumer is a depthwise conv leaves its apply pass to that conv's kernels (ops.bn_act(defer=))
        self.defer_bn_apply = True
        # U-Net up path: the concat of the up-sampled tensor and the bridge crop is written directly as fp16 planes (ops.upsample2_crop_concat)
        self.upcat_planes = True
        # the kernels that write a gradient no BatchNorm produces (bilinear backward, loss backward) also return its range, so the conv backward
        # reading it needs no pass of its own (PYLC_RUNTIME=fused_grad_ranges=0: stand-alone pylc_amax passes, A/B knob)
        self.fused_grad_ranges = True
        # the ASPP's image pool reads the backbone output as the fp16 planes it is (pylc_gap_fwd_planes) instead of converting it first, and the
        # Xception stem / exit BatchNorms write planes for the convs behind them (PYLC_RUNTIME=gap_planes=0: the round-2 forms, A/B knob)
        self.gap_planes = True
        # one queue: the split-K slab sums of all wgrads of a backward pass run as ONE launch before the gradients are read
        # (ops.flush_slab_sums); PYLC_RUNTIME=batch_slab_sums=0: one sum behind every wgrad (A/B knob)
        self.batch_slab_sums = True
        # PYLC_RUNTIME=fuse_bn_sums=1: a conv dgrad that writes the complete gradient of a BatchNorm output takes that BatchNorm's backward sums in its
        # epilogue (pylc_conv2d_dgrad_bn) and the BatchNorm skips its reduction pass.  Built, tested, measured NEGATIVE (the dgrad epilogue is the
        # exposed part of those kernels: BatchNorm passes 32.7 -> 28.7 ms per step, dgrads +5 ms; 381.6 vs 386.0 tiles/s): off by default.
        # Re-measured in round 5 with the epilogue rewritten (pl_epilogue_bn): still negative, 409 vs 414 tiles/s -- the BatchNorm input tile is
        # fetched in the epilogue, a chain of exposed round trips.  (experimental: needs a library built with EXPERIMENTAL=1.)
        self.fuse_bn_sums = False
        # precision mode 3 only: the tensors between the kernels -- conv / depthwise outputs, the gradients handed back to BatchNorm -- travel as
        # ONE fp16 plane (2 bytes per element) wherever producer and consumer both support it; PYLC_RUNTIME=half_acts=0 keeps them fp32 (A/B knob)
        self.half_acts = True
        # ... and the depthwise kernels' operands with them (PYLC_RUNTIME=half_dw=0: the tensors around the depthwise convs stay fp32) (A/B knob)
        self.half_dw = True
        # inference: the fused conv + BatchNorm kernels read and write fp16-plane tensors where producer and consumer allow it
        # (ops.conv_bn_act_eval_planes); PYLC_RUNTIME=eval_planes=0 keeps fp32 activations between the kernels (the round-3 path; A/B knob)
        self.eval_planes = True
        self.fuse_eval_bn = True      # inference: eval-mode BatchNorm (+ residual + ReLU) inside the conv epilogue (layers.conv_bn)
        self.bn_clamp_eps = False     # True = vendored SyncBN's clamp(var, eps)^-1/2 (batchnorm.py:125)
        self.dropout_enabled = True   # parity runs switch dropout off (RNG streams differ from torch's)
        # PYLC_RUNTIME=no_planes=1: keep every activation fp32 (the conv kernels split operands themselves) -- A/B and bit-identity tests
        self.no_planes = False
        # knobs that used to be environment switches of their own and were read in other modules (round 6: attributes, PYLC_RUNTIME overrides)
        self.planes_dy = True         # BatchNorm backward writes dy as fp16 planes for the conv dgrad behind it
        self.pool_planes = True       # the max-pool writes the next conv's plane operand
        self.fold_planes = True       # Xception: the depthwise conv's BatchNorm folded into the pointwise filter planes
        self.dw_wgrad_side = True     # precision mode 3: depthwise filter gradients on the wgrad side stream
        self.adamw_ranges = True      # AdamW takes the next step's filter ranges in the same pass
        self.grad_overlap = True      # data parallel: a bucket's all-reduce starts when its last gradient is enqueued (False: all after the backward)
        self.decoder_hold = (0, 0)    # wgrad launch holds of the decoder's two 3x3 convs (nets/deeplabv3p.py; two-queue schedules only)
        self.debug_planes = False     # print which convs take plane operands
        self.debug_streams = False    # print the side-stream probe's decision
        self.planes_only = ''         # debug: 'cin:cout:k,...' with * wildcards -- planes for these filters only
        self.planes_fwd_only = False  # debug: planes in the forward pass only
        self.comm = None              # native RCCL communicator handle (pylc_comm_init) when PYLC_COMM=native; None: torch.distributed carries the collectives
        self.grad_comm = None         # ... and the second one, for the gradient buckets (parallel.init_native_comm)
        self.shard_check = None       # pending equal-shard evidence: the reduced [sum b, sum b^2] pairs of EVERY data-parallel loss exchange since the host last looked (ops.note_shard_pair / ops.check_equal_shards)
        self.collectives = 0          # SyncBN / loss collectives issued (diagnostics: bench.py collectives_per_step)
        self.seed = 0x5EED
        self._counter = itertools.count(1)
        self._apply_overrides(os.environ.get('PYLC_RUNTIME', ''))

    def _apply_overrides(self, spec):
        """PYLC_RUNTIME='name=value,name=value': set attributes of this object from the environment -- the ONE switch for same-box A/B runs of
        the feature flags above (tools/ab_multi.sh), in place of an environment variable per flag.  Values: int, float, True / False, or text;
        an unknown name is an error (a typo would silently measure the default twice)."""
        for item in filter(None, (t.strip() for t in spec.split(','))):
            name, _, val = item.partition('=')
            if not hasattr(self, name) or name.startswith('_'):
                raise ValueError('PYLC_RUNTIME: no runtime attribute %r' % name)
            low = val.strip().lower()
            if low in ('true', 'false'):
                v = low == 'true'
            else:
                try:
                    v = int(val)
                except ValueError:
                    try:
                        v = float(val)
                    except ValueError:
                        v = val
            if isinstance(getattr(self, name), bool) and isinstance(v, int):
                v = bool(v)
            setattr(self, name, v)

    def sync_all_reduce(self, t, group):
        """The SUM all-reduce of a SyncBN / loss statistics message (counted).  With a native communicator (PYLC_COMM=native,
        parallel.init_native_comm) the message goes through the C ABI -- pylc_comm_allreduce, enqueued on the CURRENT stream: no work
        object, no stream hop --, else (or for a message the C ABI does not take: not fp32 / fp64, not contiguous) through torch.distributed."""
        self.collectives += 1
        if self.comm is not None and t.is_cuda and group is self.sync_group and self.native_takes(t):
            self.native_all_reduce(t)
            return
        import torch.distributed as dist
        dist.all_reduce(t, group=group)

    @staticmethod
    def native_takes(t):
        import torch
        return t.is_contiguous() and t.dtype in (torch.float32, torch.float64)

    def native_all_reduce(self, t, comm=None):
        """In-place SUM all-reduce of a contiguous fp32 / fp64 device tensor on the current stream through pylc_comm_allreduce
        (`comm`: the communicator; default the SyncBN / loss one)."""
        import torch
        from .lib import lib, check, ptr, stream
        if not self.native_takes(t):
            raise ValueError('native all-reduce takes contiguous fp32 / fp64 tensors, got %s %s' % (t.dtype, tuple(t.stride())))
        check(lib.pylc_comm_allreduce(comm if comm is not None else self.comm, ptr(t), t.numel(), 0 if t.dtype == torch.float32 else 1, stream()))

    def next_seed(self):
        """Distinct, reproducible seed per dropout call (rank-offset so data-parallel ranks draw different masks)."""
        return (self.seed * 0x9E3779B97F4A7C15 + next(self._counter) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def manual_seed(self, seed, rank=0):
        self.seed = (int(seed) * 1000003 + rank) & 0xFFFFFFFF
        self._counter = itertools.count(1)


runtime = _Runtime()
