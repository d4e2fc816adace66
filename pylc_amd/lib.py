"""ctypes binding of libpylc_hip.so (the C ABI declared in include/pylc_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  Importing this module
without the built library raises, and every op raises if it is handed a non-HIP tensor.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYLC_LIB: load another build of the library (same-box A/B of two builds: tools/ab_builds.sh)
LIB_PATH = os.environ.get('PYLC_LIB') or os.path.join(_HERE, 'libpylc_hip.so')

ABI_VERSION = 13


class PylcError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'H', 'W', 'Cin', 'Cout', 'R', 'S', 'stride', 'pad', 'dil',
                                       'OH', 'OW', 'x_pitch', 'y_pitch')] + \
               [(n, C.c_void_p) for n in ('x_amax', 'w_amax', 'dy_amax',      # operand ranges (precision mode 2)
                                          'w_planes', 'w_planes_t')] + \
               [(n, C.c_int) for n in ('x_fmt', 'dy_fmt', 'out_fmt')] + [('out_bound', C.c_void_p)] + \
               [('w_planes_fmt', C.c_int)]      # operand / output formats: 0 fp32, 1 fp16 planes; filter-plane layout bits (chunk-interleaved)


class WPrepEntry(C.Structure):
    _fields_ = [('src_offset', C.c_longlong), ('fwd_offset', C.c_longlong), ('t_offset', C.c_longlong),
                ('tile_begin', C.c_longlong), ('K', C.c_int), ('RS', C.c_int), ('C', C.c_int), ('amax_index', C.c_int)]


class SlabSum(C.Structure):
    """PylcSlabSum: one pending split-K slab sum (pylc_conv2d_wgrad_slabs -> pylc_splitk_reduce_batch)."""
    _fields_ = [('slabs', C.c_void_p), ('dw', C.c_void_p), ('n4', C.c_longlong), ('slab_stride', C.c_longlong), ('splits', C.c_int), ('reserved', C.c_int)]


class BnExtra(C.Structure):
    """PylcBnExtra: fp16-plane operands / fused dropout of the BatchNorm *_ex entry points."""
    _fields_ = [('out_planes', C.c_void_p), ('out_plane_stride', C.c_longlong), ('out_bound', C.c_void_p),
                ('res_planes', C.c_void_p), ('res_plane_stride', C.c_longlong), ('res_amax', C.c_void_p),
                ('dy_planes', C.c_void_p), ('dy_plane_stride', C.c_longlong), ('dy_bound', C.c_void_p),
                ('nplanes', C.c_int), ('drop_p', C.c_float), ('drop_seed', C.c_uint64), ('g_amax', C.c_void_p), ('relu_mask', C.c_void_p), ('y_half_bound', C.c_void_p), ('dout_half_bound', C.c_void_p)]


class BnBack(C.Structure):
    """PylcBnBack: the BatchNorm whose backward sums a conv dgrad takes in its epilogue (pylc_conv2d_dgrad_bn)."""
    _fields_ = [('y', C.c_void_p), ('mean', C.c_void_p), ('invstd', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('relu_mask', C.c_void_p), ('relu', C.c_int), ('g_amax', C.c_void_p)]


class FwdEp(C.Structure):
    """PylcFwdEp: the fused inference epilogue of pylc_conv2d_fwd_bnact_ex (plane residual / plane output / true input range)."""
    _fields_ = [('scale', C.c_void_p), ('shift', C.c_void_p), ('scale_amax', C.c_void_p), ('shift_amax', C.c_void_p), ('residual', C.c_void_p),
                ('res_fmt', C.c_int), ('res_scale_bound', C.c_void_p), ('res_amax', C.c_void_p), ('x_true_amax', C.c_void_p), ('relu', C.c_int),
                ('amax_out', C.c_void_p)]


class DwDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'H', 'W', 'C', 'stride', 'dil', 'OH', 'OW', 'x_pitch', 'y_pitch')]


_P = C.c_void_p
_I = C.c_int
_LL = C.c_longlong
_F = C.c_float
_D = C.c_double
_SZ = C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/pylc_hip.h (tests check this)
SIGNATURES = {
    'pylc_last_error': (C.c_char_p, []),
    'pylc_abi_version': (_I, []),
    'pylc_planes_stride': (_LL, [_LL, _I, _I]),
    'pylc_set_planes_interleave': (_I, [_I]),
    'pylc_experimental_build': (_I, []),
    'pylc_init': (_I, []),
    'pylc_set_conv_precision': (_I, [_I]),
    'pylc_get_conv_precision': (_I, []),
    'pylc_debug_set_big_tile': (_I, [_I]),
    'pylc_debug_pp_flags': (_I, [_I]),
    'pylc_debug_stagger': (_I, [_I]),
    'pylc_debug_wgrad_flags': (_I, [_I]),
    'pylc_debug_wgrad_max_steps': (_I, [_I]),
    'pylc_comm_available': (_I, []),
    'pylc_comm_unique_id': (_I, [_P]),
    'pylc_comm_init': (_I, [_P, _I, _I, C.POINTER(C.c_void_p)]),
    'pylc_comm_allreduce': (_I, [_P, _P, _LL, _I, _P]),
    'pylc_comm_syncbn_reduce': (_I, [_P, _P, _I, _P]),
    'pylc_comm_destroy': (_I, [_P]),
    'pylc_debug_dw_tiles': (_I, [_I]),
    'pylc_range_product': (_I, [_P, _P, _F, _P, _P, _P]),
    'pylc_maxpool_fwd_planes': (_I, [_P, _P, _LL, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pylc_upsample2_crop_concat_planes': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _LL, _I, _P, _P]),
    'pylc_conv1x1_fold_input_affine': (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    'pylc_debug_wgrad_acc1': (_I, [_I]),
    'pylc_debug_wgrad_sets': (_I, [_I]),
    'pylc_debug_wgrad_m16': (_I, [_I]),
    'pylc_debug_wgrad_dma': (_I, [_I]),
    'pylc_debug_pp_stamps': (_I, [_P]),
    'pylc_amax': (_I, [_P, _LL, _I, _I, _P, _P]),
    'pylc_amax_segments': (_I, [_P, _P, _I, _P, _P]),
    'pylc_to_planes': (_I, [_P, _I, _P, _I, _LL, _LL, _I, _P, _I, _P]),
    'pylc_from_planes': (_I, [_P, _I, _LL, _P, _I, _LL, _I, _P, _I, _P]),
    'pylc_planes_colsum_workspace_floats': (C.c_size_t, [_I]),
    'pylc_planes_colsum': (_I, [_P, _I, _LL, _I, _P, _LL, _I, _P, _P, _P]),
    'pylc_weight_prepare': (_I, [_P, _P, _I, _LL, _P, _P, _I, _P]),
    'pylc_conv2d_dgrad_needs_f32_weights': (_I, [C.POINTER(ConvDesc)]),
    'pylc_conv2d_fwd': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    'pylc_conv2d_fwd_bnact': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    'pylc_conv2d_fwd_bnact_ex': (_I, [C.POINTER(ConvDesc), _P, _P, _P, C.POINTER(FwdEp), _P, _P]),
    'pylc_conv2d_fwd_stats_floats': (_SZ, [C.POINTER(ConvDesc)]),
    'pylc_conv2d_fwd_stats': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.POINTER(_I), _P]),
    'pylc_conv2d_dgrad': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P]),
    'pylc_conv2d_dgrad_add': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _P, _P]),
    'pylc_relu_bwd_bits': (_I, [_P, _P, _P, _LL, _I, _P]),
    'pylc_conv2d_dgrad_bn_floats': (_SZ, [C.POINTER(ConvDesc)]),
    'pylc_conv2d_dgrad_bn': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _P, C.POINTER(BnBack), _P, C.POINTER(_I), _P]),
    'pylc_bn_bwd_sums_from_partial': (_I, [_P, _I, _I, _P, _P, _P, _D, _P, _P, _P]),
    'pylc_conv2d_wgrad_workspace': (_SZ, [C.POINTER(ConvDesc)]),
    'pylc_conv2d_wgrad': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _SZ, _P]),
    'pylc_conv2d_wgrad_slabs': (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _SZ, C.POINTER(SlabSum), _P]),
    'pylc_splitk_reduce_batch': (_I, [_P, _P, _I, C.c_longlong, _P]),
    'pylc_weight_transpose': (_I, [_P, _P, _I, _I, _I, _P]),
    'pylc_dwconv3x3_fwd': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P]),
    'pylc_dwconv3x3_fwd_stats_rows': (_I, [C.POINTER(DwDesc)]),
    'pylc_dwconv3x3_fwd_stats': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _P]),
    'pylc_dwconv3x3_dgrad': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P]),
    'pylc_dwconv3x3_dgrad_acc': (_I, [C.POINTER(DwDesc), _P, _P, _P, _I, _P]),
    'pylc_dwconv3x3_wgrad_workspace': (_SZ, [C.POINTER(DwDesc)]),
    'pylc_dwconv3x3_wgrad': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _SZ, _P]),
    'pylc_dwconv3x3_half_ok': (_I, [C.POINTER(DwDesc)]),
    'pylc_dwconv3x3_fwd_h_stats_rows': (_I, [C.POINTER(DwDesc)]),
    'pylc_dwconv3x3_bn_ok': (_I, [C.POINTER(DwDesc)]),
    'pylc_dwconv3x3_fwd_h_bn': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    'pylc_dwconv3x3_wgrad_h_bn': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    'pylc_dwconv3x3_dgrad_h_add_ok': (_I, [C.POINTER(DwDesc)]),
    'pylc_dwconv3x3_dgrad_h_add': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'pylc_dwconv3x3_fwd_h': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'pylc_dwconv3x3_fwd_h_eval': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'pylc_dwconv3x3_dgrad_h': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'pylc_dwconv3x3_wgrad_h': (_I, [C.POINTER(DwDesc), _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'pylc_bn_workspace_floats': (_SZ, [_LL, _I]),
    'pylc_bn_stats': (_I, [_P, _LL, _I, _I, _P, _P, _P]),
    'pylc_bn_stats_from_partial': (_I, [_P, _I, _I, _P, _P]),
    'pylc_bn_finalize': (_I, [_P, _D, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P]),
    'pylc_bn_eval_coeffs': (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    'pylc_bn_eval_coeffs_full': (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P]),
    'pylc_bn_finalize_from_partial': (_I, [_P, _I, _D, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P]),
    'pylc_bn_finalize_ex': (_I, [_P, _D, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _LL, _P, _P]),
    'pylc_bn_local_moments': (_I, [_P, _D, _I, _P, _I, _LL, _P, _P, _P]),
    'pylc_bn_finalize_moments': (_I, [_P, _D, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P]),
    'pylc_bn_finalize_from_partial_ex': (_I, [_P, _I, _D, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _LL, _P, _P]),
    'pylc_bn_apply_ex': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _LL, _I, _I, _P, C.POINTER(BnExtra), _P]),
    'pylc_bn_bwd_reduce_ex': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _LL, _I, _I, _P, _P, _P, _P, _P, _D, C.POINTER(BnExtra), _P, _P]),
    'pylc_bn_bwd_bound': (_I, [_P, _P, _P, _D, _I, _P, _P, _P]),
    'pylc_bn_bwd_apply_ex': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _D, _LL, _I, _I, _P, _I, _P, _I, _P, _P, _P,
                                  C.POINTER(BnExtra), _P]),
    'pylc_bn_apply': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _LL, _I, _I, _P, _P]),
    'pylc_bn_bwd_reduce': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _LL, _I, _I, _P, _P, _P, _P, _P]),
    'pylc_bn_bwd_apply': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _D, _LL, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P]),
    'pylc_relu_fwd': (_I, [_P, _I, _P, _I, _LL, _I, _P]),
    'pylc_relu_bwd': (_I, [_P, _I, _P, _I, _P, _I, _LL, _I, _P]),
    'pylc_maxpool_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pylc_maxpool_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pylc_maxpool_bwd_add': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    'pylc_crop_copy': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    'pylc_bilinear_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pylc_bilinear_bwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'pylc_bilinear_bwd_workspace': (C.c_size_t, [_I, _I, _I, _I]),
    'pylc_bilinear_bwd_separable': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'pylc_gap_fwd': (_I, [_P, _P, _I, _I, _I, _P]),
    'pylc_gap_fwd_planes': (_I, [_P, _LL, _I, _P, _P, _I, _I, _I, _P]),
    'pylc_gap_bwd': (_I, [_P, _P, _I, _I, _I, _P]),
    'pylc_gap_bwd_acc': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'pylc_image_pack': (_I, [_P, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P, _P]),
    'pylc_image_pack_tiles': (_I, [_P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P, _P]),
    'pylc_stitch_argmax': (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'pylc_colourize_resize': (_I, [_P, _I, _I, _P, _P, _I, _I, _P]),
    'pylc_image_pack_denom': (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _F, _P, _P]),
    'pylc_image_pack_u8': (_I, [_P, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P, _P]),
    'pylc_confusion_matrix': (_I, [_P, _I, _P, _I, _LL, _I, _I, _P, _P]),
    'pylc_nhwc_to_nchw': (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    'pylc_nchw_to_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'pylc_multiloss_workspace_floats': (_SZ, [_LL, _I]),
    'pylc_multiloss_stats': (_I, [_P, _I, _P, _LL, _I, _P, _P, _P, _P]),
    'pylc_multiloss_finalize': (_I, [_P, _D, _I, _F, _F, _F, _P, _P]),
    'pylc_multiloss_bwd': (_I, [_P, _I, _P, _LL, _I, _P, _P, _D, _F, _F, _F, _P, _P, _I, _P, _P]),
    'pylc_sqnorm_workspace_floats': (_SZ, [_LL]),
    'pylc_grad_norm_clip': (_I, [_P, _LL, _F, _P, _P, _P]),
    'pylc_adamw_step': (_I, [_P, _P, _P, _P, _LL, _P, _F, _F, _F, _F, _F, _I, _P]),
    'pylc_adamw_step_ranges': (_I, [_P, _P, _P, _P, _LL, _P, _F, _F, _F, _F, _F, _I, _P, _I, _P, _P]),
    'pylc_sgd_step': (_I, [_P, _P, _P, _LL, _P, _F, _F, _I, _P]),
    'pylc_dropout': (_I, [_P, _I, _P, _I, _LL, _I, _F, C.c_uint64, _P]),
    'pylc_stream_create_cu_mask': (_I, [_I, _I, C.POINTER(_P)]),
    'pylc_stream_destroy': (_I, [_P]),
}


# entry points that exist only in a library built with `make EXPERIMENTAL=1` (include/pylc_hip.h: #ifdef PYLC_EXPERIMENTAL): what was
# measured neutral or negative and is off in the product -- the persistent 1x1 kernels, the dgrad epilogue that takes BatchNorm-backward
# sums, CU-masked streams.  HAS_EXPERIMENTAL tells the callers (and the tests, which skip without it).
EXPERIMENTAL = ('pylc_conv2d_dgrad_bn_floats', 'pylc_conv2d_dgrad_bn', 'pylc_bn_bwd_sums_from_partial',
                'pylc_stream_create_cu_mask', 'pylc_stream_destroy')
HAS_EXPERIMENTAL = False


def _load():
    global HAS_EXPERIMENTAL
    if not os.path.exists(LIB_PATH):
        raise PylcError(
            'libpylc_hip.so is not built (%s). Build it with `python __graft_entry__.py` or '
            '`make -C pylc_amd/csrc`; there is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    HAS_EXPERIMENTAL = hasattr(lib, 'pylc_experimental_build') and lib.pylc_experimental_build() == 1
    for name, (res, args) in SIGNATURES.items():
        if name in EXPERIMENTAL and not HAS_EXPERIMENTAL:
            continue
        fn = getattr(lib, name)           # AttributeError => header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.pylc_abi_version() != ABI_VERSION:
        raise PylcError('libpylc_hip.so ABI %d != binding ABI %d' % (lib.pylc_abi_version(), ABI_VERSION))
    return lib


lib = _load()
_initialised = False


def check(rc):
    if rc != 0:
        raise PylcError('libpylc_hip: %s (code %d)' % (lib.pylc_last_error().decode(), rc))


def _need_experimental(what):
    if not HAS_EXPERIMENTAL:
        raise PylcError('%s needs a library built with `make -C pylc_amd/csrc EXPERIMENTAL=1` (this one holds the product kernels only)' % what)


def init():
    """Per-process kernel attribute setup; needs a visible GPU."""
    global _initialised
    if not _initialised:
        check(lib.pylc_init())
        mode = os.environ.get('PYLC_CONV_PRECISION')
        if mode is not None:
            check(lib.pylc_set_conv_precision(int(mode)))
        flags = os.environ.get('PYLC_DEBUG_FLAGS')
        if flags is not None:
            lib.pylc_debug_pp_flags(int(flags))
        if os.environ.get('PYLC_NO_PLANE_INTERLEAVE'):           # two-plane activations as separate plane arrays (A/B; the round-4 format)
            lib.pylc_set_planes_interleave(0)
        # PYLC_DEBUG_KNOBS='name=value,...': the library's A/B knobs and bit-identity references by the name of their entry point without the
        # pylc_debug_ prefix (include/pylc_hip.h: set_big_tile, stagger, dw_tiles, wgrad_m16, wgrad_dma, wgrad_acc1, wgrad_sets, wgrad_flags,
        # wgrad_max_steps) -- one variable instead of one per knob
        for item in filter(None, (t.strip() for t in os.environ.get('PYLC_DEBUG_KNOBS', '').split(','))):
            name, _, val = item.partition('=')
            fn = getattr(lib, 'pylc_debug_' + name, None)
            if fn is None or name in ('pp_stamps',):
                raise PylcError('PYLC_DEBUG_KNOBS: no knob %r' % name)
            fn(int(val))
        _initialised = True


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Refuses host tensors: the HIP path is the only path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PylcError('pylc_amd ops need HIP device tensors (got %s); there is no CPU fallback' % t.device)
    return t.data_ptr()
