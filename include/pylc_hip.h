/*
 * pylc_hip.h -- C ABI of libpylc_hip.so: the MI355X (gfx950) kernels behind PyLC's
 * segmentation training / inference step.
 *
 * The reference (scrose/pylc) has no FFI or plugin API: its hot path calls PyTorch ATen ops
 * from Python.  Each entry point below therefore replaces one ATen op
 * family at the call sites cited next to it (paths relative to the reference root).  A binding
 * is a ctypes stub; no torch types appear in any signature.
 *
 * Conventions
 *   - All tensors are fp32, device memory, NHWC ("channels_last"): element (b,h,w,c) lives at
 *     ((b*H + h)*W + w)*pitch + c, where `pitch` (>= C, in floats) lets a tensor be a channel
 *     slice of a wider concat buffer.  Conv weights are KRSC = [Cout][kh][kw][Cin] (the
 *     channels_last memory of a [Cout,Cin,kh,kw] tensor).  Channel counts and pitches must be
 *     multiples of 4 unless stated otherwise (16-byte vector access).
 *   - The caller owns every buffer, including workspaces; no entry point allocates, frees or
 *     synchronises.  All work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - Return value: 0 on success, non-zero on error; pylc_last_error() returns a message for
 *     the calling thread.  Shapes are validated on the host before anything is launched.
 */
#ifndef PYLC_HIP_H
#define PYLC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYLC_OK 0
#define PYLC_ERR_ARG 1
#define PYLC_ERR_HIP 2
#define PYLC_ERR_WORKSPACE 3
#define PYLC_ERR_UNSUPPORTED 4   /* an optional run-time dependency is absent (pylc_comm_*: no loadable librccl) */

const char* pylc_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int pylc_abi_version(void);
/* 1 when the library was built with EXPERIMENTAL=1, i.e. the entry points inside #ifdef PYLC_EXPERIMENTAL below exist (pylc_amd/csrc/Makefile) */
int pylc_experimental_build(void);
/* One-time per-process kernel attribute setup (dynamic LDS opt-in). Idempotent. */
int pylc_init(void);

/* ---------------------------------------------------------------------------------------------
 * Convolution (dense, groups = 1): replaces nn.Conv2d at models/backbone/resnet.py:21-26,72,92;
 * models/modules/aspp.py:18,64,67; models/decoder.py:27,30,34,38; models/backbone/xception.py:32,
 * 48,122,126; models/architectures/unet.py:78,112,116,137 -- and their autograd backward.
 * Implicit-GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), LDS-staged NHWC / KRSC tiles.
 * ------------------------------------------------------------------------------------------- */
typedef struct PylcConvDesc {
    int B, H, W;            /* input batch / height / width                                  */
    int Cin, Cout;          /* Cin % 4 == 0 (pad 3-channel images to 4, see pylc_image_pack)  */
    int R, S;               /* kernel height / width                                         */
    int stride, pad, dil;   /* symmetric stride / zero padding / dilation                    */
    int OH, OW;             /* output height / width = floor((H + 2*pad - dil*(R-1) - 1)/stride) + 1 */
    int x_pitch;            /* floats between input pixels  (>= Cin)                          */
    int y_pitch;            /* floats between output pixels (>= Cout)                         */
    /* Operand ranges, used by conv precision mode 2 only (ignored otherwise, may be NULL): DEVICE scalars holding
     * the IEEE bit pattern of a float >= max|element| of x / the weights / dy (pylc_amax produces them; any upper
     * bound within a few binades of the true maximum is as good).  fwd reads x_amax and w_amax, dgrad dy_amax and
     * w_amax, wgrad dy_amax and x_amax. */
    const unsigned int* x_amax;
    const unsigned int* w_amax;
    const unsigned int* dy_amax;
    /* Optional prepared filter (precision mode 2, pylc_weight_prepare): two fp16 planes in the forward layout
     * [2][Cout][R*S][Cin] (read by fwd) and in the dgrad layout [2][Cin][R*S][roundup4(Cout)] (read by dgrad), split
     * with the scale that w_amax implies.  NULL = the kernels split the fp32 filter themselves. */
    const void* w_planes;
    const void* w_planes_t;
    /* Operand formats (0 = fp32, the default).  1 = "planes": the tensor was written by its producer (pylc_bn_apply /
     * pylc_bn_bwd_apply with planes output, pylc_to_planes) already scaled and split into two fp16 planes -- see "fp16 planes"
     * below; the `x` (fwd, wgrad) / `dy` (dgrad, wgrad) argument then points at plane 0 and the matching *_amax must be the
     * bound the producer scaled with.  Needs precision mode >= 2, Cin resp. Cout % 8 == 0 and pitches % 8 == 0. */
    int x_fmt;
    int dy_fmt;
    /* Output format of pylc_conv2d_fwd / _fwd_stats (y) and of the stride-1 pylc_conv2d_dgrad (dx) (pylc_conv2d_fwd_bnact_ex also takes 2 =
     * two fp16 planes, the f16x3 operand format): 0 = fp32 (default); 1 = a ONE-PLANE fp16
     * tensor (precision mode 3: element = rn16(s v), 2 bytes per element, pitch in elements) whose range bound -- reduction length x the two
     * operand bounds -- is derived in the kernel and written to *out_bound for the consumer.  Needs fp16-plane operands (x_fmt / dy_fmt = 1),
     * no bias, no accumulation. */
    int out_fmt;
    unsigned int* out_bound;
    /* Layout of the prepared filter planes (round 5): bit 0 = w_planes, bit 1 = w_planes_t are CHUNK-INTERLEAVED -- the flat element
     * index e of the [rows][R*S][channels] filter maps to halves (e >> 5) * 64 + (e & 31) for plane 0 and + 32 for plane 1, i.e. the two
     * planes of a 32-channel chunk (one K-step of the conv kernels) share ONE 128-byte line; 0 = two separate plane arrays.  What
     * pylc_weight_prepare(..., interleave = 1) writes for a filter whose channel count (w_planes: Cin; w_planes_t: roundup4(Cout)) is a
     * multiple of 32.  Why: LDS-DMA moves 64-byte pieces of a row per K-step; with separate planes each piece is HALF a cache line and the
     * L2 -> LDS path delivers 21.8 TB/s, with both halves of a line requested back to back 31.7 TB/s (profiles/r04_dma_piece.txt). */
    int w_planes_fmt;
} PylcConvDesc;

/* Arithmetic of the dense conv kernels (process-wide):
 *   0 = v_mfma_f32_32x32x2_f32: bit-exact fp32 fmaf chain, 157 TFLOP/s matrix peak;
 *   1 = "bf16x6" (default): each fp32 operand is split exactly into three bf16 pieces in LDS and every product is
 *       formed from the six leading cross terms on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- fp32-grade
 *       accuracy (error <= the fp32 chain's, tests/test_ops_gpu.py::test_conv_precision_modes) at 16/6 the rate;
 *   2 = "f16x3": each fp32 operand is scaled by an exact power of two taken from the tensor's max magnitude
 *       (PylcConvDesc.*_amax) and split into two fp16 pieces h0 = rn(s*x), h1 = rn(2^11 (s*x - h0)); a product is
 *       a0b0 + 2^-11 (a1b0 + a0b1) on v_mfma_f32_32x32x16_f16, the cross terms in their own fp32 accumulator (the
 *       single-precision-recovery scheme of Ootomo & Yokota, IJHPCA 2022, moved to CDNA4).  Operand error <= 2^-23 |x|
 *       for every element within 2^29 of the tensor maximum (fp16 subnormals are honoured by the MFMA, measured), the
 *       dropped a1b1 term <= 2^-22 |ab|; measured error vs fp64 is below the fp32 chain's.  3 MFMAs instead of 6. */
int pylc_set_conv_precision(int mode);
int pylc_get_conv_precision(void);

/* out_bits[0] = IEEE bits of max|x[r*pitch + c]| over r < rows, c < roundup4(cols) when that fits the pitch (else
 * c < cols): the operand range for precision mode 2.  pylc_amax_segments does the same for `count` contiguous
 * segments base[offsets[s] .. offsets[s+1]) in one launch (all parameters of a flat arena); offsets is a DEVICE
 * array of count+1 entries. */
int pylc_amax(const float* x, long long rows, int cols, int pitch, unsigned int* out_bits, void* stream);
/* Prepare the f16x3 filter planes of `count` conv filters in ONE launch (after every optimiser step, once the ranges
 * are refreshed): entry i describes a KRSC filter base[src_offset ..] of shape [K][RS][C], whose range is
 * amax[amax_index]; its planes go to planes[fwd_offset ..] (2*K*RS*C halves) and planes[t_offset ..]
 * (2*C*RS*roundup4(K) halves).  `table` is a DEVICE array; total_tiles = the sum that continues tile_begin. */
typedef struct PylcWPrepEntry {
    long long src_offset;      /* floats from `base`   */
    long long fwd_offset;      /* halves from `planes` */
    long long t_offset;        /* halves from `planes` */
    long long tile_begin;      /* running sum of RS * ceil(K/32) * ceil(C/32) over the preceding entries */
    int K, RS, C;
    int amax_index;
} PylcWPrepEntry;
/* Inference: a per-input-channel affine x -> scale (.) x + shift in front of a 1x1 conv (the eval-mode BatchNorm between the depthwise and
 * the pointwise conv of xception.py:34-39, coefficients from pylc_bn_eval_coeffs) folded into the conv:  w_out[n][k] = w[n][k] scale[k],
 * bias_out[n] = bias_in[n] (NULL: 0) + sum_k w[n][k] shift[k].  w / w_out: [Cout][Cin] (KRSC of a 1x1 filter); amax_out (may be NULL)
 * receives max |w_out| as float bits, the filter range of the f16x3 arithmetic.  Done once per set of weights, not per batch. */
int pylc_conv1x1_fold_input_affine(const float* w, const float* scale, const float* shift, const float* bias_in, int Cout, int Cin,
                                   float* w_out, float* bias_out, unsigned int* amax_out, void* stream);
/* interleave != 0: filters whose channel count is a multiple of 32 are written chunk-interleaved (PylcConvDesc.w_planes_fmt), per layout:
 * the forward planes if C % 32 == 0, the dgrad planes if roundup4(K) % 32 == 0; the others (and everything with interleave = 0) as two
 * separate plane arrays. */
int pylc_weight_prepare(const float* base, const PylcWPrepEntry* table, int count, long long total_tiles,
                        const unsigned int* amax, void* planes, int interleave, void* stream);
/* out = bits(factor * float(a) * float(b) [+ float(add)]): a range BOUND for a tensor that is bilinear in two ranged operands -- a depthwise
 * 3x3 output, |y| <= 9 max|w| max|x|; a conv output with bias, |y| <= Cin R S max|w| max|x| + max|bias| (add_bits, may be NULL) -- instead of
 * a read pass over it (the f16x3 arithmetic needs a float >= max|element| within 2^29 of the small elements that matter, not the maximum) */
int pylc_range_product(const unsigned int* a_bits, const unsigned int* b_bits, float factor, const unsigned int* add_bits,
                       unsigned int* out_bits, void* stream);
int pylc_amax_segments(const float* base, const long long* offsets, int count, unsigned int* out_bits, void* stream);

/* fp16 planes (operand format 1 of PylcConvDesc): M pixels x C channels (C % 8 == 0, pitch P % 8 == 0 halves) stored as two
 * fp16 planes of M x P halves, `plane_stride` halves apart: plane 0 = rn16(s x), plane 1 = rn16(2^11 (s x - plane 0)), with
 * s the power of two that maps the bound behind `amax` into [2^14, 2^15) -- the two pieces the f16x3 arithmetic forms from an
 * fp32 operand, written ONCE by the tensor's producer (4 bytes per element, like fp32) so that the conv kernels copy operand
 * tiles to LDS by LDS-DMA with no arithmetic.  nplanes = 1 writes / reads plane 0 only (precision mode 3).
 * Replaces nothing in the reference: it is the HBM layout of the activations between the BatchNorm and conv kernels. */
/* Chunk-interleaved two-plane tensors (round 5).  `plane_stride == 32` in any of the entry points below (and in PylcBnExtra, pylc_maxpool_fwd_planes,
 * pylc_gap_fwd_planes, ...) says: flat element e of the dense [M][C] tensor (C % 32 == 0) is at halves (e >> 5) * 64 + (e & 31) for plane 0 and
 * + 32 for plane 1, i.e. the two planes of a 32-channel chunk -- one K-step of the conv kernels -- share ONE 128-byte line (LDS-DMA requests
 * both halves back to back: 21.8 -> 31.7 TB/s L2 -> LDS, profiles/r04_dma_piece.txt).  Any other value: two plane arrays that many halves apart.
 * pylc_planes_stride(M, C, nplanes) is THE rule every producer and consumer follows (32 iff nplanes == 2, C % 32 == 0, 4 M C < 2^31 and the
 * format is on); the conv entry points derive the strides of their plane operands from it, so a caller allocates and passes what it returns.
 * pylc_set_planes_interleave(0) switches the format off process-wide (A/B). */
long long pylc_planes_stride(long long M, int C, int nplanes);
int pylc_set_planes_interleave(int on);
int pylc_to_planes(const float* x, int x_pitch, void* planes, int p_pitch, long long plane_stride, long long M, int C,
                   const unsigned int* amax, int nplanes, void* stream);
int pylc_from_planes(const void* planes, int p_pitch, long long plane_stride, float* x, int x_pitch, long long M, int C,
                     const unsigned int* amax, int nplanes, void* stream);
/* sums[0:C] = per-channel sum over the M pixels of a planes tensor: a conv's bias gradient (nn.Conv2d(bias=True) backward,
 * unet.py:112,116) when its dy arrives as planes.  workspace: pylc_planes_colsum_workspace_floats(C) floats. */
size_t pylc_planes_colsum_workspace_floats(int C);
/* U-Net up path (unet.py:135-152): torch.cat([upsample_x2_bilinear_align_corners(z), center_crop(bridge)], 1) written directly as the fp16-plane
 * tensor the next conv reads ([nplanes][B * 2h * 2w][C1 + C2] halves, dense): no fp32 concat buffer, no range pass, no pylc_to_planes.
 * z: fp32 [B, h, w, C1] (pitch z_pitch), bridge: fp32 [B, HH, WW, C2] (HH >= 2h, WW >= 2w; the centre window is taken); bound: float bits
 * >= max(max|z|, max|bridge|).  The interpolation uses pylc_bilinear_fwd's expression (same bits).  Backward = pylc_bilinear_bwd* on the
 * first C1 channels of the concat gradient + the crop gradient (pylc_maxpool_bwd_add). */
int pylc_upsample2_crop_concat_planes(const float* z, int z_pitch, int B, int h, int w, int C1, const float* bridge, int bridge_pitch, int HH, int WW,
                                      int C2, void* planes, long long plane_stride, int nplanes, const unsigned int* bound, void* stream);
int pylc_planes_colsum(const void* planes, int p_pitch, long long plane_stride, int nplanes, const unsigned int* amax, long long M, int C,
                       float* sums, float* workspace, void* stream);

/* y = conv(x, w) + bias.  bias may be NULL.  Channels [Cout, roundup4(Cout)) of y are written as zeros
 * when they fit inside y_pitch (9/11-class heads use a 12-float pitch). */
int pylc_conv2d_fwd(const PylcConvDesc* d, const float* x, const float* w_krsc, const float* bias,
                    float* y, void* stream);
/* Inference: y = act(conv(x, w) * scale[c] + shift[c] (+ residual)), relu != 0 applies max(., 0) -- eval-mode BatchNorm
 * (scale / shift from pylc_bn_eval_coeffs), the residual add of a ResNet block and the ReLU, all in the conv epilogue
 * (models/backbone/resnet.py:36-51 in eval mode).  residual (may be NULL) has y's geometry and pitch.  amax_out (may be
 * NULL; zero-initialised by the caller) is max-accumulated with the range of y.  Bit-identical to pylc_conv2d_fwd followed
 * by pylc_bn_apply. */
int pylc_conv2d_fwd_bnact(const PylcConvDesc* d, const float* x, const float* w_krsc, const float* bias,
                          const float* scale, const float* shift, const float* residual, int relu, float* y,
                          unsigned int* amax_out, void* stream);

/* pylc_conv2d_fwd_bnact on fp16-PLANE tensors (inference without fp32 round trips between the kernels): x arrives as planes (d->x_fmt = 1,
 * d->x_amax = the bound it was scaled with), the residual as fp32 or planes, y leaves as fp32 (d->out_fmt = 0), ONE fp16 plane (1:
 * precision mode 3) or TWO planes (2: f16x3).  Replaces the same call sites as pylc_conv2d_fwd_bnact -- nn.Conv2d + BatchNorm2d.eval()
 * (+ residual add + ReLU) in models/backbone/resnet.py:36-51,92, models/backbone/xception.py:34-39,60-97, models/modules/aspp.py:18-30,
 * models/decoder.py:27-38, models/architectures/unet.py:107-126 as run by Model.eval / Model.test (models/model.py:338-382).
 * An eval-mode network has no batch statistics to re-anchor range bounds, so a plane tensor carries TWO device scalars: the bound it was
 * SCALED with (formed before the kernel writes: Cin R S x true max|x| x max|w| x max|scale| + max|shift| + max|residual|, written to
 * *d->out_bound) and its TRUE maximum (max-accumulated by the epilogue into amax_out, zero-initialised by the caller), which is what the
 * consumer's bound starts from. */
typedef struct PylcFwdEp {
    const float* scale;                    /* eval-BatchNorm coefficients (pylc_bn_eval_coeffs), [Cout] each                   */
    const float* shift;
    const unsigned int* scale_amax;        /* device scalars: float bits of max|scale|, max|shift| (plane outputs only)        */
    const unsigned int* shift_amax;
    const void* residual;                  /* y's geometry, dense; NULL = none                                                  */
    int res_fmt;                           /* 0 = fp32, 1 = one fp16 plane, 2 = two fp16 planes (plane 1 at + B OH OW Cout halves) */
    const unsigned int* res_scale_bound;   /* res_fmt 1 / 2: the bound the residual was scaled with                             */
    const unsigned int* res_amax;          /* true max|residual| (plane outputs only)                                           */
    const unsigned int* x_true_amax;       /* true max|x| (NULL: d->x_amax, the scale bound, is used for the output bound too)  */
    int relu;
    unsigned int* amax_out;                /* receives max|y| (atomic max of float bits); may be NULL for an fp32 output        */
} PylcFwdEp;
int pylc_conv2d_fwd_bnact_ex(const PylcConvDesc* d, const void* x, const float* w, const float* bias, const PylcFwdEp* ep, void* y, void* stream);
/* pylc_conv2d_fwd that additionally emits per-M-tile partial column sums of y for the BatchNorm that follows (saves a full read
 * of y): stats_partial has pylc_conv2d_fwd_stats_floats(d) floats, laid out [rows][2][roundup4(Cout)] = (sum | sum of
 * squares) of (y - bias) -- taken BEFORE the bias is added, so that a bias much larger than the spread does not cost the variance its
 * digits; pass the bias as `stat_shift` to the finalize.  *stats_rows receives the number of rows written. */
size_t pylc_conv2d_fwd_stats_floats(const PylcConvDesc* d);
int pylc_conv2d_fwd_stats(const PylcConvDesc* d, const float* x, const float* w_krsc, const float* bias,
                          float* y, float* stats_partial, int* stats_rows, void* stream);
/* dx = conv_transpose(dy, w).  w_crsk = weights re-laid-out as [Cin][R][S][Cout] by
 * pylc_weight_transpose.  accumulate != 0 adds into dx instead of overwriting it. */
int pylc_conv2d_dgrad(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx,
                      int accumulate, void* stream);
/* dx = conv_transpose(dy, w) + relu'(add_src): the data gradient of a residual block's first conv with the gradient of the block's
 * identity branch formed in the same epilogue (models/backbone/resnet.py:36-51: `out += residual; out = relu(out)` -- autograd hands
 * the ReLU-masked gradient of the block output both to bn3 and to the block input).  add_src = gradient of the block OUTPUT (fp32, the
 * geometry and pitch of dx), add_mask = the 1-bit ReLU mask that pylc_bn_apply_ex left (PylcBnExtra::relu_mask; NULL = add add_src
 * unmasked).  Saves the pass that would write the masked gradient and the read-modify-write of accumulate = 1.  Needs fp16-plane dy
 * (dy_fmt = 1), stride 1, a dense dx (x_pitch == Cin, Cin % 8 == 0) and accumulate == 0; add_src == NULL is pylc_conv2d_dgrad. */
int pylc_conv2d_dgrad_add(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                          const float* add_src, const void* add_mask, void* stream);
#ifdef PYLC_EXPERIMENTAL      /* measured negative inside the step (the y tile is fetched in the epilogue, a latency chain): off, built with EXPERIMENTAL=1 only */
/* pylc_conv2d_dgrad_add that ALSO takes the backward sums of the BatchNorm whose output this dgrad differentiates: dx is that
 * BatchNorm's `dout`, so sum g xhat and sum g (g = relu'(dout), xhat = (y - mean) invstd: what pylc_bn_bwd_reduce computes in a read
 * pass over dout and y, models/sync_batchnorm/batchnorm.py:113-125 / torch's batch_norm_backward) are taken from the output tile while
 * it is in registers.  Valid only when dx is the COMPLETE gradient of that tensor (the caller's business: pylc_amd/ops.py does it for the
 * last dgrad of a gradient link, and for a BatchNorm output with a single consumer).  sums_partial: pylc_conv2d_dgrad_bn_floats(d)
 * floats, [rows][sum g xhat (Cin) | sum g (Cin)] per 128- or 256-pixel tile, *sums_rows rows -- combine with
 * pylc_bn_bwd_sums_from_partial.  bn->g_amax (zero-initialised) is max-accumulated with max |g|.  Same conditions as
 * pylc_conv2d_dgrad_add; bn == NULL is pylc_conv2d_dgrad_add. */
typedef struct PylcBnBack {
    const float* y;            /* the BatchNorm's input: fp32, geometry and pitch of dx */
    const float* mean;         /* per channel, as pylc_bn_finalize* wrote them */
    const float* invstd;
    const float* scale;        /* mask source A (ReLU without residual): y * scale + shift > 0, the forward's own expression */
    const float* shift;
    const void* relu_mask;     /* mask source B: the 1-bit mask of PylcBnExtra::relu_mask */
    int relu;                  /* 0: no ReLU (g = dout) */
    unsigned int* g_amax;      /* may be NULL */
} PylcBnBack;
size_t pylc_conv2d_dgrad_bn_floats(const PylcConvDesc* d);
int pylc_conv2d_dgrad_bn(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                         const float* add_src, const void* add_mask, const PylcBnBack* bn, float* sums_partial, int* sums_rows,
                         void* stream);
#endif
/* g_out = dout where the 1-bit mask is set, else 0 (the fallback of pylc_conv2d_dgrad_add when the dgrad that consumes a parked
 * (dout, mask) pair does not run on the fp16-plane kernels).  M rows x C channels, C % 8 == 0, dense. */
int pylc_relu_bwd_bits(const float* dout, const void* mask, float* g_out, long long M, int C, void* stream);
/* 0 when pylc_conv2d_dgrad(d, ...) only reads d->w_planes_t (w_crsk may then be NULL and the transpose be skipped). */
int pylc_conv2d_dgrad_needs_f32_weights(const PylcConvDesc* d);
/* dw (KRSC) = sum over pixels of dy (x) x.  workspace: pylc_conv2d_wgrad_workspace(d) bytes
 * (deterministic split-K slabs, reduced in a fixed order).  dbias (may be NULL) = sum_pixels dy. */
size_t pylc_conv2d_wgrad_workspace(const PylcConvDesc* d);
int pylc_conv2d_wgrad(const PylcConvDesc* d, const float* x, const float* dy, float* dw_krsc,
                      float* dbias /* must be NULL: take sums[0:C] of pylc_bn_stats(dy) */,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same wgrad WITHOUT its split-K slab sum: the slabs stay in `workspace` (which must then outlive the call until the sum has run) and
 * `*pending` describes the sum -- splits == 0: dw is already complete (a single split).  pylc_splitk_reduce_batch sums the slabs of many
 * wgrads in ONE launch (table and tile prefix in device memory: tile_prefix[e] = sum over entries before e of ceil(n4 / 32), n + 1 values;
 * total_tiles = tile_prefix[n]); per element the association is that of pylc_conv2d_wgrad's own sum, so the results are bit-identical.
 * Why: one queue runs ~1100 kernels per training step and every dependent launch costs 2-3 us on top of the kernel (DESIGN.md 5.2 i);
 * the reference's optimiser (train.py:93-99) needs the gradients only after the whole backward pass. */
typedef struct PylcSlabSum {
    const float* slabs;       /* [splits][slab_stride] floats */
    float* dw;                /* n4 float4 columns */
    long long n4;
    long long slab_stride;    /* floats */
    int splits;
    int reserved;
} PylcSlabSum;
int pylc_conv2d_wgrad_slabs(const PylcConvDesc* d, const float* x, const float* dy, float* dw_krsc, void* workspace, size_t workspace_bytes,
                            PylcSlabSum* pending, void* stream);
int pylc_splitk_reduce_batch(const PylcSlabSum* table_dev, const long long* tile_prefix_dev, int n, long long total_tiles, void* stream);
/* [K][R*S][C] -> [C][R*S][Kp], Kp = roundup4(K), zero-filled pad columns */
int pylc_weight_transpose(const float* w_krsc, float* w_crsk, int K, int RS, int C, void* stream);

/* Depthwise 3x3 (groups = C) with the explicit TF-'SAME' padding of fixed_padding folded into the
 * index math: replaces F.pad + nn.Conv2d(groups=C) at models/backbone/xception.py:16-22,29-31,35-36.
 * w is [C][3][3] (the memory of a [C,1,3,3] tensor).  pad_beg = ((3-1)*dil)/2. */
typedef struct PylcDwDesc {
    int B, H, W, C;
    int stride, dil;
    int OH, OW;
    int x_pitch, y_pitch;
} PylcDwDesc;
int pylc_dwconv3x3_fwd(const PylcDwDesc* d, const float* x, const float* w, float* y, void* stream);
/* pylc_dwconv3x3_fwd that also emits per-block (sum | sum of squares) of y for the BatchNorm that follows the depthwise conv
 * (xception.py:34-39): stats_partial [rows][2][C] with rows = pylc_dwconv3x3_fwd_stats_rows(d) (0: this shape has no fused form --
 * stride 2 / dilated -- use pylc_bn_stats); feed to pylc_bn_finalize_from_partial(_ex). */
int pylc_dwconv3x3_fwd_stats_rows(const PylcDwDesc* d);
int pylc_dwconv3x3_fwd_stats(const PylcDwDesc* d, const float* x, const float* w_c9, float* y, float* stats_partial, void* stream);
int pylc_dwconv3x3_dgrad(const PylcDwDesc* d, const float* dy, const float* w, float* dx, void* stream);
/* accumulate != 0: dx += the data gradient (dx holds the part of the tensor's other consumers: the block input of xception.py:88-97
 * feeds both the first depthwise conv of `rep` and the skip path) */
int pylc_dwconv3x3_dgrad_acc(const PylcDwDesc* d, const float* dy, const float* w_c9, float* dx, int accumulate, void* stream);
size_t pylc_dwconv3x3_wgrad_workspace(const PylcDwDesc* d);
int pylc_dwconv3x3_wgrad(const PylcDwDesc* d, const float* x, const float* dy, float* dw,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same three passes on ONE-PLANE fp16 tensors (precision mode 3, "fp16 planes" below with nplanes = 1: element = rn16(s v), s the power
 * of two that maps the tensor's range bound into [2^14, 2^15)): x / dy / y / dx at 2 bytes per element, arithmetic and the filter in fp32.
 * *_bound: device scalars with the float bits of each tensor's range bound.  The output's bound is derived in the kernel --
 * 9 * w_amax * input bound (+ acc_bound when accumulating into a dx that already holds another consumer's part, which is re-scaled in the
 * same pass) -- and written to *_bound_out.  pylc_dwconv3x3_dgrad_h with dx_bound_out == NULL writes (and accumulates into) an fp32 dx:
 * the gradient of a block input, which other consumers add to in fp32.  Dense stride-1 / dilation-1 shapes only
 * (pylc_dwconv3x3_half_ok); stats_partial as in pylc_dwconv3x3_fwd_stats (NULL: none), taken from the fp32 values before rounding. */
int pylc_dwconv3x3_half_ok(const PylcDwDesc* d);
/* rows of the stats_partial buffer pylc_dwconv3x3_fwd_h fills ([rows][2][C]; 0: shape not eligible) -- the half kernels tile the image
 * differently from the fp32 strip kernels, so this is not pylc_dwconv3x3_fwd_stats_rows */
int pylc_dwconv3x3_fwd_h_stats_rows(const PylcDwDesc* d);
int pylc_dwconv3x3_fwd_h(const PylcDwDesc* d, const void* x_h, const unsigned int* x_bound, const float* w_c9, const unsigned int* w_amax,
                         void* y_h, unsigned int* y_bound_out, float* stats_partial, void* stream);
/* pylc_dwconv3x3_fwd_h for inference on one-plane tensors (xception.py:29-31 under Model.test, models/model.py:367-382): x was scaled with
 * x_bound but its TRUE maximum is x_true_amax (left by the producing pylc_conv2d_fwd_bnact_ex); y is scaled with, and *y_bound_out receives,
 * 9 max|w| x_true_amax -- the bound does not inherit the looseness of x's.  No statistics. */
int pylc_dwconv3x3_fwd_h_eval(const PylcDwDesc* d, const void* x_h, const unsigned int* x_bound, const unsigned int* x_true_amax, const float* w,
                              const unsigned int* w_amax, void* y_h, unsigned int* y_bound_out, void* stream);
int pylc_dwconv3x3_dgrad_h(const PylcDwDesc* d, const void* dy_h, const unsigned int* dy_bound, const float* w_c9, const unsigned int* w_amax,
                           void* dx_h, unsigned int* dx_bound_out, int accumulate, const unsigned int* acc_bound, void* stream);
/* pylc_dwconv3x3_dgrad_h with an fp32 dx that also receives a ReLU'd residual gradient: dx = dw^T(dy) + (mask ? add_src : 0), add_src fp32 of
 * dx's shape, add_mask the 1-bit ReLU mask pylc_bn_apply_ex left (PylcBnExtra.relu_mask) -- what pylc_relu_bwd_bits + an accumulating
 * dgrad would do in two passes (xception.py:53-97: the block input feeds the first depthwise conv and the residual add).  Shapes of the
 * tiled stride-1 kernel only: pylc_dwconv3x3_dgrad_h_add_ok. */
int pylc_dwconv3x3_dgrad_h_add_ok(const PylcDwDesc* d);
int pylc_dwconv3x3_dgrad_h_add(const PylcDwDesc* d, const void* dy_h, const unsigned int* dy_bound, const float* w_c9, const unsigned int* w_amax,
                               float* dx, const float* add_src, const unsigned char* add_mask, void* stream);
/* The `_bn` forms read, instead of x, the INPUT y_in of the training-mode BatchNorm (+ ReLU) that produces x (one fp16 plane scaled with
 * y_in_bound) and apply x = act(y_in * bn_scale[c] + bn_shift[c]) to their LDS patch before computing -- the pass pylc_bn_apply_ex would
 * have made, bit for bit (same operations, same fp16 rounding with x_bound's scale), without x ever being written: for a BatchNorm whose
 * only consumer is this depthwise conv (xception.py:60-97: every BatchNorm between two separable convs of a block).  bn_scale / bn_shift:
 * the coefficients of pylc_bn_finalize*_ex; x_bound: the bound that pass computed for x.  Tiled kernels only: pylc_dwconv3x3_bn_ok. */
int pylc_dwconv3x3_bn_ok(const PylcDwDesc* d);
int pylc_dwconv3x3_fwd_h_bn(const PylcDwDesc* d, const void* y_in_h, const unsigned int* y_in_bound, const float* bn_scale, const float* bn_shift,
                            int relu, const unsigned int* x_bound, const float* w_c9, const unsigned int* w_amax, void* y_h,
                            unsigned int* y_bound_out, float* stats_partial, void* stream);
int pylc_dwconv3x3_wgrad_h_bn(const PylcDwDesc* d, const void* y_in_h, const unsigned int* y_in_bound, const float* bn_scale, const float* bn_shift,
                              int relu, const unsigned int* x_bound, const void* dy_h, const unsigned int* dy_bound, float* dw, void* workspace,
                              size_t workspace_bytes, void* stream);
int pylc_dwconv3x3_wgrad_h(const PylcDwDesc* d, const void* x_h, const unsigned int* x_bound, const void* dy_h, const unsigned int* dy_bound,
                           float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d (+ fused ReLU / residual add): replaces torch.nn.BatchNorm2d (selected at
 * models/model.py:71-76) and the ReLU / `out += residual` that follow it (resnet.py:36-51,
 * aspp.py:28-31, decoder.py:42-44, unet.py:113-118, xception.py:37,60-97).
 * ------------------------------------------------------------------------------------------- */
/* Number of floats of workspace needed by the two reductions below for M rows x C channels. */
size_t pylc_bn_workspace_floats(long long M, int C);
/* Per-channel sums of y[M][C] (dense rows of pitch y_pitch): sums[0:C] = sum y, sums[C:2C] = sum y*y
 * (fp64-combined, deterministic).  With `count`: the SyncBN wire format of
 * models/sync_batchnorm/batchnorm.py:66-68 (sum, ssum, size). */
int pylc_bn_stats(const float* y, long long M, int C, int y_pitch, float* sums /*[2C]*/,
                  float* workspace, void* stream);
/* sums[0:2C] from the per-tile partials written by pylc_conv2d_fwd_stats (fp64 combine, fixed order). */
int pylc_bn_stats_from_partial(const float* partial, int n_rows, int C, float* sums /*[2C]*/, void* stream);
/* From (possibly all-reduced) sums and the GLOBAL row count n: mean, invstd = 1/sqrt(var_biased + eps)
 * (clamp_eps != 0 selects batchnorm.py:125's clamp(var, eps)^-1/2 instead), running stats update with
 * the unbiased variance (momentum), and the fused affine scale = gamma*invstd, shift = beta - mean*scale. */
int pylc_bn_finalize(const float* sums, double n, int C, const float* gamma, const float* beta,
                     float eps, float momentum, int clamp_eps,
                     float* running_mean, float* running_var,      /* may be NULL (no update) */
                     float* mean, float* invstd, float* scale, float* shift, void* stream);
/* pylc_bn_stats_from_partial + pylc_bn_finalize in one launch (single-GPU path: nothing to all-reduce in between);
 * `partial` / n_rows as produced by pylc_conv2d_fwd_stats.  Bit-identical to the two-launch path. */
int pylc_bn_finalize_from_partial(const float* partial, int n_rows, double n, int C, const float* gamma, const float* beta,
                                  float eps, float momentum, int clamp_eps, float* running_mean, float* running_var,
                                  float* mean, float* invstd, float* scale, float* shift, void* stream);
/* Eval mode: scale/shift from running statistics. */
int pylc_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma,
                        const float* beta, float eps, int C, float* scale, float* shift, void* stream);
/* The same plus mean = running_mean and invstd = 1/sqrt(running_var + eps) (what an eval-mode backward reads). */
int pylc_bn_eval_coeffs_full(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                             int C, float* scale, float* shift, float* mean, float* invstd, void* stream);
/* out = act(y*scale + shift (+ residual)); relu != 0 applies max(.,0).  y may alias out.  amax_out (may be NULL) is
 * max-accumulated with the bit pattern of max|out| -- the x_amax of the conv that consumes `out` (precision mode 2) at no
 * extra pass; the caller provides it ZERO-INITIALISED (or holding a lower bound). */
int pylc_bn_apply(const float* y, int y_pitch, const float* scale, const float* shift,
                  const float* residual, int res_pitch, float* out, int out_pitch,
                  long long M, int C, int relu, unsigned int* amax_out, void* stream);
/* Backward, training mode.  g = dout * (out > 0 if relu).  sums[0:C] = sum g * xhat (= dgamma),
 * sums[C:2C] = sum g (= dbeta), xhat = (y - mean) * invstd  -- parameter order, so `sums` may point straight at the
 * adjacent (gamma, beta) gradient slots of a flat arena.
 * The ReLU mask: from `out` when given; with out == NULL and (scale, shift) = the forward's coefficients it is
 * recomputed as y*scale + shift > 0 (bit-identical to the forward, valid when the forward had NO residual input) --
 * one tensor less to read in both backward kernels. */
int pylc_bn_bwd_reduce(const float* dout, int dout_pitch, const float* out, int out_pitch,
                       const float* y, int y_pitch, const float* mean, const float* invstd,
                       long long M, int C, int relu, float* sums /*[2C]*/, float* workspace,
                       const float* scale, const float* shift, void* stream);
/* dy = gamma*invstd*(g - sum_g/n - xhat*sum_gx/n) with the (all-reduced) sums and GLOBAL n.
 * If g_out != NULL it receives g (the gradient of the residual branch). dy may alias dout.  amax_dy (may be NULL,
 * zero-initialised by the caller) is max-accumulated with the bit pattern of max|dy| (the dy_amax of the preceding
 * conv's dgrad / wgrad in precision mode 2). */
int pylc_bn_bwd_apply(const float* dout, int dout_pitch, const float* out, int out_pitch,
                      const float* y, int y_pitch, const float* mean, const float* invstd,
                      const float* gamma, const float* sums, double n,
                      long long M, int C, int relu, float* dy, int dy_pitch, float* g_out, int g_pitch,
                      unsigned int* amax_dy, const float* scale, const float* shift, void* stream);
/* Optional extras of the BatchNorm kernels (the *_ex entry points; NULL = plain fp32 behaviour):
 *   - fp16-plane operands (see "fp16 planes"): when a *_planes pointer is set, the matching fp32 pointer argument must be NULL and the
 *     pitch argument counts halves; the scale comes from the device scalar next to it (a range BOUND written by pylc_bn_finalize*_ex /
 *     pylc_bn_bwd_reduce_ex before the apply pass runs -- Samuelson's inequality bounds a BatchNorm output from its statistics alone);
 *   - dropout fused behind the activation: out = dropout(act(...)) with pylc_dropout's counter-based mask (models/modules/aspp.py:86,
 *     models/decoder.py:33,37); the backward regenerates the mask from the seed;
 *   - g_amax: pylc_bn_bwd_reduce_ex max-accumulates max|g| into it (zero-initialised by the caller);
 *   - relu_mask: see the field. */
typedef struct PylcBnExtra {
    void* out_planes;        long long out_plane_stride;  const unsigned int* out_bound;   /* bn_apply output; the `out` the backward masks with */
    const void* res_planes;  long long res_plane_stride;  const unsigned int* res_amax;    /* bn_apply residual input */
    void* dy_planes;         long long dy_plane_stride;   const unsigned int* dy_bound;    /* bn_bwd_apply output */
    int nplanes;             /* 2, or 1 = plane 0 only (precision mode 3) */
    float drop_p;            /* 0 = no dropout */
    uint64_t drop_seed;
    unsigned int* g_amax;
    /* 1-bit ReLU mask, M * C / 8 bytes, C % 8 == 0 (bit = pre-activation > 0, torch's threshold_backward mask; byte (row * C/4 + c/4) / 2
     * holds the nibbles of two adjacent channel quads).  pylc_bn_apply_ex with relu WRITES it; pylc_bn_bwd_reduce_ex / _apply_ex READ it
     * instead of `out` / out_planes -- for a BatchNorm with a residual input (resnet.py:47-51: relu(bn3(.) + residual)), whose mask cannot
     * be recomputed from y, this replaces two 4-byte-per-element reads of `out` in the backward by two 1/8-byte ones. */
    void* relu_mask;
    /* Precision mode 3 with ONE-PLANE fp16 activations end to end (2 bytes per element): when y_half_bound is set, the `y` argument of
     * pylc_bn_apply_ex / _bwd_reduce_ex / _bwd_apply_ex points at a one-plane fp16 tensor (element = rn16(s v), s from this bound; pitch in
     * elements) instead of fp32; dout_half_bound does the same for the `dout` argument of the two backward entry points (either, both or
     * neither), and g_out, when asked for, is written in dout's format and scale.  The refinement pass of the finalize kernels
     * (ill-conditioned channels) needs an fp32 y: pass y = NULL there. */
    const unsigned int* y_half_bound;
    const unsigned int* dout_half_bound;
} PylcBnExtra;
/* pylc_bn_finalize / _from_partial that also max-accumulate into *bound_out (zero-initialised) an upper bound of
 * |act(BN(y)) (+ residual)| * bound_mul:  max_c (|gamma_c| sqrt(n - 1) + |beta_c|) + *bound_extra (the residual's range, may be NULL).
 * y (may be NULL) / y_pitch / M == n: the tensor the sums were taken over.  With it, a channel whose mean^2 exceeds 64 var -- where
 * sumsq/n - mean^2 from fp32 sums no longer carries the variance -- is re-measured in a second pass as sum((y - mean)^2), so the result
 * follows torch.nn.BatchNorm2d's two-pass / Welford statistics (what the reference's non-SyncBN layers run) at any mean / sigma ratio.
 * stat_shift (may be NULL): per-channel K when the sums are of (y - K) and (y - K)^2 -- the conv epilogues take their statistics before
 * the bias is added (a bias that dwarfs the spread, U-Net's first layer, is the common ill-conditioned case and costs nothing this
 * way); the mean returned is K + sum / n.  For all-reduced statistics use pylc_bn_local_moments / pylc_bn_finalize_moments. */
int pylc_bn_finalize_ex(const float* sums, double n, int C, const float* gamma, const float* beta, float eps, float momentum,
                        int clamp_eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                        float* shift, const unsigned int* bound_extra, float bound_mul, unsigned int* bound_out,
                        const float* y, int y_pitch, long long M, const float* stat_shift, void* stream);
int pylc_bn_finalize_from_partial_ex(const float* partial, int n_rows, double n, int C, const float* gamma, const float* beta,
                                     float eps, float momentum, int clamp_eps, float* running_mean, float* running_var,
                                     float* mean, float* invstd, float* scale, float* shift, const unsigned int* bound_extra,
                                     float bound_mul, unsigned int* bound_out, const float* y, int y_pitch, long long M,
                                     const float* stat_shift, void* stream);
/* SyncBN in two stages around ONE all-reduce (sync_batchnorm/batchnorm.py:48-125 exchanges [sum, ssum, size] per layer): stage 1 turns
 * this rank's fp32 sums over its n rows into fp64 moments [sum y | sum y^2 | n] (2C + 1 doubles), re-measuring ill-conditioned channels
 * from y as pylc_bn_finalize_ex does; the caller all-reduces the buffer (SUM); stage 2 derives the coefficients from the global moments
 * over n = the summed count.  With one rank the result equals pylc_bn_finalize_ex bit for bit.  stat_shift: as pylc_bn_finalize_ex (the
 * same K on every rank: a conv bias is a replicated parameter). */
int pylc_bn_local_moments(const float* sums, double n, int C, const float* y, int y_pitch, long long M, const float* stat_shift,
                          double* moments, void* stream);
int pylc_bn_finalize_moments(const double* moments, double n, int C, const float* gamma, const float* beta, float eps, float momentum,
                             int clamp_eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                             float* shift, const unsigned int* bound_extra, float bound_mul, unsigned int* bound_out,
                             const float* stat_shift, void* stream);
int pylc_bn_apply_ex(const float* y, int y_pitch, const float* scale, const float* shift, const float* residual, int res_pitch,
                     float* out, int out_pitch, long long M, int C, int relu, unsigned int* amax_out, const PylcBnExtra* ex,
                     void* stream);
/* dy_bound_out (may be NULL; zero-initialised; needs ex->g_amax, gamma, n): bound of the dy that pylc_bn_bwd_apply_ex will write,
 * |gamma invstd| (max|g| + |sum g| / n + sqrt(n - 1) |sum g xhat| / n), from the LOCAL sums (single GPU); with all-reduced sums
 * call pylc_bn_bwd_bound after the exchange instead. */
int pylc_bn_bwd_reduce_ex(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                          const float* mean, const float* invstd, long long M, int C, int relu, float* sums, float* workspace,
                          const float* scale, const float* shift, const float* gamma, double n, const PylcBnExtra* ex,
                          unsigned int* dy_bound_out, void* stream);
int pylc_bn_bwd_bound(const float* sums, const float* gamma, const float* invstd, double n, int C, const unsigned int* g_amax,
                      unsigned int* bound_out, void* stream);
#ifdef PYLC_EXPERIMENTAL
/* The combine stage of pylc_bn_bwd_reduce_ex alone, for per-tile partials that a conv dgrad emitted (pylc_conv2d_dgrad_bn):
 * sums = fp64 column sums of partial[rows][2C] in fixed order; dy_bound_out as in pylc_bn_bwd_reduce_ex (NULL: none). */
int pylc_bn_bwd_sums_from_partial(const float* partial, int rows, int C, float* sums, const float* gamma, const float* invstd, double n,
                                  const unsigned int* g_amax, unsigned int* dy_bound_out, void* stream);
#endif
int pylc_bn_bwd_apply_ex(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                         const float* mean, const float* invstd, const float* gamma, const float* sums, double n, long long M,
                         int C, int relu, float* dy, int dy_pitch, float* g_out, int g_pitch, unsigned int* amax_dy,
                         const float* scale, const float* shift, const PylcBnExtra* ex, void* stream);
/* Plain ReLU forward / backward on [M][C] (Xception's stand-alone ReLUs, xception.py:83-84,199-232). */
int pylc_relu_fwd(const float* x, int x_pitch, float* out, int out_pitch, long long M, int C, void* stream);
int pylc_relu_bwd(const float* dout, int dout_pitch, const float* out, int out_pitch, float* dx, int dx_pitch,
                  long long M, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pooling / resize: nn.MaxPool2d(3,2,1) resnet.py:76, F.max_pool2d(x,2) unet.py:98,
 * F.interpolate(bilinear, align_corners=True) deeplab.py:38 / decoder.py:46 / aspp.py:79 /
 * nn.Upsample unet.py:136, nn.AdaptiveAvgPool2d(1) aspp.py:63.
 * ------------------------------------------------------------------------------------------- */
/* idx (may be NULL for inference) receives, per output element, the window position kh*k + kw of the
 * FIRST maximum in scan order (PyTorch's convention); the backward routes dy through it (a gather over
 * the <= 4 windows covering each input element: deterministic, no atomics). */
int pylc_maxpool_fwd(const float* x, float* y, unsigned char* idx, int B, int H, int W, int C, int k, int stride,
                     int pad, int OH, int OW, void* stream);
/* pylc_maxpool_fwd with the pooled tensor written as fp16 planes ([nplanes][B * OH * OW][C] halves, scaled with `bound` >= max|x|, which
 * bounds the maxima too) for a conv that copies its operand tiles: saves the conversion pass over the pooled tensor (unet.py:98). */
int pylc_maxpool_fwd_planes(const float* x, void* y_planes, long long plane_stride, int nplanes, const unsigned int* bound, unsigned char* idx,
                            int B, int H, int W, int C, int k, int stride, int pad, int OH, int OW, void* stream);
int pylc_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int k,
                     int stride, int pad, int OH, int OW, void* stream);
/* U-Net skip connections (unet.py:95-101,145-152): the encoder feature map x feeds the 2x2 max-pool AND, centre-cropped,
 * the decoder's channel concat.  pylc_crop_copy writes the crop [h0, h0+TH) x [w0, w0+TW) of src straight into a
 * channel range of the concat buffer (dst points at that range, dst_pitch = the buffer's channel count);
 * pylc_maxpool_bwd_add is pylc_maxpool_bwd plus the crop's gradient `add` ([B][add_h][add_w] pixels, add_pitch floats
 * apart) summed into the window it came from -- instead of a zero-padded full-size tensor and an add pass. */
int pylc_maxpool_bwd_add(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int k,
                         int stride, int pad, int OH, int OW, const float* add, int add_pitch, int add_h0,
                         int add_w0, int add_h, int add_w, void* stream);
int pylc_crop_copy(const float* src, int src_pitch, int H, int W, int h0, int w0, float* dst, int dst_pitch,
                   int B, int TH, int TW, int C, void* stream);
int pylc_bilinear_fwd(const float* x, int x_pitch, float* y, int y_pitch, int B, int H, int W, int C,
                      int OH, int OW, void* stream);
int pylc_bilinear_bwd(const float* dy, int dy_pitch, float* dx, int dx_pitch, int B, int H, int W, int C,
                      int OH, int OW, void* stream);
/* The same gradient applied one axis at a time (along W into `workspace` [B, OH, W, C], pylc_bilinear_bwd_workspace bytes, then along
 * H): ~2 x (2 OW / W + 3) candidate taps per element instead of their product -- the form to use for up-sampling factors >= 2
 * (deeplab.py:38 logits x4, decoder.py:46 x4).  Same weights; the two-stage summation order differs from pylc_bilinear_bwd's. */
size_t pylc_bilinear_bwd_workspace(int B, int W, int C, int OH);
/* amax_bits (may be NULL): receives the IEEE bits of max|dx| (as pylc_amax would compute them) from the pass that writes dx -- the range
 * the conv backward reading dx scales its operand with, without a pass of its own. */
int pylc_bilinear_bwd_separable(const float* dy, int dy_pitch, float* dx, int dx_pitch, int B, int H, int W, int C,
                                int OH, int OW, float* workspace, unsigned int* amax_bits, void* stream);
int pylc_gap_fwd(const float* x, float* y, int B, int HW, int C, void* stream);
int pylc_gap_bwd(const float* dy, float* dx, int B, int HW, int C, void* stream);
/* pylc_gap_fwd over a tensor held as fp16 planes (dense pitch C; plane p at planes + p * plane_stride halves; `amax`: the bound it was
 * scaled with): each element rebuilt as pylc_from_planes does, summed in pylc_gap_fwd's order -- the same bits without the conversion
 * pass (aspp.py:59-63 on the planes a backbone's last BatchNorm leaves for the atrous convs). */
int pylc_gap_fwd_planes(const void* planes, long long plane_stride, int nplanes, const unsigned int* amax, float* y,
                        int B, int HW, int C, void* stream);
/* accumulate != 0: dx += the pooled gradient (dx holds the gradient parts of the tensor's other consumers: aspp.py:76-80, where the
 * encoder output feeds four atrous branches and this pooling branch) */
int pylc_gap_bwd_acc(const float* dy, float* dx, int B, int HW, int C, int accumulate, void* stream);

/* Image ingest: Model.normalize_image models/model.py:416-445 + the 1->3 channel stack model.py:310-311
 * + NCHW->NHWC, zero-padded to 4 channels.  img is [B][Cimg][H][W] raw 0..255 floats (Cimg = 1 or 3);
 * out is [B][H][W][4]: out[..., c] = ((img[c or 0] - mean[c]) / std[c]) / 255, out[..., 3] = 0.
 * mean3 / std3 are HOST arrays of 3 floats. */
int pylc_image_pack(const float* img_nchw, int B, int Cimg, int H, int W, const float* mean3,
                    const float* std3, float* out_nhwc4, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Sliding-window inference (test.py:50-110): tile extraction fused with normalisation, replacing
 * Extractor.__split utils/extract.py:279-310; tile stitching + argmax replacing utils/tools.py:209-319
 * reconstruct() (overlap quirks reproduced, see csrc/stitch.hip); palette + INTER_NEAREST resize replacing
 * colourize utils/tools.py:322-358 and the cv2.resize at :315-317.
 * ------------------------------------------------------------------------------------------- */
/* img: [Cimg][H][W] raw 0..255 (fitted: H, W multiples of stride); writes tiles first_tile .. first_tile+n_tiles-1
 * (row-major tile order) as normalised [n][tile][tile][4]. mean3/std3 are HOST arrays. */
int pylc_image_pack_tiles(const float* img, int Cimg, int H, int W, int tile, int stride, int first_tile, int n_tiles,
                          const float* mean3, const float* std3, float* out, void* stream);
/* logits: [rows*cols][tile][tile][pitch] (NHWC tiles in row-major order), stride = tile or tile/2.
 * mask: uint8 [rows*stride + (tile-stride)][cols*stride + (tile-stride)]. */
int pylc_stitch_argmax(const float* logits, int pitch, int rows, int cols, int tile, int stride, int C,
                       unsigned char* mask, void* stream);
/* out_rgb[oy][ox][3] = palette[mask[floor(oy*h/oh)][floor(ox*w/ow)]]; palette_rgb: device uint8 [n_classes][3]. */
int pylc_colourize_resize(const unsigned char* mask, int h, int w, const unsigned char* palette_rgb,
                          unsigned char* out_rgb, int oh, int ow, void* stream);

/* Confusion matrix cm[t*C + p] += 1 over n pixels of class-index masks (uint8 or int64; *_bytes = 1 or 8); the scores of
 * utils/metrics.py:64-88 (weighted F1, weighted IoU = the "mIoU", MCC, normalised matrix) are functions of it.
 * force_coverage applies Evaluator.validate()'s overwrite of the first C pixels (utils/evaluate.py:171-174). cm must be
 * zeroed by the caller; integer atomics make the result exact and order-independent. */
int pylc_confusion_matrix(const void* y_true, int true_bytes, const void* y_pred, int pred_bytes, long long n, int C,
                          int force_coverage, unsigned long long* cm, void* stream);

/* The general form: ((x - mean[c]) / std[c]) / denom on float (is_u8 = 0) or uint8 (is_u8 = 1) tiles.  denom = 255 is
 * pylc_image_pack[_u8]; denom = 1 is the reference's grayscale `default=True` branch, which omits the division by 255
 * (models/model.py:428-430). */
int pylc_image_pack_denom(const void* img_nchw, int is_u8, int B, int Cimg, int H, int W, const float* mean3, const float* std3,
                          float denom, float* out_nhwc4, void* stream);
/* pylc_image_pack for uint8 tiles (the dtype the HDF5 database stores, db/database.py:218-233; db/buffer.py:62 converts to
 * float32 on the host): the H2D copy carries 1 byte per sample instead of 4. */
int pylc_image_pack_u8(const unsigned char* img_nchw, int B, int Cimg, int H, int W, const float* mean3,
                       const float* std3, float* out_nhwc4, void* stream);

/* Layout converters for module-boundary tensors (logits): [B][H][W][pitch] <-> [B][C][H][W]. */
int pylc_nhwc_to_nchw(const float* x, int x_pitch, float* y, int B, int H, int W, int C, void* stream);
int pylc_nchw_to_nhwc(const float* x, float* y, int y_pitch, int B, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MultiLoss: models/modules/loss.py:66-69 (CE), :137-146 (Dice), :174-189 (Focal), :107-112
 * (weighted sum) as ONE per-pixel pass forward and ONE backward.
 * logits: NHWC [N][pitch] (N = B*H*W pixels), target: int64 [N].
 * ------------------------------------------------------------------------------------------- */
#define PYLC_MAX_CLASSES 16
size_t pylc_multiloss_workspace_floats(long long N, int C);
/* stats[0] = sum_n w_t * (-log p_t), stats[1] = sum_n w_t, stats[2] = sum_n focal_n,
 * stats[3 + c] = I_c, stats[3 + C + c] = sum_n p_c, stats[3 + 2C + c] = count_c   (3 + 3C floats).
 * These are the quantities a data-parallel run all-reduces before the non-linear Dice / weighted-CE
 * finalisation.  class_weights may be NULL (unweighted CE). */
int pylc_multiloss_stats(const float* logits, int pitch, const int64_t* target, long long N, int C,
                         const float* class_weights, float* stats, float* workspace, void* stream);
/* losses[0..3] = total, ce, dice, focal from (all-reduced) stats and the GLOBAL pixel count. */
int pylc_multiloss_finalize(const float* stats, double n_global, int C, float w_ce, float w_dice, float w_focal,
                            float* losses, void* stream);
/* dlogits = grad_scale[0] * d(total)/d(logits), using the same (global) stats.  amax_bits (may be NULL): receives the IEEE bits of
 * max|dlogits| (the operand range of the conv backward that reads them). */
int pylc_multiloss_bwd(const float* logits, int pitch, const int64_t* target, long long N, int C,
                       const float* class_weights, const float* stats, double n_global,
                       float w_ce, float w_dice, float w_focal, const float* grad_scale,
                       float* dlogits, int dpitch, unsigned int* amax_bits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser: torch.nn.utils.clip_grad_norm_(params, 0.5) models/model.py:326 + torch.optim.AdamW
 * models/model.py:240-245 over ONE flat fp32 arena holding every parameter.
 * ------------------------------------------------------------------------------------------- */
size_t pylc_sqnorm_workspace_floats(long long n);
/* out[0] = ||g||_2, out[1] = clip coefficient min(1, max_norm / (||g|| + 1e-6)). */
int pylc_grad_norm_clip(const float* g, long long n, float max_norm, float* out2, float* workspace, void* stream);
/* p,m,v updated in place; g is read as g * coef[1] (coef = out2 of pylc_grad_norm_clip, or NULL for 1). */
int pylc_adamw_step(float* p, const float* g, float* m, float* v, long long n, const float* coef,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
/* pylc_adamw_step that also returns, per parameter segment [seg_offsets[s], seg_offsets[s+1]) of the arena (DEVICE int64[seg_count + 1],
 * multiples of 4, seg_offsets[0] = 0, seg_offsets[seg_count] = n), the IEEE bits of max|p| AFTER the update -- what pylc_amax_segments
 * would read back in a pass of its own (same values, bit for bit).  n % 4 == 0. */
int pylc_adamw_step_ranges(float* p, const float* g, float* m, float* v, long long n, const float* coef,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           const long long* seg_offsets, int seg_count, unsigned int* seg_amax_bits, void* stream);
/* SGD with momentum (models/model.py:246-251): buf = mu*buf + g ; p -= lr*buf. */
int pylc_sgd_step(float* p, const float* g, float* buf, long long n, const float* coef, float lr, float momentum,
                  int step, void* stream);

/* Dropout (aspp.py:70, decoder.py:33,37, unet.py:120): mask from a counter-based hash of
 * (seed, element index); out = x * keep / (1 - p).  The same call with dy regenerates the mask. */
int pylc_dropout(const float* x, int x_pitch, float* out, int out_pitch, long long M, int C, float p,
                 uint64_t seed, void* stream);

#ifdef PYLC_EXPERIMENTAL      /* measured: no gain from confining the wgrad stream (profiles/r03_cumask_ab.txt) */
/* ---------------------------------------------------------------------------------------------
 * Streams confined to a subset of the compute units (no reference counterpart: the reference runs one CUDA stream).  The weight-gradient
 * kernels run on a second HIP stream beside the BatchNorm backward passes (pylc_amd/ops.py); created through this entry point that stream
 * only ever occupies `n_cus` of the 256 compute units, so that the HBM-bound passes of the main stream keep the others to themselves.
 * Bit i of HIP's mask addresses CU (i / 8) of XCD (i % 8): taking the n_cus lowest (from_top = 0) or highest (from_top = 1) bits spreads
 * the share evenly over the eight XCDs.  n_cus must be a multiple of 8 in [8, 256].  The stream is the caller's to destroy.
 * --------------------------------------------------------------------------------------------- */
int pylc_stream_create_cu_mask(int n_cus, int from_top, void** stream_out);
int pylc_stream_destroy(void* stream);
#endif

/* ---------------------------------------------------------------------------------------------
 * Profiling / A-B knobs (tools/, not needed by a caller)
 * --------------------------------------------------------------------------------------------- */
/* 0: 128x128 conv tiles only; 1: lock-step 256x128 8-wave tile; 2 (default): 256x128 ping-pong kernel for f16x3 */
int pylc_debug_set_big_tile(int mode);
/* bit 3 (8): wgrad without the uniform-geometry fast path; bit 4 (16): finer stamps (see pylc_debug_pp_stamps);
 * bit 6 (64): ping-pong kernel as persistent blocks (one per CU) instead of one block per tile; bit 7 (128): padded
 * 80-byte LDS rows and two stages instead of swizzled 64-byte rows and three; bit 8 (256): 32x32x16 instead of 16x16x32 MFMAs;
 * bit 12 (4096): ping-pong kernel walks the reduction with channel chunks innermost (the old order) instead of taps innermost;
 * bit 10 (1024): take the 256x128 tile even for launches of fewer than 192 tiles (tools/pp_stamps.py: a tile's phases with few
 * CUs active); bit 19 (524288): 1x1 launches stay on the per-tile kernel instead of the persistent one (conv_pl.hip gg_plp_kernel: the
 * bit-identity reference of tests/test_planes_gpu.py, A/B tools/plp_ab.py); bits 20-22: grid / wait variants of that kernel (tools/plp_ab.py) */
int pylc_debug_pp_flags(int flags);
/* conv_pl.hip, 128-row tiles: start delay of the second block of every CU in the first round of blocks, in units of 2048 cycles
 * (< 0: the launch heuristic, about half a tile; 0: none) -- A/B knob for tools/pl_stagger_ab.py */
int pylc_debug_stagger(int units);
/* dwconv.hip, one-plane fp16 depthwise convs (A/B knob, env PYLC_DW_TILES; default 3): bit 0 = LDS-tiled kernels for stride 1 / dilation 1
 * (else the strip kernels), bit 1 = LDS-tiled kernels for stride 2 and for dilation 2 (else those shapes are not half-eligible) */
int pylc_debug_dw_tiles(int on);
/* wgrad_pl.hip: 1 = the 128 x 128 wgrad takes its products as v_mfma_f32_16x16x32_f16 (swizzled unpadded LDS rows) instead of
 * v_mfma_f32_32x32x16_f16: same terms, fp32-rounding-level differences.  A/B knob, env PYLC_WG_M16. */
int pylc_debug_wgrad_m16(int on);
/* wgrad_pl.hip: 1 = the 16x16x32 form takes its operand tiles by LDS-DMA into two 32 KB LDS stages (no staging registers, no LDS stores,
 * one barrier per K-step).  Bit-identical to the register-staged form.  A/B knob, env PYLC_WG_DMA. */
int pylc_debug_wgrad_dma(int on);
/* wgrad_pl.hip: 1 runs the 128 x 128 f16x3 wgrad with one accumulator set under 128 registers per wave (A/B knob) */
int pylc_debug_wgrad_acc1(int on);
/* wgrad_pl.hip: operand staging register sets -- 0: one set, the loads of tile s + 1 fly while tile s is multiplied; 1 (default): two sets
 * (loads two tiles ahead, counted vmcnt) for multi-tap filters; 2: two sets for every wgrad.  A/B knob, env PYLC_WG_SETS (DESIGN.md 5.2 g). */
int pylc_debug_wgrad_sets(int mode);
/* wgrad_pl.hip rasterisation experiments (tools/wgrad_traffic.py): bit 0 = blocks in plain blockIdx order (no XCD remap), bit 1 = split
 * index fastest (the blocks that share a pixel chunk far apart), bit 2 = taps slowest (the round-3 order).  0 = the product's order
 * (taps fastest, split slowest, XCD-contiguous). */
int pylc_debug_wgrad_flags(int flags);
/* Longest reduction, in K-steps of 32 pixels, of one block of a multi-tap wgrad (default 256; 0 = no cap, the round-3 plan): shorter
 * blocks start together round after round, so that the blocks sharing a pixel chunk stay within the L2's reach of each other. */
int pylc_debug_wgrad_max_steps(int steps);
/* The next forward convs that take the ping-pong kernel record s_memtime stamps of block 0 (waves 0 and 4) at every
 * segment boundary into buf (2 x 256 uint64, device memory); NULL switches it off (tools/pp_stamps.py). */
int pylc_debug_pp_stamps(unsigned long long* buf);

/* ---- data-parallel exchange over RCCL / xGMI, one process per GPU ------------------------------------------
 * The reference has no working multi-GPU path: nn.DataParallel is commented out (models/model.py:186-188) and the vendored
 * SynchronizedBatchNorm2d (models/sync_batchnorm/batchnorm.py:48-125 + comm.py's master / slave queues: reduce [sum, sumsq, count] on
 * the master, broadcast mean / inv_std) is never constructed.  These entries are that exchange as collectives: communicator set-up, an
 * in-place SUM all-reduce (gradient buckets of the flat arena, loss-head partial sums, BatchNorm backward sums) and the SyncBN moment
 * exchange.  RCCL is resolved at run time (the librccl.so.1 already in the process, else the system's); without one the calls return
 * PYLC_ERR_UNSUPPORTED.  Everything is enqueued on the caller's stream.  pylc_amd uses them when PYLC_COMM=native (default: the same
 * collectives through torch.distributed's RCCL backend). */
#define PYLC_COMM_ID_BYTES 128
int pylc_comm_available(void);                                           /* PYLC_OK if an RCCL could be resolved in this process (no GPU call, no collective): the probe every rank runs BEFORE any rank starts a communicator hand-shake */
int pylc_comm_unique_id(void* id_out);                                   /* PYLC_COMM_ID_BYTES bytes; one rank makes it, the caller hands it to all */
int pylc_comm_init(const void* id, int rank, int world, void** comm_out);   /* collective; the current HIP device is this rank's GPU */
int pylc_comm_allreduce(void* comm, void* buf, long long count, int dtype, void* stream);      /* in-place SUM; dtype 0 = fp32, 1 = fp64 */
int pylc_comm_syncbn_reduce(void* comm, double* moments, int channels, void* stream);          /* [sum | sumsq | n]: 2 C + 1 doubles, in place */
int pylc_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* PYLC_HIP_H */
