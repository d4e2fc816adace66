// Declarations shared by the implicit-GEMM conv kernels (conv_igemm.hip: fp32 operands split in the kernel; conv_pl.hip: operands
// pre-split into fp16 planes by their producers).
#pragma once
#include "common.h"
#include <type_traits>

namespace pylc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int LDB = 80;   // bf16x6 mode: bytes per LDS row per plane (32 bf16 = 64 B + 16 B pad)

// Exact 3-way split of four fp32 values into bf16 planes by truncation: x = x0 + x1 + x2, each piece the top 16 bits of
// the running remainder (8 significant bits), packed two per dword in k order.
__device__ __forceinline__ void split3(const float (&x)[4], uint2& p0, uint2& p1, uint2& p2) {
    unsigned u0[4], u1[4], u2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u0[e] = __float_as_uint(x[e]);
        const float r1 = x[e] - __uint_as_float(u0[e] & 0xFFFF0000u);
        u1[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1[e] & 0xFFFF0000u);
        u2[e] = __float_as_uint(r2);
    }
    p0.x = (u0[0] >> 16) | (u0[1] & 0xFFFF0000u); p0.y = (u0[2] >> 16) | (u0[3] & 0xFFFF0000u);
    p1.x = (u1[0] >> 16) | (u1[1] & 0xFFFF0000u); p1.y = (u1[2] >> 16) | (u1[3] & 0xFFFF0000u);
    p2.x = (u2[0] >> 16) | (u2[1] & 0xFFFF0000u); p2.y = (u2[2] >> 16) | (u2[3] & 0xFFFF0000u);
}
__device__ __forceinline__ void split3(const float __attribute__((ext_vector_type(4))) v, uint2& p0, uint2& p1, uint2& p2) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    split3(x, p0, p1, p2);
}

// 2-way fp16 split of four scaled fp32 values: h0 = rn16(s*x), h1 = rn16(2048 (s*x - h0)); s*x = h0 + h1/2048 up to
// 2^-23 |s*x|.  s is a power of two chosen from the tensor's max magnitude, so s*x is exact.
// Scalar f32 ops on purpose: packed f32 VALU (v_pk_mul_f32 / v_pk_add_f32) issues at a third of the rate beside MFMAs
// (MI355X_MICROARCH.md, constants table); the residual is one mixed-precision FMA per element (v_fma_mix_f32).  Measured
// alternatives: v_fma_mixlo/hi_f16 (multiply + convert in one, 3 instead of 4 instructions per element) is slower.
__device__ __forceinline__ void split2(const f32x4 v, float s, uint2& p0, uint2& p1) {
    const float s2 = s * 2048.f;
    const f32x2 lo = {v.x * s, v.y * s}, hi = {v.z * s, v.w * s};
    const f16x2 l0 = __builtin_convertvector(lo, f16x2), h0 = __builtin_convertvector(hi, f16x2);
    const f32x2 rl = {__builtin_fmaf((float)l0.x, -2048.f, v.x * s2), __builtin_fmaf((float)l0.y, -2048.f, v.y * s2)};
    const f32x2 rh = {__builtin_fmaf((float)h0.x, -2048.f, v.z * s2), __builtin_fmaf((float)h0.y, -2048.f, v.w * s2)};
    const f16x2 l1 = __builtin_convertvector(rl, f16x2), h1 = __builtin_convertvector(rh, f16x2);
    p0.x = __builtin_bit_cast(unsigned, l0); p0.y = __builtin_bit_cast(unsigned, h0);
    p1.x = __builtin_bit_cast(unsigned, l1); p1.y = __builtin_bit_cast(unsigned, h1);
}

// power-of-two scale that maps a tensor with max magnitude `amax` (given as float bits) into [2^14, 2^15): exact to apply
// and to undo, keeps the fp16 pieces clear of overflow with the low piece inside the (sub)normal range for 40 binades.
__device__ __forceinline__ float pow2_scale_for(unsigned amax_bits) {
    int e = (int)((amax_bits >> 23) & 0xFFu);                  // biased exponent of amax (0: zero / denormal)
    int se = 127 + 14 - (e - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}

// 1 / s for a power of two s in [2^-126, 2^127] by exponent arithmetic (2^-127 is the subnormal 0x00400000)
__device__ __forceinline__ float pow2_inv(float s) {
    const unsigned e = (__float_as_uint(s) >> 23) & 0xFFu;
    return __uint_as_float(e >= 254u ? 0x00400000u : (254u - e) << 23);
}

constexpr int BK = 32;    // reduction depth per LDS stage
constexpr int LDT = 36;   // padded LDS row (floats): 144 B rows -> conflict-free ds_read_b128 of k-slices

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (observed round-robin dispatch), so give every
// XCD a contiguous range of logical tiles -> neighbouring tiles (which share halos / weight panels) share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct GatherGemmArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int M, P, Q;            // M = B*P*Q logical output pixels
    int IH, IW, Cin, x_pitch;
    int in_sh, in_sw;       // input coordinate of pixel (p,q), tap (r,s): (p*in_sh + dh0 + r*dh_step, ...)
    int TR, TS;
    int dh0, dh_step, dw0, dw_step;
    int w_off0, w_step_r, w_step_s, w_row_stride;
    int N, N_store;         // valid output channels / channels written (N rounded up to 4 inside the pitch)
    int OH, OW, out_sh, out_sw, oh0, ow0, y_pitch;
    int accumulate;
    // dgrad of a block's first conv whose input also feeds the block's residual add (resnet.py:36-51): y = result + relu'(add_src), i.e. the
    // residual branch's gradient is formed HERE from the gradient of the block output (add_src, same geometry and pitch as y, dense) and
    // the 1-bit ReLU mask of bn.hip (add_mask; NULL = no mask) instead of being written by the BatchNorm backward and re-read here
    const float* add_src;
    const unsigned char* add_mask;
    // dgrad that also takes the backward sums of the BatchNorm whose OUTPUT gradient it produces (conv_pl.hip pl_epilogue "BatchNorm-backward
    // mode"): bn_y = that BatchNorm's input (fp32, the geometry and pitch of this launch's output, dense), per-channel mean / invstd, and its
    // ReLU's mask source -- bn_scale / bn_shift (mask recomputed as y * scale + shift > 0) or the 1-bit mask bn_mask; bn_relu = 0: no ReLU.
    // The per-tile partials go to `stats` ([tiles_m][sum g xhat | sum g]), max |g| to bn_gmax.
    // output written as a ONE-PLANE fp16 tensor (precision mode 3, conv_pl.hip): element = rn16(s v) with s from the bound
    // out_bound_k * amax_x * amax_w (reduction length x operand bounds: loose by the usual few binades, which fp16's exponent range absorbs);
    // the bound is written to *out_bound for the consumer
    int out_half;
    float out_bound_k;
    unsigned* out_bound;
    const float* bn_y;
    const float* bn_mean;
    const float* bn_invstd;
    const float* bn_scale;
    const float* bn_shift;
    const unsigned char* bn_mask;
    unsigned* bn_gmax;
    int bn_relu;
    // fused inference epilogue (pylc_conv2d_fwd_bnact): val = relu(val * ep_scale[n] + ep_shift[n] + ep_res[...]); ep_amax
    // (zero-initialised by the caller) is max-accumulated with the range of what is stored
    const float* ep_scale;
    const float* ep_shift;
    const float* ep_res;    // same geometry and pitch as y
    unsigned* ep_amax;
    int ep_relu;
    int ep_vec_ok;          // host: ep_scale, ep_shift and the bias (if any) are 16-byte aligned -- the lean inference epilogue reads them as float4
    // ... on fp16-plane tensors (conv_pl.hip, EP instantiations; pylc_conv2d_fwd_bnact_ex): the residual may arrive as planes
    // (ep_res_fmt 1 = one plane, 2 = two planes at + ep_res_plane_stride halves, scaled with the bound behind ep_res_scale), the output may
    // leave as planes (out_half = one plane; out_planes2 = two planes at + out_plane_stride halves).  The output's scale comes from the bound
    //     out_bound_k x TRUE max|x| (bound_x; amax_x is the -- looser -- bound x was SCALED with) x max|w| x max|scale| + max|shift| + max|residual|
    // formed in the kernel from device scalars and written to *out_bound; the TRUE maximum of what is stored is max-accumulated into
    // ep_amax for the next layer's bound, so the looseness of one layer's bound never compounds into the next
    const unsigned* bound_x;
    const unsigned* ep_scale_amax;
    const unsigned* ep_shift_amax;
    const unsigned* ep_res_amax;
    const unsigned* ep_res_scale;
    int ep_res_fmt;
    long long ep_res_plane_stride;
    int out_planes2;
    long long out_plane_stride;
    int tiles_n;
    int n_tiles;            // tiles_m * tiles_n (the persistent ping-pong kernel walks them)
    const unsigned* amax_x; // PREC 2: device scalars holding the float bits of max|x| and max|w| (upper bounds are fine)
    const unsigned* amax_w;
    long long x_bytes, w_bytes;   // extents of the x / w buffers (raw buffer loads of the ping-pong kernel)
    const void* w_planes;         // optional: the filter already split into two fp16 planes (pylc_weight_prepare), same
                                  // indexing as w, plane 1 at + w_plane_stride halves; scaled with the amax behind amax_w
    long long w_plane_stride;
    int w_il;                     // the filter planes are chunk-interleaved (PylcConvDesc.w_planes_fmt): plane 1 at + 32 halves, a 32-channel chunk every 64 halves
    const void* x_planes;         // optional: the A operand (activations / incoming gradient) already split into two fp16 planes by its
                                  // producer with the scale behind amax_x: plane p at + p * x_plane_stride halves, element (pixel, c) at
                                  // pixel * x_pitch + c (conv_pl.hip takes these launches)
    long long x_plane_stride;
    int nterms;                   // conv_pl.hip: 3 = f16x3 (both planes of both operands), 1 = plain fp16 (plane 0 only)
    int tile_bm;                  // conv_pl.hip (set by launch_gg_pl): pixel rows per tile = rows per statistics partial
    int halo_tiles_m;             // conv_pl.hip halo kernel: number of 16 x 16 patches (> M / 256 for ragged sizes); 0 otherwise
    int ident;                    // conv_pl.hip (set by launch_gg_pl): 1x1 / stride 1 / no padding -- input pixel == output pixel
    int stagger, stagger_blocks;  // conv_pl.hip (set by launch_gg_pl): start delay (units of 2048 cycles) of the SECOND block of each CU among
                                  // the first `stagger_blocks` blocks of the grid (128-row tiles, several rounds of blocks)
    int dbg_flags;           // tools/pp_stamps.py: 16 = finer stamps inside the store segment (STAMPS build only)
    unsigned long long* dbg; // debug builds of the ping-pong kernel: per-segment clock stamps of block 0 (else null)
    float* stats;           // optional [tiles_m][2][N_store]: per-M-tile column sums / sums of squares of the stored values (BatchNorm)
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// Sum over the 16 lanes of a DPP row (same fixed order in every lane): quad butterflies, then the two mirrors.
__device__ __forceinline__ float row_sum16(float v) {
#define PYLC_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    PYLC_DPP_ADD(0xB1);      // quad_perm [1,0,3,2]
    PYLC_DPP_ADD(0x4E);      // quad_perm [2,3,0,1]
    PYLC_DPP_ADD(0x141);     // row_half_mirror: lane i <-> 7 - i of each half row
    PYLC_DPP_ADD(0x140);     // row_mirror: lane i <-> 15 - i
#undef PYLC_DPP_ADD
    return v;
}


// ---- wgrad (conv_igemm.hip: fp32 operands; wgrad_pl.hip: fp16-plane operands) ----
struct WgradArgs {
    const float* x;
    const float* dy;
    float* out;             // dw, or split-K slab base
    int M, P, Q;            // pixels of dy (dense, pitch dy_pitch)
    int IH, IW, Cin, x_pitch, in_sh, in_sw;
    int TR, TS, dh0, dh_step, dw0, dw_step;
    int N, N_ld, dy_pitch;  // N valid couts; N_ld = couts readable from dy (rounded up to 4)
    int out_row_stride;     // floats between couts in dw = T*Cin
    int tiles_n, tiles_c;   // tiles over cout / (cin or taps*4)
    int splits, m_per_split;
    long long slab_stride;
    const unsigned* amax_dy;   // f16x3: device scalars with the float bits of max|dy| and max|x|
    const unsigned* amax_x;
    long long x_bytes, dy_bytes;   // buffer extents (FAST path: raw buffer loads)
    const void* x_planes;          // wgrad_pl.hip: both operands as fp16 planes (plane 1 at + *_plane_stride halves)
    const void* dy_planes;
    long long x_plane_stride, dy_plane_stride;
    int nterms;                    // 3 = f16x3, 1 = plain fp16 (plane 0 only)
    int dbg_flags;                 // wgrad_pl.hip rasterisation experiments (pylc_debug_wgrad_flags): 1 = no XCD remap, 2 = split index fastest
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__host__ __device__ constexpr int wg_rowb(int w) { return w * 2 + (w == 32 ? 0 : 64); }

__device__ __forceinline__ bf16x8 tr_frag(const char* p, int rowb) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * rowb));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}


int launch_wg_pl(WgradArgs& a, int cfg, long long grid, hipStream_t st);      // cfg: plan_wgrad's tile configuration (0, 1, 2)
int wgrad_pl_init();

// conv_pl.hip: the gather-GEMM whose A operand arrives as fp16 planes (GatherGemmArgs::x_planes != nullptr)
bool takes_pl(const GatherGemmArgs& a);
int launch_gg_pl(GatherGemmArgs& a, hipStream_t st);
int conv_pl_init();
// conv_stem.hip: the ResNet stem (7x7 / stride 2 / pad 3, 4 -> 64) as a patch kernel
bool takes_stem(const GatherGemmArgs& a);
int launch_stem_fwd(GatherGemmArgs& a, hipStream_t st);
int conv_stem_init();

extern int g_conv_precision;
extern int g_wg_flags;
extern int g_pp_flags;

}  // namespace pylc
