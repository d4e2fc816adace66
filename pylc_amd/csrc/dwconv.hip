// Depthwise 3x3 convolution (groups = C) for the Aligned-Xception separable convs, NHWC fp32.
// HBM-bound (9 MACs per element): no MFMA.  A thread owns one float4 channel vector, keeps its 36 filter
// taps in registers and walks output pixels of a row slab; the explicit TF-'SAME' padding of
// fixed_padding (models/backbone/xception.py:16-22; for k = 3: pad_beg = pad_end = dilation) is folded
// into the index math instead of materialising a padded tensor.
// Replaces F.pad + nn.Conv2d(groups=C) at xception.py:29-31,35-36 and their backward.
#include "common.h"
#include "slab.h"

namespace pylc {

struct DwGeom { int B, H, W, C, stride, dil, OH, OW, x_pitch, y_pitch; };

// w is [C][9]; gather the 9 taps of channels 4cv..4cv+3 into 9 float4 registers
__device__ __forceinline__ void load_taps(const float* __restrict__ w, int cv, f32x4 (&k)[9]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        k[t].x = w[(4 * cv + 0) * 9 + t]; k[t].y = w[(4 * cv + 1) * 9 + t];
        k[t].z = w[(4 * cv + 2) * 9 + t]; k[t].w = w[(4 * cv + 3) * 9 + t];
    }
}

__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, DwGeom d,
                                                     Slab g) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    if (ty >= g.RL) return;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    for (int cv = tx; cv < g.CV; cv += g.cols) {
        f32x4 k[9];
        load_taps(w, cv, k);
        for (long long r = r_begin + ty; r < r_end; r += g.RL) {
            const unsigned r32 = (unsigned)r;                 // < 2^31 pixels (host check): 32-bit divisions, not 64-bit ones
            const int ow = (int)(r32 % (unsigned)d.OW);
            const unsigned t = r32 / (unsigned)d.OW;
            const int oh = (int)(t % (unsigned)d.OH), b = (int)(t / (unsigned)d.OH);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kr = 0; kr < 3; ++kr) {
                const int h = oh * d.stride + (kr - 1) * d.dil;
                if ((unsigned)h >= (unsigned)d.H) continue;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const int ww = ow * d.stride + (ks - 1) * d.dil;
                    if ((unsigned)ww >= (unsigned)d.W) continue;
                    acc += k[kr * 3 + ks] * ld4(x + ((size_t)(b * d.H + h) * d.W + ww) * d.x_pitch + 4 * cv);
                }
            }
            st4(y + r * d.y_pitch + 4 * cv, acc);
        }
    }
}

// dx[b,h,w,c] = sum_{r,s} dy[b, (h + dil - r*dil)/stride, (w + dil - s*dil)/stride, c] * k[c][r][s]   (where divisible)
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, DwGeom d,
                                                       Slab g, int accumulate) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    if (ty >= g.RL) return;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    for (int cv = tx; cv < g.CV; cv += g.cols) {
        f32x4 k[9];
        load_taps(w, cv, k);
        for (long long r = r_begin + ty; r < r_end; r += g.RL) {
            const unsigned r32 = (unsigned)r;
            const int wi = (int)(r32 % (unsigned)d.W);
            const unsigned t = r32 / (unsigned)d.W;
            const int hi = (int)(t % (unsigned)d.H), b = (int)(t / (unsigned)d.H);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kr = 0; kr < 3; ++kr) {
                const int hn = hi - (kr - 1) * d.dil;
                if (hn < 0 || hn % d.stride != 0) continue;
                const int oh = hn / d.stride;
                if (oh >= d.OH) continue;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const int wn = wi - (ks - 1) * d.dil;
                    if (wn < 0 || wn % d.stride != 0) continue;
                    const int ow = wn / d.stride;
                    if (ow >= d.OW) continue;
                    acc += k[kr * 3 + ks] * ld4(dy + ((size_t)(b * d.OH + oh) * d.OW + ow) * d.y_pitch + 4 * cv);
                }
            }
            if (accumulate) acc += ld4(dx + r * d.x_pitch + 4 * cv);      // dx holds the other consumers' part of the gradient (ops.ResidualLink)
            st4(dx + r * d.x_pitch + 4 * cv, acc);
        }
    }
}

// partial[slab][9][C]
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                                                       DwGeom d, Slab g) {
    __shared__ f32x4 red[256];
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    for (int cb = 0; cb < g.CV; cb += g.cols) {
        const int cv = cb + tx;
        const bool active = ty < g.RL && cv < g.CV;
        f32x4 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active) {
            for (long long r = r_begin + ty; r < r_end; r += g.RL) {
                const unsigned r32 = (unsigned)r;
                const int ow = (int)(r32 % (unsigned)d.OW);
                const unsigned t = r32 / (unsigned)d.OW;
                const int oh = (int)(t % (unsigned)d.OH), b = (int)(t / (unsigned)d.OH);
                const f32x4 gy = ld4(dy + r * d.y_pitch + 4 * cv);
#pragma unroll
                for (int kr = 0; kr < 3; ++kr) {
                    const int h = oh * d.stride + (kr - 1) * d.dil;
                    if ((unsigned)h >= (unsigned)d.H) continue;
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        const int ww = ow * d.stride + (ks - 1) * d.dil;
                        if ((unsigned)ww >= (unsigned)d.W) continue;
                        acc[kr * 3 + ks] += gy * ld4(x + ((size_t)(b * d.H + h) * d.W + ww) * d.x_pitch + 4 * cv);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            red[threadIdx.x] = acc[t];
            __syncthreads();
            if (ty == 0 && cv < g.CV) {
                f32x4 s = acc[t];
                for (int k = 1; k < g.RL; ++k) s += red[k * g.cols + tx];
                st4(partial + ((size_t)blockIdx.x * 9 + t) * d.C + 4 * cv, s);
            }
            __syncthreads();
        }
    }
}

// ---- stride-1, dilation-1 fast path (all but a handful of the Aligned-Xception depthwise convs) ------------------------
// The pixel-per-thread kernels above issue 9 (10 for wgrad) 16-byte loads per output vector, nearly all of them L2 hits:
// they run at the L2 request rate, ~1.6 TB/s of useful traffic.  Here a thread walks a strip of TWO output rows along W
// with the 4x3 input window in registers: each step loads one new 4-row column (prefetched one step ahead) and produces
// two outputs -- 2 loads per output (3 for wgrad).  Tap order of the sums is the one of the general kernels, so forward
// and dgrad results are bit-identical to them.
struct DwStrip { int nseg, seg, pairs, n_strips, strips_per_block; };

// HIN / HOUT: the inputs (x, dy) / the output (y, dx) are one-plane fp16 tensors (precision mode 3: 2 bytes per element, scaled per tensor)
// instead of fp32.  HIN && !HOUT: the dgrad of a block's first depthwise conv, whose fp32 output accumulates into the block input's gradient.
struct DwHalf {
    const unsigned* in_bound;     // range bound (float bits) the `in` tensor was scaled with
    const unsigned* aux_bound;    // MODE 2: bound of dy
    const unsigned* w_amax;       // MODE 0 / 1: max |filter tap| (float bits): the output's bound is 9 * w_amax * in_bound (+ the old bound when accumulating)
    const unsigned* acc_bound;    // MODE 1 with accumulate: bound the values already in `out` were scaled with
    unsigned* out_bound;          // MODE 0 / 1: receives the output's bound (written by one thread; every thread derives the same value)
    // MODE 1 with an fp32 output (tiled stride-1 kernel): dx = dw^T(dy) + (mask bit ? add_src : 0) -- the gradient of a block input whose
    // other consumer is the ReLU'd residual add of the block's last BatchNorm (bn.hip "1-bit ReLU masks": one nibble per float4 vector)
    const float* add_src = nullptr;
    const unsigned char* add_mask = nullptr;
    // MODE 0 / 2 of the tiled kernels: `in` is not x but the INPUT y of the training-mode BatchNorm (+ ReLU) that produces x, scaled with
    // y_bound; the block applies x = act(y * in_scale[c] + in_shift[c]) to its LDS patch before it computes -- the BatchNorm's apply pass,
    // bit for bit (same operations, same fp16 rounding with in_bound's scale), without the tensor x ever being written (dw_bn_patch)
    const float* in_scale = nullptr;
    const float* in_shift = nullptr;
    const unsigned* y_bound = nullptr;
    int in_relu = 0;
    // inference (pylc_dwconv3x3_fwd_h_eval): the TRUE max|in| when the tensor was scaled with a looser bound -- the output's bound is then
    // 9 * w_amax * in_true, so that the looseness of one layer's bound does not compound into the next (conv_common.h GatherGemmArgs::bound_x)
    const unsigned* in_true = nullptr;
};

template <bool HALF>
__device__ __forceinline__ void dw_load_col(f32x4 (&c)[4], const void* in, const size_t (&rowo)[4], const bool (&rv)[4], bool col_ok, int col, int pitch,
                                            float inv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = (col_ok && rv[j]) ? ldq<HALF>(in, rowo[j] + (size_t)col * pitch, inv) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// MODE 0: y = dw(x) (in = x, aux = filter, out = y); MODE 1: dx = dw^T(dy) (in = dy, aux = filter, out = dx);
// MODE 2: partial[block][9][C] = sum over this block's pixels of dy (x) window(x) (in = x, aux = dy, out = partial)
// STATS (MODE 0): additionally stats[block][2C] = per-block (sum | sum of squares) of the outputs this block wrote, for the BatchNorm that
// follows every depthwise conv (xception.py:34-39) -- the layout pylc_bn_finalize_from_partial reads; saves that layer's statistics pass.
template <int MODE, bool STATS = false, bool HIN = false, bool HOUT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 2 : 3, 8))) void dw_strip_kernel(const void* __restrict__ in, const void* __restrict__ aux, void* __restrict__ out,
                                                       DwGeom d, DwStrip s, int cols, int RL, int CV, int accumulate, float* __restrict__ stats = nullptr,
                                                       DwHalf hf = DwHalf{}) {
    __shared__ f32x4 red[(MODE == 2 || STATS) ? 256 : 1];
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    const int in_pitch = MODE == 1 ? d.y_pitch : d.x_pitch;
    const int out_pitch = MODE == 1 ? d.x_pitch : d.y_pitch;      // MODE 2: pitch of dy
    const int s_begin = blockIdx.x * s.strips_per_block;
    const int s_end = s_begin + s.strips_per_block < s.n_strips ? s_begin + s.strips_per_block : s.n_strips;
    float in_inv = 1.f, aux_inv = 1.f, out_scale = 1.f, acc_inv = 1.f;
    if constexpr (HIN) {
        in_inv = 1.f / half_scale_for(*hf.in_bound);
        if (MODE == 2) aux_inv = 1.f / half_scale_for(*hf.aux_bound);
    }
    if constexpr (HOUT && MODE != 2) {
        float b = 9.f * __uint_as_float(*hf.w_amax) * __uint_as_float(*(hf.in_true != nullptr ? hf.in_true : hf.in_bound));
        if (MODE == 1 && accumulate) { b += __uint_as_float(*hf.acc_bound); acc_inv = 1.f / half_scale_for(*hf.acc_bound); }
        out_scale = half_scale_for(__float_as_uint(b));
        if (blockIdx.x == 0 && threadIdx.x == 0) *hf.out_bound = __float_as_uint(b);
    }
    const float* const auxf = static_cast<const float*>(aux);
    for (int cb = 0; cb < CV; cb += cols) {
        const int cv = cb + tx;
        const bool active = ty < RL && cv < CV;
        f32x4 k[9];                                                 // filter taps (MODE 0/1) or accumulators (MODE 2)
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) k[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active) {
            if (MODE != 2) load_taps(auxf, cv, k);
            for (int sid = s_begin + ty; sid < s_end; sid += RL) {
                const int sgi = sid % s.nseg;
                const int t_ = sid / s.nseg;
                const int p = t_ % s.pairs, b = t_ / s.pairs;
                const int h0 = 2 * p, w0 = sgi * s.seg;
                const int w1 = w0 + s.seg < d.W ? w0 + s.seg : d.W;
                const bool two = h0 + 1 < d.H;
                size_t rowo[4];
                bool rv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int hh = h0 - 1 + j;
                    rv[j] = (unsigned)hh < (unsigned)d.H;
                    rowo[j] = ((size_t)(b * d.H + (rv[j] ? hh : 0)) * d.W) * in_pitch + 4 * cv;
                }
                const size_t o0 = ((size_t)(b * d.H + h0) * d.W) * out_pitch + 4 * cv;      // output rows h0, h0 + 1 (dy rows in MODE 2)
                const size_t o1 = o0 + (size_t)d.W * out_pitch;
                f32x4 ca[4], cb_[4], cc[4], cd[4];
                dw_load_col<HIN>(ca, in, rowo, rv, w0 > 0, w0 - 1, in_pitch, in_inv);
                dw_load_col<HIN>(cb_, in, rowo, rv, true, w0, in_pitch, in_inv);
                dw_load_col<HIN>(cc, in, rowo, rv, w0 + 1 < d.W, w0 + 1, in_pitch, in_inv);
                int ww = w0;
                // one step: L/M/R = window columns ww-1, ww, ww+1 (already loaded); N receives column ww+2 for the next step
#define PYLC_DW_STEP(L, M, R, N)                                                                                          \
    {                                                                                                                     \
        dw_load_col<HIN>(N, in, rowo, rv, ww + 2 < d.W && ww + 1 < w1, ww + 2, in_pitch, in_inv);                        \
        if (MODE == 2) {                                                                                                  \
            const f32x4 g0 = ldq<HIN>(aux, o0 + (size_t)ww * out_pitch, aux_inv);                                        \
            const f32x4 g1 = two ? ldq<HIN>(aux, o1 + (size_t)ww * out_pitch, aux_inv) : f32x4{0.f, 0.f, 0.f, 0.f};      \
            _Pragma("unroll") for (int kr = 0; kr < 3; ++kr) {                                                            \
                k[kr * 3 + 0] += g0 * L[kr]; k[kr * 3 + 1] += g0 * M[kr]; k[kr * 3 + 2] += g0 * R[kr];                    \
            }                                                                                                             \
            _Pragma("unroll") for (int kr = 0; kr < 3; ++kr) {                                                            \
                k[kr * 3 + 0] += g1 * L[kr + 1]; k[kr * 3 + 1] += g1 * M[kr + 1]; k[kr * 3 + 2] += g1 * R[kr + 1];        \
            }                                                                                                             \
        } else {                                                                                                          \
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};                                                   \
            _Pragma("unroll") for (int kr = 0; kr < 3; ++kr) {                                                            \
                const int j = MODE == 0 ? kr : 2 - kr;      /* dgrad: dy row h + 1 - kr, column w + 1 - ks */             \
                a0 += k[kr * 3 + 0] * (MODE == 0 ? L[j] : R[j]);                                                          \
                a0 += k[kr * 3 + 1] * M[j];                                                                               \
                a0 += k[kr * 3 + 2] * (MODE == 0 ? R[j] : L[j]);                                                          \
                a1 += k[kr * 3 + 0] * (MODE == 0 ? L[j + 1] : R[j + 1]);                                                  \
                a1 += k[kr * 3 + 1] * M[j + 1];                                                                           \
                a1 += k[kr * 3 + 2] * (MODE == 0 ? R[j + 1] : L[j + 1]);                                                  \
            }                                                                                                             \
            if (MODE == 1 && accumulate) {          /* dx += : the other consumers' part of the gradient is already there */  \
                a0 += ldq<HOUT>(out, o0 + (size_t)ww * out_pitch, acc_inv);                                               \
                if (two) a1 += ldq<HOUT>(out, o1 + (size_t)ww * out_pitch, acc_inv);                                      \
            }                                                                                                             \
            stq<HOUT>(out, o0 + (size_t)ww * out_pitch, a0, out_scale);                                                   \
            if (two) stq<HOUT>(out, o1 + (size_t)ww * out_pitch, a1, out_scale);                                          \
            if (STATS) {                                                                                                  \
                if (HOUT) {        /* statistics of the ROUNDED halves: the tensor the BatchNorm will read */               \
                    a0 = half4_to_f32(f32_to_half4(a0 * out_scale)) * (1.f / out_scale);                                  \
                    if (two) a1 = half4_to_f32(f32_to_half4(a1 * out_scale)) * (1.f / out_scale);                         \
                }                                                                                                         \
                st1 += a0; st2 += a0 * a0;                                                                                \
                if (two) { st1 += a1; st2 += a1 * a1; }                                                                   \
            }                                                                                                             \
        }                                                                                                                 \
        if (++ww >= w1) break;                                                                                            \
    }
                for (;;) {
                    PYLC_DW_STEP(ca, cb_, cc, cd)
                    PYLC_DW_STEP(cb_, cc, cd, ca)
                    PYLC_DW_STEP(cc, cd, ca, cb_)
                    PYLC_DW_STEP(cd, ca, cb_, cc)
                }
#undef PYLC_DW_STEP
            }
        }
        float* const outf = static_cast<float*>(out);
        if (STATS) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                red[threadIdx.x] = t == 0 ? st1 : st2;
                __syncthreads();
                if (ty == 0 && cv < CV) {
                    f32x4 sum = t == 0 ? st1 : st2;
                    for (int q = 1; q < RL; ++q) sum += red[q * cols + tx];
                    st4(stats + (size_t)blockIdx.x * 2 * d.C + t * d.C + 4 * cv, sum);
                }
                __syncthreads();
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                red[threadIdx.x] = k[t];
                __syncthreads();
                if (ty == 0 && cv < CV) {
                    f32x4 sum = k[t];
                    for (int q = 1; q < RL; ++q) sum += red[q * cols + tx];
                    st4(outf + ((size_t)blockIdx.x * 9 + t) * d.C + 4 * cv, sum);
                }
                __syncthreads();
            }
        }
    }
}

// ---- LDS-tiled stride-1 / dilation-1 kernels for one-plane fp16 tensors (precision mode 3) ---------------------------------------
// The strip kernels above run at the L2 REQUEST rate: two 4-row column loads per pair of outputs, and on a half tensor each request
// carries 8 bytes per lane instead of 16 -- 1.3-1.5 TB/s of useful traffic on the Aligned-Xception shapes (25 of configs[4]'s 80 ms).
// Here a block brings a (8 + 2) x (16 + 2) pixel patch of one 128-channel chunk to LDS by DMA -- every element fetched once with
// 16-byte requests, 256 contiguous bytes per pixel, padding pixels and channels past C arriving as zeros (out-of-range offsets) -- and
// computes the 8 x 16 outputs from LDS: a thread owns 4 channels and 2 adjacent columns and walks down the patch with a 3 x 4 window in
// registers (4 ds_read_b64 and 72 FMAs per output pair).  Neighbouring lanes then swap one of their two results (DPP), so that each lane
// holds 8 consecutive channels of ONE pixel: 16-byte stores.  MODE 2 stages the dy tile the same way.
// grid = groups x chunks; a block walks the tiles group, group + groups, ... of its chunk (statistics / filter-gradient partials: one
// row per group, as the strip kernels' per-block rows).
constexpr int DT_TH = 8, DT_TW = 16, DT_CC = 128;
constexpr int DT_IH = DT_TH + 2, DT_IW = DT_TW + 2;
constexpr int DT_PIX = DT_IH * DT_IW;                    // 180 patch pixels
constexpr int DT_PIXB = DT_CC * 2;                       // bytes per pixel of a chunk
constexpr int DT_IN_BYTES = DT_PIX * DT_PIXB;            // 46080
constexpr int DT_DY_BYTES = DT_TH * DT_TW * DT_PIXB;     // 32768
typedef __attribute__((address_space(3))) void* dw_lds_vptr;
struct DwTiles { int tiles_w, tiles_h, n_tiles, groups, chunks; };

template <int MODE>
constexpr int dw_tile_lds_bytes() {
    return MODE == 2 ? DT_IN_BYTES + DT_DY_BYTES : (DT_IN_BYTES > 2 * 256 * 32 ? DT_IN_BYTES : 2 * 256 * 32);
}

__device__ __forceinline__ f32x4 dw_fma4(f32x4 a, f32x4 b, f32x4 c) {      // the file is built with -ffp-contract=off: fused explicitly here
    return f32x4{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y), __builtin_fmaf(a.z, b.z, c.z), __builtin_fmaf(a.w, b.w, c.w)};
}
__device__ __forceinline__ float dw_swap1(float v) {        // value of lane ^ 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
}

// The deferred BatchNorm apply (DwHalf::in_scale): every thread transforms 16-byte pieces (8 channels of one pixel; a thread's channel
// piece is fixed: 256 threads, 16 pieces per pixel) of the patch in place -- the raw halves of y become the halves pylc_bn_apply_ex would
// have written for x: v = (h * y_inv) * scale + shift, ReLU, rn16(v * x_scale).  Pixels outside the tensor keep the zeros the DMA put there.
__device__ __forceinline__ void dw_bn_patch(char* lds, int npx, int pw, int ph0, int pw0, int SH, int SW, int c0, int C, const DwHalf& hf, float y_inv,
                                            float x_scale) {
    const int piece = threadIdx.x & 15;
    const int c = c0 + 8 * piece;
    if (c < C) {
        const f32x4 sc0 = ld4(hf.in_scale + c), sc1 = ld4(hf.in_scale + c + 4), sh0 = ld4(hf.in_shift + c), sh1 = ld4(hf.in_shift + c + 4);
        for (int px = threadIdx.x >> 4; px < npx; px += 16) {
            const int iy = px / pw, ix = px - iy * pw;
            if ((unsigned)(ph0 + iy) >= (unsigned)SH || (unsigned)(pw0 + ix) >= (unsigned)SW) continue;
            uint4* p = reinterpret_cast<uint4*>(lds + px * DT_PIXB + piece * 16);
            const uint4 raw = *p;
            f32x4 a = half4_to_f32(uint2{raw.x, raw.y}) * y_inv, b = half4_to_f32(uint2{raw.z, raw.w}) * y_inv;
            a = a * sc0 + sh0;
            b = b * sc1 + sh1;
            if (hf.in_relu) {
                a = f32x4{fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)};
                b = f32x4{fmaxf(b.x, 0.f), fmaxf(b.y, 0.f), fmaxf(b.z, 0.f), fmaxf(b.w, 0.f)};
            }
            const uint2 lo = f32_to_half4(a * x_scale), hi = f32_to_half4(b * x_scale);
            *p = uint4{lo.x, lo.y, hi.x, hi.y};
        }
    }
    __syncthreads();
}

// What both tiled kernels do with a thread's two results of one output row (columns 2 pt and 2 pt + 1, 4 channels each, in half units):
// lane pairs (q, q ^ 1) swap one result each, so that the even lane keeps column 2 pt and the odd lane column 2 pt + 1 with 8 consecutive
// channels (16-byte accesses); then scale, add what is accumulated into (MODE 1: the old dx, fp32 or half; the ReLU-masked residual
// gradient), collect the statistics, store.  `e`: element offset of this lane's 8 channels in the written tensor; `ok`: inside it.
template <int MODE, bool STATS, bool HOUT>
__device__ __forceinline__ void dw_emit_pair(const f32x4 a0, const f32x4 a1, int odd, bool ok, size_t e, void* __restrict__ out, int accumulate, float in_inv,
                                             float out_scale, float acc_inv, const DwHalf& hf, float (&st1)[8], float (&st2)[8]) {
    const f32x4 give = odd ? a0 : a1;
    const f32x4 got = {dw_swap1(give.x), dw_swap1(give.y), dw_swap1(give.z), dw_swap1(give.w)};
    const f32x4 mine = odd ? a1 : a0;
    float v[8];
    v[0] = odd ? got.x : mine.x; v[1] = odd ? got.y : mine.y; v[2] = odd ? got.z : mine.z; v[3] = odd ? got.w : mine.w;
    v[4] = odd ? mine.x : got.x; v[5] = odd ? mine.y : got.y; v[6] = odd ? mine.z : got.z; v[7] = odd ? mine.w : got.w;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= in_inv;
    if (!ok) return;
    if (MODE == 1 && accumulate) {
        if constexpr (HOUT) {
            const uint4 o = *reinterpret_cast<const uint4*>(static_cast<const _Float16*>(out) + e);
            const f32x4 lo = half4_to_f32(uint2{o.x, o.y}), hi = half4_to_f32(uint2{o.z, o.w});
            v[0] += lo.x * acc_inv; v[1] += lo.y * acc_inv; v[2] += lo.z * acc_inv; v[3] += lo.w * acc_inv;
            v[4] += hi.x * acc_inv; v[5] += hi.y * acc_inv; v[6] += hi.z * acc_inv; v[7] += hi.w * acc_inv;
        } else {
            const f32x4 lo = ld4(static_cast<const float*>(out) + e), hi = ld4(static_cast<const float*>(out) + e + 4);
            v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
        }
    }
    if constexpr (MODE == 1 && !HOUT) {
        if (hf.add_src != nullptr) {          // this lane's 8 channels = two float4 vectors = one mask byte
            const f32x4 lo = ld4(hf.add_src + e), hi = ld4(hf.add_src + e + 4);
            const unsigned m = hf.add_mask[e >> 3];
            v[0] += (m & 1u) ? lo.x : 0.f; v[1] += (m & 2u) ? lo.y : 0.f; v[2] += (m & 4u) ? lo.z : 0.f; v[3] += (m & 8u) ? lo.w : 0.f;
            v[4] += (m & 16u) ? hi.x : 0.f; v[5] += (m & 32u) ? hi.y : 0.f; v[6] += (m & 64u) ? hi.z : 0.f; v[7] += (m & 128u) ? hi.w : 0.f;
        }
    }
    if constexpr (HOUT) {
        const uint2 lo = f32_to_half4(f32x4{v[0], v[1], v[2], v[3]} * out_scale), hi = f32_to_half4(f32x4{v[4], v[5], v[6], v[7]} * out_scale);
        if (STATS) {
            // the BatchNorm that follows normalises the ROUNDED halves: its statistics are those of the tensor it will read
            const float inv = 1.f / out_scale;          // a power of two: exact
            const f32x4 rl = half4_to_f32(lo) * inv, rh = half4_to_f32(hi) * inv;
            v[0] = rl.x; v[1] = rl.y; v[2] = rl.z; v[3] = rl.w; v[4] = rh.x; v[5] = rh.y; v[6] = rh.z; v[7] = rh.w;
#pragma unroll
            for (int i = 0; i < 8; ++i) { st1[i] += v[i]; st2[i] += v[i] * v[i]; }
        }
        *reinterpret_cast<uint4*>(static_cast<_Float16*>(out) + e) = uint4{lo.x, lo.y, hi.x, hi.y};
    } else {
        if (STATS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { st1[i] += v[i]; st2[i] += v[i] * v[i]; }
        }
        st4(static_cast<float*>(out) + e, f32x4{v[0], v[1], v[2], v[3]});
        st4(static_cast<float*>(out) + e + 4, f32x4{v[4], v[5], v[6], v[7]});
    }
}

// Block reduction of the per-lane statistics (after dw_emit_pair every lane holds 8 channels: 16 contributions -- 8 pixel threads x 2
// lanes of a pair -- per 8-channel group, summed in a fixed order) into row `group` of the partials [groups][2][C].  LDS: 2 x 256 x 8 floats.
__device__ __forceinline__ void dw_reduce_stats(float* red, const float (&st1)[8], const float (&st2)[8], int tid, int c0, int C, int group, float* __restrict__ stats) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[tid * 8 + i] = st1[i]; red[(256 + tid) * 8 + i] = st2[i]; }
    __syncthreads();
    if (tid < 32) {
        const int o = tid & 15, which = tid >> 4;
        const int c = c0 + 8 * o;
        if (c < C) {
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int od = 0; od < 2; ++od) {
                    const int src = which * 256 + p * 32 + 2 * o + od;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s[i] += red[src * 8 + i];
                }
            float* dst = stats + (size_t)group * 2 * C + which * C + c;
            st4(dst, f32x4{s[0], s[1], s[2], s[3]});
            st4(dst + 4, f32x4{s[4], s[5], s[6], s[7]});
        }
    }
}

// Block reduction of the nine filter-gradient accumulators over the 8 pixel threads into partial[group][9][C] (LDS: 256 float4).
__device__ __forceinline__ void dw_reduce_wgrad(f32x4* red, const f32x4 (&k)[9], int tid, int q, int pt, bool c_ok, int cq, int C, int group, float sc,
                                                float* __restrict__ partial) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        red[tid] = k[i];
        __syncthreads();
        if (pt == 0 && c_ok) {
            f32x4 sum = k[i];
            for (int p = 1; p < 8; ++p) sum += red[p * 32 + q];
            st4(partial + ((size_t)group * 9 + i) * C + cq, sum * sc);
        }
        __syncthreads();
    }
}

template <int MODE, bool STATS, bool HOUT>
__global__ __launch_bounds__(256) void dw_tile_kernel(const void* __restrict__ in, const void* __restrict__ aux, void* __restrict__ out, DwGeom d,
                                                      DwTiles t, int accumulate, float* __restrict__ stats, DwHalf hf) {
    extern __shared__ __attribute__((aligned(16))) char dt_lds[];
    char* const lin = dt_lds;
    char* const ldy = dt_lds + DT_IN_BYTES;
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.x % t.chunks, group = blockIdx.x / t.chunks;
    const int c0 = chunk * DT_CC;
    const int q = tid & 31, pt = tid >> 5;
    const int cq = c0 + 4 * q;                              // this thread's 4 channels
    const bool c_ok = cq < d.C;
    const float in_inv = 1.f / half_scale_for(*hf.in_bound);
    float aux_inv = 1.f, out_scale = 1.f, acc_inv = 1.f;
    if (MODE == 2) aux_inv = 1.f / half_scale_for(*hf.aux_bound);
    if constexpr (HOUT && MODE != 2) {
        float b = 9.f * __uint_as_float(*hf.w_amax) * __uint_as_float(*(hf.in_true != nullptr ? hf.in_true : hf.in_bound));
        if (MODE == 1 && accumulate) { b += __uint_as_float(*hf.acc_bound); acc_inv = 1.f / half_scale_for(*hf.acc_bound); }
        out_scale = half_scale_for(__float_as_uint(b));
        if (blockIdx.x == 0 && threadIdx.x == 0) *hf.out_bound = __float_as_uint(b);
    }
    f32x4 k[9];                                             // filter taps (MODE 0 / 1, dgrad: flipped) or accumulators (MODE 2)
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE != 2 && c_ok) {
        f32x4 kt[9];
        load_taps(static_cast<const float*>(aux), cq >> 2, kt);
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i] = MODE == 0 ? kt[i] : kt[8 - i];      // dx = dy correlated with the filter rotated by 180 degrees
    }
    const long long in_bytes = (long long)d.B * d.H * d.W * d.C * 2;
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(in), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(MODE == 2 ? aux : in), 0, (int)in_bytes, 0x00020000);
    const int piece_c = c0 + 8 * (lane & 15);              // loader: 16 lanes x 16 bytes per pixel, 4 pixels per DMA instruction
    const bool piece_ok = piece_c < d.C;
    float st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st1[i] = st2[i] = 0.f;
    const int oct = q >> 1, odd = q & 1;
    const int cp = c0 + 8 * oct;                            // after the swap: this lane's 8 channels ...
    const bool cp_ok = cp < d.C;

    for (int tile = group; tile < t.n_tiles; tile += t.groups) {
        const int tx = tile % t.tiles_w, t1 = tile / t.tiles_w;
        const int ty = t1 % t.tiles_h, b = t1 / t.tiles_h;
        const int h0 = ty * DT_TH, w0 = tx * DT_TW;
        for (int i = wave; i < DT_PIX / 4; i += 4) {
            const int pi = 4 * i + (lane >> 4);
            const int iy = pi / DT_IW, ix = pi - iy * DT_IW;
            const int h = h0 - 1 + iy, w = w0 - 1 + ix;
            const bool ok = piece_ok & ((unsigned)h < (unsigned)d.H) & ((unsigned)w < (unsigned)d.W);
            const unsigned off = ok ? ((unsigned)((b * d.H + h) * d.W + w) * (unsigned)d.C + (unsigned)piece_c) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (dw_lds_vptr)(lin + i * 1024), 16, off, 0, 0, 0);
        }
        if (MODE == 2) {
            for (int i = wave; i < DT_TH * DT_TW / 4; i += 4) {
                const int pi = 4 * i + (lane >> 4);
                const int h = h0 + (pi >> 4), w = w0 + (pi & 15);
                const bool ok = piece_ok & (h < d.H) & (w < d.W);
                const unsigned off = ok ? ((unsigned)((b * d.H + h) * d.W + w) * (unsigned)d.C + (unsigned)piece_c) * 2u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_dy, (dw_lds_vptr)(ldy + i * 1024), 16, off, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE != 1 && hf.in_scale != nullptr)
            dw_bn_patch(lin, DT_PIX, DT_IW, h0 - 1, w0 - 1, d.H, d.W, c0, d.C, hf, 1.f / half_scale_for(*hf.y_bound), half_scale_for(*hf.in_bound));

        const char* const pin = lin + (2 * pt) * DT_PIXB + q * 8;
        f32x4 win[3][4];
        auto ldrow = [&](int iy, f32x4 (&row)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) row[j] = half4_to_f32(*reinterpret_cast<const uint2*>(pin + (iy * DT_IW + j) * DT_PIXB));
        };
        ldrow(0, win[0]);
        ldrow(1, win[1]);
        const int colx = w0 + 2 * pt + odd;                // ... of the pixel in this column
#pragma unroll
        for (int r = 0; r < DT_TH; ++r) {
            ldrow(r + 2, win[(r + 2) % 3]);
            if (MODE == 2) {
                const f32x4 g0 = half4_to_f32(*reinterpret_cast<const uint2*>(ldy + (r * DT_TW + 2 * pt) * DT_PIXB + q * 8));
                const f32x4 g1 = half4_to_f32(*reinterpret_cast<const uint2*>(ldy + (r * DT_TW + 2 * pt + 1) * DT_PIXB + q * 8));
#pragma unroll
                for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        k[kr * 3 + ks] = dw_fma4(g0, win[(r + kr) % 3][ks], k[kr * 3 + ks]);
                        k[kr * 3 + ks] = dw_fma4(g1, win[(r + kr) % 3][ks + 1], k[kr * 3 + ks]);
                    }
            } else {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        a0 = dw_fma4(k[kr * 3 + ks], win[(r + kr) % 3][ks], a0);
                        a1 = dw_fma4(k[kr * 3 + ks], win[(r + kr) % 3][ks + 1], a1);
                    }
                const int oh = h0 + r;
                const bool ok = cp_ok & (oh < d.H) & (colx < d.W);
                const size_t e = ((size_t)(b * d.H + oh) * d.W + colx) * d.C + cp;
                dw_emit_pair<MODE, STATS, HOUT>(a0, a1, odd, ok, e, out, accumulate, in_inv, out_scale, acc_inv, hf, st1, st2);
            }
        }
        __syncthreads();          // the patch is overwritten by the next tile's DMA (or by the reductions below)
    }

    if (STATS) dw_reduce_stats(reinterpret_cast<float*>(dt_lds), st1, st2, tid, c0, d.C, group, stats);
    if (MODE == 2) dw_reduce_wgrad(reinterpret_cast<f32x4*>(dt_lds), k, tid, q, pt, c_ok, cq, d.C, group, in_inv * aux_inv, static_cast<float*>(out));
}

// The same scheme for the Aligned Xception's other two geometries: stride 2 (the last separable conv of the entry-flow blocks) and
// dilation 2 at stride 1 (exit flow at output stride 16).  No register window -- a thread reads every tap of its two output columns
// from LDS (18 ds_read_b64 per output pair) -- since these are six layers of the network.  MODE 0 / 2 tile the OUTPUT (4 x 16 at
// stride 2, patch 9 x 33; 8 x 16 at dilation 2, patch 12 x 20); MODE 1 at stride 1 is MODE 0 with the filter rotated; MODE 1 at
// stride 2 tiles dx (8 x 16) over a 5 x 9 patch of dy and takes, per pixel parity, the one, two or four taps that reach it.
template <int S, int D> struct DwG {
    static constexpr int TH = S == 2 ? 4 : 8, TW = 16;
    static constexpr int PH = (TH - 1) * S + 2 * D + 1, PW = (TW - 1) * S + 2 * D + 1;
    static constexpr int NPX = PH * PW;
    static constexpr int IN_BYTES = ((NPX + 3) / 4) * 4 * DT_PIXB;
    static constexpr int DY_BYTES = TH * TW * DT_PIXB;
    // MODE 1 at stride 2: dx tile 8 x 16, dy patch 5 x 9
    static constexpr int BH = 8, BW = 16, QH = BH / 2 + 1, QW = BW / 2 + 1, NQ = QH * QW;
    static constexpr int BWD_BYTES = ((NQ + 3) / 4) * 4 * DT_PIXB;
};
template <int MODE, int S, int D>
constexpr int dw_tileg_lds_bytes() {
    const int red = 2 * 256 * 32;
    const int b = MODE == 2 ? DwG<S, D>::IN_BYTES + DwG<S, D>::DY_BYTES : ((MODE == 1 && S == 2) ? DwG<S, D>::BWD_BYTES : DwG<S, D>::IN_BYTES);
    return b > red ? b : red;
}

template <int MODE, int S, int D, bool STATS, bool HOUT>
__global__ __launch_bounds__(256) void dw_tileg_kernel(const void* __restrict__ in, const void* __restrict__ aux, void* __restrict__ out, DwGeom d,
                                                       DwTiles t, int accumulate, float* __restrict__ stats, DwHalf hf) {
    typedef DwG<S, D> G;
    constexpr bool BWD2 = MODE == 1 && S == 2;
    constexpr int TH = BWD2 ? G::BH : G::TH, TW = BWD2 ? G::BW : G::TW;       // tile of the tensor this kernel WRITES (MODE 2: of dy)
    constexpr int PW = BWD2 ? G::QW : G::PW, NPX = BWD2 ? G::NQ : G::NPX;      // patch of the tensor it reads through LDS
    extern __shared__ __attribute__((aligned(16))) char dt_lds[];
    char* const lin = dt_lds;
    char* const ldy = dt_lds + G::IN_BYTES;
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.x % t.chunks, group = blockIdx.x / t.chunks;
    const int c0 = chunk * DT_CC;
    const int q = tid & 31, pt = tid >> 5;
    const int cq = c0 + 4 * q;
    const bool c_ok = cq < d.C;
    // geometry of the tensor read through LDS (SH x SW) and of the one written / tiled (DH x DW)
    const int SH = MODE == 1 ? d.OH : d.H, SW = MODE == 1 ? d.OW : d.W;
    const int DH = MODE == 1 ? d.H : d.OH, DW_ = MODE == 1 ? d.W : d.OW;
    const float in_inv = 1.f / half_scale_for(*hf.in_bound);
    float aux_inv = 1.f, out_scale = 1.f, acc_inv = 1.f;
    if (MODE == 2) aux_inv = 1.f / half_scale_for(*hf.aux_bound);
    if constexpr (HOUT && MODE != 2) {
        float b = 9.f * __uint_as_float(*hf.w_amax) * __uint_as_float(*(hf.in_true != nullptr ? hf.in_true : hf.in_bound));
        if (MODE == 1 && accumulate) { b += __uint_as_float(*hf.acc_bound); acc_inv = 1.f / half_scale_for(*hf.acc_bound); }
        out_scale = half_scale_for(__float_as_uint(b));
        if (blockIdx.x == 0 && threadIdx.x == 0) *hf.out_bound = __float_as_uint(b);
    }
    f32x4 k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE != 2 && c_ok) {
        f32x4 kt[9];
        load_taps(static_cast<const float*>(aux), cq >> 2, kt);
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i] = (MODE == 0 || BWD2) ? kt[i] : kt[8 - i];
    }
    const long long src_bytes = (long long)d.B * SH * SW * d.C * 2;
    const long long dy_bytes = (long long)d.B * d.OH * d.OW * d.C * 2;
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(in), 0, (int)src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(MODE == 2 ? aux : in), 0, (int)(MODE == 2 ? dy_bytes : src_bytes), 0x00020000);
    const int piece_c = c0 + 8 * (lane & 15);
    const bool piece_ok = piece_c < d.C;
    float st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st1[i] = st2[i] = 0.f;
    const int oct = q >> 1, odd = q & 1;
    const int cp = c0 + 8 * oct;
    const bool cp_ok = cp < d.C;

    for (int tile = group; tile < t.n_tiles; tile += t.groups) {
        const int tx = tile % t.tiles_w, t1 = tile / t.tiles_w;
        const int ty = t1 % t.tiles_h, b = t1 / t.tiles_h;
        const int h0 = ty * TH, w0 = tx * TW;                 // origin of the written tile
        // origin of the patch in the tensor read through LDS
        const int ph0 = BWD2 ? h0 / 2 : h0 * S - D, pw0 = BWD2 ? w0 / 2 : w0 * S - D;
        for (int i = wave; i < (NPX + 3) / 4; i += 4) {
            const int pi = 4 * i + (lane >> 4);
            const int iy = pi / PW, ix = pi - iy * PW;
            const int h = ph0 + iy, w = pw0 + ix;
            const bool ok = piece_ok & (pi < NPX) & ((unsigned)h < (unsigned)SH) & ((unsigned)w < (unsigned)SW);
            const unsigned off = ok ? ((unsigned)((b * SH + h) * SW + w) * (unsigned)d.C + (unsigned)piece_c) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (dw_lds_vptr)(lin + i * 1024), 16, off, 0, 0, 0);
        }
        if (MODE == 2) {
            for (int i = wave; i < TH * TW / 4; i += 4) {
                const int pi = 4 * i + (lane >> 4);
                const int h = h0 + (pi >> 4), w = w0 + (pi & 15);
                const bool ok = piece_ok & (h < d.OH) & (w < d.OW);
                const unsigned off = ok ? ((unsigned)((b * d.OH + h) * d.OW + w) * (unsigned)d.C + (unsigned)piece_c) * 2u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_dy, (dw_lds_vptr)(ldy + i * 1024), 16, off, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE != 1 && hf.in_scale != nullptr)
            dw_bn_patch(lin, NPX, PW, ph0, pw0, SH, SW, c0, d.C, hf, 1.f / half_scale_for(*hf.y_bound), half_scale_for(*hf.in_bound));

        const char* const pin = lin + q * 8;
        auto px = [&](int iy, int ix) { return half4_to_f32(*reinterpret_cast<const uint2*>(pin + (iy * PW + ix) * DT_PIXB)); };
        const int colx = w0 + 2 * pt + odd;
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            if (MODE == 2) {
                const f32x4 g0 = half4_to_f32(*reinterpret_cast<const uint2*>(ldy + (r * TW + 2 * pt) * DT_PIXB + q * 8));
                const f32x4 g1 = half4_to_f32(*reinterpret_cast<const uint2*>(ldy + (r * TW + 2 * pt + 1) * DT_PIXB + q * 8));
#pragma unroll
                for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        k[kr * 3 + ks] = dw_fma4(g0, px(r * S + kr * D, (2 * pt) * S + ks * D), k[kr * 3 + ks]);
                        k[kr * 3 + ks] = dw_fma4(g1, px(r * S + kr * D, (2 * pt + 1) * S + ks * D), k[kr * 3 + ks]);
                    }
            } else {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (BWD2) {
                    // dx[h, w] = sum over (kr, ks) with h + 1 - kr and w + 1 - ks even of dy[(h + 1 - kr) / 2, (w + 1 - ks) / 2] k[kr][ks];
                    // h0, w0 even: row parity = r & 1, column 2 pt even, 2 pt + 1 odd
#pragma unroll
                    for (int kr = (r & 1) ? 0 : 1; kr < 3; kr += 2) {
                        const int prow = (r + 1 - kr) / 2;
                        const f32x4 e0 = px(prow, pt), e1 = px(prow, pt + 1);
                        a0 = dw_fma4(k[kr * 3 + 1], e0, a0);
                        a1 = dw_fma4(k[kr * 3 + 0], e1, a1);
                        a1 = dw_fma4(k[kr * 3 + 2], e0, a1);
                    }
                } else {
#pragma unroll
                    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                        for (int ks = 0; ks < 3; ++ks) {
                            a0 = dw_fma4(k[kr * 3 + ks], px(r * S + kr * D, (2 * pt) * S + ks * D), a0);
                            a1 = dw_fma4(k[kr * 3 + ks], px(r * S + kr * D, (2 * pt + 1) * S + ks * D), a1);
                        }
                }
                const int oh = h0 + r;
                const bool ok = cp_ok & (oh < DH) & (colx < DW_);
                const size_t e = ((size_t)(b * DH + oh) * DW_ + colx) * d.C + cp;
                dw_emit_pair<MODE, STATS, HOUT>(a0, a1, odd, ok, e, out, accumulate, in_inv, out_scale, acc_inv, hf, st1, st2);
            }
        }
        __syncthreads();
    }

    if (STATS) dw_reduce_stats(reinterpret_cast<float*>(dt_lds), st1, st2, tid, c0, d.C, group, stats);
    if (MODE == 2) dw_reduce_wgrad(reinterpret_cast<f32x4*>(dt_lds), k, tid, q, pt, c_ok, cq, d.C, group, in_inv * aux_inv, static_cast<float*>(out));
}

static int g_dw_tiles = 3;          // A/B knob, bit 0: the stride-1 tiled kernels (else strips), bit 1: the stride-2 / dilation-2 ones (else fp32 kernels)
extern "C" int pylc_debug_dw_tiles(int on) { g_dw_tiles = on; return PYLC_OK; }

static bool dw_tile_dense(const PylcDwDesc* d) {
    return d->C % 8 == 0 && d->x_pitch == d->C && d->y_pitch == d->C && (long long)d->B * d->H * d->W * d->C * 2 < (1ll << 31);
}
static bool dw_tile_ok(const PylcDwDesc* d) { return (g_dw_tiles & 1) && dw_tile_dense(d) && d->stride == 1 && d->dil == 1; }
// the other two geometries of the Aligned Xception: stride 2 (even input sizes: the dx tiles' parity logic) and dilation 2 at stride 1
static bool dw_tileg_ok(const PylcDwDesc* d) {
    return (g_dw_tiles & 2) && dw_tile_dense(d) && ((d->stride == 2 && d->dil == 1 && d->H % 2 == 0 && d->W % 2 == 0) || (d->stride == 1 && d->dil == 2));
}
static DwTiles make_tiles_hw(const PylcDwDesc* d, int H, int W, int th, int tw) {
    DwTiles t;
    t.tiles_w = cdiv(W, tw);
    t.tiles_h = cdiv(H, th);
    t.n_tiles = d->B * t.tiles_h * t.tiles_w;
    t.chunks = cdiv(d->C, DT_CC);
    t.groups = t.n_tiles < kDefaultSlabs ? t.n_tiles : kDefaultSlabs;
    return t;
}
static DwTiles make_tiles(const PylcDwDesc* d) { return make_tiles_hw(d, d->H, d->W, DT_TH, DT_TW); }
// tiles of the generic kernels: over the output for the forward / filter gradient, over the input for the data gradient
static DwTiles make_tiles_g(const PylcDwDesc* d, int mode) {
    if (mode == 1) return make_tiles_hw(d, d->H, d->W, 8, 16);
    return make_tiles_hw(d, d->OH, d->OW, d->stride == 2 ? 4 : 8, 16);
}
template <typename K>
static hipError_t dw_opt_in(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
static int dw_tile_init() {          // the filter-gradient form keeps two tiles in LDS: above the 64 KB a kernel gets without asking
    static bool done = false;
    if (!done) {
        PYLC_HIP(dw_opt_in((dw_tile_kernel<2, false, false>), dw_tile_lds_bytes<2>()));
        done = true;
    }
    return PYLC_OK;
}

static bool dw_fast(const PylcDwDesc* d) { return d->stride == 1 && d->dil == 1 && d->W >= 2; }

static DwStrip make_strips(const PylcDwDesc* d, int cols, int RL) {
    DwStrip s;
    s.pairs = (d->H + 1) / 2;
    int seg = 32;
    const int waves = (cols * RL + 63) / 64;      // waves per block that carry work
    for (;;) {
        s.nseg = cdiv(d->W, seg);
        s.seg = cdiv(d->W, s.nseg);
        s.n_strips = d->B * s.pairs * s.nseg;
        // enough strips to give every SIMD of the chip a few waves; shorter strips pay 2 extra column loads each
        if (seg <= 8 || (long long)cdiv(s.n_strips, RL) * waves >= 256 * 12) break;
        seg /= 2;
    }
    s.strips_per_block = RL * cdiv(s.n_strips, RL * kDefaultSlabs);
    return s;
}

// dw[c][t] = sum_slab partial[slab][t][c], fp64, fixed order: 8 columns x 32 slab lanes per block (a single thread walking
// all slabs of its column is latency-bound: 250 us for 27 MB).
__global__ __launch_bounds__(256) void dw_wgrad_combine_kernel(const float* __restrict__ partial, int nslab, int C, float* __restrict__ dw) {
    __shared__ double red[32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + tx;                        // over 9*C, i = t*C + c
    double acc = 0.0;
    if (i < 9 * C)
        for (int s = ty; s < nslab; s += 32) acc += (double)partial[(size_t)s * 9 * C + i];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && i < 9 * C) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) sum += red[k][tx];
        const int t = i / C, c = i % C;
        dw[c * 9 + t] = (float)sum;
    }
}

static int check_dw(const PylcDwDesc* d) {
    PYLC_REQUIRE(d != nullptr, "null depthwise descriptor");
    PYLC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % 4 == 0, "dwconv: bad dims");
    PYLC_REQUIRE((d->stride == 1 || d->stride == 2) && d->dil >= 1, "dwconv: stride 1|2, dil >= 1");
    PYLC_REQUIRE(d->OH == (d->H - 1) / d->stride + 1 && d->OW == (d->W - 1) / d->stride + 1, "dwconv: OH/OW must be (H-1)/stride+1");
    PYLC_REQUIRE(d->x_pitch >= d->C && d->y_pitch >= d->C && d->x_pitch % 4 == 0 && d->y_pitch % 4 == 0, "dwconv: bad pitch");
    PYLC_REQUIRE((long long)d->B * d->H * d->W < (1ll << 31), "dwconv: more than 2^31 pixels");
    return PYLC_OK;
}
static DwGeom geom(const PylcDwDesc* d) { return DwGeom{d->B, d->H, d->W, d->C, d->stride, d->dil, d->OH, d->OW, d->x_pitch, d->y_pitch}; }

}  // namespace pylc

using namespace pylc;

extern "C" int pylc_dwconv3x3_fwd(const PylcDwDesc* d, const float* x, const float* w, float* y, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(x && w && y, "dwconv_fwd: null pointer");
    const Slab g = make_slab((long long)d->B * d->OH * d->OW, d->C);
    if (dw_fast(d)) {
        const DwStrip s = make_strips(d, g.cols, g.RL);
        hipLaunchKernelGGL((dw_strip_kernel<0>), dim3(cdiv(s.n_strips, s.strips_per_block)), dim3(256), 0, as_stream(stream), x, w, y, geom(d), s,
                           g.cols, g.RL, g.CV, 0);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    hipLaunchKernelGGL(dw_fwd_kernel, dim3(g.nslab), dim3(256), 0, as_stream(stream), x, w, y, geom(d), g);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_dwconv3x3_fwd_stats_rows(const PylcDwDesc* d) {
    if (check_dw(d) || !dw_fast(d)) return 0;
    const Slab g = make_slab((long long)d->B * d->OH * d->OW, d->C);
    const DwStrip s = make_strips(d, g.cols, g.RL);
    return cdiv(s.n_strips, s.strips_per_block);
}

extern "C" int pylc_dwconv3x3_fwd_stats(const PylcDwDesc* d, const float* x, const float* w, float* y, float* stats_partial, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(x && w && y && stats_partial && dw_fast(d), "dwconv_fwd_stats: null pointer / not a stride-1 dilation-1 shape (pylc_dwconv3x3_fwd_stats_rows == 0)");
    const Slab g = make_slab((long long)d->B * d->OH * d->OW, d->C);
    const DwStrip s = make_strips(d, g.cols, g.RL);
    hipLaunchKernelGGL((dw_strip_kernel<0, true>), dim3(cdiv(s.n_strips, s.strips_per_block)), dim3(256), 0, as_stream(stream), x, w, y, geom(d), s,
                       g.cols, g.RL, g.CV, 0, stats_partial);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_dwconv3x3_dgrad_acc(const PylcDwDesc* d, const float* dy, const float* w, float* dx, int accumulate, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(dy && w && dx, "dwconv_dgrad: null pointer");
    const Slab g = make_slab((long long)d->B * d->H * d->W, d->C);
    if (dw_fast(d)) {
        const DwStrip s = make_strips(d, g.cols, g.RL);
        hipLaunchKernelGGL((dw_strip_kernel<1>), dim3(cdiv(s.n_strips, s.strips_per_block)), dim3(256), 0, as_stream(stream), dy, w, dx, geom(d), s,
                           g.cols, g.RL, g.CV, accumulate);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    hipLaunchKernelGGL(dw_dgrad_kernel, dim3(g.nslab), dim3(256), 0, as_stream(stream), dy, w, dx, geom(d), g, accumulate);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_dwconv3x3_dgrad(const PylcDwDesc* d, const float* dy, const float* w, float* dx, void* stream) {
    return pylc_dwconv3x3_dgrad_acc(d, dy, w, dx, 0, stream);
}

extern "C" size_t pylc_dwconv3x3_wgrad_workspace(const PylcDwDesc* d) {
    if (check_dw(d)) return 0;
    return (size_t)kDefaultSlabs * 9 * (size_t)d->C * sizeof(float);
}

extern "C" int pylc_dwconv3x3_wgrad(const PylcDwDesc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(x && dy && dw && workspace, "dwconv_wgrad: null pointer");
    const Slab g = make_slab((long long)d->B * d->OH * d->OW, d->C);
    if ((size_t)g.nslab * 9 * d->C * sizeof(float) > workspace_bytes)
        return fail(PYLC_ERR_WORKSPACE, "dwconv_wgrad workspace too small");
    hipStream_t st = as_stream(stream);
    int nslab = g.nslab;
    if (dw_fast(d)) {
        const DwStrip s = make_strips(d, g.cols, g.RL);
        nslab = cdiv(s.n_strips, s.strips_per_block);          // <= kDefaultSlabs by construction
        hipLaunchKernelGGL((dw_strip_kernel<2>), dim3(nslab), dim3(256), 0, st, x, dy, static_cast<float*>(workspace), geom(d), s, g.cols, g.RL,
                           g.CV, 0);
    } else {
        hipLaunchKernelGGL(dw_wgrad_kernel, dim3(g.nslab), dim3(256), 0, st, x, dy, static_cast<float*>(workspace), geom(d), g);
    }
    PYLC_LAUNCH_CHECK();
    hipLaunchKernelGGL(dw_wgrad_combine_kernel, dim3(cdiv(9 * d->C, 8)), dim3(256), 0, st, static_cast<const float*>(workspace), nslab, d->C, dw);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

// ---- one-plane fp16 operands (precision mode 3) ------------------------------------------------------------------------------------
extern "C" int pylc_dwconv3x3_half_ok(const PylcDwDesc* d) {
    return (check_dw(d) == PYLC_OK && (dw_fast(d) || dw_tileg_ok(d)) && d->x_pitch == d->C && d->y_pitch == d->C && d->C % 4 == 0) ? 1 : 0;
}

// launches of the generic tiled kernels (stride 2 / dilation 2), LDS opt-in on first use
template <int MODE, bool STATS, bool HOUT>
static int launch_tileg(const PylcDwDesc* d, const void* in, const void* aux, void* out, int accumulate, float* stats, const DwHalf& hf, hipStream_t st) {
    const DwTiles t = make_tiles_g(d, MODE);
    const dim3 grid(t.groups * t.chunks);
    if (d->stride == 2) {
        static bool opted = false;
        if (!opted) { PYLC_HIP(dw_opt_in((dw_tileg_kernel<MODE, 2, 1, STATS, HOUT>), dw_tileg_lds_bytes<MODE, 2, 1>())); opted = true; }
        constexpr int lds = dw_tileg_lds_bytes<MODE, 2, 1>();
        hipLaunchKernelGGL((dw_tileg_kernel<MODE, 2, 1, STATS, HOUT>), grid, dim3(256), lds, st, in, aux, out, geom(d), t, accumulate, stats, hf);
    } else {
        static bool opted = false;
        if (!opted) { PYLC_HIP(dw_opt_in((dw_tileg_kernel<MODE, 1, 2, STATS, HOUT>), dw_tileg_lds_bytes<MODE, 1, 2>())); opted = true; }
        constexpr int lds = dw_tileg_lds_bytes<MODE, 1, 2>();
        hipLaunchKernelGGL((dw_tileg_kernel<MODE, 1, 2, STATS, HOUT>), grid, dim3(256), lds, st, in, aux, out, geom(d), t, accumulate, stats, hf);
    }
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_dwconv3x3_fwd_h_stats_rows(const PylcDwDesc* d) {
    if (!pylc_dwconv3x3_half_ok(d)) return 0;
    if (dw_tile_ok(d)) return make_tiles(d).groups;
    if (dw_tileg_ok(d)) return make_tiles_g(d, 0).groups;
    return pylc_dwconv3x3_fwd_stats_rows(d);
}

static int dw_fwd_h_impl(const PylcDwDesc* d, const void* x_h, const float* w, void* y_h, float* stats_partial, const DwHalf& hf, void* stream);

extern "C" int pylc_dwconv3x3_fwd_h(const PylcDwDesc* d, const void* x_h, const unsigned int* x_bound, const float* w, const unsigned int* w_amax,
                                    void* y_h, unsigned int* y_bound_out, float* stats_partial, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(x_h && x_bound && w && w_amax && y_h && y_bound_out && pylc_dwconv3x3_half_ok(d),
                 "dwconv_fwd_h: null pointer, or not a dense stride-1 / dilation-1 shape (pylc_dwconv3x3_half_ok)");
    const DwHalf hf{x_bound, nullptr, w_amax, nullptr, y_bound_out};
    return dw_fwd_h_impl(d, x_h, w, y_h, stats_partial, hf, stream);
}

// pylc_dwconv3x3_fwd_h for inference on plane tensors: x_true_amax = the true max|x| (x_bound is what x was SCALED with); the output is
// scaled with -- and *y_bound_out receives -- 9 max|w| x_true_amax.  No statistics.
extern "C" int pylc_dwconv3x3_fwd_h_eval(const PylcDwDesc* d, const void* x_h, const unsigned int* x_bound, const unsigned int* x_true_amax, const float* w,
                                         const unsigned int* w_amax, void* y_h, unsigned int* y_bound_out, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(x_h && x_bound && x_true_amax && w && w_amax && y_h && y_bound_out && pylc_dwconv3x3_half_ok(d),
                 "dwconv_fwd_h_eval: null pointer, or not a shape of the half kernels (pylc_dwconv3x3_half_ok)");
    DwHalf hf{x_bound, nullptr, w_amax, nullptr, y_bound_out};
    hf.in_true = x_true_amax;
    return dw_fwd_h_impl(d, x_h, w, y_h, nullptr, hf, stream);
}

// pylc_dwconv3x3_fwd_h on the INPUT of the BatchNorm (+ ReLU) that produces x: see DwHalf::in_scale.  Tiled kernels only.
extern "C" int pylc_dwconv3x3_bn_ok(const PylcDwDesc* d) { return (check_dw(d) == PYLC_OK && (dw_tile_ok(d) || dw_tileg_ok(d))) ? 1 : 0; }

extern "C" int pylc_dwconv3x3_fwd_h_bn(const PylcDwDesc* d, const void* y_in_h, const unsigned int* y_in_bound, const float* bn_scale, const float* bn_shift,
                                       int relu, const unsigned int* x_bound, const float* w, const unsigned int* w_amax, void* y_h,
                                       unsigned int* y_bound_out, float* stats_partial, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(y_in_h && y_in_bound && bn_scale && bn_shift && x_bound && w && w_amax && y_h && y_bound_out && pylc_dwconv3x3_bn_ok(d),
                 "dwconv_fwd_h_bn: null pointer, or not a shape of the tiled kernels (pylc_dwconv3x3_bn_ok)");
    DwHalf hf{x_bound, nullptr, w_amax, nullptr, y_bound_out};
    hf.in_scale = bn_scale; hf.in_shift = bn_shift; hf.y_bound = y_in_bound; hf.in_relu = relu;
    return dw_fwd_h_impl(d, y_in_h, w, y_h, stats_partial, hf, stream);
}

static int dw_fwd_h_impl(const PylcDwDesc* d, const void* x_h, const float* w, void* y_h, float* stats_partial, const DwHalf& hf, void* stream) {
    if (dw_tile_ok(d)) {
        const DwTiles t = make_tiles(d);
        const dim3 tgrid(t.groups * t.chunks);
        if (stats_partial != nullptr)
            hipLaunchKernelGGL((dw_tile_kernel<0, true, true>), tgrid, dim3(256), dw_tile_lds_bytes<0>(), as_stream(stream), x_h, w, y_h, geom(d), t, 0, stats_partial, hf);
        else
            hipLaunchKernelGGL((dw_tile_kernel<0, false, true>), tgrid, dim3(256), dw_tile_lds_bytes<0>(), as_stream(stream), x_h, w, y_h, geom(d), t, 0, nullptr, hf);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    if (dw_tileg_ok(d))
        return stats_partial != nullptr ? launch_tileg<0, true, true>(d, x_h, w, y_h, 0, stats_partial, hf, as_stream(stream))
                                        : launch_tileg<0, false, true>(d, x_h, w, y_h, 0, nullptr, hf, as_stream(stream));
    const Slab g = make_slab((long long)d->B * d->OH * d->OW, d->C);
    const DwStrip s = make_strips(d, g.cols, g.RL);
    const dim3 grid(cdiv(s.n_strips, s.strips_per_block));
    if (stats_partial != nullptr)
        hipLaunchKernelGGL((dw_strip_kernel<0, true, true, true>), grid, dim3(256), 0, as_stream(stream), x_h, w, y_h, geom(d), s, g.cols, g.RL, g.CV, 0, stats_partial, hf);
    else
        hipLaunchKernelGGL((dw_strip_kernel<0, false, true, true>), grid, dim3(256), 0, as_stream(stream), x_h, w, y_h, geom(d), s, g.cols, g.RL, g.CV, 0, nullptr, hf);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_dwconv3x3_dgrad_h(const PylcDwDesc* d, const void* dy_h, const unsigned int* dy_bound, const float* w, const unsigned int* w_amax,
                                      void* dx_h, unsigned int* dx_bound_out, int accumulate, const unsigned int* acc_bound, void* stream) {
    if (int rc = check_dw(d)) return rc;
    const bool out_f32 = dx_bound_out == nullptr;          // dx (and what it accumulates into) stays fp32: the block-input gradient
    PYLC_REQUIRE(dy_h && dy_bound && w && w_amax && dx_h && (out_f32 || !accumulate || acc_bound) && pylc_dwconv3x3_half_ok(d),
                 "dwconv_dgrad_h: null pointer, or not a dense stride-1 / dilation-1 shape (pylc_dwconv3x3_half_ok)");
    PYLC_REQUIRE(out_f32 || !accumulate || acc_bound != dx_bound_out, "dwconv_dgrad_h: the new bound needs its own scalar (the old one is read by every block)");
    const DwHalf hf{dy_bound, nullptr, w_amax, acc_bound, dx_bound_out};
    if (dw_tile_ok(d)) {
        const DwTiles t = make_tiles(d);
        const dim3 tgrid(t.groups * t.chunks);
        if (out_f32)
            hipLaunchKernelGGL((dw_tile_kernel<1, false, false>), tgrid, dim3(256), dw_tile_lds_bytes<1>(), as_stream(stream), dy_h, w, dx_h, geom(d), t, accumulate, nullptr, hf);
        else
            hipLaunchKernelGGL((dw_tile_kernel<1, false, true>), tgrid, dim3(256), dw_tile_lds_bytes<1>(), as_stream(stream), dy_h, w, dx_h, geom(d), t, accumulate, nullptr, hf);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    if (dw_tileg_ok(d))
        return out_f32 ? launch_tileg<1, false, false>(d, dy_h, w, dx_h, accumulate, nullptr, hf, as_stream(stream))
                       : launch_tileg<1, false, true>(d, dy_h, w, dx_h, accumulate, nullptr, hf, as_stream(stream));
    const Slab g = make_slab((long long)d->B * d->H * d->W, d->C);
    const DwStrip s = make_strips(d, g.cols, g.RL);
    const dim3 grid(cdiv(s.n_strips, s.strips_per_block));
    if (out_f32)
        hipLaunchKernelGGL((dw_strip_kernel<1, false, true, false>), grid, dim3(256), 0, as_stream(stream), dy_h, w, dx_h, geom(d), s, g.cols, g.RL, g.CV, accumulate,
                           nullptr, hf);
    else
        hipLaunchKernelGGL((dw_strip_kernel<1, false, true, true>), grid, dim3(256), 0, as_stream(stream), dy_h, w, dx_h, geom(d), s, g.cols, g.RL, g.CV, accumulate,
                           nullptr, hf);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_dwconv3x3_dgrad_h_add_ok(const PylcDwDesc* d) { return (check_dw(d) == PYLC_OK && dw_tile_ok(d)) ? 1 : 0; }

extern "C" int pylc_dwconv3x3_dgrad_h_add(const PylcDwDesc* d, const void* dy_h, const unsigned int* dy_bound, const float* w, const unsigned int* w_amax,
                                          float* dx, const float* add_src, const unsigned char* add_mask, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(dy_h && dy_bound && w && w_amax && dx && add_src && add_mask && dw_tile_ok(d),
                 "dwconv_dgrad_h_add: null pointer, or not a shape of the tiled stride-1 kernel (pylc_dwconv3x3_dgrad_h_add_ok)");
    DwHalf hf{dy_bound, nullptr, w_amax, nullptr, nullptr};
    hf.add_src = add_src;
    hf.add_mask = add_mask;
    const DwTiles t = make_tiles(d);
    hipLaunchKernelGGL((dw_tile_kernel<1, false, false>), dim3(t.groups * t.chunks), dim3(256), dw_tile_lds_bytes<1>(), as_stream(stream), dy_h, w, dx, geom(d), t, 0, nullptr, hf);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

static int dw_wgrad_h_impl(const PylcDwDesc* d, const void* x_h, const void* dy_h, float* dw, void* workspace, size_t workspace_bytes, const DwHalf& hf,
                           void* stream);

extern "C" int pylc_dwconv3x3_wgrad_h(const PylcDwDesc* d, const void* x_h, const unsigned int* x_bound, const void* dy_h, const unsigned int* dy_bound,
                                      float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(x_h && x_bound && dy_h && dy_bound && dw && workspace && pylc_dwconv3x3_half_ok(d),
                 "dwconv_wgrad_h: null pointer, or not a dense stride-1 / dilation-1 shape (pylc_dwconv3x3_half_ok)");
    const DwHalf hf{x_bound, dy_bound, nullptr, nullptr, nullptr};
    return dw_wgrad_h_impl(d, x_h, dy_h, dw, workspace, workspace_bytes, hf, stream);
}

extern "C" int pylc_dwconv3x3_wgrad_h_bn(const PylcDwDesc* d, const void* y_in_h, const unsigned int* y_in_bound, const float* bn_scale,
                                         const float* bn_shift, int relu, const unsigned int* x_bound, const void* dy_h, const unsigned int* dy_bound,
                                         float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_dw(d)) return rc;
    PYLC_REQUIRE(y_in_h && y_in_bound && bn_scale && bn_shift && x_bound && dy_h && dy_bound && dw && workspace && pylc_dwconv3x3_bn_ok(d),
                 "dwconv_wgrad_h_bn: null pointer, or not a shape of the tiled kernels (pylc_dwconv3x3_bn_ok)");
    DwHalf hf{x_bound, dy_bound, nullptr, nullptr, nullptr};
    hf.in_scale = bn_scale; hf.in_shift = bn_shift; hf.y_bound = y_in_bound; hf.in_relu = relu;
    return dw_wgrad_h_impl(d, y_in_h, dy_h, dw, workspace, workspace_bytes, hf, stream);
}

static int dw_wgrad_h_impl(const PylcDwDesc* d, const void* x_h, const void* dy_h, float* dw, void* workspace, size_t workspace_bytes, const DwHalf& hf,
                           void* stream) {
    hipStream_t st = as_stream(stream);
    if (dw_tile_ok(d)) {
        const DwTiles t = make_tiles(d);
        if ((size_t)t.groups * 9 * d->C * sizeof(float) > workspace_bytes) return fail(PYLC_ERR_WORKSPACE, "dwconv_wgrad_h workspace too small");
        if (int rc = dw_tile_init()) return rc;
        hipLaunchKernelGGL((dw_tile_kernel<2, false, false>), dim3(t.groups * t.chunks), dim3(256), dw_tile_lds_bytes<2>(), st, x_h, dy_h, workspace, geom(d), t, 0,
                           nullptr, hf);
        PYLC_LAUNCH_CHECK();
        hipLaunchKernelGGL(dw_wgrad_combine_kernel, dim3(cdiv(9 * d->C, 8)), dim3(256), 0, st, static_cast<const float*>(workspace), t.groups, d->C, dw);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    if (dw_tileg_ok(d)) {
        const DwTiles t = make_tiles_g(d, 2);
        if ((size_t)t.groups * 9 * d->C * sizeof(float) > workspace_bytes) return fail(PYLC_ERR_WORKSPACE, "dwconv_wgrad_h workspace too small");
        if (int rc = launch_tileg<2, false, false>(d, x_h, dy_h, workspace, 0, nullptr, hf, st)) return rc;
        hipLaunchKernelGGL(dw_wgrad_combine_kernel, dim3(cdiv(9 * d->C, 8)), dim3(256), 0, st, static_cast<const float*>(workspace), t.groups, d->C, dw);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    const Slab g = make_slab((long long)d->B * d->OH * d->OW, d->C);
    const DwStrip s = make_strips(d, g.cols, g.RL);
    const int nslab = cdiv(s.n_strips, s.strips_per_block);
    if ((size_t)nslab * 9 * d->C * sizeof(float) > workspace_bytes) return fail(PYLC_ERR_WORKSPACE, "dwconv_wgrad_h workspace too small");
    hipLaunchKernelGGL((dw_strip_kernel<2, false, true, false>), dim3(nslab), dim3(256), 0, st, x_h, dy_h, workspace, geom(d), s, g.cols, g.RL, g.CV, 0, nullptr, hf);
    PYLC_LAUNCH_CHECK();
    hipLaunchKernelGGL(dw_wgrad_combine_kernel, dim3(cdiv(9 * d->C, 8)), dim3(256), 0, st, static_cast<const float*>(workspace), nslab, d->C, dw);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
