// BatchNorm2d forward / backward with fused ReLU and residual add, NHWC fp32 (HBM-bound kernels).
//
// Every kernel walks a [M rows][C channels] matrix in row slabs.  A thread owns ONE float4 channel vector
// (so gamma/beta/scale/shift live in registers, no per-element division) and strides over rows; consecutive
// threads read consecutive 16-byte vectors of a row, so every wave instruction touches whole contiguous rows.
// Reductions are two-level: per-slab fp32 partials -> one fp64 combine per channel (deterministic order).
//
// Replaces torch.nn.BatchNorm2d (models/model.py:71-76) + ReLU + residual add at resnet.py:36-51,
// aspp.py:28-31,81-84, decoder.py:42-44 and last_conv, unet.py:113-118, xception.py:37,60-97; the
// synchronized variant's wire format follows models/sync_batchnorm/batchnorm.py:48-125.
#include "conv_common.h"
#include "slab.h"

namespace pylc {

// ---- fp16-plane tensors (include/pylc_hip.h "fp16 planes"): a thread's float4 channel vector <-> 4 halves per plane ----------
struct PlanesRef {            // where a planes tensor lives and how it was scaled
    _Float16* base;           // plane 0; nullptr = the tensor is fp32
    long long plane_stride;   // halves between the planes
    int nplanes;              // 2, or 1 (precision mode 3: plane 0 only)
    float scale;              // power of two the values were multiplied with (pow2_scale_for(bound))
};

// Store a float4 channel vector as planes.  Two planes: lanes 2k / 2k+1 own ADJACENT channel quads of one row (row-slab kernels: an even
// number of vector columns per row, pairs never straddle a row), so they swap halves through DPP and each issues ONE 16-byte store
// -- the even lane 8 halves of plane 0, the odd lane 8 halves of plane 1 -- instead of two 8-byte stores each (measured: the 8-byte
// form made bn_bwd_apply 15 % slower than its fp32 version).  Both lanes of a pair must call this together (same row validity).
__device__ __forceinline__ void planes_store4(const PlanesRef& p, long long elem, f32x4 v) {
    uint2 p0, p1;
    split2(v, p.scale, p0, p1);
    if (p.nplanes == 2) {
        const bool odd = threadIdx.x & 1;
        const uint2 send = odd ? p0 : p1;              // what the partner stores
        uint2 recv;
        recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]: lane ^ 1
        recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, true);
        const uint4 val = odd ? make_uint4(recv.x, recv.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, recv.x, recv.y);
        // (chunk-interleaved tensor, plane_stride == 32: the pair's 8 channels sit in one 32-channel chunk, plane 1 is 32 halves on)
        const bool il = planes_il(p.plane_stride);
        _Float16* dst = odd ? p.base + p.plane_stride + planes_phys(elem - 4, il) : p.base + planes_phys(elem, il);
        *reinterpret_cast<uint4*>(dst) = val;
    } else {
        *reinterpret_cast<uint2*>(p.base + elem) = p0;
    }
}

// raw pieces of a float4 channel vector (loaded in a walk's load pass, decoded in its math pass)
struct PlanesRaw { uint2 h0, h1; };
__device__ __forceinline__ PlanesRaw planes_raw4(const PlanesRef& p, long long elem) {
    PlanesRaw r;
    const long long e = planes_phys(elem, p.nplanes == 2 && planes_il(p.plane_stride));
    r.h0 = *reinterpret_cast<const uint2*>(p.base + e);
    r.h1 = p.nplanes == 2 ? *reinterpret_cast<const uint2*>(p.base + p.plane_stride + e) : make_uint2(0u, 0u);
    return r;
}
__device__ __forceinline__ f32x4 halves4(uint2 u) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 h = __builtin_bit_cast(f16x4, u);
    const f32x4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return v;
}
// value: (h0 + 2^-11 h1) / s
__device__ __forceinline__ f32x4 planes_value4(const PlanesRaw& r, float inv_scale) {
    return (halves4(r.h0) + halves4(r.h1) * (1.f / 2048.f)) * inv_scale;
}
// sign carrier for the ReLU mask: positive iff the element is (one of its pieces is; both are >= 0 for a ReLU output)
__device__ __forceinline__ f32x4 planes_sign4(const PlanesRaw& r) {
    const f32x4 a = halves4(r.h0), b = halves4(r.h1);
    const f32x4 v = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
    return v;
}

// optional extras of the BatchNorm kernels (the *_ex entry points): fp16-plane operands and the dropout fused behind the activation
struct BnEx {
    _Float16* out_pl; long long out_ps; const unsigned* out_bound;        // bn_apply's output / the `out` the backward reads its mask from
    const _Float16* res_pl; long long res_ps; const unsigned* res_amax;   // bn_apply's residual input
    _Float16* dy_pl; long long dy_ps; const unsigned* dy_bound;           // bn_bwd_apply's dy
    int nplanes;
    unsigned drop_thresh; float keep_scale; unsigned long long seed;       // drop_thresh == 0: no dropout
    unsigned* g_amax;                                                      // bn_bwd_reduce: max |g| (zero-initialised by the caller)
    unsigned char* mask;                                                   // 1-bit ReLU mask (relu_nibble): written by bn_apply, read by the backward
    // YH instantiations (precision mode 3): y and dout arrive as ONE-PLANE fp16 tensors (slab.h ldq) scaled with these bounds; the residual
    // gradient g_out leaves in the same format with dout's bound
    const unsigned* y_bound;
    const unsigned* dout_bound;
};

// ---- 1-bit ReLU masks ---------------------------------------------------------------------------------------------------------------
// A BatchNorm with a residual input cannot recompute its ReLU mask from y alone (out = relu(y*scale + shift + residual)), so its
// backward passes used to re-read `out` for the sign only: 4 B per element in bn_bwd_reduce AND in bn_bwd_apply (bn3 of every
// bottleneck, the widest tensors of the network).  The apply pass now leaves one BIT per element -- the nibble of a thread's four
// channels, two threads per byte, byte index (row * CV + cv) / 2 (C % 8 == 0: an even number of vector columns per row) -- and the
// backward reads 1/8 B per element instead.  Exact: the bit is (pre-activation > 0), torch's threshold_backward mask.
__device__ __forceinline__ unsigned relu_nibble(f32x4 v) {
    return (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
}
// lanes 2k / 2k+1 own adjacent vector columns of one row (as planes_store4): the even lane stores the pair's byte
__device__ __forceinline__ void mask_store(unsigned char* mask, long long vec_index, unsigned nib) {
    const unsigned partner = (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]: lane ^ 1
    if (!(threadIdx.x & 1)) mask[vec_index >> 1] = (unsigned char)(nib | (partner << 4));
}
__device__ __forceinline__ f32x4 mask_apply(f32x4 g, unsigned byte, int cv) {
    const unsigned nib = byte >> ((cv & 1) * 4);
    f32x4 r;
    r.x = (nib & 1u) ? g.x : 0.f; r.y = (nib & 2u) ? g.y : 0.f; r.z = (nib & 4u) ? g.z : 0.f; r.w = (nib & 8u) ? g.w : 0.f;
    return r;
}

// counter-based hash RNG: one 64-bit mix per float4 -> 4 x 16-bit uniform thresholds (dropout; the backward regenerates the mask)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct DropRef { unsigned thresh16; float keep_scale; unsigned long long seed; };     // thresh16 == 0: no dropout
__device__ __forceinline__ f32x4 drop4(const DropRef& d, long long r, int CV, int cv, f32x4 v) {
    const unsigned long long h = mix64(d.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(r * CV + cv + 1));
    v.x = ((h & 0xFFFF) >= d.thresh16) ? v.x * d.keep_scale : 0.f;
    v.y = (((h >> 16) & 0xFFFF) >= d.thresh16) ? v.y * d.keep_scale : 0.f;
    v.z = (((h >> 32) & 0xFFFF) >= d.thresh16) ? v.z * d.keep_scale : 0.f;
    v.w = (((h >> 48) & 0xFFFF) >= d.thresh16) ? v.w * d.keep_scale : 0.f;
    return v;
}

__device__ __forceinline__ f32x4 relu_mask(f32x4 g, f32x4 o) {
    f32x4 r;
    r.x = o.x > 0.f ? g.x : 0.f; r.y = o.y > 0.f ? g.y : 0.f; r.z = o.z > 0.f ? g.z : 0.f; r.w = o.w > 0.f ? g.w : 0.f;
    return r;
}

// ---- reductions --------------------------------------------------------------------------------
// MODE 0: (sum y, sum y^2).  MODE 1: (sum g, sum g*xhat) with g = [dropout mask * keep scale *] dout * (out > 0).
// EX (MODE 1): `out` may be an fp16-plane tensor, dropout is regenerated from its seed, max|g| goes to ex.g_amax.
// MS: where a ReLU's mask comes from (MODE 1; the launch picks it, so that only that source's registers are live):
//   MS_Y    none, or recomputed from y (scale / shift given: out = max(y*scale + shift, 0), the forward's own expression -- no residual)
//   MS_OUT  the fp32 `out`        MS_PLANES  the fp16-plane `out`        MS_BITS  the 1-bit mask bn_apply left (ex.mask)
enum { MS_Y = 0, MS_OUT = 1, MS_PLANES = 2, MS_BITS = 3 };
template <int MODE, bool EX = false, bool DROP = false, int MS = MS_Y, bool YH = false, bool AH = false>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ a, int a_pitch,
                                                        const float* __restrict__ out, int out_pitch,
                                                        const float* __restrict__ y, int y_pitch,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        int relu, Slab g, int C, float* __restrict__ partial,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, BnEx ex) {
    // Accumulation: a thread sums each batch of kRowBatch rows in fp32 and folds the batch sums into fp64 accumulators; the block combines
    // in fp64 and rounds ONCE to the fp32 partial.  (Before: fp32 all the way to the partial -- hundreds of terms per thread, an error of
    // ~sqrt(rows) 2^-24 of sum |g xhat| on sums that cancel to a small fraction of that; torch's CPU BatchNorm, the parity yardstick,
    // accumulates these sums in double.  Measured on the fixtures: tests/test_nets_gpu.py::test_error_against_fp64_truth.)
    // [sum][component][thread]: consecutive lanes write consecutive doubles (the [thread][component] layout of rounds 1-5 put a wave's 64 lanes
    // 32 bytes apart: PMC SQ_LDS_BANK_CONFLICT 0.37 of this kernel's LDS cycles)
    __shared__ double red[2][4][256];
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    PlanesRef po{};
    DropRef dr{};
    if constexpr (EX) {
        if (ex.out_pl != nullptr) po = PlanesRef{ex.out_pl, ex.out_ps, ex.nplanes, 1.f};
        dr = DropRef{ex.drop_thresh, ex.keep_scale, ex.seed};
    }
    float gmax = 0.f;
    float a_inv = 1.f, y_inv = 1.f;
    if constexpr (AH) a_inv = 1.f / half_scale_for(*ex.dout_bound);
    if constexpr (YH) y_inv = 1.f / half_scale_for(*ex.y_bound);
    for (int cb = 0; cb < g.CV; cb += g.cols) {
        const int cv = cb + tx;
        const bool active = ty < g.RL && cv < g.CV;
        double d0[4] = {0.0, 0.0, 0.0, 0.0}, d1[4] = {0.0, 0.0, 0.0, 0.0};
        if (active) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};      // sums of the current batch of rows
            f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
            const bool remask = MODE == 1 && MS == MS_Y && relu;      // ReLU mask recomputed from y
            if (MODE == 1) { mu = ld4(mean + 4 * cv); is = ld4(invstd + 4 * cv); }
            if (remask) { sc = ld4(scale + 4 * cv); sh = ld4(shift + 4 * cv); }
            f32x4 va[kRowBatch], vy[kRowBatch], vo[MS == MS_OUT ? kRowBatch : 1];
            PlanesRaw vp[MS == MS_PLANES ? kRowBatch : 1];
            unsigned vm[MS == MS_BITS ? kRowBatch : 1];
            walk_rows(r_begin + ty, r_end, g.RL,
                [&](int u, long long r) {
                    va[u] = ldq<AH>(a, (size_t)(r * a_pitch + 4 * cv), a_inv);
                    if (MODE == 1) vy[u] = ldq<YH>(y, (size_t)(r * y_pitch + 4 * cv), y_inv);
                    if constexpr (MS == MS_OUT) vo[u] = ld4(out + r * out_pitch + 4 * cv);
                    if constexpr (MS == MS_PLANES) vp[u] = planes_raw4(po, r * out_pitch + 4 * cv);
                    if constexpr (MS == MS_BITS) vm[u] = ex.mask[(r * g.CV + cv) >> 1];
                },
                [&](int u, long long r, bool valid) {
                    if (MODE == 0) {
                        if (valid) { s0 += va[u]; s1 += va[u] * va[u]; }
                    } else {
                        f32x4 gg = va[u];
                        if constexpr (DROP) gg = drop4(dr, r, g.CV, cv, gg);
                        if (remask) gg = relu_mask(gg, vy[u] * sc + sh);   // the forward's own expression: identical bits
                        if constexpr (MS == MS_OUT) gg = relu_mask(gg, vo[u]);
                        if constexpr (MS == MS_PLANES) gg = relu_mask(gg, planes_sign4(vp[u]));
                        if constexpr (MS == MS_BITS) gg = mask_apply(gg, vm[u], cv);
                        const f32x4 xh = (vy[u] - mu) * is;
                        if (valid) {
                            s0 += gg; s1 += gg * xh;
                            if constexpr (EX) gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(gg.x), fabsf(gg.y))), fmaxf(fabsf(gg.z), fabsf(gg.w)));
                        }
                    }
                },
                [](int, long long) {},
                [&] {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { d0[k] += (double)s0[k]; d1[k] += (double)s1[k]; }
                    s0 = f32x4{0.f, 0.f, 0.f, 0.f};
                    s1 = s0;
                });
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[0][k][threadIdx.x] = d0[k]; red[1][k][threadIdx.x] = d1[k]; }
        __syncthreads();
        if (ty == 0 && cv < g.CV) {
            for (int j = 1; j < g.RL; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { d0[k] += red[0][k][j * g.cols + tx]; d1[k] += red[1][k][j * g.cols + tx]; }
            }
            const f32x4 f0 = {(float)d0[0], (float)d0[1], (float)d0[2], (float)d0[3]};
            const f32x4 f1 = {(float)d1[0], (float)d1[1], (float)d1[2], (float)d1[3]};
            float* p = partial + (size_t)blockIdx.x * 2 * C;
            // MODE 0: [sum | sumsq].  MODE 1: [sum g*xhat (dgamma) | sum g (dbeta)] -- the parameter order, so the
            // result can land directly in the adjacent (gamma, beta) slots of the flat gradient arena
            st4(p + 4 * cv, MODE == 0 ? f0 : f1);
            st4(p + C + 4 * cv, MODE == 0 ? f1 : f0);
        }
        __syncthreads();
    }
    if constexpr (EX) { if (ex.g_amax != nullptr) amax_commit(gmax, ex.g_amax); }
}

// Range bound of a BatchNorm output for the fp16-plane format, from what the statistics already give: by Samuelson's inequality
// |y - mean| <= sigma sqrt(n - 1), so |xhat| <= sqrt(n - 1) and |gamma xhat + beta| <= |gamma| sqrt(n - 1) + |beta| -- loose by
// sqrt(n) / (the true max |xhat|, a handful), i.e. 5-8 binades of the 29 the format has to spare, but free:
// no min / max pass, and known BEFORE the apply pass writes the planes.  `extra` = range of the residual input (float bits, may
// be NULL); `mul` = dropout's 1 / (1 - p).
__device__ __forceinline__ void bn_commit_bound(float gamma, float beta, double n, const unsigned* extra, float mul, unsigned* bound_out) {
    float b = fabsf(gamma) * (float)sqrt(n > 1.0 ? n - 1.0 : 1.0) + fabsf(beta);
    if (extra != nullptr) b += __uint_as_float(*extra);
    b *= mul;
    atomicMax(bound_out, __float_as_uint(b));
}

// Where the statistics' second moment cannot carry the variance: sum(x^2)/n - mean^2 computed from fp32 partial sums loses
// (mean/sigma)^2 * 2^-24 of relative accuracy, i.e. everything once |mean| ~ 10^3 sigma -- torch.nn.BatchNorm2d (two-pass on the CPU,
// Welford on the GPU; what the reference runs outside SyncBN) does not.  A channel with mean^2 > kRefineRatio * var is therefore
// re-measured here in a second pass over its column of y: sum(y - mean) and sum((y - mean)^2), differences exact in fp32 (Sterbenz),
// sums in fp64.  Costs nothing for well-conditioned channels (all of them in the networks of this package); the pass itself is a
// strided column read by the 32 row lanes of the finalize block.
constexpr double kRefineRatio = 64.0;
constexpr double kRefineFloor = 1e-5;    // the BatchNorm eps of every layer of this path: a variance far below it does not need its digits

struct RefineSrc {
    const float* y;       // NULL: no second pass
    int pitch;
    long long M;
    const float* shift;   // NULL, or per-channel K: the sums are of (y - K) and (y - K)^2 (a conv's statistics are taken before its bias)
};

// First half of a finalize (all 256 threads of the block: 8 channels (tx) x 32 row lanes (ty); the ty == 0 lanes arrive with the
// channel's fp32-rounded sum and sum of squares): the channel's LOCAL moments (S1, S2) = (sum y, sum y^2) in fp64 -- the sums as they
// are for a well-conditioned channel, rebuilt from the second pass for a refined one (sum y^2 = sum d^2 + 2 m sum d + n m^2 with
// d = y - m: no cancellation left in fp64).  Result valid in the ty == 0 lanes.
__device__ __forceinline__ void bn_local_moments(double sum, double sumsq, double (*red)[32][9], int tx, int ty, int c, double n, int C,
                                                 RefineSrc src, double& S1, double& S2) {
    __shared__ float ref_mu[8];
    __shared__ int ref_need[8];
    S1 = sum;
    S2 = sumsq;
    if (src.y == nullptr) return;            // kernel-uniform
    const double K = (src.shift != nullptr && c < C) ? (double)src.shift[c] : 0.0;
    if (ty == 0) {
        const double mu = sum / n;           // mean of (y - K)
        double var = sumsq / n - mu * mu;
        if (var < 0.0) var = 0.0;
        // the rounding of sum(x^2) is ~2^-22 mu^2: re-measure where that is more than 2^-16 of what the coefficient depends on
        ref_need[tx] = (c < C && mu * mu > kRefineRatio * (var + kRefineFloor)) ? 1 : 0;
        ref_mu[tx] = (float)(K + mu);
    }
    __syncthreads();
    const int need = ref_need[tx];
    if (!__syncthreads_or(need)) return;     // block-uniform
    double a0 = 0.0, a1 = 0.0;
    const float mf = ref_mu[tx];
    if (need) {
        const float* col = src.y + c;
        long long r = ty;
        for (; r + 7 * 32 < src.M; r += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(r + u * 32) * src.pitch];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float d = v[u] - mf; a0 += (double)d; a1 += (double)d * (double)d; }
        }
        for (; r < src.M; r += 32) { const float d = col[r * src.pitch] - mf; a0 += (double)d; a1 += (double)d * (double)d; }
    }
    __syncthreads();                         // `red` may still hold the caller's first-stage values until every lane has read them
    red[0][ty][tx] = a0;
    red[1][ty][tx] = a1;
    __syncthreads();
    if (ty == 0 && need) {
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) { d0 += red[0][k][tx]; d1 += red[1][k][tx]; }
        const double m = (double)mf - K;     // back to moments of (y - K)
        S1 = n * m + d0;
        S2 = d1 + 2.0 * m * d0 + n * m * m;
    }
}

// Second half: one thread per channel, from the (local, or all-reduced) moments over n values.
__device__ __forceinline__ void bn_coeffs_from_moments(double S1, double S2, double K, int c, double n, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float momentum, int clamp_eps,
                                                       float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                                       float* shift, const unsigned* bound_extra, float bound_mul, unsigned* bound_out) {
    if (bound_out != nullptr) bn_commit_bound(gamma[c], beta[c], n, bound_extra, bound_mul, bound_out);
    const double mk = S1 / n;                // moments of (y - K)
    double var = S2 / n - mk * mk;
    if (var < 0.0) var = 0.0;
    const double mu = K + mk;
    // torch.nn.BatchNorm2d: 1/sqrt(var + eps); vendored SyncBN (batchnorm.py:125): clamp(var, eps)^-1/2
    const double is = clamp_eps ? 1.0 / sqrt(var > (double)eps ? var : (double)eps) : 1.0 / sqrt(var + (double)eps);
    const float mu_f = (float)mu, is_f = (float)is;
    mean[c] = mu_f;
    invstd[c] = is_f;
    const float sc = gamma[c] * is_f;
    scale[c] = sc;
    shift[c] = beta[c] - mu_f * sc;
    if (running_mean != nullptr) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu_f;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__device__ __forceinline__ void bn_finalize_tail(double sum, double sumsq, double (*red)[32][9], int tx, int ty, int c, double n, int C,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                 int clamp_eps, float* running_mean, float* running_var, float* mean, float* invstd,
                                                 float* scale, float* shift, const unsigned* bound_extra, float bound_mul,
                                                 unsigned* bound_out, RefineSrc src) {
    double S1, S2;
    bn_local_moments(sum, sumsq, red, tx, ty, c, n, C, src, S1, S2);
    if (ty != 0 || c >= C) return;
    bn_coeffs_from_moments(S1, S2, src.shift != nullptr ? (double)src.shift[c] : 0.0, c, n, gamma, beta, eps, momentum, clamp_eps, running_mean,
                           running_var, mean, invstd, scale, shift, bound_extra, bound_mul, bound_out);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, double n, int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum, int clamp_eps,
                                                          float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                                          float* shift, const unsigned* bound_extra, float bound_mul, unsigned* bound_out,
                                                          RefineSrc src) {
    __shared__ double red[2][32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx;
    const bool mine = ty == 0 && c < C;
    bn_finalize_tail(mine ? (double)sums[c] : 0.0, mine ? (double)sums[C + c] : 0.0, red, tx, ty, c, n, C, gamma, beta, eps, momentum, clamp_eps,
                     running_mean, running_var, mean, invstd, scale, shift, bound_extra, bound_mul, bound_out, src);
}

// SyncBN, stage 1: this rank's moments in fp64 ([S1 | S2 | count], the buffer ONE all-reduce sums over the ranks), refined like the
// single-GPU statistics.  Stage 2 (bn_finalize_moments_kernel) is the second half of bn_finalize_kernel, so a one-rank group computes
// bit for bit what the group-less path does.
__global__ __launch_bounds__(256) void bn_local_moments_kernel(const float* __restrict__ sums, double n, int C, RefineSrc src,
                                                               double* __restrict__ moments) {
    __shared__ double red[2][32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx;
    const bool mine = ty == 0 && c < C;
    double S1, S2;
    bn_local_moments(mine ? (double)sums[c] : 0.0, mine ? (double)sums[C + c] : 0.0, red, tx, ty, c, n, C, src, S1, S2);
    if (mine) {
        moments[c] = S1;
        moments[C + c] = S2;
        if (c == 0) moments[2 * C] = n;
    }
}

__global__ void bn_finalize_moments_kernel(const double* __restrict__ moments, double n, int C, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float eps, float momentum, int clamp_eps, float* running_mean,
                                           float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                           const unsigned* bound_extra, float bound_mul, unsigned* bound_out, const float* stat_shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    bn_coeffs_from_moments(moments[c], moments[C + c], stat_shift != nullptr ? (double)stat_shift[c] : 0.0, c, n, gamma, beta, eps, momentum,
                           clamp_eps, running_mean, running_var, mean, invstd, scale, shift, bound_extra, bound_mul, bound_out);
}

// bn_finalize fed directly by the conv epilogue's per-tile partials: the fp64 column combine (column_sum_kernel's
// arithmetic, 8 channels x 32 row lanes per block, fixed order) and the coefficient math in one launch.
__global__ __launch_bounds__(256) void bn_finalize_partial_kernel(const float* __restrict__ partial, int nrows, double n, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                  float momentum, int clamp_eps, float* running_mean, float* running_var,
                                                                  float* mean, float* invstd, float* scale, float* shift,
                                                                  const unsigned* bound_extra, float bound_mul, unsigned* bound_out,
                                                                  RefineSrc src) {
    __shared__ double red[2][32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        // eight rows (16 loads) in flight per thread, tail batch included (rows past the end add an exact 0); row order of the adds unchanged
        for (int r = ty; r < nrows; r += 8 * 32) {
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + u * 32;
                const size_t row = rr < nrows ? (size_t)rr : 0;
                v0[u] = partial[row * 2 * C + c];
                v1[u] = partial[row * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = r + u * 32 < nrows;
                a0 += ok ? (double)v0[u] : 0.0;
                a1 += ok ? (double)v1[u] : 0.0;
            }
        }
    }
    red[0][ty][tx] = a0;
    red[1][ty][tx] = a1;
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    if (ty == 0) {
#pragma unroll
        for (int k = 0; k < 32; ++k) { s0 += red[0][k][tx]; s1 += red[1][k][tx]; }
    }
    // the two-launch path rounds the sums to fp32 in between
    bn_finalize_tail((double)(float)s0, (double)(float)s1, red, tx, ty, c, n, C, gamma, beta, eps, momentum, clamp_eps, running_mean, running_var,
                     mean, invstd, scale, shift, bound_extra, bound_mul, bound_out, src);
}

__global__ void bn_eval_coeffs_kernel(const float* rm, const float* rv, const float* gamma, const float* beta, float eps,
                                      int C, float* scale, float* shift, float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
    if (mean != nullptr) {                   // what an eval-mode backward needs (running statistics are constants)
        mean[c] = rm[c];
        invstd[c] = 1.f / sqrtf(rv[c] + eps);
    }
}

// ---- elementwise -------------------------------------------------------------------------------
template <bool EX = false, bool DROP = false, bool BITS = false, bool YH = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int y_pitch,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ res, int res_pitch, float* __restrict__ out,
                                                       int out_pitch, int relu, Slab g, unsigned* __restrict__ amax_out, BnEx ex) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    float amax = 0.f;
    PlanesRef po{}, pr{};
    DropRef dr{};
    float res_inv = 1.f;
    if constexpr (EX) {
        if (ex.out_pl != nullptr) po = PlanesRef{ex.out_pl, ex.out_ps, ex.nplanes, pow2_scale_for(*ex.out_bound)};
        if (ex.res_pl != nullptr) {
            pr = PlanesRef{const_cast<_Float16*>(ex.res_pl), ex.res_ps, ex.nplanes, 1.f};
            res_inv = 1.f / pow2_scale_for(*ex.res_amax);
        }
        dr = DropRef{ex.drop_thresh, ex.keep_scale, ex.seed};
    }
    const bool res_pl = EX && pr.base != nullptr, out_pl = EX && po.base != nullptr;
    float y_inv = 1.f;
    if constexpr (YH) y_inv = 1.f / half_scale_for(*ex.y_bound);
    if (ty < g.RL) {
        for (int cv = tx; cv < g.CV; cv += g.cols) {
            const f32x4 sc = ld4(scale + 4 * cv), sh = ld4(shift + 4 * cv);
            f32x4 vy[kRowBatch], vr[kRowBatch];
            PlanesRaw vp[EX ? kRowBatch : 1];
            unsigned nb[BITS ? kRowBatch : 1];
            walk_rows(r_begin + ty, r_end, g.RL,
                [&](int u, long long r) {
                    vy[u] = ldq<YH>(y, (size_t)(r * y_pitch + 4 * cv), y_inv);
                    if (res != nullptr) vr[u] = ld4(res + r * res_pitch + 4 * cv);
                    if constexpr (EX) { if (res_pl) vp[u] = planes_raw4(pr, r * res_pitch + 4 * cv); }
                },
                [&](int u, long long r, bool valid) {
                    f32x4 v = vy[u] * sc + sh;
                    if (res != nullptr) v += vr[u];
                    if constexpr (EX) { if (res_pl) v += planes_value4(vp[u], res_inv); }
                    if constexpr (BITS) nb[u] = relu_nibble(v);
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if constexpr (DROP) v = drop4(dr, r, g.CV, cv, v);
                    vy[u] = v;
                    if (valid) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                },
                [&](int u, long long r) {
                    if (out_pl) { if constexpr (EX) planes_store4(po, r * out_pitch + 4 * cv, vy[u]); }
                    else st4(out + r * out_pitch + 4 * cv, vy[u]);
                    if constexpr (BITS) mask_store(ex.mask, r * g.CV + cv, nb[u]);
                });
        }
    }
    if (amax_out != nullptr) amax_commit(amax, amax_out);      // range of the output for the f16x3 conv that consumes it
}

// EX: `out` (ReLU mask) may be an fp16-plane tensor, the forward's dropout is regenerated, dy may be written as fp16 planes scaled
// with the bound in ex.dy_bound (bn_bwd_sums_kernel).
template <bool EX = false, bool DROP = false, int MS = MS_Y, bool YH = false, bool AH = false>
__global__ __launch_bounds__(256, 3) void bn_bwd_apply_kernel(const float* __restrict__ dout, int dout_pitch,
                                                           const float* __restrict__ out, int out_pitch,
                                                           const float* __restrict__ y, int y_pitch,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           float inv_n, int C, int relu, float* __restrict__ dy, int dy_pitch,
                                                           float* __restrict__ g_out, int g_pitch, Slab g,
                                                           unsigned* __restrict__ amax_dy, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, BnEx ex) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    float amax = 0.f;
    PlanesRef po{}, pd{};
    DropRef dr{};
    if constexpr (EX) {
        if (ex.out_pl != nullptr) po = PlanesRef{ex.out_pl, ex.out_ps, ex.nplanes, 1.f};
        if (ex.dy_pl != nullptr) pd = PlanesRef{ex.dy_pl, ex.dy_ps, ex.nplanes, pow2_scale_for(*ex.dy_bound)};
        dr = DropRef{ex.drop_thresh, ex.keep_scale, ex.seed};
    }
    const bool dy_pl = EX && pd.base != nullptr;
    float a_inv = 1.f, y_inv = 1.f, g_scale = 1.f;
    if constexpr (AH) { a_inv = 1.f / half_scale_for(*ex.dout_bound); g_scale = half_scale_for(*ex.dout_bound); }
    if constexpr (YH) y_inv = 1.f / half_scale_for(*ex.y_bound);
    if (ty < g.RL) {
        for (int cv = tx; cv < g.CV; cv += g.cols) {
            const f32x4 mu = ld4(mean + 4 * cv), is = ld4(invstd + 4 * cv);
            const f32x4 k = ld4(gamma + 4 * cv) * is;
            const f32x4 sgx = ld4(sums + 4 * cv) * inv_n, sg = ld4(sums + C + 4 * cv) * inv_n;
            const bool remask = MS == MS_Y && relu;
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
            if (remask) { sc = ld4(scale + 4 * cv); sh = ld4(shift + 4 * cv); }
            constexpr int NB = 4;            // 4 rows x 2-3 tensors in flight per wave; <= 168 registers = 3 waves per SIMD = the 3 blocks per CU of a 768-slab launch
            f32x4 vg[NB], vy[NB], vo[MS == MS_OUT ? NB : 1];
            PlanesRaw vp[MS == MS_PLANES ? NB : 1];
            unsigned vm[MS == MS_BITS ? NB : 1];
            walk_rows<NB>(r_begin + ty, r_end, g.RL,
                [&](int u, long long r) {
                    vg[u] = ldq<AH>(dout, (size_t)(r * dout_pitch + 4 * cv), a_inv);
                    vy[u] = ldq<YH>(y, (size_t)(r * y_pitch + 4 * cv), y_inv);
                    if constexpr (MS == MS_OUT) vo[u] = ld4(out + r * out_pitch + 4 * cv);
                    if constexpr (MS == MS_PLANES) vp[u] = planes_raw4(po, r * out_pitch + 4 * cv);
                    if constexpr (MS == MS_BITS) vm[u] = ex.mask[(r * g.CV + cv) >> 1];
                },
                [&](int u, long long r, bool valid) {
                    f32x4 gg = vg[u];
                    if constexpr (DROP) gg = drop4(dr, r, g.CV, cv, gg);
                    if (remask) gg = relu_mask(gg, vy[u] * sc + sh);
                    if constexpr (MS == MS_OUT) gg = relu_mask(gg, vo[u]);
                    if constexpr (MS == MS_PLANES) gg = relu_mask(gg, planes_sign4(vp[u]));
                    if constexpr (MS == MS_BITS) gg = mask_apply(gg, vm[u], cv);
                    const f32x4 xh = (vy[u] - mu) * is;
                    const f32x4 v = k * (gg - sg - xh * sgx);
                    vg[u] = gg;                                      // the masked gradient (residual branch) and dy, kept for the store pass
                    vy[u] = v;
                    if (valid) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                },
                [&](int u, long long r) {
                    if (g_out != nullptr) stq<AH>(g_out, (size_t)(r * g_pitch + 4 * cv), vg[u], g_scale);
                    if (dy_pl) { if constexpr (EX) planes_store4(pd, r * dy_pitch + 4 * cv, vy[u]); }
                    else st4(dy + r * dy_pitch + 4 * cv, vy[u]);
                });
        }
    }
    if (amax_dy != nullptr) amax_commit(amax, amax_dy);
}

// Column combine of bn_reduce_kernel<1>'s partials (column_sum_kernel's arithmetic: fp64, fixed order) for BOTH halves of 8 channels
// per block, plus the range bound of the dy that bn_bwd_apply is about to write as fp16 planes:
//     dy = k (g - sg - xhat sgx),  k = gamma invstd  =>  |dy| <= |k| (max|g| + |sg| + sqrt(n - 1) |sgx|)       (|xhat| <= sqrt(n - 1))
// with the actual per-channel means sg = sum g / n, sgx = sum g xhat / n (sgx is O(max|g| / sqrt(n)) for uncorrelated g, so the last
// term stays O(max|g|)).  `sums` = [sum g xhat | sum g] (parameter order).
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ partial, int nrows, int C, float* __restrict__ sums,
                                                          const float* __restrict__ gamma, const float* __restrict__ invstd, double n,
                                                          const unsigned* __restrict__ g_amax, unsigned* __restrict__ bound_out) {
    __shared__ double red[2][32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        // sixteen rows (32 loads) in flight per thread, the tail batch included: rows past the end load row 0 and add an exact 0, so the
        // 512-row combine of a 512-slab reduce (slab.h kRowSlabs) is ONE load round trip; row order of the adds unchanged
        for (int r = ty; r < nrows; r += 16 * 32) {
            float v0[16], v1[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int rr = r + u * 32;
                const size_t row = rr < nrows ? (size_t)rr : 0;
                v0[u] = partial[row * 2 * C + c];
                v1[u] = partial[row * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool ok = r + u * 32 < nrows;
                a0 += ok ? (double)v0[u] : 0.0;
                a1 += ok ? (double)v1[u] : 0.0;
            }
        }
    }
    red[0][ty][tx] = a0;
    red[1][ty][tx] = a1;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) { s0 += red[0][k][tx]; s1 += red[1][k][tx]; }
    const float sgx = (float)s0, sg = (float)s1;
    sums[c] = sgx;
    sums[C + c] = sg;
    if (bound_out != nullptr) {
        const float gmax = __uint_as_float(*g_amax);
        const float inv_n = (float)(1.0 / n);
        const float b = fabsf(gamma[c] * invstd[c]) * (gmax + fabsf(sg) * inv_n + (float)sqrt(n > 1.0 ? n - 1.0 : 1.0) * fabsf(sgx) * inv_n);
        atomicMax(bound_out, __float_as_uint(b));
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void relu_kernel(const float* __restrict__ a, int a_pitch, const float* __restrict__ o, int o_pitch,
                                                   float* __restrict__ dst, int dst_pitch, Slab g) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    if (ty >= g.RL) return;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    for (int cv = tx; cv < g.CV; cv += g.cols) {
        f32x4 va[kRowBatch], vo[kRowBatch];
        walk_rows(r_begin + ty, r_end, g.RL,
            [&](int u, long long r) {
                va[u] = ld4(a + r * a_pitch + 4 * cv);
                if (MODE != 0) vo[u] = ld4(o + r * o_pitch + 4 * cv);
            },
            [&](int u, long long, bool) {
                f32x4 v = va[u];
                if (MODE == 0) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else v = relu_mask(v, vo[u]);
                va[u] = v;
            },
            [&](int u, long long r) { st4(dst + r * dst_pitch + 4 * cv, va[u]); });
    }
}

__global__ __launch_bounds__(256) void relu_bits_kernel(const float* __restrict__ a, const unsigned char* __restrict__ mask, float* __restrict__ dst, Slab g) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    if (ty >= g.RL) return;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    const int C = 4 * g.CV;
    for (int cv = tx; cv < g.CV; cv += g.cols) {
        f32x4 va[kRowBatch];
        unsigned vm[kRowBatch];
        walk_rows(r_begin + ty, r_end, g.RL,
            [&](int u, long long r) { va[u] = ld4(a + r * C + 4 * cv); vm[u] = mask[(r * g.CV + cv) >> 1]; },
            [&](int u, long long, bool) { va[u] = mask_apply(va[u], vm[u], cv); },
            [&](int u, long long r) { st4(dst + r * C + 4 * cv, va[u]); });
    }
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int x_pitch, float* __restrict__ out, int out_pitch,
                                                      unsigned thresh16, float keep_scale, unsigned long long seed, Slab g) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    if (ty >= g.RL) return;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    for (int cv = tx; cv < g.CV; cv += g.cols) {
        f32x4 vx[kRowBatch];
        walk_rows(r_begin + ty, r_end, g.RL,
            [&](int u, long long r) { vx[u] = ld4(x + r * x_pitch + 4 * cv); },
            [&](int u, long long r, bool) {
                const unsigned long long h = mix64(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(r * g.CV + cv + 1));
                f32x4 v = vx[u];
                v.x = ((h & 0xFFFF) >= thresh16) ? v.x * keep_scale : 0.f;
                v.y = (((h >> 16) & 0xFFFF) >= thresh16) ? v.y * keep_scale : 0.f;
                v.z = (((h >> 32) & 0xFFFF) >= thresh16) ? v.z * keep_scale : 0.f;
                v.w = (((h >> 48) & 0xFFFF) >= thresh16) ? v.w * keep_scale : 0.f;
                vx[u] = v;
            },
            [&](int u, long long r) { st4(out + r * out_pitch + 4 * cv, vx[u]); });
    }
}

static BnEx make_ex(const PylcBnExtra* e) {
    BnEx x{};
    if (e == nullptr) return x;
    x.out_pl = static_cast<_Float16*>(e->out_planes); x.out_ps = e->out_plane_stride; x.out_bound = e->out_bound;
    x.res_pl = static_cast<const _Float16*>(e->res_planes); x.res_ps = e->res_plane_stride; x.res_amax = e->res_amax;
    x.dy_pl = static_cast<_Float16*>(e->dy_planes); x.dy_ps = e->dy_plane_stride; x.dy_bound = e->dy_bound;
    x.nplanes = e->nplanes == 1 ? 1 : 2;
    if (e->drop_p > 0.f) {
        x.drop_thresh = (unsigned)(e->drop_p * 65536.0f + 0.5f);      // == pylc_dropout's threshold and scale
        x.keep_scale = 1.0f / (1.0f - e->drop_p);
        x.seed = (unsigned long long)e->drop_seed;
    }
    x.g_amax = e->g_amax;
    x.mask = static_cast<unsigned char*>(e->relu_mask);
    x.y_bound = e->y_half_bound;
    x.dout_bound = e->dout_half_bound;
    return x;
}

static int check_ex(const PylcBnExtra* e, int C, const char* what) {
    if (e == nullptr) return PYLC_OK;
    PYLC_REQUIRE(e->drop_p >= 0.f && e->drop_p < 1.f, "%s: dropout p must be in [0,1)", what);
    PYLC_REQUIRE(!e->out_planes || (e->nplanes == 1 || e->out_plane_stride > 0), "%s: bad out plane stride", what);
    PYLC_REQUIRE(!e->out_planes || (reinterpret_cast<uintptr_t>(e->out_planes) & 7) == 0, "%s: planes must be 8-byte aligned", what);
    PYLC_REQUIRE(!e->res_planes || e->res_amax, "%s: residual planes need their range (res_amax)", what);
    PYLC_REQUIRE(!e->dy_planes || e->dy_bound, "%s: dy planes need dy_bound", what);
    PYLC_REQUIRE(!e->relu_mask || C % 8 == 0, "%s: the 1-bit ReLU mask needs C %% 8 == 0 (C=%d)", what, C);
    return PYLC_OK;
}

static int check_mc(long long M, int C, int pitch, const char* what) {
    PYLC_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "%s: need M > 0 and C %% 4 == 0 (M=%lld C=%d)", what, M, C);
    PYLC_REQUIRE(pitch >= C && pitch % 4 == 0, "%s: pitch %d invalid for C=%d", what, pitch, C);
    return PYLC_OK;
}

}  // namespace pylc

using namespace pylc;

extern "C" size_t pylc_bn_workspace_floats(long long M, int C) {
    (void)M;
    return (size_t)kMaxSlabs * 2 * (size_t)C;
}

extern "C" int pylc_bn_stats(const float* y, long long M, int C, int y_pitch, float* sums, float* workspace, void* stream) {
    if (int rc = check_mc(M, C, y_pitch, "bn_stats")) return rc;
    PYLC_REQUIRE(y && sums && workspace, "bn_stats: null pointer");
    const Slab g = make_slab(M, C);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(g.nslab), dim3(256), 0, st, y, y_pitch, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, g, C,
                       workspace, nullptr, nullptr, BnEx{});
    PYLC_LAUNCH_CHECK();
    hipLaunchKernelGGL(column_sum_kernel, dim3(cdiv(2 * C, 8)), dim3(256), 0, st, workspace, g.nslab, 2 * C, sums);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_stats_from_partial(const float* partial, int n_rows, int C, float* sums, void* stream) {
    PYLC_REQUIRE(partial && sums && n_rows > 0 && C > 0, "bn_stats_from_partial: bad arguments");
    hipLaunchKernelGGL(column_sum_kernel, dim3(cdiv(2 * C, 8)), dim3(256), 0, as_stream(stream), partial, n_rows, 2 * C, sums);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_finalize_ex(const float* sums, double n, int C, const float* gamma, const float* beta, float eps, float momentum,
                                   int clamp_eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                   float* shift, const unsigned int* bound_extra, float bound_mul, unsigned int* bound_out,
                                   const float* y, int y_pitch, long long M, const float* stat_shift, void* stream) {
    PYLC_REQUIRE(sums && gamma && beta && mean && invstd && scale && shift && C > 0 && n > 0, "bn_finalize: bad arguments");
    PYLC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must both be set or both NULL");
    PYLC_REQUIRE(y == nullptr || (y_pitch >= C && M > 0 && (double)M == n), "bn_finalize: the refinement source must be the n rows the sums cover");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 8)), dim3(256), 0, as_stream(stream), sums, n, C, gamma, beta, eps, momentum,
                       clamp_eps, running_mean, running_var, mean, invstd, scale, shift, bound_extra, bound_mul, bound_out,
                       RefineSrc{y, y_pitch, M, stat_shift});
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_finalize(const float* sums, double n, int C, const float* gamma, const float* beta, float eps, float momentum,
                                int clamp_eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                float* shift, void* stream) {
    return pylc_bn_finalize_ex(sums, n, C, gamma, beta, eps, momentum, clamp_eps, running_mean, running_var, mean, invstd, scale, shift,
                               nullptr, 1.f, nullptr, nullptr, 0, 0, nullptr, stream);
}

extern "C" int pylc_bn_finalize_from_partial_ex(const float* partial, int n_rows, double n, int C, const float* gamma, const float* beta,
                                                float eps, float momentum, int clamp_eps, float* running_mean, float* running_var,
                                                float* mean, float* invstd, float* scale, float* shift, const unsigned int* bound_extra,
                                                float bound_mul, unsigned int* bound_out, const float* y, int y_pitch, long long M,
                                                const float* stat_shift, void* stream) {
    PYLC_REQUIRE(partial && n_rows > 0 && gamma && beta && mean && invstd && scale && shift && C > 0 && n > 0,
                 "bn_finalize_from_partial: bad arguments");
    PYLC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_from_partial: running stats must both be set or both NULL");
    PYLC_REQUIRE(y == nullptr || (y_pitch >= C && M > 0 && (double)M == n),
                 "bn_finalize_from_partial: the refinement source must be the n rows the partials cover");
    hipLaunchKernelGGL(bn_finalize_partial_kernel, dim3(cdiv(C, 8)), dim3(256), 0, as_stream(stream), partial, n_rows, n, C, gamma, beta, eps,
                       momentum, clamp_eps, running_mean, running_var, mean, invstd, scale, shift, bound_extra, bound_mul, bound_out,
                       RefineSrc{y, y_pitch, M, stat_shift});
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_finalize_from_partial(const float* partial, int n_rows, double n, int C, const float* gamma, const float* beta,
                                             float eps, float momentum, int clamp_eps, float* running_mean, float* running_var,
                                             float* mean, float* invstd, float* scale, float* shift, void* stream) {
    return pylc_bn_finalize_from_partial_ex(partial, n_rows, n, C, gamma, beta, eps, momentum, clamp_eps, running_mean, running_var, mean,
                                            invstd, scale, shift, nullptr, 1.f, nullptr, nullptr, 0, 0, nullptr, stream);
}

extern "C" int pylc_bn_local_moments(const float* sums, double n, int C, const float* y, int y_pitch, long long M, const float* stat_shift,
                                     double* moments, void* stream) {
    PYLC_REQUIRE(sums && moments && C > 0 && n > 0, "bn_local_moments: bad arguments");
    PYLC_REQUIRE(y == nullptr || (y_pitch >= C && M > 0 && (double)M == n), "bn_local_moments: the refinement source must be the n rows the sums cover");
    hipLaunchKernelGGL(bn_local_moments_kernel, dim3(cdiv(C, 8)), dim3(256), 0, as_stream(stream), sums, n, C, RefineSrc{y, y_pitch, M, stat_shift},
                       moments);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_finalize_moments(const double* moments, double n, int C, const float* gamma, const float* beta, float eps,
                                        float momentum, int clamp_eps, float* running_mean, float* running_var, float* mean, float* invstd,
                                        float* scale, float* shift, const unsigned int* bound_extra, float bound_mul,
                                        unsigned int* bound_out, const float* stat_shift, void* stream) {
    PYLC_REQUIRE(moments && gamma && beta && mean && invstd && scale && shift && C > 0 && n > 0, "bn_finalize_moments: bad arguments");
    PYLC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_moments: running stats must both be set or both NULL");
    hipLaunchKernelGGL(bn_finalize_moments_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), moments, n, C, gamma, beta, eps, momentum,
                       clamp_eps, running_mean, running_var, mean, invstd, scale, shift, bound_extra, bound_mul, bound_out, stat_shift);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_eval_coeffs(const float* rm, const float* rv, const float* gamma, const float* beta, float eps, int C, float* scale,
                                   float* shift, void* stream) {
    PYLC_REQUIRE(rm && rv && gamma && beta && scale && shift && C > 0, "bn_eval_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), rm, rv, gamma, beta, eps, C, scale, shift,
                       nullptr, nullptr);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_eval_coeffs_full(const float* rm, const float* rv, const float* gamma, const float* beta, float eps, int C, float* scale,
                                        float* shift, float* mean, float* invstd, void* stream) {
    PYLC_REQUIRE(rm && rv && gamma && beta && scale && shift && mean && invstd && C > 0, "bn_eval_coeffs_full: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), rm, rv, gamma, beta, eps, C, scale, shift,
                       mean, invstd);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_apply_ex(const float* y, int y_pitch, const float* scale, const float* shift, const float* residual, int res_pitch,
                                float* out, int out_pitch, long long M, int C, int relu, unsigned int* amax_out, const PylcBnExtra* ex,
                                void* stream) {
    if (int rc = check_mc(M, C, y_pitch, "bn_apply")) return rc;
    if (int rc = check_mc(M, C, out_pitch, "bn_apply(out)")) return rc;
    if (int rc = check_ex(ex, C, "bn_apply")) return rc;
    const bool out_pl = ex && ex->out_planes, res_pl = ex && ex->res_planes;
    PYLC_REQUIRE(y && scale && shift && (out != nullptr) != out_pl, "bn_apply: null pointer (exactly one of out / ex->out_planes)");
    PYLC_REQUIRE(!out_pl || ex->out_bound, "bn_apply: planes output needs out_bound");
    PYLC_REQUIRE(!(residual && res_pl), "bn_apply: residual given twice");
    PYLC_REQUIRE((residual == nullptr && !res_pl) || (res_pitch >= C && res_pitch % 4 == 0), "bn_apply: bad residual pitch");
    const Slab g = make_slab(M, C);
    if (ex != nullptr && ex->y_half_bound != nullptr) {          // y is a one-plane fp16 tensor (precision mode 3)
        const bool bits = ex->relu_mask != nullptr && relu, drop = ex->drop_p > 0.f;
        PYLC_REQUIRE(!(bits && drop), "bn_apply: the 1-bit ReLU mask and fused dropout are not combined");
#define PYLC_BN_APPLY_H(DROPV, BITSV)                                                                                                    \
        hipLaunchKernelGGL((bn_apply_kernel<true, DROPV, BITSV, true>), dim3(g.nslab), dim3(256), 0, as_stream(stream), y, y_pitch, scale, shift, residual, \
                           res_pitch, out, out_pitch, relu, g, amax_out, make_ex(ex))
        if (bits) PYLC_BN_APPLY_H(false, true); else if (drop) PYLC_BN_APPLY_H(true, false); else PYLC_BN_APPLY_H(false, false);
#undef PYLC_BN_APPLY_H
    } else if (ex != nullptr && ex->relu_mask != nullptr && relu) {
        PYLC_REQUIRE(!(ex->drop_p > 0.f), "bn_apply: the 1-bit ReLU mask and fused dropout are not combined");
        hipLaunchKernelGGL((bn_apply_kernel<true, false, true>), dim3(g.nslab), dim3(256), 0, as_stream(stream), y, y_pitch, scale, shift, residual, res_pitch,
                           out, out_pitch, relu, g, amax_out, make_ex(ex));
    } else if (ex != nullptr && ex->drop_p > 0.f)
        hipLaunchKernelGGL((bn_apply_kernel<true, true>), dim3(g.nslab), dim3(256), 0, as_stream(stream), y, y_pitch, scale, shift, residual, res_pitch,
                           out, out_pitch, relu, g, amax_out, make_ex(ex));
    else if (ex != nullptr)
        hipLaunchKernelGGL((bn_apply_kernel<true, false>), dim3(g.nslab), dim3(256), 0, as_stream(stream), y, y_pitch, scale, shift, residual, res_pitch,
                           out, out_pitch, relu, g, amax_out, make_ex(ex));
    else
        hipLaunchKernelGGL((bn_apply_kernel<false>), dim3(g.nslab), dim3(256), 0, as_stream(stream), y, y_pitch, scale, shift, residual, res_pitch,
                           out, out_pitch, relu, g, amax_out, BnEx{});
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_apply(const float* y, int y_pitch, const float* scale, const float* shift, const float* residual, int res_pitch,
                             float* out, int out_pitch, long long M, int C, int relu, unsigned int* amax_out, void* stream) {
    return pylc_bn_apply_ex(y, y_pitch, scale, shift, residual, res_pitch, out, out_pitch, M, C, relu, amax_out, nullptr, stream);
}

extern "C" int pylc_bn_bwd_reduce_ex(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                                     const float* mean, const float* invstd, long long M, int C, int relu, float* sums, float* workspace,
                                     const float* scale, const float* shift, const float* gamma, double n, const PylcBnExtra* ex,
                                     unsigned int* dy_bound_out, void* stream) {
    if (int rc = check_mc(M, C, dout_pitch, "bn_bwd_reduce")) return rc;
    if (int rc = check_mc(M, C, y_pitch, "bn_bwd_reduce(y)")) return rc;
    if (int rc = check_ex(ex, C, "bn_bwd_reduce")) return rc;
    PYLC_REQUIRE(dout && y && mean && invstd && sums && workspace, "bn_bwd_reduce: null pointer");
    const bool out_pl = ex && ex->out_planes;
    PYLC_REQUIRE(!relu || (out && out_pitch >= C && out_pitch % 4 == 0) || out_pl || (ex && ex->relu_mask) || (!out && scale && shift),
                 "bn_bwd_reduce: relu needs `out` (fp32 or planes), the 1-bit mask, or scale and shift to recompute the mask from y");
    PYLC_REQUIRE(!dy_bound_out || (ex && ex->g_amax && gamma && n > 0), "bn_bwd_reduce: the dy bound needs ex->g_amax, gamma and n");
    const Slab g = make_slab(M, C);
    hipStream_t st = as_stream(stream);
    // where the ReLU mask comes from picks the instantiation (only that source's registers are live in the kernel)
    const int ms = !relu ? MS_Y : (ex && ex->relu_mask) ? MS_BITS : out_pl ? MS_PLANES : out ? MS_OUT : MS_Y;
    const bool drop = ex != nullptr && ex->drop_p > 0.f;
    PYLC_REQUIRE(!(drop && ms == MS_BITS), "bn_bwd_reduce: the 1-bit ReLU mask and fused dropout are not combined");
    const bool yh = ex != nullptr && ex->y_half_bound != nullptr, ah = ex != nullptr && ex->dout_half_bound != nullptr;      // one-plane fp16 y / dout (precision mode 3)
#define PYLC_BN_REDUCE(EXV, DROPV, MSV)                                                                                                   \
    hipLaunchKernelGGL((bn_reduce_kernel<1, EXV, DROPV, MSV>), dim3(g.nslab), dim3(256), 0, st, dout, dout_pitch, out, out_pitch, y, y_pitch, mean, \
                       invstd, relu, g, C, workspace, scale, shift, ex ? make_ex(ex) : BnEx{})
#define PYLC_BN_REDUCE_H2(DROPV, MSV, YHV, AHV)                                                                                           \
    hipLaunchKernelGGL((bn_reduce_kernel<1, true, DROPV, MSV, YHV, AHV>), dim3(g.nslab), dim3(256), 0, st, dout, dout_pitch, out, out_pitch, y, y_pitch, mean, \
                       invstd, relu, g, C, workspace, scale, shift, make_ex(ex))
#define PYLC_BN_REDUCE_H(DROPV, MSV)                                                                                                      \
    { if (yh && ah) PYLC_BN_REDUCE_H2(DROPV, MSV, true, true); else if (yh) PYLC_BN_REDUCE_H2(DROPV, MSV, true, false); else PYLC_BN_REDUCE_H2(DROPV, MSV, false, true); }
    if (yh || ah) {
        PYLC_REQUIRE(ms != MS_OUT && !(drop && ms != MS_Y), "bn_bwd_reduce (half operands): mask from y, the 1-bit mask or fp16-plane out; dropout with the y mask only");
        if (drop) PYLC_BN_REDUCE_H(true, MS_Y) else if (ms == MS_BITS) PYLC_BN_REDUCE_H(false, MS_BITS)
        else if (ms == MS_PLANES) PYLC_BN_REDUCE_H(false, MS_PLANES) else PYLC_BN_REDUCE_H(false, MS_Y)
    } else
    if (ex == nullptr) { if (ms == MS_OUT) PYLC_BN_REDUCE(false, false, MS_OUT); else PYLC_BN_REDUCE(false, false, MS_Y); }
    else if (drop) { if (ms == MS_OUT) PYLC_BN_REDUCE(true, true, MS_OUT); else if (ms == MS_PLANES) PYLC_BN_REDUCE(true, true, MS_PLANES); else PYLC_BN_REDUCE(true, true, MS_Y); }
    else if (ms == MS_BITS) PYLC_BN_REDUCE(true, false, MS_BITS);
    else if (ms == MS_PLANES) PYLC_BN_REDUCE(true, false, MS_PLANES);
    else if (ms == MS_OUT) PYLC_BN_REDUCE(true, false, MS_OUT);
    else PYLC_BN_REDUCE(true, false, MS_Y);
#undef PYLC_BN_REDUCE
#undef PYLC_BN_REDUCE_H
#undef PYLC_BN_REDUCE_H2
    PYLC_LAUNCH_CHECK();
    if (dy_bound_out != nullptr)
        hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(cdiv(C, 8)), dim3(256), 0, st, workspace, g.nslab, C, sums, gamma, invstd, n, ex->g_amax,
                           dy_bound_out);
    else
        hipLaunchKernelGGL(column_sum_kernel, dim3(cdiv(2 * C, 8)), dim3(256), 0, st, workspace, g.nslab, 2 * C, sums);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_bwd_reduce(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                                  const float* mean, const float* invstd, long long M, int C, int relu, float* sums, float* workspace,
                                  const float* scale, const float* shift, void* stream) {
    return pylc_bn_bwd_reduce_ex(dout, dout_pitch, out, out_pitch, y, y_pitch, mean, invstd, M, C, relu, sums, workspace, scale, shift,
                                 nullptr, 0.0, nullptr, nullptr, stream);
}

// dy bound from ALL-REDUCED sums (data parallel: the local bound of bn_bwd_reduce_ex would use local means)
__global__ void bn_bwd_bound_kernel(const float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                    double n, int C, const unsigned* __restrict__ g_amax, unsigned* __restrict__ bound_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv_n = (float)(1.0 / n);
    const float b = fabsf(gamma[c] * invstd[c]) *
                    (__uint_as_float(*g_amax) + fabsf(sums[C + c]) * inv_n + (float)sqrt(n > 1.0 ? n - 1.0 : 1.0) * fabsf(sums[c]) * inv_n);
    atomicMax(bound_out, __float_as_uint(b));
}

#ifdef PYLC_EXPERIMENTAL
extern "C" int pylc_bn_bwd_sums_from_partial(const float* partial, int rows, int C, float* sums, const float* gamma, const float* invstd, double n,
                                             const unsigned int* g_amax, unsigned int* dy_bound_out, void* stream) {
    PYLC_REQUIRE(partial && sums && rows > 0 && C > 0, "bn_bwd_sums_from_partial: bad arguments");
    PYLC_REQUIRE(!dy_bound_out || (g_amax && gamma && invstd && n > 0), "bn_bwd_sums_from_partial: the dy bound needs g_amax, gamma, invstd and n");
    hipStream_t st = as_stream(stream);
    if (dy_bound_out != nullptr)
        hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(cdiv(C, 8)), dim3(256), 0, st, partial, rows, C, sums, gamma, invstd, n, g_amax, dy_bound_out);
    else
        hipLaunchKernelGGL(column_sum_kernel, dim3(cdiv(2 * C, 8)), dim3(256), 0, st, partial, rows, 2 * C, sums);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
#endif

extern "C" int pylc_bn_bwd_bound(const float* sums, const float* gamma, const float* invstd, double n, int C, const unsigned int* g_amax,
                                 unsigned int* bound_out, void* stream) {
    PYLC_REQUIRE(sums && gamma && invstd && g_amax && bound_out && C > 0 && n > 0, "bn_bwd_bound: bad arguments");
    hipLaunchKernelGGL(bn_bwd_bound_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), sums, gamma, invstd, n, C, g_amax, bound_out);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_bwd_apply_ex(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                                    const float* mean, const float* invstd, const float* gamma, const float* sums, double n, long long M,
                                    int C, int relu, float* dy, int dy_pitch, float* g_out, int g_pitch, unsigned int* amax_dy,
                                    const float* scale, const float* shift, const PylcBnExtra* ex, void* stream) {
    if (int rc = check_mc(M, C, dout_pitch, "bn_bwd_apply")) return rc;
    if (int rc = check_mc(M, C, dy_pitch, "bn_bwd_apply(dy)")) return rc;
    if (int rc = check_ex(ex, C, "bn_bwd_apply")) return rc;
    const bool out_pl = ex && ex->out_planes, dy_pl = ex && ex->dy_planes;
    PYLC_REQUIRE(dout && y && mean && invstd && gamma && sums && n > 0 && (dy != nullptr) != dy_pl,
                 "bn_bwd_apply: bad arguments (exactly one of dy / ex->dy_planes)");
    PYLC_REQUIRE(!relu || (out && out_pitch >= C) || out_pl || (ex && ex->relu_mask) || (!out && scale && shift),
                 "bn_bwd_apply: relu needs `out` (fp32 or planes), the 1-bit mask, or scale and shift to recompute the mask from y");
    PYLC_REQUIRE(g_out == nullptr || (g_pitch >= C && g_pitch % 4 == 0), "bn_bwd_apply: bad g pitch");
    const Slab g = make_slab(M, C);
    const int ms = !relu ? MS_Y : (ex && ex->relu_mask) ? MS_BITS : out_pl ? MS_PLANES : out ? MS_OUT : MS_Y;
    const bool drop = ex != nullptr && ex->drop_p > 0.f;
    PYLC_REQUIRE(!(drop && ms == MS_BITS), "bn_bwd_apply: the 1-bit ReLU mask and fused dropout are not combined");
    const bool yh = ex != nullptr && ex->y_half_bound != nullptr, ah = ex != nullptr && ex->dout_half_bound != nullptr;      // one-plane fp16 y / dout (+ g_out)
#define PYLC_BN_BWD_APPLY_H2(DROPV, MSV, YHV, AHV)                                                                                        \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, DROPV, MSV, YHV, AHV>), dim3(g.nslab), dim3(256), 0, as_stream(stream), dout, dout_pitch, out, out_pitch, y, \
                       y_pitch, mean, invstd, gamma, sums, (float)(1.0 / n), C, relu, dy, dy_pitch, g_out, g_pitch, g, amax_dy, scale, shift, make_ex(ex))
#define PYLC_BN_BWD_APPLY_H(DROPV, MSV)                                                                                                   \
    { if (yh && ah) PYLC_BN_BWD_APPLY_H2(DROPV, MSV, true, true); else if (yh) PYLC_BN_BWD_APPLY_H2(DROPV, MSV, true, false); else PYLC_BN_BWD_APPLY_H2(DROPV, MSV, false, true); }
#define PYLC_BN_BWD_APPLY(EXV, DROPV, MSV)                                                                                                \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<EXV, DROPV, MSV>), dim3(g.nslab), dim3(256), 0, as_stream(stream), dout, dout_pitch, out, out_pitch, y, \
                       y_pitch, mean, invstd, gamma, sums, (float)(1.0 / n), C, relu, dy, dy_pitch, g_out, g_pitch, g, amax_dy, scale, shift,       \
                       ex ? make_ex(ex) : BnEx{})
    if (yh || ah) {
        PYLC_REQUIRE(ms != MS_OUT && !(drop && ms != MS_Y), "bn_bwd_apply (half operands): mask from y, the 1-bit mask or fp16-plane out; dropout with the y mask only");
        if (drop) PYLC_BN_BWD_APPLY_H(true, MS_Y) else if (ms == MS_BITS) PYLC_BN_BWD_APPLY_H(false, MS_BITS)
        else if (ms == MS_PLANES) PYLC_BN_BWD_APPLY_H(false, MS_PLANES) else PYLC_BN_BWD_APPLY_H(false, MS_Y)
    } else
    if (ex == nullptr) { if (ms == MS_OUT) PYLC_BN_BWD_APPLY(false, false, MS_OUT); else PYLC_BN_BWD_APPLY(false, false, MS_Y); }
    else if (drop) { if (ms == MS_OUT) PYLC_BN_BWD_APPLY(true, true, MS_OUT); else if (ms == MS_PLANES) PYLC_BN_BWD_APPLY(true, true, MS_PLANES); else PYLC_BN_BWD_APPLY(true, true, MS_Y); }
    else if (ms == MS_BITS) PYLC_BN_BWD_APPLY(true, false, MS_BITS);
    else if (ms == MS_PLANES) PYLC_BN_BWD_APPLY(true, false, MS_PLANES);
    else if (ms == MS_OUT) PYLC_BN_BWD_APPLY(true, false, MS_OUT);
    else PYLC_BN_BWD_APPLY(true, false, MS_Y);
#undef PYLC_BN_BWD_APPLY
#undef PYLC_BN_BWD_APPLY_H
#undef PYLC_BN_BWD_APPLY_H2
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_bwd_apply(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                                 const float* mean, const float* invstd, const float* gamma, const float* sums, double n, long long M,
                                 int C, int relu, float* dy, int dy_pitch, float* g_out, int g_pitch, unsigned int* amax_dy,
                                 const float* scale, const float* shift, void* stream) {
    return pylc_bn_bwd_apply_ex(dout, dout_pitch, out, out_pitch, y, y_pitch, mean, invstd, gamma, sums, n, M, C, relu, dy, dy_pitch, g_out,
                                g_pitch, amax_dy, scale, shift, nullptr, stream);
}

extern "C" int pylc_relu_fwd(const float* x, int x_pitch, float* out, int out_pitch, long long M, int C, void* stream) {
    if (int rc = check_mc(M, C, x_pitch, "relu_fwd")) return rc;
    if (int rc = check_mc(M, C, out_pitch, "relu_fwd(out)")) return rc;
    const Slab g = make_slab(M, C);
    hipLaunchKernelGGL((relu_kernel<0>), dim3(g.nslab), dim3(256), 0, as_stream(stream), x, x_pitch, nullptr, 0, out, out_pitch, g);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_relu_bwd(const float* dout, int dout_pitch, const float* out, int out_pitch, float* dx, int dx_pitch, long long M,
                             int C, void* stream) {
    if (int rc = check_mc(M, C, dout_pitch, "relu_bwd")) return rc;
    if (int rc = check_mc(M, C, out_pitch, "relu_bwd(out)")) return rc;
    if (int rc = check_mc(M, C, dx_pitch, "relu_bwd(dx)")) return rc;
    const Slab g = make_slab(M, C);
    hipLaunchKernelGGL((relu_kernel<1>), dim3(g.nslab), dim3(256), 0, as_stream(stream), dout, dout_pitch, out, out_pitch, dx, dx_pitch, g);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_relu_bwd_bits(const float* dout, const void* mask, float* g_out, long long M, int C, void* stream) {
    if (int rc = check_mc(M, C, C, "relu_bwd_bits")) return rc;
    PYLC_REQUIRE(dout && mask && g_out && C % 8 == 0, "relu_bwd_bits: null pointer or C %% 8 != 0");
    const Slab g = make_slab(M, C);
    hipLaunchKernelGGL(relu_bits_kernel, dim3(g.nslab), dim3(256), 0, as_stream(stream), dout, static_cast<const unsigned char*>(mask), g_out, g);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_dropout(const float* x, int x_pitch, float* out, int out_pitch, long long M, int C, float p, uint64_t seed,
                            void* stream) {
    if (int rc = check_mc(M, C, x_pitch, "dropout")) return rc;
    if (int rc = check_mc(M, C, out_pitch, "dropout(out)")) return rc;
    PYLC_REQUIRE(p >= 0.f && p < 1.f, "dropout: p must be in [0,1)");
    const Slab g = make_slab(M, C);
    const unsigned thresh = (unsigned)(p * 65536.0f + 0.5f);
    hipLaunchKernelGGL(dropout_kernel, dim3(g.nslab), dim3(256), 0, as_stream(stream), x, x_pitch, out, out_pitch, thresh,
                       1.0f / (1.0f - p), (unsigned long long)seed, g);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
