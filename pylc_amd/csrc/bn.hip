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
#include "common.h"
#include "slab.h"

namespace pylc {

__device__ __forceinline__ f32x4 relu_mask(f32x4 g, f32x4 o) {
    f32x4 r;
    r.x = o.x > 0.f ? g.x : 0.f; r.y = o.y > 0.f ? g.y : 0.f; r.z = o.z > 0.f ? g.z : 0.f; r.w = o.w > 0.f ? g.w : 0.f;
    return r;
}

// ---- reductions --------------------------------------------------------------------------------
// MODE 0: (sum y, sum y^2).  MODE 1: (sum g, sum g*xhat) with g = dout * (out > 0).
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ a, int a_pitch,
                                                        const float* __restrict__ out, int out_pitch,
                                                        const float* __restrict__ y, int y_pitch,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        int relu, Slab g, int C, float* __restrict__ partial,
                                                        const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr) {
    __shared__ f32x4 red[2][256];
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    for (int cb = 0; cb < g.CV; cb += g.cols) {
        const int cv = cb + tx;
        const bool active = ty < g.RL && cv < g.CV;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
            const bool remask = MODE == 1 && relu && out == nullptr;      // ReLU mask recomputed from y: out = max(y*scale + shift, 0)
            if (MODE == 1) { mu = ld4(mean + 4 * cv); is = ld4(invstd + 4 * cv); }
            if (remask) { sc = ld4(scale + 4 * cv); sh = ld4(shift + 4 * cv); }
            const bool use_out = MODE == 1 && relu && !remask;
            f32x4 va[kRowBatch], vy[kRowBatch], vo[kRowBatch];
            walk_rows(r_begin + ty, r_end, g.RL,
                [&](int u, long long r) {
                    va[u] = ld4(a + r * a_pitch + 4 * cv);
                    if (MODE == 1) vy[u] = ld4(y + r * y_pitch + 4 * cv);
                    if (use_out) vo[u] = ld4(out + r * out_pitch + 4 * cv);
                },
                [&](int u, long long, bool valid) {
                    if (MODE == 0) {
                        if (valid) { s0 += va[u]; s1 += va[u] * va[u]; }
                    } else {
                        f32x4 gg = va[u];
                        if (remask) gg = relu_mask(gg, vy[u] * sc + sh);   // the forward's own expression: identical bits
                        else if (relu) gg = relu_mask(gg, vo[u]);
                        const f32x4 xh = (vy[u] - mu) * is;
                        if (valid) { s0 += gg; s1 += gg * xh; }
                    }
                },
                [](int, long long) {});
        }
        red[0][threadIdx.x] = s0;
        red[1][threadIdx.x] = s1;
        __syncthreads();
        if (ty == 0 && cv < g.CV) {
            for (int k = 1; k < g.RL; ++k) { s0 += red[0][k * g.cols + tx]; s1 += red[1][k * g.cols + tx]; }
            float* p = partial + (size_t)blockIdx.x * 2 * C;
            // MODE 0: [sum | sumsq].  MODE 1: [sum g*xhat (dgamma) | sum g (dbeta)] -- the parameter order, so the
            // result can land directly in the adjacent (gamma, beta) slots of the flat gradient arena
            st4(p + 4 * cv, MODE == 0 ? s0 : s1);
            st4(p + C + 4 * cv, MODE == 0 ? s1 : s0);
        }
        __syncthreads();
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, double n, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, int clamp_eps,
                                   float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                   float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = (double)sums[c] / n;
    double var = (double)sums[C + c] / n - mu * mu;
    if (var < 0.0) var = 0.0;
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

// bn_finalize fed directly by the conv epilogue's per-tile partials: the fp64 column combine (column_sum_kernel's
// arithmetic, 8 channels x 32 row lanes per block, fixed order) and the coefficient math in one launch.
__global__ __launch_bounds__(256) void bn_finalize_partial_kernel(const float* __restrict__ partial, int nrows, double n, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                  float momentum, int clamp_eps, float* running_mean, float* running_var,
                                                                  float* mean, float* invstd, float* scale, float* shift) {
    __shared__ double red[2][32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        int r = ty;
        for (; r + 3 * 32 < nrows; r += 4 * 32) {          // 8 loads in flight; row order of the adds unchanged
            float v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v0[u] = partial[(size_t)(r + u * 32) * 2 * C + c];
                v1[u] = partial[(size_t)(r + u * 32) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += (double)v0[u]; a1 += (double)v1[u]; }
        }
        for (; r < nrows; r += 32) {
            a0 += (double)partial[(size_t)r * 2 * C + c];
            a1 += (double)partial[(size_t)r * 2 * C + C + c];
        }
    }
    red[0][ty][tx] = a0;
    red[1][ty][tx] = a1;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) { s0 += red[0][k][tx]; s1 += red[1][k][tx]; }
    const double sum = (double)(float)s0, sumsq = (double)(float)s1;      // the two-launch path rounds the sums to fp32 in between
    const double mu = sum / n;
    double var = sumsq / n - mu * mu;
    if (var < 0.0) var = 0.0;
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

__global__ void bn_eval_coeffs_kernel(const float* rm, const float* rv, const float* gamma, const float* beta, float eps,
                                      int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// ---- elementwise -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int y_pitch,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ res, int res_pitch, float* __restrict__ out,
                                                       int out_pitch, int relu, Slab g, unsigned* __restrict__ amax_out) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    float amax = 0.f;
    if (ty < g.RL) {
        for (int cv = tx; cv < g.CV; cv += g.cols) {
            const f32x4 sc = ld4(scale + 4 * cv), sh = ld4(shift + 4 * cv);
            f32x4 vy[kRowBatch], vr[kRowBatch];
            walk_rows(r_begin + ty, r_end, g.RL,
                [&](int u, long long r) {
                    vy[u] = ld4(y + r * y_pitch + 4 * cv);
                    if (res != nullptr) vr[u] = ld4(res + r * res_pitch + 4 * cv);
                },
                [&](int u, long long, bool valid) {
                    f32x4 v = vy[u] * sc + sh;
                    if (res != nullptr) v += vr[u];
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    vy[u] = v;
                    if (valid) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                },
                [&](int u, long long r) { st4(out + r * out_pitch + 4 * cv, vy[u]); });
        }
    }
    if (amax_out != nullptr) amax_commit(amax, amax_out);      // range of the output for the f16x3 conv that consumes it
}

__global__ __launch_bounds__(256, 3) void bn_bwd_apply_kernel(const float* __restrict__ dout, int dout_pitch,
                                                           const float* __restrict__ out, int out_pitch,
                                                           const float* __restrict__ y, int y_pitch,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           float inv_n, int C, int relu, float* __restrict__ dy, int dy_pitch,
                                                           float* __restrict__ g_out, int g_pitch, Slab g,
                                                           unsigned* __restrict__ amax_dy, const float* __restrict__ scale,
                                                           const float* __restrict__ shift) {
    const int tx = threadIdx.x % g.cols, ty = threadIdx.x / g.cols;
    const long long r_begin = (long long)blockIdx.x * g.rows_per_slab;
    long long r_end = r_begin + g.rows_per_slab;
    if (r_end > g.M) r_end = g.M;
    float amax = 0.f;
    if (ty < g.RL) {
        for (int cv = tx; cv < g.CV; cv += g.cols) {
            const f32x4 mu = ld4(mean + 4 * cv), is = ld4(invstd + 4 * cv);
            const f32x4 k = ld4(gamma + 4 * cv) * is;
            const f32x4 sgx = ld4(sums + 4 * cv) * inv_n, sg = ld4(sums + C + 4 * cv) * inv_n;
            const bool remask = relu && out == nullptr;
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
            if (remask) { sc = ld4(scale + 4 * cv); sh = ld4(shift + 4 * cv); }
            const bool use_out = relu && !remask;
            constexpr int NB = 4;            // 4 rows x 3 tensors in flight per wave; 136 registers = 3 waves per SIMD = the 3 blocks per CU of a 768-slab launch
            f32x4 vg[NB], vy[NB], vo[NB];
            walk_rows<NB>(r_begin + ty, r_end, g.RL,
                [&](int u, long long r) {
                    vg[u] = ld4(dout + r * dout_pitch + 4 * cv);
                    vy[u] = ld4(y + r * y_pitch + 4 * cv);
                    if (use_out) vo[u] = ld4(out + r * out_pitch + 4 * cv);
                },
                [&](int u, long long, bool valid) {
                    f32x4 gg = vg[u];
                    if (remask) gg = relu_mask(gg, vy[u] * sc + sh);
                    else if (relu) gg = relu_mask(gg, vo[u]);
                    const f32x4 xh = (vy[u] - mu) * is;
                    const f32x4 v = k * (gg - sg - xh * sgx);
                    vg[u] = gg;                                      // the masked gradient (residual branch) and dy, kept for the store pass
                    vy[u] = v;
                    if (valid) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                },
                [&](int u, long long r) {
                    if (g_out != nullptr) st4(g_out + r * g_pitch + 4 * cv, vg[u]);
                    st4(dy + r * dy_pitch + 4 * cv, vy[u]);
                });
        }
    }
    if (amax_dy != nullptr) amax_commit(amax, amax_dy);
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

// counter-based hash RNG: one 64-bit mix per float4 -> 4 x 16-bit uniform thresholds
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
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
                       workspace);
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

extern "C" int pylc_bn_finalize(const float* sums, double n, int C, const float* gamma, const float* beta, float eps, float momentum,
                                int clamp_eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                float* shift, void* stream) {
    PYLC_REQUIRE(sums && gamma && beta && mean && invstd && scale && shift && C > 0 && n > 0, "bn_finalize: bad arguments");
    PYLC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must both be set or both NULL");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), sums, n, C, gamma, beta, eps, momentum,
                       clamp_eps, running_mean, running_var, mean, invstd, scale, shift);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_finalize_from_partial(const float* partial, int n_rows, double n, int C, const float* gamma, const float* beta,
                                             float eps, float momentum, int clamp_eps, float* running_mean, float* running_var,
                                             float* mean, float* invstd, float* scale, float* shift, void* stream) {
    PYLC_REQUIRE(partial && n_rows > 0 && gamma && beta && mean && invstd && scale && shift && C > 0 && n > 0,
                 "bn_finalize_from_partial: bad arguments");
    PYLC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_from_partial: running stats must both be set or both NULL");
    hipLaunchKernelGGL(bn_finalize_partial_kernel, dim3(cdiv(C, 8)), dim3(256), 0, as_stream(stream), partial, n_rows, n, C, gamma, beta, eps,
                       momentum, clamp_eps, running_mean, running_var, mean, invstd, scale, shift);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_eval_coeffs(const float* rm, const float* rv, const float* gamma, const float* beta, float eps, int C, float* scale,
                                   float* shift, void* stream) {
    PYLC_REQUIRE(rm && rv && gamma && beta && scale && shift && C > 0, "bn_eval_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), rm, rv, gamma, beta, eps, C, scale, shift);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_apply(const float* y, int y_pitch, const float* scale, const float* shift, const float* residual, int res_pitch,
                             float* out, int out_pitch, long long M, int C, int relu, unsigned int* amax_out, void* stream) {
    if (int rc = check_mc(M, C, y_pitch, "bn_apply")) return rc;
    if (int rc = check_mc(M, C, out_pitch, "bn_apply(out)")) return rc;
    PYLC_REQUIRE(y && scale && shift && out, "bn_apply: null pointer");
    PYLC_REQUIRE(residual == nullptr || (res_pitch >= C && res_pitch % 4 == 0), "bn_apply: bad residual pitch");
    const Slab g = make_slab(M, C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(g.nslab), dim3(256), 0, as_stream(stream), y, y_pitch, scale, shift, residual, res_pitch, out,
                       out_pitch, relu, g, amax_out);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_bwd_reduce(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                                  const float* mean, const float* invstd, long long M, int C, int relu, float* sums, float* workspace,
                                  const float* scale, const float* shift, void* stream) {
    if (int rc = check_mc(M, C, dout_pitch, "bn_bwd_reduce")) return rc;
    if (int rc = check_mc(M, C, y_pitch, "bn_bwd_reduce(y)")) return rc;
    PYLC_REQUIRE(dout && y && mean && invstd && sums && workspace, "bn_bwd_reduce: null pointer");
    PYLC_REQUIRE(!relu || (out && out_pitch >= C && out_pitch % 4 == 0) || (!out && scale && shift),
                 "bn_bwd_reduce: relu needs `out`, or scale and shift to recompute the mask from y");
    const Slab g = make_slab(M, C);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(g.nslab), dim3(256), 0, st, dout, dout_pitch, out, out_pitch, y, y_pitch, mean, invstd,
                       relu, g, C, workspace, scale, shift);
    PYLC_LAUNCH_CHECK();
    hipLaunchKernelGGL(column_sum_kernel, dim3(cdiv(2 * C, 8)), dim3(256), 0, st, workspace, g.nslab, 2 * C, sums);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bn_bwd_apply(const float* dout, int dout_pitch, const float* out, int out_pitch, const float* y, int y_pitch,
                                 const float* mean, const float* invstd, const float* gamma, const float* sums, double n, long long M,
                                 int C, int relu, float* dy, int dy_pitch, float* g_out, int g_pitch, unsigned int* amax_dy,
                                 const float* scale, const float* shift, void* stream) {
    if (int rc = check_mc(M, C, dout_pitch, "bn_bwd_apply")) return rc;
    if (int rc = check_mc(M, C, dy_pitch, "bn_bwd_apply(dy)")) return rc;
    PYLC_REQUIRE(dout && y && mean && invstd && gamma && sums && dy && n > 0, "bn_bwd_apply: bad arguments");
    PYLC_REQUIRE(!relu || (out && out_pitch >= C) || (!out && scale && shift),
                 "bn_bwd_apply: relu needs `out`, or scale and shift to recompute the mask from y");
    PYLC_REQUIRE(g_out == nullptr || (g_pitch >= C && g_pitch % 4 == 0), "bn_bwd_apply: bad g pitch");
    const Slab g = make_slab(M, C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(g.nslab), dim3(256), 0, as_stream(stream), dout, dout_pitch, out, out_pitch, y, y_pitch,
                       mean, invstd, gamma, sums, (float)(1.0 / n), C, relu, dy, dy_pitch, g_out, g_pitch, g, amax_dy, scale, shift);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
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
