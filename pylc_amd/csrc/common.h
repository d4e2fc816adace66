// Shared host/device helpers for libpylc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/pylc_hip.h"

namespace pylc {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define PYLC_REQUIRE(cond, ...)                                     \
    do {                                                            \
        if (!(cond)) return ::pylc::fail(PYLC_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define PYLC_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return ::pylc::fail(PYLC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));  \
    } while (0)

#define PYLC_LAUNCH_CHECK() PYLC_HIP(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
__host__ __device__ inline T cdiv(T a, T b) { return (a + b - 1) / b; }

// Chunk-interleaved two-plane tensors (round 5; include/pylc_hip.h "fp16 planes"): `plane_stride == 32` says that flat element e of the
// dense [pixels][C] tensor (C % 32 == 0) lives at halves (e >> 5) * 64 + (e & 31) for plane 0 and + 32 for plane 1 -- the two planes of a
// 32-channel chunk (one K-step of the conv kernels) share ONE 128-byte line.  Any other stride: two separate plane arrays.
constexpr long long kPlanesIL = 32;
__host__ __device__ inline bool planes_il(long long plane_stride) { return plane_stride == kPlanesIL; }
__host__ __device__ inline long long planes_phys(long long e, bool il) { return il ? ((e >> 5) << 6) + (e & 31) : e; }
__host__ __device__ inline long long planes_p1(long long plane_stride) { return plane_stride; }      // (interleaved: 32 == the sentinel itself)

// THE rule (one place): a two-plane tensor of M pixels x C channels is chunk-interleaved iff its rows are dense, C is a multiple of 32 and
// both planes fit one 2 GiB buffer descriptor; everything that allocates, writes or reads such a tensor derives its plane stride from here
// (C ABI: pylc_planes_stride; pylc_set_planes_interleave(0) switches the format off process-wide -- A/B, env PYLC_NO_PLANE_INTERLEAVE).
extern int g_planes_interleave;
inline long long planes_stride_rule(long long M, int C, int pitch, int nplanes) {
    const bool il = nplanes == 2 && g_planes_interleave && pitch == C && C % 32 == 0 && M * C * 4 < (1ll << 31);
    return il ? kPlanesIL : M * pitch;
}

constexpr int kWave = 64;
constexpr int kNumCU = 256;

// Block-wide sum of `v` over 256 threads; result valid in thread 0. `red` needs >= 4 floats.
__device__ inline float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Fold this thread's running max |value| into the device scalar `out` (float bit pattern, compared as unsigned: valid
// for non-negative floats).  Max is order-independent, so the atomics keep the result deterministic; the plain read
// first skips the atomic once the stored bound is already large enough.
__device__ inline void amax_commit(float m, unsigned* out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) {
        const unsigned bits = __float_as_uint(m);
        if (bits > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, bits);
    }
}

// out[c] = sum_r partial[r][c] for r < nrows, accumulated in fp64 in a fixed order (deterministic).
// Launch with 256 threads and grid = cdiv(ncols, 8): 8 columns x 32 row lanes per block.
__global__ __launch_bounds__(256) static void column_sum_kernel(const float* __restrict__ partial, int nrows, int ncols,
                                                                 float* __restrict__ out) {
    __shared__ double red[32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx;
    double acc = 0.0;
    if (c < ncols) {
        // sixteen independent loads in flight per thread (the kernel is pure latency: ~1000 rows, a few dozen blocks); the adds
        // keep the row order, so the result does not depend on the unrolling
        for (int r = ty; r < nrows; r += 16 * 32) {       // the tail batch too: rows past the end load row 0 and add an exact 0
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(r + u * 32 < nrows ? r + u * 32 : 0) * ncols + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += r + u * 32 < nrows ? (double)v[u] : 0.0;
        }
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < ncols) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += red[k][tx];
        out[c] = (float)s;
    }
}

}  // namespace pylc
