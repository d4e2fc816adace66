// fp16-plane tensor format: conversions to and from fp32 (HBM-bound, one pass each).
//
// A "planes" tensor holds M pixels x C channels (channel pitch P, C % 8 == 0, P % 8 == 0) as two fp16 planes of M x P halves:
//      plane 0: h0 = rn16(s x)          plane 1: h1 = rn16(2^11 (s x - h0))          s = pow2_scale_for(bound), bound >= max|x|
// i.e. exactly the two pieces the f16x3 conv arithmetic forms from an fp32 operand (conv_common.h split2 / wprep.hip wsplit),
// stored once by the producer instead of re-derived by every consumer; 4 bytes per element like fp32.  s x = h0 + 2^-11 h1 to
// 2^-23 relative.  Precision mode 3 (plain fp16 operands) writes and reads plane 0 only.
//
// The hot producers write planes directly (bn.hip); these two kernels serve the other producers / consumers (concat buffers,
// pooled or interpolated tensors feeding a conv; reductions that want fp32).
#include "conv_common.h"

namespace pylc {

typedef _Float16 f16;

__device__ __forceinline__ void psplit(float x, float s, f16& h0, f16& h1) {      // == wprep.hip wsplit == conv_common.h split2
    const float xs = x * s;
    h0 = (f16)xs;
    h1 = (f16)((xs - (float)h0) * 2048.f);
}

template <int NPL>
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ x, int x_pitch, f16* __restrict__ planes, int p_pitch,
                                                        long long plane_stride, long long M, int C8, const unsigned* __restrict__ amax) {
    const float s = pow2_scale_for(*amax);
    const long long total = M * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C8;
        const int c = (int)(i - r * C8) * 8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(x + r * x_pitch + c);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(x + r * x_pitch + c + 4);
        f16x8 h0, h1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f16 a0, a1, b0, b1;
            psplit(lo[e], s, a0, a1);
            psplit(hi[e], s, b0, b1);
            h0[e] = a0; h1[e] = a1; h0[4 + e] = b0; h1[4 + e] = b1;
        }
        *reinterpret_cast<f16x8*>(planes + r * p_pitch + c) = h0;
        if constexpr (NPL == 2) *reinterpret_cast<f16x8*>(planes + plane_stride + r * p_pitch + c) = h1;
    }
}

template <int NPL>
__global__ __launch_bounds__(256) void from_planes_kernel(const f16* __restrict__ planes, int p_pitch, long long plane_stride,
                                                          float* __restrict__ x, int x_pitch, long long M, int C8,
                                                          const unsigned* __restrict__ amax) {
    const float inv = 1.f / pow2_scale_for(*amax);
    const long long total = M * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C8;
        const int c = (int)(i - r * C8) * 8;
        const f16x8 h0 = *reinterpret_cast<const f16x8*>(planes + r * p_pitch + c);
        f16x8 h1 = {};
        if constexpr (NPL == 2) h1 = *reinterpret_cast<const f16x8*>(planes + plane_stride + r * p_pitch + c);
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = ((float)h0[e] + (float)h1[e] * (1.f / 2048.f)) * inv;
            hi[e] = ((float)h0[4 + e] + (float)h1[4 + e] * (1.f / 2048.f)) * inv;
        }
        *reinterpret_cast<f32x4*>(x + r * x_pitch + c) = lo;
        *reinterpret_cast<f32x4*>(x + r * x_pitch + c + 4) = hi;
    }
}

// Per-channel sum over the pixels of a planes tensor (a conv's bias gradient when its dy arrives as planes): partial[block][C] in fp32
// per row slab, combined in fp64 / fixed order by column_sum_kernel (common.h) -- the arithmetic of pylc_bn_stats's first half.
constexpr int kColsumSlabs = 768;
template <int NPL>
__global__ __launch_bounds__(256) void planes_colsum_kernel(const f16* __restrict__ planes, int p_pitch, long long plane_stride, long long M, int C8,
                                                            int cols, int RL, long long rows_per_slab, const unsigned* __restrict__ amax,
                                                            float* __restrict__ partial) {
    __shared__ float red[256][9];
    const float inv = 1.f / pow2_scale_for(*amax);
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    const long long r_begin = (long long)blockIdx.x * rows_per_slab;
    long long r_end = r_begin + rows_per_slab;
    if (r_end > M) r_end = M;
    for (int cb = 0; cb < C8; cb += cols) {
        const int c8 = cb + tx;
        const bool active = ty < RL && c8 < C8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (active) {
            for (long long r = r_begin + ty; r < r_end; r += RL) {
                const f16x8 h0 = *reinterpret_cast<const f16x8*>(planes + r * p_pitch + 8 * c8);
                f16x8 h1 = {};
                if constexpr (NPL == 2) h1 = *reinterpret_cast<const f16x8*>(planes + plane_stride + r * p_pitch + 8 * c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ((float)h0[e] + (float)h1[e] * (1.f / 2048.f)) * inv;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
        __syncthreads();
        if (ty == 0 && c8 < C8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sum = acc[e];
                for (int q = 1; q < RL; ++q) sum += red[q * cols + tx][e];
                partial[(size_t)blockIdx.x * (8 * C8) + 8 * c8 + e] = sum;
            }
        }
        __syncthreads();
    }
}

static int check_planes(const void* planes, int p_pitch, long long plane_stride, long long M, int C, int nplanes, const char* what) {
    PYLC_REQUIRE(planes && M > 0 && C > 0 && C % 8 == 0, "%s: need M > 0 and C %% 8 == 0 (M=%lld C=%d)", what, M, C);
    PYLC_REQUIRE(p_pitch >= C && p_pitch % 8 == 0, "%s: plane pitch %d invalid for C=%d (multiple of 8)", what, p_pitch, C);
    PYLC_REQUIRE(nplanes == 1 || nplanes == 2, "%s: nplanes must be 1 or 2", what);
    PYLC_REQUIRE(nplanes == 1 || (plane_stride >= M * p_pitch - (p_pitch - C) && plane_stride % 8 == 0), "%s: bad plane stride", what);
    PYLC_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "%s: planes must be 16-byte aligned", what);
    return PYLC_OK;
}

}  // namespace pylc

using namespace pylc;

extern "C" int pylc_to_planes(const float* x, int x_pitch, void* planes, int p_pitch, long long plane_stride, long long M, int C,
                              const unsigned int* amax, int nplanes, void* stream) {
    if (int rc = check_planes(planes, p_pitch, plane_stride, M, C, nplanes, "to_planes")) return rc;
    PYLC_REQUIRE(x && amax && x_pitch >= C && x_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "to_planes: bad fp32 source");
    const long long total = M * (C / 8);
    const int blocks = (int)(cdiv<long long>(total, 256 * 4) < 4096 ? cdiv<long long>(total, 256 * 4) : 4096);
    const dim3 g(blocks > 0 ? blocks : 1), b(256);
    if (nplanes == 2) hipLaunchKernelGGL((to_planes_kernel<2>), g, b, 0, as_stream(stream), x, x_pitch, static_cast<f16*>(planes), p_pitch, plane_stride, M, C / 8, amax);
    else hipLaunchKernelGGL((to_planes_kernel<1>), g, b, 0, as_stream(stream), x, x_pitch, static_cast<f16*>(planes), p_pitch, plane_stride, M, C / 8, amax);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_from_planes(const void* planes, int p_pitch, long long plane_stride, float* x, int x_pitch, long long M, int C,
                                const unsigned int* amax, int nplanes, void* stream) {
    if (int rc = check_planes(planes, p_pitch, plane_stride, M, C, nplanes, "from_planes")) return rc;
    PYLC_REQUIRE(x && amax && x_pitch >= C && x_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "from_planes: bad fp32 destination");
    const long long total = M * (C / 8);
    const int blocks = (int)(cdiv<long long>(total, 256 * 4) < 4096 ? cdiv<long long>(total, 256 * 4) : 4096);
    const dim3 g(blocks > 0 ? blocks : 1), b(256);
    if (nplanes == 2) hipLaunchKernelGGL((from_planes_kernel<2>), g, b, 0, as_stream(stream), static_cast<const f16*>(planes), p_pitch, plane_stride, x, x_pitch, M, C / 8, amax);
    else hipLaunchKernelGGL((from_planes_kernel<1>), g, b, 0, as_stream(stream), static_cast<const f16*>(planes), p_pitch, plane_stride, x, x_pitch, M, C / 8, amax);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" size_t pylc_planes_colsum_workspace_floats(int C) { return (size_t)kColsumSlabs * (size_t)C; }

extern "C" int pylc_planes_colsum(const void* planes, int p_pitch, long long plane_stride, int nplanes, const unsigned int* amax, long long M, int C,
                                  float* sums, float* workspace, void* stream) {
    if (int rc = check_planes(planes, p_pitch, plane_stride, M, C, nplanes, "planes_colsum")) return rc;
    PYLC_REQUIRE(amax && sums && workspace, "planes_colsum: null pointer");
    const int C8 = C / 8;
    const int cols = C8 < 256 ? C8 : 256;
    const int RL = 256 / cols;
    long long rps = cdiv<long long>(M, kColsumSlabs);
    if (rps < (long long)RL * 8) rps = (long long)RL * 8;
    rps = cdiv<long long>(rps, RL) * RL;
    const int nslab = (int)cdiv<long long>(M, rps);
    hipStream_t st = as_stream(stream);
    const f16* p = static_cast<const f16*>(planes);
    if (nplanes == 2)
        hipLaunchKernelGGL(planes_colsum_kernel<2>, dim3(nslab), dim3(256), 0, st, p, p_pitch, plane_stride, M, C8, cols, RL, rps, amax, workspace);
    else
        hipLaunchKernelGGL(planes_colsum_kernel<1>, dim3(nslab), dim3(256), 0, st, p, p_pitch, plane_stride, M, C8, cols, RL, rps, amax, workspace);
    PYLC_LAUNCH_CHECK();
    hipLaunchKernelGGL(column_sum_kernel, dim3(cdiv(C, 8)), dim3(256), 0, st, workspace, nslab, C, sums);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
