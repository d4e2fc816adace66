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
