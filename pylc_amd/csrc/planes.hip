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
        const long long e = planes_phys(r * p_pitch + c, NPL == 2 && planes_il(plane_stride));      // (8 channels stay inside a 32-channel chunk)
        *reinterpret_cast<f16x8*>(planes + e) = h0;
        if constexpr (NPL == 2) *reinterpret_cast<f16x8*>(planes + plane_stride + e) = h1;
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
        const long long e = planes_phys(r * p_pitch + c, NPL == 2 && planes_il(plane_stride));
        const f16x8 h0 = *reinterpret_cast<const f16x8*>(planes + e);
        f16x8 h1 = {};
        if constexpr (NPL == 2) h1 = *reinterpret_cast<const f16x8*>(planes + plane_stride + e);
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

// U-Net up path (unet.py:135-152): torch.cat([upsample_x2(z), center_crop(bridge)], 1) written DIRECTLY as the planes tensor the next conv
// reads -- channels [0, C1) bilinear (align_corners) from z [B, h, w, C1], channels [C1, C1 + C2) the crop window of bridge -- instead of
// an fp32 concat buffer + a range pass + pylc_to_planes over it.  One thread: 8 channels of one output pixel (16-byte plane stores);
// row and batch from the grid.  `bound` >= max(max|z|, max|bridge|) (the interpolation is a convex combination).
struct UpLerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ UpLerp up_lerp(int dst, float scale, int in_size) {      // pool_resize.hip lerp_of: PyTorch's align_corners rule
    const float src = scale * (float)dst;
    UpLerp l;
    l.i0 = (int)src;
    if (l.i0 > in_size - 1) l.i0 = in_size - 1;
    l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

template <int NPL>
__global__ __launch_bounds__(256) void upcat_planes_kernel(const float* __restrict__ z, int z_pitch, int h, int w, const float* __restrict__ bridge,
                                                           int b_pitch, int HH, int WW, int h0, int w0, f16* __restrict__ planes,
                                                           long long plane_stride, int OH, int OW, int C1, int C2, float sh, float sw,
                                                           const unsigned* __restrict__ bound) {
    const int C8 = (C1 + C2) / 8;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)(OW * C8)) return;
    const int c = (int)(i % (unsigned)C8) * 8, ow = (int)(i / (unsigned)C8);
    const int oh = blockIdx.y, b = blockIdx.z;
    const float s = pow2_scale_for(*bound);
    f32x4 lo, hi;
    if (c < C1) {
        const UpLerp lh = up_lerp(oh, sh, h), lw = up_lerp(ow, sw, w);
        const float* base = z + (size_t)b * h * w * z_pitch + c;
        const float* p00 = base + ((size_t)lh.i0 * w + lw.i0) * z_pitch;
        const float* p01 = base + ((size_t)lh.i0 * w + lw.i1) * z_pitch;
        const float* p10 = base + ((size_t)lh.i1 * w + lw.i0) * z_pitch;
        const float* p11 = base + ((size_t)lh.i1 * w + lw.i1) * z_pitch;
        // the expression of bilinear_fwd_kernel (pool_resize.hip), so that both paths give the same bits
        lo = lh.w0 * (lw.w0 * *reinterpret_cast<const f32x4*>(p00) + lw.w1 * *reinterpret_cast<const f32x4*>(p01)) +
             lh.w1 * (lw.w0 * *reinterpret_cast<const f32x4*>(p10) + lw.w1 * *reinterpret_cast<const f32x4*>(p11));
        hi = lh.w0 * (lw.w0 * *reinterpret_cast<const f32x4*>(p00 + 4) + lw.w1 * *reinterpret_cast<const f32x4*>(p01 + 4)) +
             lh.w1 * (lw.w0 * *reinterpret_cast<const f32x4*>(p10 + 4) + lw.w1 * *reinterpret_cast<const f32x4*>(p11 + 4));
    } else {
        const float* src = bridge + ((size_t)(b * HH + h0 + oh) * WW + w0 + ow) * b_pitch + (c - C1);
        lo = *reinterpret_cast<const f32x4*>(src);
        hi = *reinterpret_cast<const f32x4*>(src + 4);
    }
    f16x8 q0, q1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f16 a0, a1, b0, b1;
        psplit(lo[e], s, a0, a1);
        psplit(hi[e], s, b0, b1);
        q0[e] = a0; q1[e] = a1; q0[4 + e] = b0; q1[4 + e] = b1;
    }
    f16* dst = planes + planes_phys((long long)((size_t)(b * OH + oh) * OW + ow) * (C1 + C2) + c, NPL == 2 && planes_il(plane_stride));
    *reinterpret_cast<f16x8*>(dst) = q0;
    if constexpr (NPL == 2) *reinterpret_cast<f16x8*>(dst + plane_stride) = q1;
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
                const long long e = planes_phys(r * p_pitch + 8 * c8, NPL == 2 && planes_il(plane_stride));
                const f16x8 h0 = *reinterpret_cast<const f16x8*>(planes + e);
                f16x8 h1 = {};
                if constexpr (NPL == 2) h1 = *reinterpret_cast<const f16x8*>(planes + plane_stride + e);
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
    // two planes: separate arrays `plane_stride` halves apart, or chunk-interleaved (plane_stride == 32: dense rows of a multiple of 32 channels)
    PYLC_REQUIRE(nplanes == 1 || (plane_stride >= M * p_pitch - (p_pitch - C) && plane_stride % 8 == 0) ||
                     (planes_il(plane_stride) && p_pitch == C && C % 32 == 0),
                 "%s: bad plane stride (32 = chunk-interleaved needs a dense pitch and C %% 32 == 0)", what);
    PYLC_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "%s: planes must be 16-byte aligned", what);
    return PYLC_OK;
}

}  // namespace pylc

using namespace pylc;

int pylc::g_planes_interleave = 1;
extern "C" int pylc_set_planes_interleave(int on) { pylc::g_planes_interleave = on ? 1 : 0; return PYLC_OK; }
extern "C" long long pylc_planes_stride(long long M, int C, int nplanes) { return pylc::planes_stride_rule(M, C, C, nplanes); }

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

extern "C" int pylc_upsample2_crop_concat_planes(const float* z, int z_pitch, int B, int h, int w, int C1, const float* bridge, int bridge_pitch, int HH,
                                                 int WW, int C2, void* planes, long long plane_stride, int nplanes, const unsigned int* bound,
                                                 void* stream) {
    const int OH = 2 * h, OW = 2 * w;
    PYLC_REQUIRE(z && bridge && planes && bound && B > 0 && h > 0 && w > 0 && C1 > 0 && C2 > 0 && C1 % 8 == 0 && C2 % 8 == 0 && (nplanes == 1 || nplanes == 2),
                 "upsample2_crop_concat_planes: bad arguments (channel counts must be multiples of 8)");
    PYLC_REQUIRE(z_pitch >= C1 && bridge_pitch >= C2 && z_pitch % 4 == 0 && bridge_pitch % 4 == 0 && HH >= OH && WW >= OW && B <= 65535 && OH <= 65535 &&
                     (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && plane_stride % 8 == 0,
                 "upsample2_crop_concat_planes: bad pitch / the bridge is smaller than the up-sampled tensor / unaligned planes");
    const int h0 = (HH - OH) / 2, w0 = (WW - OW) / 2;
    const float sh = OH > 1 ? (float)(h - 1) / (float)(OH - 1) : 0.f, sw = OW > 1 ? (float)(w - 1) / (float)(OW - 1) : 0.f;
    const dim3 grid(cdiv(OW * ((C1 + C2) / 8), 256), OH, B);
    if (nplanes == 2)
        hipLaunchKernelGGL(upcat_planes_kernel<2>, grid, dim3(256), 0, as_stream(stream), z, z_pitch, h, w, bridge, bridge_pitch, HH, WW, h0, w0,
                           static_cast<f16*>(planes), plane_stride, OH, OW, C1, C2, sh, sw, bound);
    else
        hipLaunchKernelGGL(upcat_planes_kernel<1>, grid, dim3(256), 0, as_stream(stream), z, z_pitch, h, w, bridge, bridge_pitch, HH, WW, h0, w0,
                           static_cast<f16*>(planes), plane_stride, OH, OW, C1, C2, sh, sw, bound);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
