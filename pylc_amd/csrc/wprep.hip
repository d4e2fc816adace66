// Filter preparation for the f16x3 conv arithmetic: once per optimiser step every conv filter is scaled by its power-of-two
// range factor and split into two fp16 planes (h0 = rn16(s*w), h1 = rn16(2048 (s*w - h0))), in the forward layout
// [Cout][R*S][Cin] and in the dgrad layout [Cin][R*S][Kp] (Kp = roundup4(Cout), zero padded).  The conv kernels then copy
// filter tiles to LDS without touching the vector ALU, and the backward needs no per-layer transpose launch.
#include "common.h"

namespace pylc {

typedef _Float16 f16;

__device__ __forceinline__ float wprep_scale(unsigned amax_bits) {      // = conv_igemm.hip pow2_scale_for
    int e = (int)((amax_bits >> 23) & 0xFFu);
    int se = 127 + 14 - (e - 127);
    se = se < 1 ? 1 : (se > 253 ? 253 : se);
    return __uint_as_float((unsigned)se << 23);
}

__device__ __forceinline__ void wsplit(float w, float s, f16& h0, f16& h1) {
    const float xs = w * s;
    h0 = (f16)xs;
    h1 = (f16)((xs - (float)h0) * 2048.f);
}

// One block per 32(k) x 32(c) tile of one (filter, tap): read coalesced along c, write the forward planes in the same order
// and the dgrad planes transposed through LDS (coalesced along k).  Blocks find their filter by bisection over tile_begin.
// `il`: chunk-interleaved layout (include/pylc_hip.h PylcConvDesc.w_planes_fmt) for the layouts whose channel count is a multiple of 32:
// flat element e -> halves (e >> 5) * 64 + (e & 31), plane 1 at + 32 -- both planes of a K-step chunk in one 128-byte line.
__device__ __forceinline__ long long il_phys(long long e) { return ((e >> 5) << 6) + (e & 31); }

__global__ __launch_bounds__(256) void weight_prepare_kernel(const float* __restrict__ base, const PylcWPrepEntry* __restrict__ table,
                                                             int count, const unsigned* __restrict__ amax, f16* __restrict__ planes, int il) {
    __shared__ f16 t0s[32][34], t1s[32][34];
    int lo = 0, hi = count - 1;
    const long long tile = blockIdx.x;
    while (lo < hi) {                                 // last entry with tile_begin <= tile
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile_begin <= tile) lo = mid; else hi = mid - 1;
    }
    const PylcWPrepEntry e = table[lo];
    const float s = wprep_scale(amax[e.amax_index]);
    const int Kp = (e.K + 3) & ~3;
    const int tk = (e.K + 31) / 32, tc = (e.C + 31) / 32;
    int id = (int)(tile - e.tile_begin);
    const int ct = id % tc; id /= tc;
    const int kt = id % tk; id /= tk;
    const int rs = id;
    const int k0 = kt * 32, c0 = ct * 32;
    const float* w = base + e.src_offset;
    const long long n = (long long)e.K * e.RS * e.C, nt = (long long)e.C * e.RS * Kp;
    f16* f0 = planes + e.fwd_offset;                 // [2][K][RS][C]
    f16* t0 = planes + e.t_offset;                   // [2][C][RS][Kp]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const bool il_f = il && e.C % 32 == 0, il_t = il && Kp % 32 == 0;
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        f16 h0 = (f16)0.f, h1 = (f16)0.f;
        if (k < e.K && c < e.C) {
            const long long src = ((long long)k * e.RS + rs) * e.C + c;
            wsplit(w[src], s, h0, h1);
            if (il_f) {
                f0[il_phys(src)] = h0;
                f0[il_phys(src) + 32] = h1;
            } else {
                f0[src] = h0;
                f0[n + src] = h1;
            }
        }
        t0s[i][tx] = h0;                              // zeros outside the filter: they become the Kp padding
        t1s[i][tx] = h1;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (c < e.C && k < Kp) {
            const long long dst = ((long long)c * e.RS + rs) * Kp + k;
            if (il_t) {
                t0[il_phys(dst)] = t0s[tx][i];
                t0[il_phys(dst) + 32] = t1s[tx][i];
            } else {
                t0[dst] = t0s[tx][i];
                t0[nt + dst] = t1s[tx][i];
            }
        }
    }
}

// A per-input-channel affine in FRONT of a 1x1 conv folded into the conv (inference: the eval-mode BatchNorm between the depthwise and
// the pointwise conv of a separable conv, xception.py:34-39):  W (s (.) x + t) = (W diag s) x + W t.  One block per output channel:
// w_out[n][k] = w[n][k] * scale[k], bias_out[n] = bias_in[n] + sum_k w[n][k] * shift[k] (fp64 sum, fixed order); amax_out receives
// max |w_out| as float bits (the range scalar of the f16x3 arithmetic).
__global__ __launch_bounds__(256) void fold_affine_kernel(const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ bias_in, int Cin, float* __restrict__ w_out,
                                                          float* __restrict__ bias_out, unsigned* __restrict__ amax_out) {
    __shared__ double red[256];
    __shared__ float redm[256];
    const int n = blockIdx.x;
    double acc = 0.0;
    float m = 0.f;
    for (int k = threadIdx.x; k < Cin; k += 256) {
        const float v = w[(size_t)n * Cin + k];
        const float o = v * scale[k];
        w_out[(size_t)n * Cin + k] = o;
        acc += (double)v * (double)shift[k];
        m = fmaxf(m, fabsf(o));
    }
    red[threadIdx.x] = acc;
    redm[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        float mm = 0.f;
        for (int i = 0; i < 256; ++i) { sum += red[i]; mm = fmaxf(mm, redm[i]); }
        bias_out[n] = (float)(sum + (bias_in != nullptr ? (double)bias_in[n] : 0.0));
        if (amax_out != nullptr) atomicMax(amax_out, __float_as_uint(mm));
    }
}

}  // namespace pylc

using namespace pylc;

extern "C" int pylc_conv1x1_fold_input_affine(const float* w, const float* scale, const float* shift, const float* bias_in, int Cout, int Cin,
                                              float* w_out, float* bias_out, unsigned int* amax_out, void* stream) {
    PYLC_REQUIRE(w && scale && shift && w_out && bias_out && Cout > 0 && Cin > 0, "conv1x1_fold_input_affine: bad arguments");
    if (amax_out != nullptr) PYLC_HIP(hipMemsetAsync(amax_out, 0, sizeof(unsigned), as_stream(stream)));
    hipLaunchKernelGGL(fold_affine_kernel, dim3(Cout), dim3(256), 0, as_stream(stream), w, scale, shift, bias_in, Cin, w_out, bias_out, amax_out);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_weight_prepare(const float* base, const PylcWPrepEntry* table, int count, long long total_tiles,
                                   const unsigned int* amax, void* planes, int interleave, void* stream) {
    PYLC_REQUIRE(base && table && amax && planes && count > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "weight_prepare: bad arguments");
    hipLaunchKernelGGL(weight_prepare_kernel, dim3((unsigned)total_tiles), dim3(256), 0, as_stream(stream), base, table, count, amax,
                       static_cast<f16*>(planes), interleave);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
