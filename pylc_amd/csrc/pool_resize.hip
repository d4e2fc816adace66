// Pooling, bilinear resize (align_corners=True), global average pool and image ingest, NHWC fp32.
// All HBM-bound; every thread moves 16-byte channel vectors.  Backward passes are written as gathers
// (each input element sums the outputs that reference it) so results are deterministic -- no float atomics.
//
// Replaces nn.MaxPool2d(3,2,1) resnet.py:76; F.max_pool2d(x,2) unet.py:98; F.interpolate(bilinear,
// align_corners=True) deeplab.py:38, decoder.py:46, aspp.py:79; nn.Upsample unet.py:136;
// nn.AdaptiveAvgPool2d(1) aspp.py:63; Model.normalize_image model.py:416-445 (+ the x3 stack :310-311).
#include "common.h"

namespace pylc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

static inline int grid_for(long long n, int cap = 8192) {
    long long b = cdiv<long long>(n, 256);
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

// ---- max pool ----------------------------------------------------------------------------------
// planes != NULL: the pooled tensor leaves as fp16 planes ([nplanes][pixels][C] halves: plane 0 = rn16(sc v), plane 1 = rn16(2^11 (sc v -
// plane 0)), sc from `bound` -- planes.hip) for a conv that copies its operand tiles, instead of fp32 + a conversion pass.
struct PoolPlanes { _Float16* planes; long long plane_stride; const unsigned* bound; int nplanes; };
__device__ __forceinline__ float pool_pow2_scale(unsigned bits) {      // conv_common.h pow2_scale_for: bound -> [2^14, 2^15)
    int e = (int)((bits >> 23) & 0xFFu);
    int se = 127 + 14 - (e - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}

__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int B, int H, int W,
                                   int C, int k, int s, int pad, int OH, int OW, PoolPlanes pp) {
    const int CV = C / 4;
    const long long total = (long long)B * OH * OW * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int b = (int)(t / OH);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int kh = 0; kh < k; ++kh) {
            const int h = oh * s - pad + kh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int w = ow * s - pad + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                const f32x4 v = ld4(x + ((size_t)(b * H + h) * W + w) * C + 4 * cv);
                const int code = kh * k + kw;
                // first maximum in scan order wins (PyTorch: val > maxval); the first valid element seeds it
                if (first || v.x > best.x) { best.x = v.x; bi[0] = code; }
                if (first || v.y > best.y) { best.y = v.y; bi[1] = code; }
                if (first || v.z > best.z) { best.z = v.z; bi[2] = code; }
                if (first || v.w > best.w) { best.w = v.w; bi[3] = code; }
                first = false;
            }
        }
        if (pp.planes != nullptr) {
            typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
            const float sc = pool_pow2_scale(*pp.bound);
            const f32x4 xs = best * sc;
            const f16x4_ h0 = {(_Float16)xs.x, (_Float16)xs.y, (_Float16)xs.z, (_Float16)xs.w};
            const long long pe = planes_phys(4 * i, pp.nplanes == 2 && planes_il(pp.plane_stride));      // (chunk-interleaved: common.h)
            *reinterpret_cast<uint2*>(pp.planes + pe) = __builtin_bit_cast(uint2, h0);
            if (pp.nplanes == 2) {
                const f32x4 r = {(xs.x - (float)h0.x) * 2048.f, (xs.y - (float)h0.y) * 2048.f, (xs.z - (float)h0.z) * 2048.f, (xs.w - (float)h0.w) * 2048.f};
                const f16x4_ h1 = {(_Float16)r.x, (_Float16)r.y, (_Float16)r.z, (_Float16)r.w};
                *reinterpret_cast<uint2*>(pp.planes + pp.plane_stride + pe) = __builtin_bit_cast(uint2, h1);
            }
        } else {
            st4(y + 4 * i, best);
        }
        if (idx != nullptr) {
            uchar4 c4 = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
            *reinterpret_cast<uchar4*>(idx + 4 * i) = c4;
        }
    }
}

// `add` (may be NULL): a second gradient of the pooled tensor that covers only the window [h0, h0+AH) x [w0, w0+AW) -- the
// centre crop a U-Net skip connection reads (unet.py:145-148) -- summed here instead of in a zero-padded full-size pass.
struct PoolAdd { const float* p; int pitch, h0, w0, AH, AW; };

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx, int B, int H,
                                   int W, int C, int k, int s, int pad, int OH, int OW, PoolAdd add) {
    const int CV = C / 4;
    const long long total = (long long)B * H * W * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // windows covering (h,w): oh*s - pad <= h <= oh*s - pad + k - 1
        int oh_lo = (h + pad - k + 1 + s - 1); oh_lo = oh_lo < 0 ? 0 : oh_lo / s;
        int oh_hi = (h + pad) / s; if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = (w + pad - k + 1 + s - 1); ow_lo = ow_lo < 0 ? 0 : ow_lo / s;
        int ow_hi = (w + pad) / s; if (ow_hi > OW - 1) ow_hi = OW - 1;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const int kh = h - (oh * s - pad);
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int code = kh * k + (w - (ow * s - pad));
                const size_t o = ((size_t)(b * OH + oh) * OW + ow) * C + 4 * cv;
                const uchar4 c4 = *reinterpret_cast<const uchar4*>(idx + o);
                const f32x4 g = ld4(dy + o);
                if (c4.x == code) acc.x += g.x;
                if (c4.y == code) acc.y += g.y;
                if (c4.z == code) acc.z += g.z;
                if (c4.w == code) acc.w += g.w;
            }
        }
        if (add.p != nullptr && (unsigned)(h - add.h0) < (unsigned)add.AH && (unsigned)(w - add.w0) < (unsigned)add.AW)
            acc += ld4(add.p + ((size_t)(b * add.AH + (h - add.h0)) * add.AW + (w - add.w0)) * add.pitch + 4 * cv);
        st4(dx + 4 * i, acc);
    }
}

// 2 x 2 / stride 2 / no padding (the U-Net's four pools, unet.py:98): windows do not overlap, so a thread takes one WINDOW and one channel
// vector -- reads its dy vector and argmax codes once and writes the window's four dx vectors -- instead of one input pixel per thread
// gathering from the (single) window that covers it: a quarter of the index arithmetic and of the dy / idx requests, and the row /
// batch come from the grid instead of 64-bit divisions (the general kernel ran at 1.5 TB/s on the 1.06 GB dx of the first pool).
// The last row / column of an odd input lies in no window: dx = 0 (+ add).  grid = (ceil(PW * CV / 256), PH, B), PH = ceil(H / 2).
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx,
                                                          int H, int W, int C, int OH, int OW, int PW, PoolAdd add) {
    const int CV = C / 4;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)(PW * CV)) return;
    const int cv = (int)(i % (unsigned)CV), pw = (int)(i / (unsigned)CV);
    const int ph = blockIdx.y, b = blockIdx.z;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    uchar4 c4 = make_uchar4(255, 255, 255, 255);
    if (ph < OH && pw < OW) {
        const size_t o = ((size_t)(b * OH + ph) * OW + pw) * C + 4 * cv;
        g = ld4(dy + o);
        c4 = *reinterpret_cast<const uchar4*>(idx + o);
    }
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int kw = 0; kw < 2; ++kw) {
            const int h = 2 * ph + kh, w = 2 * pw + kw;
            if (h >= H || w >= W) continue;
            const int code = kh * 2 + kw;
            f32x4 acc = {c4.x == code ? g.x : 0.f, c4.y == code ? g.y : 0.f, c4.z == code ? g.z : 0.f, c4.w == code ? g.w : 0.f};
            if (add.p != nullptr && (unsigned)(h - add.h0) < (unsigned)add.AH && (unsigned)(w - add.w0) < (unsigned)add.AW)
                acc += ld4(add.p + ((size_t)(b * add.AH + (h - add.h0)) * add.AW + (w - add.w0)) * add.pitch + 4 * cv);
            st4(dx + ((size_t)(b * H + h) * W + w) * C + 4 * cv, acc);
        }
}

static bool launch_maxpool2_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int k, int stride, int pad, int OH,
                                int OW, PoolAdd add, hipStream_t st) {
    if (!(k == 2 && stride == 2 && pad == 0 && OH == (H - 2) / 2 + 1 && OW == (W - 2) / 2 + 1 && B <= 65535 && (H + 1) / 2 <= 65535)) return false;
    const int PH = (H + 1) / 2, PW = (W + 1) / 2;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(cdiv(PW * (C / 4), 256), PH, B), dim3(256), 0, st, dy, idx, dx, H, W, C, OH, OW, PW, add);
    return true;
}

// dst[b, y, x, 0:C] = src[b, h0 + y, w0 + x, 0:C] (both NHWC with their own channel pitch)
__global__ void crop_copy_kernel(const float* __restrict__ src, int src_pitch, int H, int W, int h0, int w0, float* __restrict__ dst,
                                 int dst_pitch, int B, int TH, int TW, int C) {
    const int CV = C / 4;
    const long long total = (long long)B * TH * TW * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int x = (int)(t % TW); t /= TW;
        const int y = (int)(t % TH);
        const int b = (int)(t / TH);
        st4(dst + ((size_t)(b * TH + y) * TW + x) * dst_pitch + 4 * cv,
            ld4(src + ((size_t)(b * H + h0 + y) * W + w0 + x) * src_pitch + 4 * cv));
    }
}

// ---- bilinear, align_corners=True --------------------------------------------------------------
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_size) {
    // PyTorch: src = scale * dst (scale = (in-1)/(out-1), 0 if out == 1); i0 = floor(src); i1 = i0 + (i0 < in-1)
    const float src = scale * (float)dst;
    Lerp l;
    l.i0 = (int)src;
    if (l.i0 > in_size - 1) l.i0 = in_size - 1;
    l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ x, int x_pitch, float* __restrict__ y, int y_pitch, int B, int H, int W, int C,
                                    int OH, int OW, float sh, float sw) {
    const int CV = C / 4;
    const long long total = (long long)B * OH * OW * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int b = (int)(t / OH);
        const Lerp lh = lerp_of(oh, sh, H), lw = lerp_of(ow, sw, W);
        const float* base = x + (size_t)b * H * W * x_pitch + 4 * cv;
        const f32x4 v00 = ld4(base + ((size_t)lh.i0 * W + lw.i0) * x_pitch), v01 = ld4(base + ((size_t)lh.i0 * W + lw.i1) * x_pitch);
        const f32x4 v10 = ld4(base + ((size_t)lh.i1 * W + lw.i0) * x_pitch), v11 = ld4(base + ((size_t)lh.i1 * W + lw.i1) * x_pitch);
        const f32x4 r = lh.w0 * (lw.w0 * v00 + lw.w1 * v01) + lh.w1 * (lw.w0 * v10 + lw.w1 * v11);
        st4(y + ((size_t)(b * OH + oh) * OW + ow) * y_pitch + 4 * cv, r);
    }
}

// The same interpolation with the output row in the GRID (blockIdx.y = oh, blockIdx.z = b): the flat form above spends three 64-bit divisions
// per 16 bytes written on its index; here a thread pays one 32-bit division and the row's vertical weights are block-uniform: 107 -> 100 us
// per launch on the step's three up-samplings (0.94 GB written; the same form of the separable gradient was neutral and is not kept).  Same
// expression per element: bit-identical.  (launcher: B, OH <= 65535)
__global__ __launch_bounds__(256) void bilinear_fwd_rows_kernel(const float* __restrict__ x, int x_pitch, float* __restrict__ y, int y_pitch, int H, int W,
                                                                  int C, int OH, int OW, float sh, float sw) {
    const unsigned CV = (unsigned)C / 4;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)OW * CV) return;
    const unsigned ow = idx / CV, cv = idx - ow * CV;
    const int oh = blockIdx.y, b = blockIdx.z;
    const Lerp lh = lerp_of(oh, sh, H), lw = lerp_of((int)ow, sw, W);
    const float* base = x + (size_t)b * H * W * x_pitch + 4 * cv;
    const f32x4 v00 = ld4(base + ((size_t)lh.i0 * W + lw.i0) * x_pitch), v01 = ld4(base + ((size_t)lh.i0 * W + lw.i1) * x_pitch);
    const f32x4 v10 = ld4(base + ((size_t)lh.i1 * W + lw.i0) * x_pitch), v11 = ld4(base + ((size_t)lh.i1 * W + lw.i1) * x_pitch);
    const f32x4 r = lh.w0 * (lw.w0 * v00 + lw.w1 * v01) + lh.w1 * (lw.w0 * v10 + lw.w1 * v11);
    st4(y + ((size_t)(b * OH + oh) * OW + ow) * y_pitch + 4 * cv, r);
}

__device__ __forceinline__ void cand_range(int i, float scale, int out_size, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

__global__ void bilinear_bwd_kernel(const float* __restrict__ dy, int dy_pitch, float* __restrict__ dx, int dx_pitch, int B, int H, int W,
                                    int C, int OH, int OW, float sh, float sw) {
    const int CV = C / 4;
    const long long total = (long long)B * H * W * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        int oh_lo, oh_hi, ow_lo, ow_hi;
        cand_range(h, sh, OH, oh_lo, oh_hi);
        cand_range(w, sw, OW, ow_lo, ow_hi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const Lerp lh = lerp_of(oh, sh, H);
            float wh = 0.f;
            if (lh.i0 == h) wh += lh.w0;
            if (lh.i1 == h) wh += lh.w1;      // i1 == i0 at the border: both weights land on the same pixel
            if (lh.i0 != h && lh.i1 != h) continue;
            const float* row = dy + (size_t)(b * OH + oh) * OW * dy_pitch + 4 * cv;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const Lerp lw = lerp_of(ow, sw, W);
                if (lw.i0 != w && lw.i1 != w) continue;
                float ww = 0.f;
                if (lw.i0 == w) ww += lw.w0;
                if (lw.i1 == w) ww += lw.w1;
                acc += (wh * ww) * ld4(row + (size_t)ow * dy_pitch);
            }
        }
        st4(dx + ((size_t)(b * H + h) * W + w) * dx_pitch + 4 * cv, acc);
    }
}

// Separable form of the same gradient: the interpolation matrix is Wh (x) Ww, so its transpose can be applied one axis at a time --
// dy [B,OH,OW,C] -> tmp [B,OH,W,C] (along W), tmp -> dx [B,H,W,C] (along H).  Per element ~(2/scale + 3) candidates per pass instead of
// their product in one (100 -> 2 x 10 for the 4x up-sampling of the logits), and the second pass reads the OW/W times smaller tmp.
// AXIS 0: reduce along the last spatial axis (n_out = OW -> n_in = W), rows = B*OH.  AXIS 1: along H (n_out = OH -> n_in = H) of a
// [B, OH, W, C] tensor, rows = B.
template <int AXIS>
__global__ void bilinear_bwd_axis_kernel(const float* __restrict__ src, int src_pitch, float* __restrict__ dst, int dst_pitch, int rows, int n_in,
                                         int n_out, int inner, int C, float scale, unsigned* __restrict__ amax_out) {
    const int CV = C / 4;
    // AXIS 0: index = ((row * n_in + i) * CV + cv), inner == 1.   AXIS 1: index = (((row * n_in + i) * inner + w) * CV + cv), inner == W
    const long long total = (long long)rows * n_in * inner * CV;
    float m = 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % CV);
        long long t = idx / CV;
        const int w = (int)(t % inner); t /= inner;
        const int i = (int)(t % n_in);
        const int row = (int)(t / n_in);
        int lo, hi;
        cand_range(i, scale, n_out, lo, hi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* base = src + ((size_t)row * n_out * inner + w) * src_pitch + 4 * cv;
        for (int o = lo; o <= hi; ++o) {
            const Lerp l = lerp_of(o, scale, n_in);
            if (l.i0 != i && l.i1 != i) continue;
            float wt = 0.f;
            if (l.i0 == i) wt += l.w0;
            if (l.i1 == i) wt += l.w1;          // i1 == i0 at the border: both weights land on the same pixel
            acc += wt * ld4(base + (size_t)o * inner * src_pitch);
        }
        st4(dst + (((size_t)row * n_in + i) * inner + w) * dst_pitch + 4 * cv, acc);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    }
    if (amax_out != nullptr) amax_commit(m, amax_out);        // range of the gradient written (AXIS 1: the final one), for the conv backward that reads it
}

// ---- global average pool -----------------------------------------------------------------------
// 16 channel vectors x 16 row lanes per block (256 contiguous bytes per row and block), eight rows in flight per thread: the ASPP's
// image pool reads 268 MB at configs[4] -- with 64 x 4 lanes and one load in flight it ran at 0.56 TB/s on 64 blocks.  Fixed summation order.
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C) {
    __shared__ f32x4 red[256];
    const int CV = C / 4;
    const int b = blockIdx.y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int cv = blockIdx.x * 16 + tx;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (cv < CV) {
        const float* base = x + (size_t)b * HW * C + 4 * cv;
        for (int r = ty; r < HW; r += 16 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + 16 * u;
                v[u] = ld4(base + (size_t)(rr < HW ? rr : r) * C);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + 16 * u < HW) s += v[u];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && cv < CV) {
        for (int k = 1; k < 16; ++k) s += red[16 * k + tx];
        st4(y + (size_t)b * C + 4 * cv, s * (1.f / (float)HW));
    }
}

// The same pool over a tensor held as fp16 planes (the last block of a backbone leaves its output in the format the ASPP convs read): an
// element is (h0 + 2^-11 h1) / s exactly as pylc_from_planes rebuilds it, summed in gap_fwd_kernel's order -- bit-identical to converting
// first, without the conversion pass (2048 channels x 32768 pixels: 0.8 GB of traffic per step).
typedef _Float16 f16x4p __attribute__((ext_vector_type(4)));
template <int NPL>
__global__ __launch_bounds__(256) void gap_fwd_planes_kernel(const _Float16* __restrict__ planes, long long plane_stride, const unsigned* __restrict__ amax,
                                                             float* __restrict__ y, int HW, int C) {
    __shared__ f32x4 red[256];
    const int CV = C / 4;
    const int b = blockIdx.y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int cv = blockIdx.x * 16 + tx;
    int e = (int)((*amax >> 23) & 0xFFu);                      // 1 / (the power-of-two scale of the tensor's range bound): conv_common.h pow2_scale_for
    int se = 127 + 14 - (e - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    const float inv = 1.f / __uint_as_float((unsigned)se << 23);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (cv < CV) {
        const bool il = NPL == 2 && planes_il(plane_stride);      // chunk-interleaved (common.h): C % 32 == 0, a row starts a chunk
        const long long e0 = (long long)b * HW * C + 4 * cv;
        for (int r = ty; r < HW; r += 16 * 8) {
            f16x4p h0[8], h1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + 16 * u;
                const long long off = planes_phys(e0 + (long long)(rr < HW ? rr : r) * C, il);
                h0[u] = *reinterpret_cast<const f16x4p*>(planes + off);
                if constexpr (NPL == 2) h1[u] = *reinterpret_cast<const f16x4p*>(planes + plane_stride + off);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + 16 * u < HW) {
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = ((float)h0[u][k] + (NPL == 2 ? (float)h1[u][k] : 0.f) * (1.f / 2048.f)) * inv;
                    s += v;
                }
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && cv < CV) {
        for (int k = 1; k < 16; ++k) s += red[16 * k + tx];
        st4(y + (size_t)b * C + 4 * cv, s * (1.f / (float)HW));
    }
}

template <bool ACC>
__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int HW, int C) {
    const int CV = C / 4;
    const long long total = (long long)B * HW * CV;
    const float inv = 1.f / (float)HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const int b = (int)(i / ((long long)HW * CV));
        f32x4 v = ld4(dy + (size_t)b * C + 4 * cv) * inv;
        if (ACC) v = v + ld4(dx + 4 * i);       // dx already holds the other consumers' part of the gradient (ops.ResidualLink)
        st4(dx + 4 * i, v);
    }
}

// ---- image ingest / layout ---------------------------------------------------------------------
__global__ void image_pack_kernel(const float* __restrict__ img, int B, int Cimg, int HW, f32x4 mean, f32x4 istd, float* __restrict__ out,
                                  float denom) {
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW, p = i % HW;
        const float* src = img + b * Cimg * HW + p;
        f32x4 v;
        v.x = src[0];
        v.y = Cimg == 3 ? src[HW] : v.x;
        v.z = Cimg == 3 ? src[2 * (long long)HW] : v.x;
        v.w = 0.f;
        // ((x - mean) / std) / 255, evaluated in the reference's operation order (model.py:444-445)
        f32x4 r;
        r.x = ((v.x - mean.x) / istd.x) / denom;          // denom = 255; 1 for the grayscale defaults branch (model.py:428-430)
        r.y = ((v.y - mean.y) / istd.y) / denom;
        r.z = ((v.z - mean.z) / istd.z) / denom;
        r.w = 0.f;
        st4(out + 4 * i, r);
    }
}

// same as image_pack_kernel for uint8 tiles as stored in the HDF5 database (db/database.py:218-233): 4x fewer PCIe bytes
__global__ void image_pack_u8_kernel(const unsigned char* __restrict__ img, int B, int Cimg, int HW, f32x4 mean, f32x4 sd,
                                     float* __restrict__ out, float denom) {
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW, p = i % HW;
        const unsigned char* src = img + b * Cimg * HW + p;
        const float x = (float)src[0];
        const float y = Cimg == 3 ? (float)src[HW] : x;
        const float z = Cimg == 3 ? (float)src[2 * (long long)HW] : x;
        f32x4 r;
        r.x = ((x - mean.x) / sd.x) / denom;
        r.y = ((y - mean.y) / sd.y) / denom;
        r.z = ((z - mean.z) / sd.z) / denom;
        r.w = 0.f;
        st4(out + 4 * i, r);
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int x_pitch, float* __restrict__ y, int B, int HW, int C) {
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i % HW;
        const long long t = i / HW;
        const int c = (int)(t % C);
        const long long b = t / C;
        y[i] = x[(b * HW + p) * x_pitch + c];
    }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int y_pitch, int B, int HW, int C) {
    const long long total = (long long)B * HW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long t = i / C;
        const long long p = t % HW, b = t / HW;
        y[(b * HW + p) * y_pitch + c] = x[(b * C + c) * HW + p];
    }
}

}  // namespace pylc

using namespace pylc;

extern "C" int pylc_maxpool_fwd(const float* x, float* y, unsigned char* idx, int B, int H, int W, int C, int k, int stride, int pad, int OH,
                                int OW, void* stream) {
    PYLC_REQUIRE(x && y && B > 0 && C > 0 && C % 4 == 0 && k >= 1 && k <= 15 && stride >= 1 && pad >= 0 && pad <= k / 2, "maxpool_fwd: bad arguments");
    PYLC_REQUIRE(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1 && OH > 0 && OW > 0, "maxpool_fwd: bad output size");
    const long long total = (long long)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, y, idx, B, H, W, C, k, stride, pad, OH, OW,
                       PoolPlanes{nullptr, 0, nullptr, 0});
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_maxpool_fwd_planes(const float* x, void* y_planes, long long plane_stride, int nplanes, const unsigned int* bound, unsigned char* idx,
                                       int B, int H, int W, int C, int k, int stride, int pad, int OH, int OW, void* stream) {
    PYLC_REQUIRE(x && y_planes && bound && B > 0 && C > 0 && C % 8 == 0 && k >= 1 && k <= 15 && stride >= 1 && pad >= 0 && pad <= k / 2 &&
                     (nplanes == 1 || nplanes == 2) && plane_stride % 4 == 0, "maxpool_fwd_planes: bad arguments (C must be a multiple of 8)");
    PYLC_REQUIRE(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1 && OH > 0 && OW > 0, "maxpool_fwd_planes: bad output size");
    const long long total = (long long)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, nullptr, idx, B, H, W, C, k, stride, pad, OH, OW,
                       PoolPlanes{static_cast<_Float16*>(y_planes), plane_stride, bound, nplanes});
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int k, int stride, int pad,
                                int OH, int OW, void* stream) {
    PYLC_REQUIRE(dy && idx && dx && B > 0 && C > 0 && C % 4 == 0 && k >= 1 && k <= 15 && stride >= 1, "maxpool_bwd: bad arguments");
    const long long total = (long long)B * H * W * (C / 4);
    if (!launch_maxpool2_bwd(dy, idx, dx, B, H, W, C, k, stride, pad, OH, OW, PoolAdd{nullptr, 0, 0, 0, 0, 0}, as_stream(stream)))
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, idx, dx, B, H, W, C, k, stride, pad, OH, OW,
                           PoolAdd{nullptr, 0, 0, 0, 0, 0});
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_maxpool_bwd_add(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int k, int stride, int pad,
                                    int OH, int OW, const float* add, int add_pitch, int add_h0, int add_w0, int add_h, int add_w, void* stream) {
    PYLC_REQUIRE(dy && idx && dx && add && B > 0 && C > 0 && C % 4 == 0 && k >= 1 && k <= 15 && stride >= 1, "maxpool_bwd_add: bad arguments");
    PYLC_REQUIRE(add_pitch >= C && add_pitch % 4 == 0 && add_h0 >= 0 && add_w0 >= 0 && add_h > 0 && add_w > 0 && add_h0 + add_h <= H &&
                     add_w0 + add_w <= W, "maxpool_bwd_add: the added window must lie inside the %dx%d input", H, W);
    const long long total = (long long)B * H * W * (C / 4);
    if (!launch_maxpool2_bwd(dy, idx, dx, B, H, W, C, k, stride, pad, OH, OW, PoolAdd{add, add_pitch, add_h0, add_w0, add_h, add_w}, as_stream(stream)))
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, idx, dx, B, H, W, C, k, stride, pad, OH, OW,
                           PoolAdd{add, add_pitch, add_h0, add_w0, add_h, add_w});
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_crop_copy(const float* src, int src_pitch, int H, int W, int h0, int w0, float* dst, int dst_pitch, int B, int TH, int TW,
                              int C, void* stream) {
    PYLC_REQUIRE(src && dst && B > 0 && C > 0 && C % 4 == 0 && TH > 0 && TW > 0, "crop_copy: bad arguments");
    PYLC_REQUIRE(src_pitch >= C && dst_pitch >= C && src_pitch % 4 == 0 && dst_pitch % 4 == 0, "crop_copy: bad pitch");
    PYLC_REQUIRE(h0 >= 0 && w0 >= 0 && h0 + TH <= H && w0 + TW <= W, "crop_copy: window outside the %dx%d source", H, W);
    const long long total = (long long)B * TH * TW * (C / 4);
    hipLaunchKernelGGL(crop_copy_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), src, src_pitch, H, W, h0, w0, dst, dst_pitch, B, TH,
                       TW, C);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int pylc_bilinear_fwd(const float* x, int x_pitch, float* y, int y_pitch, int B, int H, int W, int C, int OH, int OW, void* stream) {
    PYLC_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, "bilinear_fwd: bad arguments");
    PYLC_REQUIRE(x_pitch >= C && y_pitch >= C && x_pitch % 4 == 0 && y_pitch % 4 == 0, "bilinear_fwd: bad pitch");
    const long long total = (long long)B * OH * OW * (C / 4);
    if (B <= 65535 && OH <= 65535 && (long long)OW * (C / 4) < (1ll << 31))
        hipLaunchKernelGGL(bilinear_fwd_rows_kernel, dim3(cdiv(OW * (C / 4), 256), OH, B), dim3(256), 0, as_stream(stream), x, x_pitch, y, y_pitch, H, W, C, OH, OW,
                           ac_scale(H, OH), ac_scale(W, OW));
    else
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, x_pitch, y, y_pitch, B, H, W, C, OH, OW,
                       ac_scale(H, OH), ac_scale(W, OW));
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_bilinear_bwd(const float* dy, int dy_pitch, float* dx, int dx_pitch, int B, int H, int W, int C, int OH, int OW, void* stream) {
    PYLC_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, "bilinear_bwd: bad arguments");
    PYLC_REQUIRE(dy_pitch >= C && dx_pitch >= C && dy_pitch % 4 == 0 && dx_pitch % 4 == 0, "bilinear_bwd: bad pitch");
    const long long total = (long long)B * H * W * (C / 4);
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, dy_pitch, dx, dx_pitch, B, H, W, C, OH, OW,
                       ac_scale(H, OH), ac_scale(W, OW));
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" size_t pylc_bilinear_bwd_workspace(int B, int W, int C, int OH) { return (size_t)B * OH * W * C * sizeof(float); }

extern "C" int pylc_bilinear_bwd_separable(const float* dy, int dy_pitch, float* dx, int dx_pitch, int B, int H, int W, int C, int OH, int OW,
                                           float* workspace, unsigned int* amax_bits, void* stream) {
    PYLC_REQUIRE(dy && dx && workspace && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, "bilinear_bwd_separable: bad arguments");
    PYLC_REQUIRE(dy_pitch >= C && dx_pitch >= C && dy_pitch % 4 == 0 && dx_pitch % 4 == 0, "bilinear_bwd_separable: bad pitch");
    const long long t0 = (long long)B * OH * W * (C / 4), t1 = (long long)B * H * W * (C / 4);
    if (amax_bits != nullptr) PYLC_HIP(hipMemsetAsync(amax_bits, 0, sizeof(unsigned), as_stream(stream)));
    hipLaunchKernelGGL(bilinear_bwd_axis_kernel<0>, dim3(grid_for(t0)), dim3(256), 0, as_stream(stream), dy, dy_pitch, workspace, C, B * OH, W, OW, 1, C,
                       ac_scale(W, OW), static_cast<unsigned*>(nullptr));
    PYLC_LAUNCH_CHECK();
    hipLaunchKernelGGL(bilinear_bwd_axis_kernel<1>, dim3(grid_for(t1)), dim3(256), 0, as_stream(stream), workspace, C, dx, dx_pitch, B, H, OH, W, C,
                       ac_scale(H, OH), amax_bits);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_gap_fwd(const float* x, float* y, int B, int HW, int C, void* stream) {
    PYLC_REQUIRE(x && y && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "gap_fwd: bad arguments");
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(cdiv(C / 4, 16), B), dim3(256), 0, as_stream(stream), x, y, HW, C);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_gap_fwd_planes(const void* planes, long long plane_stride, int nplanes, const unsigned int* amax, float* y, int B, int HW, int C,
                                   void* stream) {
    PYLC_REQUIRE(planes && amax && y && B > 0 && HW > 0 && C > 0 && C % 8 == 0 && (nplanes == 1 || nplanes == 2) &&
                 (nplanes == 1 || plane_stride >= (long long)B * HW * C || (planes_il(plane_stride) && C % 32 == 0)), "gap_fwd_planes: bad arguments");
    const _Float16* p = static_cast<const _Float16*>(planes);
    if (nplanes == 2)
        hipLaunchKernelGGL(gap_fwd_planes_kernel<2>, dim3(cdiv(C / 4, 16), B), dim3(256), 0, as_stream(stream), p, plane_stride, amax, y, HW, C);
    else
        hipLaunchKernelGGL(gap_fwd_planes_kernel<1>, dim3(cdiv(C / 4, 16), B), dim3(256), 0, as_stream(stream), p, plane_stride, amax, y, HW, C);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_gap_bwd_acc(const float* dy, float* dx, int B, int HW, int C, int accumulate, void* stream) {
    PYLC_REQUIRE(dy && dx && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "gap_bwd: bad arguments");
    const dim3 grid(grid_for((long long)B * HW * (C / 4)));
    if (accumulate)
        hipLaunchKernelGGL(gap_bwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), dy, dx, B, HW, C);
    else
        hipLaunchKernelGGL(gap_bwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), dy, dx, B, HW, C);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_gap_bwd(const float* dy, float* dx, int B, int HW, int C, void* stream) {
    return pylc_gap_bwd_acc(dy, dx, B, HW, C, 0, stream);
}

extern "C" int pylc_image_pack_denom(const void* img, int is_u8, int B, int Cimg, int H, int W, const float* mean3, const float* std3,
                                     float denom, float* out, void* stream) {
    PYLC_REQUIRE(img && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && (Cimg == 1 || Cimg == 3) && denom > 0.f, "image_pack: bad arguments");
    f32x4 mean = {mean3[0], mean3[1], mean3[2], 0.f}, sd = {std3[0], std3[1], std3[2], 1.f};
    if (is_u8)
        hipLaunchKernelGGL(image_pack_u8_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, as_stream(stream),
                           static_cast<const unsigned char*>(img), B, Cimg, H * W, mean, sd, out, denom);
    else
        hipLaunchKernelGGL(image_pack_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, as_stream(stream),
                           static_cast<const float*>(img), B, Cimg, H * W, mean, sd, out, denom);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_image_pack(const float* img, int B, int Cimg, int H, int W, const float* mean3, const float* std3, float* out, void* stream) {
    return pylc_image_pack_denom(img, 0, B, Cimg, H, W, mean3, std3, 255.f, out, stream);
}

extern "C" int pylc_image_pack_u8(const unsigned char* img, int B, int Cimg, int H, int W, const float* mean3, const float* std3, float* out,
                                  void* stream) {
    return pylc_image_pack_denom(img, 1, B, Cimg, H, W, mean3, std3, 255.f, out, stream);
}

extern "C" int pylc_nhwc_to_nchw(const float* x, int x_pitch, float* y, int B, int H, int W, int C, void* stream) {
    PYLC_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && x_pitch >= C, "nhwc_to_nchw: bad arguments");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0, as_stream(stream), x, x_pitch, y, B, H * W, C);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_nchw_to_nhwc(const float* x, float* y, int y_pitch, int B, int H, int W, int C, void* stream) {
    PYLC_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && y_pitch >= C, "nchw_to_nhwc: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0, as_stream(stream), x, y, y_pitch, B, H * W, C);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
