// Sliding-window inference helpers: tile extraction fused with input normalisation, and the reference's tile
// stitching (overlap blend + argmax) + palette colourize + nearest resize, all on the GPU.
//
// Replaces Extractor.__split utils/extract.py:279-310 (+ Model.normalize_image models/model.py:416-445),
// utils/tools.py:209-319 reconstruct() and :322-358 colourize().  The reference copies every logit tile to the host
// (tools.py:221: n_tiles x n_classes x 512^2 x 4 B, 1.5 GB for one photo) and stitches in numpy; here one kernel reads
// the <= 4 tiles that cover an output pixel and writes one byte.
//
// reconstruct()'s arithmetic is reproduced exactly, quirks included: horizontal overlaps hold
// the mean of the two tiles' softmax PROBABILITIES while interiors hold raw logits; vertical overlaps apply softmax
// again to whatever the strip holds and average; the class is the argmax (first maximum) of that mixture.
#include "common.h"

namespace pylc {

constexpr int SMAXC = PYLC_MAX_CLASSES;

struct StitchGeom { int rows, cols, tile, stride, C, pitch, h, w; };

template <int C>
__device__ __forceinline__ void softmax_c(float (&v)[C]) {
    float m = v[0];
#pragma unroll
    for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] *= inv;
}

template <int C>
__device__ __forceinline__ void load_px(const float* __restrict__ logits, const StitchGeom& g, int ti, int tj, int r, int c0, float (&v)[C]) {
    const float* p = logits + (((size_t)(ti * g.cols + tj) * g.tile + r) * g.tile + c0) * g.pitch;
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = p[c];
}

// value of strip `i` (tile row) at local row r and global column x
template <int C>
__device__ __forceinline__ void strip_val(const float* __restrict__ logits, const StitchGeom& g, int i, int r, int x, float (&v)[C]) {
    const int S = g.stride;
    const int kx = x / S, rx = x - kx * S;
    if (kx == 0) { load_px<C>(logits, g, i, 0, r, rx, v); return; }
    if (kx == g.cols) { load_px<C>(logits, g, i, g.cols - 1, r, S + rx, v); return; }
    float b[C];
    load_px<C>(logits, g, i, kx - 1, r, S + rx, v);
    load_px<C>(logits, g, i, kx, r, rx, b);
    softmax_c<C>(v);
    softmax_c<C>(b);
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = (v[c] + b[c]) * 0.5f;
}

template <int C>
__global__ __launch_bounds__(256) void stitch_argmax_kernel(const float* __restrict__ logits, StitchGeom g, unsigned char* __restrict__ mask) {
    const long long total = (long long)g.h * g.w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / g.w), x = (int)(i % g.w);
        float v[C];
        if (g.stride == g.tile) {
            load_px<C>(logits, g, y / g.tile, x / g.tile, y % g.tile, x % g.tile, v);
        } else {
            const int S = g.stride;
            const int ky = y / S, ry = y - ky * S;
            if (ky == 0) strip_val<C>(logits, g, 0, ry, x, v);
            else if (ky == g.rows) strip_val<C>(logits, g, g.rows - 1, S + ry, x, v);
            else {
                float b[C];
                strip_val<C>(logits, g, ky, ry, x, v);            // top half of strip ky
                strip_val<C>(logits, g, ky - 1, S + ry, x, b);    // bottom half of strip ky-1
                softmax_c<C>(v);
                softmax_c<C>(b);
#pragma unroll
                for (int c = 0; c < C; ++c) v[c] = (v[c] + b[c]) * 0.5f;
            }
        }
        int best = 0;
        float bv = v[0];
#pragma unroll
        for (int c = 1; c < C; ++c) if (v[c] > bv) { bv = v[c]; best = c; }     // first maximum (np.argmax)
        mask[i] = (unsigned char)best;
    }
}

__global__ void colourize_resize_kernel(const unsigned char* __restrict__ mask, int h, int w, const unsigned char* __restrict__ palette,
                                        unsigned char* __restrict__ out, int oh, int ow, float fy, float fx) {
    const long long total = (long long)oh * ow;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int oy = (int)(i / ow), ox = (int)(i % ow);
        int sy = (int)floorf(oy * fy), sx = (int)floorf(ox * fx);     // cv2.INTER_NEAREST
        sy = sy < h - 1 ? sy : h - 1;
        sx = sx < w - 1 ? sx : w - 1;
        const int cls = mask[(size_t)sy * w + sx];
        out[3 * i + 0] = palette[3 * cls + 0];
        out[3 * i + 1] = palette[3 * cls + 1];
        out[3 * i + 2] = palette[3 * cls + 2];
    }
}

// tiles [n][tile][tile][4] <- normalised window of img [Cimg][H][W] (raw 0..255); tile order row-major (extract.py:302-308)
__global__ void pack_tiles_kernel(const float* __restrict__ img, int Cimg, int H, int W, int tile, int stride, int cols, int first, int count,
                                  float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const long long total = (long long)count * tile * tile;
    const size_t plane = (size_t)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % tile);
        long long t = i / tile;
        const int r = (int)(t % tile);
        const int k = first + (int)(t / tile);
        const int y = (k / cols) * stride + r, x = (k % cols) * stride + c0;
        const float* src = img + (size_t)y * W + x;
        const float a = src[0];
        const float b = Cimg == 3 ? src[plane] : a;
        const float c = Cimg == 3 ? src[2 * plane] : a;
        f32x4 v;
        v.x = ((a - m0) / s0) / 255.f;
        v.y = ((b - m1) / s1) / 255.f;
        v.z = ((c - m2) / s2) / 255.f;
        v.w = 0.f;
        *reinterpret_cast<f32x4*>(out + 4 * i) = v;
    }
}

static inline int grid_for(long long n) {
    long long b = cdiv<long long>(n, 256);
    return (int)(b < 8192 ? (b < 1 ? 1 : b) : 8192);
}

}  // namespace pylc

using namespace pylc;

extern "C" int pylc_stitch_argmax(const float* logits, int pitch, int rows, int cols, int tile, int stride, int C, unsigned char* mask,
                                  void* stream) {
    PYLC_REQUIRE(logits && mask && rows > 0 && cols > 0 && tile > 0 && pitch >= C, "stitch_argmax: bad arguments");
    PYLC_REQUIRE(stride == tile || (tile % 2 == 0 && stride == tile / 2), "stitch_argmax: stride must be tile or tile/2 (test.py:63)");
    const int olap = tile - stride;
    StitchGeom g{rows, cols, tile, stride, C, pitch, rows * stride + olap, cols * stride + olap};
    const int blocks = grid_for((long long)g.h * g.w);
    hipStream_t st = as_stream(stream);
#define LAUNCH_ST(CC) hipLaunchKernelGGL((stitch_argmax_kernel<CC>), dim3(blocks), dim3(256), 0, st, logits, g, mask)
    switch (C) {
        case 2: LAUNCH_ST(2); break; case 3: LAUNCH_ST(3); break; case 4: LAUNCH_ST(4); break; case 5: LAUNCH_ST(5); break;
        case 6: LAUNCH_ST(6); break; case 7: LAUNCH_ST(7); break; case 8: LAUNCH_ST(8); break; case 9: LAUNCH_ST(9); break;
        case 10: LAUNCH_ST(10); break; case 11: LAUNCH_ST(11); break; case 12: LAUNCH_ST(12); break; case 13: LAUNCH_ST(13); break;
        case 14: LAUNCH_ST(14); break; case 15: LAUNCH_ST(15); break; case 16: LAUNCH_ST(16); break;
        default: return fail(PYLC_ERR_ARG, "stitch_argmax: n_classes=%d unsupported (2..%d)", C, SMAXC);
    }
#undef LAUNCH_ST
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_colourize_resize(const unsigned char* mask, int h, int w, const unsigned char* palette_rgb, unsigned char* out_rgb, int oh,
                                     int ow, void* stream) {
    PYLC_REQUIRE(mask && palette_rgb && out_rgb && h > 0 && w > 0 && oh > 0 && ow > 0, "colourize_resize: bad arguments");
    hipLaunchKernelGGL(colourize_resize_kernel, dim3(grid_for((long long)oh * ow)), dim3(256), 0, as_stream(stream), mask, h, w, palette_rgb,
                       out_rgb, oh, ow, (float)h / (float)oh, (float)w / (float)ow);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

extern "C" int pylc_image_pack_tiles(const float* img, int Cimg, int H, int W, int tile, int stride, int first_tile, int n_tiles,
                                     const float* mean3, const float* std3, float* out, void* stream) {
    PYLC_REQUIRE(img && out && mean3 && std3 && (Cimg == 1 || Cimg == 3) && tile > 0 && stride > 0 && H >= tile && W >= tile,
                 "image_pack_tiles: bad arguments");
    const int rows = (H - tile) / stride + 1, cols = (W - tile) / stride + 1;
    PYLC_REQUIRE(first_tile >= 0 && n_tiles > 0 && first_tile + n_tiles <= rows * cols, "image_pack_tiles: tile range outside the image");
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(grid_for((long long)n_tiles * tile * tile)), dim3(256), 0, as_stream(stream), img, Cimg, H, W, tile,
                       stride, cols, first_tile, n_tiles, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Evaluation: confusion matrix of class-index masks (replaces the sklearn passes over ~1e7-pixel flattened arrays in
// utils/metrics.py:64-88; the scores are simple functions of the n_cls x n_cls count matrix).  Integer counts in LDS,
// merged with 64-bit integer atomics: exact and order-independent.  Evaluator.validate()'s coverage quirk
// (utils/evaluate.py:171-174: the first n_classes pixels of both arrays are overwritten with 0..n_classes-1) is applied
// on the fly when force_coverage != 0.
// ---------------------------------------------------------------------------------------------------------------------
namespace pylc {
template <typename TT, typename TP>
__global__ __launch_bounds__(256) void confusion_kernel(const TT* __restrict__ yt, const TP* __restrict__ yp, long long n, int C,
                                                         int force_coverage, unsigned long long* __restrict__ cm) {
    __shared__ unsigned int hist[SMAXC * SMAXC];
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int t = (int)yt[i], p = (int)yp[i];
        if (force_coverage && i < C) { t = (int)i; p = (int)i; }
        if ((unsigned)t < (unsigned)C && (unsigned)p < (unsigned)C) atomicAdd(&hist[t * C + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256)
        if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}
}  // namespace pylc

extern "C" int pylc_confusion_matrix(const void* y_true, int true_bytes, const void* y_pred, int pred_bytes, long long n, int C,
                                     int force_coverage, unsigned long long* cm /* [C*C], zeroed by the caller */, void* stream) {
    PYLC_REQUIRE(y_true && y_pred && cm && n > 0 && C >= 2 && C <= SMAXC, "confusion_matrix: bad arguments");
    PYLC_REQUIRE((true_bytes == 1 || true_bytes == 8) && (pred_bytes == 1 || pred_bytes == 8), "confusion_matrix: masks must be uint8 or int64");
    const int blocks = grid_for(n) < 1024 ? grid_for(n) : 1024;
    hipStream_t st = as_stream(stream);
    if (true_bytes == 1 && pred_bytes == 1)
        hipLaunchKernelGGL((confusion_kernel<unsigned char, unsigned char>), dim3(blocks), dim3(256), 0, st, (const unsigned char*)y_true,
                           (const unsigned char*)y_pred, n, C, force_coverage, cm);
    else if (true_bytes == 8 && pred_bytes == 1)
        hipLaunchKernelGGL((confusion_kernel<long long, unsigned char>), dim3(blocks), dim3(256), 0, st, (const long long*)y_true,
                           (const unsigned char*)y_pred, n, C, force_coverage, cm);
    else if (true_bytes == 1 && pred_bytes == 8)
        hipLaunchKernelGGL((confusion_kernel<unsigned char, long long>), dim3(blocks), dim3(256), 0, st, (const unsigned char*)y_true,
                           (const long long*)y_pred, n, C, force_coverage, cm);
    else
        hipLaunchKernelGGL((confusion_kernel<long long, long long>), dim3(blocks), dim3(256), 0, st, (const long long*)y_true,
                           (const long long*)y_pred, n, C, force_coverage, cm);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
