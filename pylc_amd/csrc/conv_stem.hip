// The ResNet stem convolution -- 7x7, stride 2, padding 3, a 4-channel image pack in, 64 channels out (models/backbone/resnet.py:72
// `nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)`; the product feeds the 4-channel pack of pylc_image_pack) -- as a PATCH
// kernel in the f16x3 arithmetic.
//
// Why its own kernel: the generic thin-input path (conv_igemm.hip gather_gemm_kernel<.., CIN4>) gathers 16-byte pixels tap by tap -- eight
// taps of one output pixel per K-step, every piece its own bounds-checked load, the fp32 -> fp16 split inside the loop -- and writes its 537 MB
// of output (bs 32, 512^2) in 0.38 ms: 1.8 TB/s of HBM traffic for a conv whose matrix work is 77 us (profiles/r06_kernel_stats.csv).  The conv
// is a fixed shape; here
//   * a block owns a 16 x 32 patch of output pixels (512 pixels x 64 channels, eight waves of 2 x 32 pixels x 64 channels on 4 x 4 tiles of
//     v_mfma_f32_16x16x32_f16, as conv_pl.hip) and walks 16 patches as a persistent block, one block per CU;
//   * the patch's 37 x 69 input pixels are fetched ONCE (each pixel is used by up to 16 taps), split ONCE into the two scaled fp16 planes
//     (h0 = rn16(s x), h1 = rn16(2^11 (s x - h0)): conv_common.h split2) and kept in LDS, 8 bytes per pixel and plane; the next patch's pixels are
//     in registers while this one is multiplied (two LDS patch buffers, one barrier per patch);
//   * the filter is split once per block into LDS planes [64][7 rows][8 taps][4 channels] -- every filter ROW padded to eight taps (tap 7 zero)
//     -- so that a K-step is one filter row and a lane's 8 reduction elements (two taps x 4 channels) are 16 contiguous bytes of the patch AND
//     of the filter: fragment reads are ds_read_b128 like conv_pl.hip's, conflict-free (pixels: lanes 16 bytes apart; filter rows: a pitch of
//     29 x 16 bytes);
//   * seven K-steps (filter rows) per patch, no barrier between them: everything they read is already in LDS.
// The reduction order differs from the generic path's (8 consecutive taps per K-step there, one padded filter row here): same products, fp32
// rounding-level differences (tests/test_ops_gpu.py::test_stem_patch_kernel_matches_generic: <= 2e-7 of sum |a||b|; statistics likewise).
#include "conv_common.h"

namespace pylc {

namespace {

constexpr int ST_PH = 16, ST_PW = 32;                   // output patch: 512 pixels, eight waves of 2 x 32
constexpr int ST_IH = 2 * ST_PH + 5, ST_IW = 2 * ST_PW + 5;      // 37 x 69 input pixels
constexpr int ST_THR = 512;
constexpr int ST_ROWB = 72 * 8;                         // bytes per patch row and plane (72 pixels x 4 halves; 69 used)
constexpr int ST_PATCH = ST_IH * ST_ROWB;               // bytes per plane: 21312
constexpr int ST_WPITCH = 29 * 16;                      // bytes per filter (cout) row and plane: 7 x 8 x 4 halves = 448, padded to an odd multiple of 16
constexpr int ST_WPL = 64 * ST_WPITCH;                  // bytes per filter plane: 29696
constexpr int ST_NPIX = (ST_IH * ST_IW + ST_THR - 1) / ST_THR;    // patch pixels per thread: 5
constexpr int ST_LDS = 2 * 2 * ST_PATCH + 2 * ST_WPL + 8 * 64 * 2 * 4 + 64;      // two patch buffers (two planes each), the filter planes, statistics partials: 148.7 KB, one block of 8 waves per CU

struct StemArgs {
    const float* x;       // [B][IH][IW][4] fp32
    const float* w;       // [64][7][7][4] fp32
    float* y;             // [B][OH][OW][64] fp32
    float* stats;         // optional [patches][2][64]
    const unsigned* amax_x;
    const unsigned* amax_w;
    int B, IH, IW, OH, OW;
    int patches_w, patches_per_image, n_patches;
};

typedef __attribute__((address_space(3))) void* lds_vptr_;

__global__ __launch_bounds__(ST_THR, 1) void stem_fwd_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    char* const patch = lds;                                 // [2 buffers][2 planes][ST_IH][ST_ROWB]
    char* const wpl = lds + 4 * ST_PATCH;                    // [2 planes][64][ST_WPITCH]
    float* const sred = reinterpret_cast<float*>(lds + 4 * ST_PATCH + 2 * ST_WPL);      // [8 waves][64][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float sx = pow2_scale_for(*a.amax_x), sw = pow2_scale_for(*a.amax_w);
    const float c = pow2_inv(sx) * pow2_inv(sw);

    // ---- the filter: split once, rows padded to eight taps ----
    for (int i = tid; i < 64 * 7 * 8; i += ST_THR) {
        const int s = i & 7, r = (i >> 3) % 7, n = i / 56;
        uint2 p0 = {0u, 0u}, p1 = {0u, 0u};
        if (s < 7) split2(*reinterpret_cast<const f32x4*>(a.w + ((size_t)(n * 7 + r) * 7 + s) * 4), sw, p0, p1);
        char* dst = wpl + n * ST_WPITCH + (r * 8 + s) * 8;
        *reinterpret_cast<uint2*>(dst) = p0;
        *reinterpret_cast<uint2*>(dst + ST_WPL) = p1;
    }

    // the pad pixels 69 .. 71 of every patch row (both buffers, both planes): tap 7 of the last column reads pixel 69 -- against a zero weight,
    // but 0 x garbage must not be NaN
    for (int i = tid; i < 4 * ST_IH; i += ST_THR) {
        char* row = patch + (i / ST_IH) * ST_PATCH + (i % ST_IH) * ST_ROWB + ST_IW * 8;
        *reinterpret_cast<uint2*>(row) = uint2{0u, 0u};
        *reinterpret_cast<uint2*>(row + 8) = uint2{0u, 0u};
        *reinterpret_cast<uint2*>(row + 16) = uint2{0u, 0u};
    }

    // ---- patch loader: this thread's pixels of a patch (row-major over the 21 x 69 window), zeros outside the image ----
    f32x4 px[ST_NPIX];
    auto fetch = [&](int patch_id) {
        const int b = patch_id / a.patches_per_image, q = patch_id - b * a.patches_per_image;
        const int py = q / a.patches_w, pxw = q - py * a.patches_w;
        const int ih0 = 2 * (py * ST_PH) - 3, iw0 = 2 * (pxw * ST_PW) - 3;
        const float* img = a.x + (size_t)b * a.IH * a.IW * 4;
#pragma unroll
        for (int u = 0; u < ST_NPIX; ++u) {
            const int i = tid + ST_THR * u;
            const int r = i / ST_IW, cc = i - r * ST_IW;
            const int ih = ih0 + r, iw = iw0 + cc;
            const bool ok = i < ST_IH * ST_IW && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW;
            px[u] = ok ? *reinterpret_cast<const f32x4*>(img + ((size_t)ih * a.IW + iw) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        char* base = patch + buf * 2 * ST_PATCH;
#pragma unroll
        for (int u = 0; u < ST_NPIX; ++u) {
            const int i = tid + ST_THR * u;
            if (i < ST_IH * ST_IW) {
                const int r = i / ST_IW, cc = i - r * ST_IW;
                uint2 p0, p1;
                split2(px[u], sx, p0, p1);
                char* dst = base + r * ST_ROWB + cc * 8;
                *reinterpret_cast<uint2*>(dst) = p0;
                *reinterpret_cast<uint2*>(dst + ST_PATCH) = p1;
            }
        }
    };

    // fragment geometry: lane l = row (pixel / filter row) l & 15 of a 16-row fragment, reduction elements 8 (l >> 4) .. + 7 = taps 2 (l >> 4), + 1
    const int frow = lane & 15, fg = lane >> 4;
    // this wave: output rows 2 wave, 2 wave + 1 of the patch; fragment i: row 2 wave + (i >> 1), columns 16 (i & 1) .. + 15
    // pixel (oy, ox), filter row r, taps 2 fg, 2 fg + 1 -> patch row 2 oy + r, patch pixel 2 ox + 2 fg
    int a_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = 2 * wave + (i >> 1), ox = 16 * (i & 1) + frow;
        a_off[i] = (2 * oy) * ST_ROWB + (2 * ox + 2 * fg) * 8;
    }
    const int b_off = frow * ST_WPITCH + (2 * fg) * 8;       // + 16 j rows of the filter, + r * 64 bytes per filter row

    const int G = (int)gridDim.x;
    int pid = (int)blockIdx.x;
    if (pid < a.n_patches) fetch(pid);
    __syncthreads();                                          // the filter planes
    int buf = 0;
    if (pid < a.n_patches) stash(0);
    for (; pid < a.n_patches; pid += G, buf ^= 1) {
        __syncthreads();                                      // this patch's planes are in LDS; every wave is done with the other buffer
        const int nxt = pid + G;
        if (nxt < a.n_patches) fetch(nxt);                    // in flight under the products below

        f32x4v acc[4][4], acc_lo[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc_lo[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
        const char* pa = patch + buf * 2 * ST_PATCH;
#pragma unroll 1
        for (int r = 0; r < 7; ++r) {
            f16x8 fb[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(wpl + pl * ST_WPL + b_off + j * 16 * ST_WPITCH + r * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x8 fa[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fa[pl] = *reinterpret_cast<const f16x8*>(pa + pl * ST_PATCH + a_off[i] + r * ST_ROWB);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // the filter fragment is the FIRST operand (conv_pl.hip): lane l of the result holds pixel l & 15, channels 4 (l >> 4) .. + 3
                    acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[1], acc_lo[i][j], 0, 0, 0);
                    acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][1], fa[0], acc_lo[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[0], acc[i][j], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: fold, statistics on the raw sums (the scale is a power of two), unscale, store ----
        const int b = pid / a.patches_per_image, q = pid - b * a.patches_per_image;
        const int py = q / a.patches_w, pxw = q - py * a.patches_w;
        const bool do_stats = a.stats != nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4v cs = {0.f, 0.f, 0.f, 0.f}, css = cs;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4v t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(acc_lo[i][j][e], 1.f / 2048.f, acc[i][j][e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[e] = i == 0 ? t[e] : cs[e] + t[e]; css[e] = i == 0 ? t[e] * t[e] : css[e] + t[e] * t[e]; }
                const int oy = py * ST_PH + 2 * wave + (i >> 1), ox = pxw * ST_PW + 16 * (i & 1) + frow;
                f32x4v v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = t[e] * c;
                *reinterpret_cast<f32x4v*>(a.y + (((size_t)b * a.OH + oy) * a.OW + ox) * 64 + 16 * j + 4 * fg) = v;
            }
            if (do_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[e] = row_sum16(cs[e]); css[e] = row_sum16(css[e]); }
                if (frow == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sred[(wave * 64 + 16 * j + 4 * fg + e) * 2] = cs[e] * c;
                        sred[(wave * 64 + 16 * j + 4 * fg + e) * 2 + 1] = css[e] * c * c;
                    }
                }
            }
        }
        if (do_stats) {
            __syncthreads();
            if (tid < 64) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) { sm += sred[(wv * 64 + tid) * 2]; sq += sred[(wv * 64 + tid) * 2 + 1]; }
                float* dst = a.stats + (size_t)pid * 128;
                dst[tid] = sm;
                dst[64 + tid] = sq;
            }
        }
        if (nxt < a.n_patches) stash(buf ^ 1);                // (the other buffer: every wave left it at this iteration's first barrier)
    }
}

}  // namespace

// conv2d_fwd_impl's hook: true if the launch is the stem and was taken here (statistics rows = patches)
bool takes_stem(const GatherGemmArgs& a) {
    return g_conv_precision == 2 && a.x != nullptr && a.x_planes == nullptr && a.w != nullptr && a.Cin == 4 && a.x_pitch == 4 && a.TR == 7 && a.TS == 7 &&
           a.in_sh == 2 && a.in_sw == 2 && a.dh0 == -3 && a.dw0 == -3 && a.dh_step == 1 && a.dw_step == 1 && a.N == 64 && a.N_store == 64 && a.y_pitch == 64 &&
           a.P % ST_PH == 0 && a.Q % ST_PW == 0 && a.OH == a.P && a.OW == a.Q && a.out_sh == 1 && a.out_sw == 1 && a.oh0 == 0 && a.ow0 == 0 &&
           a.IH == 2 * a.P && a.IW == 2 * a.Q && a.bias == nullptr && !a.accumulate && !a.out_half && !a.out_planes2 && a.ep_scale == nullptr &&
           a.ep_res == nullptr && a.ep_amax == nullptr && a.add_src == nullptr && a.bn_y == nullptr && a.amax_x != nullptr && a.amax_w != nullptr &&
           a.w_row_stride == 196 && a.w_off0 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 && !(g_pp_flags & 33554432);      // pylc_debug_pp_flags bit 25: the generic thin-input path (A/B, reference)
}

int launch_stem_fwd(GatherGemmArgs& g, hipStream_t st) {
    StemArgs a{};
    a.x = g.x; a.w = g.w; a.y = g.y; a.stats = g.stats; a.amax_x = g.amax_x; a.amax_w = g.amax_w;
    a.IH = g.IH; a.IW = g.IW; a.OH = g.P; a.OW = g.Q;
    a.B = g.M / (g.P * g.Q);
    a.patches_w = g.Q / ST_PW;
    a.patches_per_image = (g.P / ST_PH) * a.patches_w;
    a.n_patches = a.B * a.patches_per_image;
    g.halo_tiles_m = a.n_patches;                            // rows of the statistics partial (conv2d_fwd_impl reports it)
    const int grid = a.n_patches < kNumCU ? a.n_patches : kNumCU;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)grid), dim3(ST_THR), ST_LDS, st, a);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

int conv_stem_init() {
    PYLC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS));
    return PYLC_OK;
}

}  // namespace pylc
