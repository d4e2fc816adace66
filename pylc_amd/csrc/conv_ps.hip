// 1x1 / stride-1 convolution (forward, and the dgrads that are one) on fp16-plane operands: a PERSISTENT kernel with SPECIALISED WAVES.
//
// gg_pl_kernel's 128 x 128 tile on a short reduction (K = Cin / 32 = 2 .. 32 steps) spends as long outside its main loop as inside it:
// geometry + first operand round trip in front, the store burst behind (tools/pl_ablate.py on 256 -> 1024 @32^2: 77 us, 40 of them with
// the DMA and the MFMAs removed), and inside the loop a wave that waits for room in the load queue issues no MFMAs.  conv_p1.hip tried
// to trickle the stores under the next tile's loop from the SAME waves and ran into gfx9's single in-order vmcnt: a wave that has stores
// in flight cannot wait for its younger operand DMA without waiting for the stores first.
//
// Here the two jobs sit in different waves of one block per CU (8 waves, 256 registers each):
//   * waves 4-7, LOADERS.  Waves 4 / 5 bring the pixel tiles (A), 6 / 7 the filter tiles (B) global -> LDS by LDS-DMA into two rings
//     (NSA and NSB stages of 16 KB), as ONE stream of K-steps that runs through tile boundaries: the next tile's first operands arrive
//     while the current tile is still being multiplied, and each ring's depth is counted by its own waves' vmcnt.  Between their DMA
//     issues they also combine the BatchNorm statistics partials of the tile that was just stored.
//   * waves 0-3, COMPUTE (one per SIMD, 64 x 64 outputs each on 4 x 4 v_mfma_f32_16x16x32_f16, as gg_pl_kernel): fragment reads and MFMAs
//     only; the fragments of step g+1's first MFMA group are read while step g is multiplied.  When a tile's last step is done the wave folds
//     its accumulators in place, takes the statistics, issues its sixteen 16-byte stores and goes straight on: its vmcnt holds nothing but
//     stores, and it never waits for it.
// One s_barrier per K-step hands stages over: B(k) tells the compute waves that step k's operands have landed and the loaders that the
// stage of step k-2 has been read.
//
// Same pieces, same MFMA order per accumulator, same fold / statistics arithmetic as gg_pl_kernel: results are bit-identical to it
// (tests/test_planes_gpu.py::test_specialised_wave_1x1_kernel_is_bit_identical).  Plain 1x1 / stride-1 launches with Cin >= 64 and no
// bias / accumulation / fused inference epilogue / one-plane output; ADD = the dgrad that adds the ReLU-masked residual gradient
// (pylc_conv2d_dgrad_add) -- its extra operand is fetched by the compute waves in two halves around the last K-step.
//
// Reference call sites replaced: nn.Conv2d 1x1 forward / backward in models/backbone/resnet.py:21-26,92 (bottleneck conv1 / conv3,
// downsample), models/modules/aspp.py:64,67,89, models/decoder.py:27, models/backbone/xception.py:48 (pointwise).
#include "conv_common.h"

namespace pylc {

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int PS_BM = 128, PS_BN = 128, PS_ROW = 64;
constexpr int PS_NSA = 5, PS_NSB = 4;                         // ring depths (stages of 128 rows x 64 B per plane)
template <int NTERMS>
constexpr int ps_ring_bytes() { return (NTERMS == 3 ? 2 : 1) * PS_BM * PS_ROW; }      // one stage of one operand
template <int NTERMS>
constexpr int ps_lds_bytes() { return (PS_NSA + PS_NSB) * ps_ring_bytes<NTERMS>() + 2 * PS_BN * 2 * 4 + 64; }

// s_waitcnt vmcnt(n) for a runtime n out of the few values the loaders need (the immediate must be a constant)
__device__ __forceinline__ void ps_wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int NTERMS, bool ADD>
__global__ __launch_bounds__(512, 1) void gg_ps_kernel(const GatherGemmArgs a) {
    constexpr int BM = PS_BM, BN = PS_BN, WM = 64, WN = 64, AT = 4, ROW = PS_ROW;
    constexpr int NPL = NTERMS == 3 ? 2 : 1;
    constexpr int RING = ps_ring_bytes<NTERMS>();
    constexpr int NSA = PS_NSA, NSB = PS_NSB;
    constexpr int OFF_B = NSA * RING;
    constexpr int PIECES = 4;                               // 16-row DMA pieces per loader wave and step (64 rows)
    constexpr int NDMA = PIECES * NPL;                      // DMA instructions per loader wave and step
    constexpr unsigned OOB = 0x80000000u;                   // >= num_records of every descriptor (takes_pl: buffers below 2 GiB)
    extern __shared__ __attribute__((aligned(16))) float smem[];      // ONE LDS object (a second one makes hipcc drain the DMA early)
    char* lds = reinterpret_cast<char*>(smem);
    float* sred = reinterpret_cast<float*>(lds + (NSA + NSB) * RING);  // [BM / WM][BN][2] statistics partials of the tile just finished

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // provably wave-uniform: LDS-DMA destinations live in M0
    const int S = (a.Cin + BK - 1) / BK;                              // K-steps per tile (>= 2: launch_gg_ps)
    const int n_tiles = a.n_tiles;
    const int stride_v = (int)gridDim.x;
    if ((int)blockIdx.x >= n_tiles) return;
    const int my_tiles = (n_tiles - 1 - (int)blockIdx.x) / stride_v + 1;
    const int T = my_tiles * S;                                       // K-steps of this block: barriers B(0) .. B(T + 1) in both roles
    const bool do_stats = a.stats != nullptr;

    if (wave >= 4) {
        // =========================================== LOADERS ===========================================
        const int lw = wave - 4;
        const bool isA = lw < 2;
        const int half = lw & 1;                                       // which 64 rows of the 128-row operand tile
        const int NS = isA ? NSA : NSB;
        const int lrow = lane >> 2;                                    // 4 lanes per 64-byte row, 16 rows per DMA instruction
        const int lc = (lane & 3) ^ (((lane >> 4) & 1) << 1);          // logical chunk this lane fetches for its LDS position
        const __amdgpu_buffer_rsrc_t r0 = isA
            ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)a.x_bytes, 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_planes), 0, (int)(a.w_plane_stride * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = isA
            ? __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(a.x_planes) + a.x_plane_stride * 2), 0, (int)a.x_bytes, 0x00020000)
            : r0;
        const unsigned plane1 = isA ? 0u : (unsigned)(a.w_plane_stride * 2);      // B: plane 1 through the same descriptor at this offset
        const unsigned sbase = isA ? 0u : (unsigned)(a.w_off0 * 2);
        char* const dst0 = lds + (isA ? 0 : OFF_B) + (64 * half) * ROW;           // + stage * RING + plane * BM * ROW + 16 i * ROW

        unsigned roff[PIECES];
        int ld_v = (int)blockIdx.x, ld_chunk = 0, ld_stage = 0;
        auto set_tile = [&](int v) {
            const int tile = xcd_remap(v, n_tiles);
            const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int r = 64 * half + 16 * i + lrow;
                if (isA) {
                    const int m = m0 + r;
                    roff[i] = m < a.M ? ((unsigned)m * (unsigned)a.x_pitch + 8u * lc) * 2u : OOB;
                } else {
                    const int n = n0 + r;
                    roff[i] = n < a.N ? ((unsigned)n * (unsigned)a.w_row_stride + 8u * lc) * 2u : OOB;
                }
            }
        };
        // the DMA of one K-step (this wave's 64 rows, NPL planes) into the ring's next stage; masked lanes fetch zeros through an out-of-range offset
        const bool no_dma = (a.dbg_flags & 64) != 0;                      // timing ablations (results are garbage): 64 = no DMA, 128 = no MFMAs, 256 = no stores
        auto issue = [&]() {
            const bool cok = ld_chunk * BK + 8 * lc < a.Cin && !no_dma;      // Cin % 8 == 0; only the last chunk can be partial
            const unsigned so = sbase + (unsigned)(ld_chunk * BK * 2);
            char* const d = dst0 + ld_stage * RING;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const bool ok = cok & (roff[i] != OOB);
                const unsigned vo = ok ? roff[i] : OOB;
                const unsigned vo1 = ok ? roff[i] + plane1 : OOB;
                if (no_dma) continue;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_vptr)(d + 16 * i * ROW), 16, vo, so, 0, 0);
                if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_vptr)(d + BM * ROW + 16 * i * ROW), 16, vo1, so, 0, 0);
            }
            if (++ld_stage == NS) ld_stage = 0;
            if (++ld_chunk == S) {
                ld_chunk = 0;
                ld_v += stride_v;
                if (ld_v < n_tiles) set_tile(ld_v);
            }
        };
        // statistics of a finished tile: the compute waves leave [2][BN][2] partials in sred behind the step's barrier; 128 loader threads add
        // the two row halves and write the tile's row of the partials buffer (gg_pl_kernel's combine, same order)
        auto combine = [&](int v) {
            const int t128 = tid - 256;
            if (t128 < BN) {
                const int tile = xcd_remap(v, n_tiles);
                const int n = (tile % a.tiles_n) * BN + t128;
                if (n < a.N_store) {
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int wm = 0; wm < BM / WM; ++wm) { sm += sred[(wm * BN + t128) * 2]; sq += sred[(wm * BN + t128) * 2 + 1]; }
                    float* dst = a.stats + (size_t)(tile / a.tiles_n) * 2 * a.N_store;
                    dst[n] = sm;
                    dst[a.N_store + n] = sq;
                }
            }
        };

        set_tile(ld_v);
        int issued = 0;                                                  // K-steps requested so far
        for (; issued < NS && issued < T; ++issued) issue();
        // B(k), k = 0 .. T + 1.  Before arriving at B(k) this wave's share of step k must have landed: all but the `issued - k - 1` younger
        // steps.  Behind B(k), k >= 2, every compute wave has finished reading step k - 2, whose stage takes step k - 2 + NS.
        int cv = (int)blockIdx.x, cs = 0;                                // tile / step-in-tile of compute step k - 2 (for the statistics)
        for (int k = 0; k <= T + 1; ++k) {
            if (k < T) {
                const int younger = issued - k - 1;
                ps_wait_vm((younger > 0 ? younger : 0) * NDMA);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 2) {
                if (issued < T) { issue(); ++issued; }
                // compute step k - 2 is complete; if it was a tile's last step, its statistics were written before B(k - 1)... they are written
                // DURING step k - 2's epilogue, i.e. between B(k - 1) and B(k): visible now
                if (++cs == S) {
                    cs = 0;
                    if (do_stats) combine(cv);
                    cv += stride_v;
                }
            }
        }
        return;
    }

    // =========================================== COMPUTE ===========================================
    __builtin_amdgcn_s_setprio(2);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((long long)a.M * a.y_pitch * 4), 0x00020000);      // < 2 GiB: takes_ps
    // fragment reads: lane l = row (l & 15) of a 16-row fragment, reduction elements 8 (l >> 4) .. +7 of the 32-deep step
    const int koff = 16 * ((lane >> 4) ^ (((lane >> 2) & 1) << 1));
    const char* const ra_base = lds + (wave_m * WM + (lane & 15)) * ROW + koff;
    const char* const rb_base = lds + OFF_B + (wave_n * WN + (lane & 15)) * ROW + koff;
    const float scale_a = a.amax_x ? pow2_scale_for(*a.amax_x) : 1.f;
    const float scale_b = a.amax_w ? pow2_scale_for(*a.amax_w) : 1.f;
    const float unscale_a = 1.f / scale_a, unscale_b = 1.f / scale_b;

    f32x4v acc[AT][AT];
    f32x4v acc_lo[NTERMS == 3 ? AT : 1][NTERMS == 3 ? AT : 1];
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (NTERMS == 3) acc_lo[i][j][r] = 0.f;
            }

    int sa = 0, sb = 0;                                       // ring stages of the step whose first fragments are read next
    int cv = (int)blockIdx.x;                                 // tile being multiplied
    constexpr int NPRE = 2;                                   // filter fragments read one step ahead (with the first pixel fragment): what the
                                                              // first MFMAs behind a barrier need; the rest is read right behind the barrier

    auto read_first = [&](f16x8 (&fb)[NPRE][NPL], f16x8 (&fa)[NPL]) {        // first pixel fragment + first NPRE filter fragments of stage (sa, sb)
        const char* pa = ra_base + sa * RING;
        const char* pb = rb_base + sb * RING;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) fa[pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * ROW);
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * ROW + j * 16 * ROW);
    };
    // the filter fragment is the FIRST operand: the 16x16 result comes out transposed (lane l: pixel l & 15, channels 4 (l >> 4) .. +3)
    // -> 16-byte epilogue stores.  Same term order per accumulator as gg_pl_kernel (bit-identical sums).
    // `first`: a tile's first K-step starts from zero (inline-constant C operand) instead of from the accumulators -- no zeroing pass
    const bool no_mfma = (a.dbg_flags & 128) != 0, no_store = (a.dbg_flags & 256) != 0;
    auto mfma1 = [&](auto first, int i, int j, const f16x8 (&fb)[NPL], const f16x8 (&fa)[NPL]) {
        constexpr bool FIRST = decltype(first)::value;
        if (no_mfma) { if (FIRST) { acc[i][j] = fa[0][0] == (_Float16)77.f ? acc[i][j] + 1.f : acc[i][j]; } return; }
        const f32x4v z = {0.f, 0.f, 0.f, 0.f};
        if constexpr (NTERMS == 3) {
            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0], fa[1], FIRST ? z : acc_lo[i][j], 0, 0, 0);
            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1], fa[0], acc_lo[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0], fa[0], FIRST ? z : acc[i][j], 0, 0, 0);
    };

    // ---- tile epilogue: fold, statistics, stores (gg_pl_kernel's arithmetic); ADD: + the ReLU-masked residual gradient ----
    auto epilogue = [&]() {
        const int tile = xcd_remap(cv, n_tiles);
        const int tile_m = tile / a.tiles_n;
        const int m0 = tile_m * BM, n0 = (tile % a.tiles_n) * BN;
        unsigned eoff[AT][AT];                                // element offsets (OOB: nothing stored)
        bool stored[AT][AT];
#pragma unroll
        for (int i = 0; i < AT; ++i) {
            const int row = m0 + wave_m * WM + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < AT; ++j) {
                const int n4 = n0 + wave_n * WN + j * 16 + 4 * (lane >> 4);
                stored[i][j] = row < a.M && n4 < a.N_store;                 // N_store % 4 == 0: all four channels or none
                eoff[i][j] = stored[i][j] ? (unsigned)row * (unsigned)a.y_pitch + (unsigned)n4 : OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int j = 0; j < AT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (NTERMS == 3) acc[i][j][r] = (acc[i][j][r] + acc_lo[i][j][r] * (1.f / 2048.f)) * unscale_a * unscale_b;
                    else acc[i][j][r] = acc[i][j][r] * unscale_a * unscale_b;
                    acc[i][j][r] = acc[i][j][r] + 0.f;                      // gg_pl_kernel adds its (absent) bias here: -0 -> +0 like it
                }
        float* sdst = sred + ((wave_m * BN) + wave_n * WN + 4 * (lane >> 4)) * 2;
        constexpr int PJ = 2;
        auto finish = [&](auto j0c, auto has_prev, const f32x4v (&prev)[AT][PJ]) {
            constexpr int j0 = decltype(j0c)::value;
#pragma unroll
            for (int jj = 0; jj < PJ; ++jj) {
                const int j = j0 + jj;
                float cs4[4] = {0.f, 0.f, 0.f, 0.f}, css4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < AT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float val = acc[i][j][r];
                        if constexpr (decltype(has_prev)::value) val += prev[i][jj][r];
                        acc[i][j][r] = val;
                        const float cvv = stored[i][j] ? val : 0.f;         // statistics of (value - bias), bias absent
                        cs4[r] += cvv;
                        css4[r] += cvv * cvv;
                    }
                if (do_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { cs4[r] = row_sum16(cs4[r]); css4[r] = row_sum16(css4[r]); }
                    if ((lane & 15) == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { sdst[(j * 16 + r) * 2] = cs4[r]; sdst[(j * 16 + r) * 2 + 1] = css4[r]; }
                    }
                }
            }
            // raw buffer stores, unconditional as instructions: rows / channel quads outside the tensor carry an out-of-range offset
#pragma unroll
            for (int jj = 0; jj < PJ; ++jj)
#pragma unroll
                for (int i = 0; i < AT; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, acc[i][j0 + jj]), ry,
                                                           (int)((eoff[i][j0 + jj] == OOB || no_store) ? OOB : eoff[i][j0 + jj] * 4u), 0, 0);
        };
        const f32x4v none[AT][PJ] = {};
        if constexpr (!ADD) {
            finish(std::integral_constant<int, 0>{}, std::false_type{}, none);
            finish(std::integral_constant<int, 2>{}, std::false_type{}, none);
        } else {
            // y = result + relu'(add_src): add_src has y's geometry and pitch (dense), add_mask one nibble per float4 vector
            f32x4v prev[AT][PJ];
            auto fetch = [&](int j0) {
                unsigned mb[AT][PJ];
#pragma unroll
                for (int i = 0; i < AT; ++i)
#pragma unroll
                    for (int jj = 0; jj < PJ; ++jj) {
                        const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
                        const bool ok = stored[i][j0 + jj];
                        prev[i][jj] = ok ? *reinterpret_cast<const f32x4v*>(a.add_src + eoff[i][j0 + jj]) : zero;
                        mb[i][jj] = (ok && a.add_mask != nullptr) ? a.add_mask[eoff[i][j0 + jj] >> 3] : 0xFFu;
                    }
#pragma unroll
                for (int i = 0; i < AT; ++i)
#pragma unroll
                    for (int jj = 0; jj < PJ; ++jj) {
                        const unsigned nib = mb[i][jj] >> (((eoff[i][j0 + jj] >> 2) & 1) * 4);
                        prev[i][jj][0] = (nib & 1u) ? prev[i][jj][0] : 0.f;
                        prev[i][jj][1] = (nib & 2u) ? prev[i][jj][1] : 0.f;
                        prev[i][jj][2] = (nib & 4u) ? prev[i][jj][2] : 0.f;
                        prev[i][jj][3] = (nib & 8u) ? prev[i][jj][3] : 0.f;
                    }
            };
            fetch(0);
            __builtin_amdgcn_sched_barrier(0);
            finish(std::integral_constant<int, 0>{}, std::true_type{}, prev);
            __builtin_amdgcn_sched_barrier(0);
            fetch(2);
            __builtin_amdgcn_sched_barrier(0);
            finish(std::integral_constant<int, 2>{}, std::true_type{}, prev);
        }
    };

    // one K-step: B(g + 1); the remaining fragments of step g; first fragments of step g + 1 -> (fbN, faN); 48 (16) MFMAs
    f16x8 fbC[NPRE][NPL], faC[NPL], fbN[NPRE][NPL], faN[NPL];
    int g = 0;
    auto step = [&](auto first) {
        __builtin_amdgcn_s_barrier();                         // B(g + 1): step g + 1 has landed; the loaders may refill step g - 1's stage
        __builtin_amdgcn_sched_barrier(0);
        const char* pa = ra_base + sa * RING;                 // stage of step g (sa / sb still point at it)
        const char* pb = rb_base + sb * RING;
        if (++sa == NSA) sa = 0;
        if (++sb == NSB) sb = 0;
        f16x8 fb2[NPL], fb3[NPL], fa[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            fb2[pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * ROW + 2 * 16 * ROW);
            fb3[pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * ROW + 3 * 16 * ROW);
        }
        mfma1(first, 0, 0, fbC[0], faC);
        mfma1(first, 0, 1, fbC[1], faC);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) fa[pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * ROW + 1 * 16 * ROW);
        mfma1(first, 0, 2, fb2, faC);
        mfma1(first, 0, 3, fb3, faC);
#pragma unroll
        for (int i = 1; i < AT; ++i) {
            f16x8 fan[NPL];
            if (i + 1 < AT) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fan[pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * ROW + (i + 1) * 16 * ROW);
            }
            if (i == 2 && g + 1 < T) read_first(fbN, faN);    // (sa, sb) now point at step g + 1
            mfma1(first, i, 0, fbC[0], fa);
            mfma1(first, i, 1, fbC[1], fa);
            mfma1(first, i, 2, fb2, fa);
            mfma1(first, i, 3, fb3, fa);
            if (i + 1 < AT) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fa[pl] = fan[pl];
            }
        }
        // the prefetched fragments become the current ones (24 register moves in the shadow of the MFMAs)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            faC[pl] = faN[pl];
#pragma unroll
            for (int j = 0; j < NPRE; ++j) fbC[j][pl] = fbN[j][pl];
        }
        ++g;
    };

    __builtin_amdgcn_s_barrier();                             // B(0): step 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    read_first(fbC, faC);
    for (; cv < n_tiles; cv += stride_v) {
        step(std::true_type{});
        for (int s = 1; s < S; ++s) step(std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        epilogue();
    }
    __builtin_amdgcn_s_barrier();                             // B(T + 1): the last tile's statistics partials are visible to the loaders
}

// OFF / ON: pylc_debug_ps(0 / 1), PYLC_PS=1.  Bit 1 (value 2): also take the dgrads with a masked residual source (ADD).
int g_ps = 0;
extern "C" int pylc_debug_ps(int on) { g_ps = on; return PYLC_OK; }

bool takes_ps(const GatherGemmArgs& a) {
    const long long tiles = (long long)cdiv(a.M, PS_BM) * cdiv(a.N_store, PS_BN);
    if (!(g_ps & 1) || a.w_il || planes_il(a.x_plane_stride) || !a.ident || a.accumulate || a.bias != nullptr || a.ep_scale != nullptr || a.ep_amax != nullptr || a.ep_res != nullptr || a.out_planes2 || a.dbg != nullptr || a.out_half || a.bn_y != nullptr)
        return false;
    if (a.add_src != nullptr && !(g_ps & 2)) return false;
    return a.Cin >= 2 * BK && a.N_store > 64 && tiles >= kNumCU &&      // at least a tile per CU; narrow outputs stay on the NARROW per-tile form
           (long long)a.M * a.y_pitch * 4 < (1ll << 31);                   // y addressed through a 32-bit buffer offset
}

template <int NTERMS>
static void launch_ps(const GatherGemmArgs& a, unsigned grid, hipStream_t st) {
    constexpr int lds_bytes = ps_lds_bytes<NTERMS>();
    if (a.add_src != nullptr) hipLaunchKernelGGL((gg_ps_kernel<NTERMS, true>), dim3(grid), dim3(512), lds_bytes, st, a);
    else hipLaunchKernelGGL((gg_ps_kernel<NTERMS, false>), dim3(grid), dim3(512), lds_bytes, st, a);
}

int launch_gg_ps(GatherGemmArgs& a, hipStream_t st) {
    PYLC_REQUIRE(takes_pl(a) && takes_ps(a), "conv (specialised-wave 1x1 kernel): not a plain 1x1 / stride-1 launch on fp16-plane operands");
    a.tile_bm = PS_BM;
    a.tiles_n = cdiv(a.N_store, PS_BN);
    const long long n_tiles = (long long)cdiv(a.M, PS_BM) * a.tiles_n;
    PYLC_REQUIRE(n_tiles > 0 && n_tiles < (1ll << 31), "conv grid out of range");
    a.n_tiles = (int)n_tiles;
    const unsigned grid = (unsigned)(n_tiles < kNumCU ? n_tiles : kNumCU);
    if (a.nterms == 1) launch_ps<1>(a, grid, st); else launch_ps<3>(a, grid, st);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

template <typename K>
static hipError_t opt_in_ps(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int conv_ps_init() {
    PYLC_HIP(opt_in_ps(gg_ps_kernel<3, false>, ps_lds_bytes<3>()));
    PYLC_HIP(opt_in_ps(gg_ps_kernel<3, true>, ps_lds_bytes<3>()));
    PYLC_HIP(opt_in_ps(gg_ps_kernel<1, false>, ps_lds_bytes<1>()));
    PYLC_HIP(opt_in_ps(gg_ps_kernel<1, true>, ps_lds_bytes<1>()));
    return PYLC_OK;
}

template __global__ void gg_ps_kernel<3, false>(const GatherGemmArgs);
template __global__ void gg_ps_kernel<3, true>(const GatherGemmArgs);
template __global__ void gg_ps_kernel<1, false>(const GatherGemmArgs);
template __global__ void gg_ps_kernel<1, true>(const GatherGemmArgs);

}  // namespace pylc
