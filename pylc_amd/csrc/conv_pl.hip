// Implicit-GEMM convolution (forward and dgrad "gather GEMM") whose operands arrive PRE-SPLIT as fp16 planes.
//
// conv_igemm.hip's f16x3 kernels read fp32 activations and split every element into two scaled fp16 pieces inside the main
// loop (64 vector instructions per wave and K-step next to 48 MFMAs: the matrix pipe was busy half the time).  Here the
// producer of a tensor (BatchNorm-apply, BatchNorm-backward, pylc_to_planes) has already written
//      plane 0: h0 = rn16(s x)            plane 1: h1 = rn16(2^11 (s x - h0))           s = 2^k from the tensor's range bound
// (4 bytes per element, as fp32) and the filter planes come from pylc_weight_prepare, so BOTH operand tiles are plain copies:
// they go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), no staging registers, no ds_write, no vector ALU work.
// The wave's main loop is: 8 DMA issues, 16 ds_read_b128 fragment reads, 48 MFMAs (NTERMS = 3) or 16 (NTERMS = 1).
//
// Two tile shapes, one body (BM = 128 / 256 pixel rows x 128 channels x 32 deep; every wave owns 64 x 64 outputs on 4 x 4
// v_mfma_f32_16x16x32_f16 tiles):
//   * 128 x 128, 4 waves, two LDS stages of 32 KB, TWO blocks per CU.  The blocks are independent, so one block's prologue
//     (geometry, first operand round trip) and epilogue (stores) overlap the other's main loop -- what a one-block-per-CU kernel
//     cannot hide on the short-K 1x1 convs.  DMA runs one K-step ahead.
//   * 256 x 128, 8 waves, three LDS stages of 48 KB, one block per CU.  DMA runs TWO K-steps ahead behind a counted vmcnt and one
//     raw s_barrier per step.  25 % fewer operand bytes per MFMA: what counts on the long reductions, because the L2 -> LDS path
//     (~70 GB/s per CU in practice, MI355X_MICROARCH.md "Indexed rows: gather into LDS") is what bounds the 128 x 128 tile
//     (32 KB per 192 MFMAs = 80-100 GB/s per CU at the full matrix rate).
//
// What bounds it (tools/pl_ablate.py, pl_stamps.py on 3x3 256->256 @128^2, bs 32): the MFMA phase alone runs at 520 TFLOP/s
// (algorithmic; the chip holds ~1.5-1.9 GHz under it), the operand DMA alone takes as long (19-20 TB/s L2 -> LDS chip-wide is
// all the CUs take: 32 KB per block and K-step), and the two overlap only partly -- a wave whose DMA issue blocks on the full
// queue issues no MFMAs.  Measured and dropped: handing a stage back in the middle of a step (second barrier, DMA 1.7 steps
// ahead with two stages); spreading the DMA pieces over the MFMA phase (one or two per 12 MFMAs).  Both neutral: the limit is
// bytes per MFMA, not latency or burstiness.
//
// LDS image: unpadded 64-byte rows (32 halves), 16-byte chunk index XORed with ((row >> 2) & 1) << 1: conflict-free for the
// 16-row ds_read_b128 fragments.  LDS-DMA writes lane-linearly (wave base + lane * 16), so the swizzle is applied on the SOURCE
// address (each lane fetches the chunk that belongs at its position) and again on the read.
//
// NTERMS = 3: a0 b0 + 2^-11 (a1 b0 + a0 b1), identical instruction order to conv_igemm.hip's kernels -> bit-identical results.
// NTERMS = 1: plain fp16 operands (plane 0 only), fp32 accumulation: precision mode 3 (BASELINE configs[4] "bf16 MFMA" class).
//
// Reference call sites replaced: as conv_igemm.hip (nn.Conv2d forward/backward in models/backbone/resnet.py:21-26,72,92,
// models/modules/aspp.py:18,64,67, models/decoder.py:27-38, models/backbone/xception.py:32,48,122,126).
#include "conv_common.h"

namespace pylc {

typedef __attribute__((address_space(3))) void* lds_vptr;

constexpr int PL_BN = 128;
constexpr int PL_ROW = 64;                                  // bytes per LDS row per plane
// gg_pl_kernel's LDS row: 64 bytes (a 32-deep K-step) for the two-plane f16x3 operands; 128 bytes (a 64-deep K-step, the same stage
// size) for the one-plane operands of precision mode 3.  Why: a 64-byte piece is HALF of a 128-byte cache line, and the other half -- the
// next K-step's chunk -- is requested again one step later, after the 32 KB L1 has turned over; tools/micro/dma_piece.hip measures the
// L2 -> LDS rate of LDS-DMA at 21.8 TB/s for 16 rows x 64 B per instruction against 33.0 TB/s for 8 rows x 128 B
// (profiles/r04_dma_piece.txt).  The one-plane loop is bound by exactly that rate (16 KB per 16 MFMAs per wave); the f16x3 loop has
// no LDS left for 128-byte rows (two blocks x two stages x 64 KB).
template <int NTERMS>
constexpr int pl_row_bytes() { return NTERMS == 1 ? 128 : PL_ROW; }
template <int NTERMS, int BM>
constexpr int pl_stage_bytes() { return (NTERMS == 3 ? 2 : 1) * (BM + PL_BN) * pl_row_bytes<NTERMS>(); }
template <int BM>
constexpr int pl_stages() { return BM == 256 ? 3 : 2; }
template <int NTERMS, int BM>
constexpr int pl_lds_bytes() { return pl_stages<BM>() * pl_stage_bytes<NTERMS, BM>() + BM * 4 + 64; }

// ---- the LEAN epilogue: what a tile of the training step almost always is ----------------------------------------------------------
// Round 4's ablation (profiles/r04_ps_ab.txt) put the short-K class's time outside the MFMAs on the epilogue's ARITHMETIC: ~1500 vector
// instructions per wave and tile -- as long as the 384 MFMAs of a K = 256 tile.  Most of them served cases a ResNet / Xception tile never is:
// a per-element bias add and subtract, rows or channels past the tensor's edge (16 `stored` selects per element group), and they applied
// the two unscale factors and the bias BEFORE taking the statistics.  This path is taken, wave by wave, when the tile has no edge:
//     t   = acc + 2^-11 lo                      one fma (two per instruction where PYLC_EPI_PK packs them; done by pl_epilogue for both paths)
//     sum += t,  sumsq += t t                   on the RAW accumulators; the scale c = 1 / (s_x s_w) is a power of two, so
//                                               c sum(t) == sum(c t) and c^2 sum(t t) == sum((c t)^2) bit for bit (same order of additions)
//                                               and it is applied to the 8 reduced column sums of a 16-row fragment instead of to 64 elements
//     y   = c t (+ masked residual gradient)    one multiply, unconditional 16-byte stores
// One-plane output (precision mode 3): h = rn16((c s_y) t) is formed ONCE, stored as it is, and the statistics the following BatchNorm
// needs -- those of the ROUNDED tensor -- are taken from h by mixed-precision fmas (h h is exact in fp32), scaled by 1 / s_y at the end.
// Everything else (ragged tiles, accumulate + statistics, the BatchNorm-backward and inference epilogues) takes the general path
// below, unchanged.  Results are bit-identical to it (tests/test_planes_gpu.py pins both against the fp32-operand kernels).
#ifndef PYLC_EPI_FULL_LINES
#define PYLC_EPI_FULL_LINES 1      // 0: each lane stores its own quads (16 rows x 64 B per instruction; A/B build)
#endif
#ifndef PYLC_EPI_PK
#define PYLC_EPI_PK 0      // 1: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 forms (A/B: tools/epi_ab.sh)
#endif
// (vector-typed f32 arithmetic is legalised into v_pk_* instructions on gfx950 whatever -fno-slp-vectorize says, so the scalar forms
// are written out element by element)
__device__ __forceinline__ f32x4v epi_fma(f32x4v a, float b, f32x4v c) {
#if PYLC_EPI_PK
    const f32x4v bb = {b, b, b, b};
    return __builtin_elementwise_fma(a, bb, c);
#else
    return f32x4v{__builtin_fmaf(a[0], b, c[0]), __builtin_fmaf(a[1], b, c[1]), __builtin_fmaf(a[2], b, c[2]), __builtin_fmaf(a[3], b, c[3])};
#endif
}
__device__ __forceinline__ f32x4v epi_mul(f32x4v a, float b) {
#if PYLC_EPI_PK
    return a * b;
#else
    return f32x4v{a[0] * b, a[1] * b, a[2] * b, a[3] * b};
#endif
}
__device__ __forceinline__ f32x4v epi_sq(f32x4v a) {
#if PYLC_EPI_PK
    return a * a;
#else
    return f32x4v{a[0] * a[0], a[1] * a[1], a[2] * a[2], a[3] * a[3]};
#endif
}
__device__ __forceinline__ f32x4v epi_add(f32x4v a, f32x4v b) {
#if PYLC_EPI_PK
    return a + b;
#else
    return f32x4v{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
#endif
}

// The K-step's instruction order, said to the scheduler.  Left to itself hipcc reads six fragments, waits for ALL of them, issues four MFMAs, reads four
// more, waits ... -- an LDS round trip in front of every few products, on a wave that shares its SIMD with one other wave only.  With the
// groups below a step is: the filter fragments and the pixel fragments of rows 0 and 1 (one wait), row 0's MFMAs, row 2's fragments, row 1's MFMAs, row
// 3's fragments, the rest -- every later fragment lands behind 12 MFMAs (+3-7 % on every 3x3 launch of gg_plhn_kernel: profiles/r06_sched_groups_ab.txt).
// Same instructions, same MFMA order per accumulator: bit-identical.  AM: 16-row pixel fragments per wave (4, or 2 in the NARROW form).
template <int AM, int NPL, int NTERMS>
__device__ __forceinline__ void pl_step_schedule() {
    constexpr int AT = 4;
    if constexpr (AM >= 3) {
        __builtin_amdgcn_sched_group_barrier(0x100, (AT + 2) * NPL, 0);      // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, AT * NTERMS, 0);         // MFMAs
#pragma unroll
        for (int r = 2; r < AM; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, AT * NTERMS, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, AT * NTERMS, 0);
    } else {
        __builtin_amdgcn_sched_group_barrier(0x100, (AT + AM) * NPL, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, AM * AT * NTERMS, 0);
    }
}

// STATS: 0 none, 1 column sums / sums of squares.  PREV: 0 none, 1 `extra` added (accumulate / residual-gradient source), 2 ... under the
// 1-bit mask `amask`.  HALF: one-plane fp16 output (NTERMS == 1 launches only).
template <int AM, int STATS, int PREV, bool HALF>
__device__ __forceinline__ void pl_epilogue_lean(const GatherGemmArgs& a, f32x4v (&acc)[AM][4], const int* rowoff,
                                                 float* sdst, int nb, int wave_m, int lane, float c, float hscale,
                                                 const float* extra, const unsigned char* amask, const float* bias) {
    constexpr int AT = 4, WM = 16 * AM;
    typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    unsigned offs[AM];                                      // element offsets of this lane's channel quad in its AM rows (all valid here)
#pragma unroll
    for (int i = 0; i < AM; ++i) offs[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + (lane & 15)] + nb);
    // FULL-LINE STORES (fp32 output).  The accumulator layout gives a store instruction 16 rows x 64 B -- half of each 128-byte line, the other
    // half following four instructions later.  Lanes l and l ^ 8 (rows r and r + 8 of a fragment, same channel quad) swap one value each per
    // pair of column groups -- the lower lane's odd group against the upper lane's even group -- so that an instruction writes 8 rows x 128 B:
    // rows 0-7 first (lower lanes their own even group, upper lanes the odd group they received), then rows 8-15.
    // tools/micro/store_patterns.hip: whole-line patterns are 8-12 % ahead on pure stores, and a short-K tile's time is its stores (DESIGN 5.2 a).
    constexpr bool FULL = !HALF && PYLC_EPI_FULL_LINES != 0;
    unsigned offsP[FULL ? AM : 1];                          // ... of the partner row (r ^ 8)
    if constexpr (FULL) {
#pragma unroll
        for (int i = 0; i < AM; ++i) offsP[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + ((lane & 15) ^ 8)] + nb);
    }
    const bool upper = (lane & 8) != 0;
    uint2 hq[(HALF && PYLC_EPI_FULL_LINES != 0) ? AM : 1][4];      // one-plane output: this lane's halves of all four column groups of a row fragment
    f32x4v hold[FULL ? AM : 1];                             // the even column group's values, waiting for their odd neighbour
    f32x4v prevn[(FULL && PREV != 0) ? AM : 1];             // the odd column group's residual-gradient values, fetched with the even group's
    const float k = HALF ? c * hscale : c;                  // powers of two: exact
    const float post = HALF ? pow2_inv(hscale) : c;         // what the column sums are multiplied with (sums of squares: twice)
    const bool odd_row = (lane >> 4) & 1;
#pragma unroll
    for (int j = 0; j < AT; ++j) {
        f32x4v prev[PREV ? AM : 1];
        if constexpr (PREV != 0) {
            unsigned mb[AM];
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                if constexpr (!FULL) prev[i] = *reinterpret_cast<const f32x4v*>(extra + (offs[i] + j * 16));
                if constexpr (PREV == 2) mb[i] = amask[(offs[i] + j * 16) >> 3];
            }
            if constexpr (FULL) {
                // full-line LOADS of the residual-gradient source, the stores' swap in reverse: at an even column group both groups of the pair
                // are fetched as 8 rows x 128 B per instruction (lower lanes their even group, upper lanes the lower rows' odd group; then rows
                // 8-15), and lanes l / l ^ 8 hand each other what belongs to the other
                if ((j & 1) == 0) {
                    const unsigned col = (upper ? j + 1 : j) * 16;
                    f32x4v la[AM], lb[AM];
#pragma unroll
                    for (int i = 0; i < AM; ++i) {
                        la[i] = *reinterpret_cast<const f32x4v*>(extra + ((upper ? offsP[i] : offs[i]) + col));      // rows 0-7 of the fragment
                        lb[i] = *reinterpret_cast<const f32x4v*>(extra + ((upper ? offs[i] : offsP[i]) + col));      // rows 8-15
                    }
#pragma unroll
                    for (int i = 0; i < AM; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float send = upper ? la[i][r] : lb[i][r];
                            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xF, 0xF, true));
                            prev[i][r] = upper ? recv : la[i][r];          // this lane's even group
                            prevn[i][r] = upper ? lb[i][r] : recv;         // ... and its odd group, for the next iteration
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < AM; ++i) prev[i] = prevn[i];
                }
            }
            if constexpr (PREV == 2) {
#pragma unroll
                for (int i = 0; i < AM; ++i) {
                    // the quad's four mask bits -> all-ones / zero words (v_bfe_i32), ANDed onto the residual gradient
                    const int nib = (int)(mb[i] >> ((((offs[i] + j * 16) >> 2) & 1u) * 4u));
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        prev[i][r] = __uint_as_float(__float_as_uint(prev[i][r]) & (unsigned)__builtin_amdgcn_sbfe(nib, r, 1));
                }
            }
        }
        // a bias (the U-Net's convs) rides in the fma that applies the scale: c t is exact, so fma(t, c, b) == c t + b in one rounding, as the
        // general path's mul-then-add; the statistics are those of (value - bias) = c t either way
        f32x4v bv = {0.f, 0.f, 0.f, 0.f};
        if (!HALF && bias != nullptr) bv = *reinterpret_cast<const f32x4v*>(bias + nb + j * 16);
        f32x4v cs = {0.f, 0.f, 0.f, 0.f}, css = cs;
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            const f32x4v t = acc[i][j];                      // (cross terms already folded in by the caller)
            if constexpr (HALF) {
                const f32x4v v = epi_mul(t, k);
                const f16x4_ h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                if constexpr (STATS != 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {            // v_fma_mix_f32: (float)h is exact, h h is exact -> the sums of the rounded tensor
                        cs[r] = i == 0 ? (float)h[r] : __builtin_fmaf((float)h[r], 1.f, cs[r]);
                        css[r] = i == 0 ? (float)h[r] * (float)h[r] : __builtin_fmaf((float)h[r], (float)h[r], css[r]);
                    }
                }
                if constexpr (PYLC_EPI_FULL_LINES != 0) {
                    hq[i][j] = __builtin_bit_cast(uint2, h);        // stored after the last column group, whole lines at a time (below)
                } else {
                    // lanes l and l + 16 (same pixel, the next channel quad) swap halves: even DPP rows store 8 channels = 16 bytes
                    const uint2 hu = __builtin_bit_cast(uint2, h);
                    const u32x2_ sx = __builtin_amdgcn_permlane16_swap(hu.x, hu.x, false, false);
                    const u32x2_ sy = __builtin_amdgcn_permlane16_swap(hu.y, hu.y, false, false);
                    if (!odd_row) *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(a.y) + (offs[i] + j * 16)) = uint4{hu.x, hu.y, sx.y, sy.y};
                }
            } else {
                if constexpr (STATS != 0) {
                    cs = i == 0 ? t : epi_add(cs, t);        // (0 + t == t: the general path's order of additions)
                    css = i == 0 ? epi_sq(t) : epi_add(css, epi_sq(t));
                }
                f32x4v v = epi_fma(t, k, bv);
                if constexpr (PREV != 0) v = epi_add(v, prev[i]);
                if constexpr (!FULL) {
                    *reinterpret_cast<f32x4v*>(a.y + (offs[i] + j * 16)) = v;
                } else if ((j & 1) == 0) {
                    hold[i] = v;
                } else {
                    f32x4v recv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float send = upper ? hold[i][r] : v[r];          // upper lanes give their even group, lower lanes their odd group
                        recv[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xF, 0xF, true));      // row_ror:8 = lane ^ 8
                    }
                    const unsigned col = (upper ? j : j - 1) * 16;
                    f32x4v da, db;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { da[r] = upper ? recv[r] : hold[i][r]; db[r] = upper ? v[r] : recv[r]; }
                    *reinterpret_cast<f32x4v*>(a.y + ((upper ? offsP[i] : offs[i]) + col)) = da;      // rows 0-7 of the fragment, 128 B each
                    *reinterpret_cast<f32x4v*>(a.y + ((upper ? offs[i] : offsP[i]) + col)) = db;      // rows 8-15
                }
            }
        }
        if constexpr (STATS != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { cs[r] = row_sum16(cs[r]); css[r] = row_sum16(css[r]); }
            if ((lane & 15) == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { sdst[(j * 16 + r) * 2] = cs[r] * post; sdst[(j * 16 + r) * 2 + 1] = css[r] * post * post; }
            }
        }
    }
    if constexpr (HALF && PYLC_EPI_FULL_LINES != 0) {
        // One-plane output, whole lines: a lane holds 4 channels = 8 bytes per column group, a fragment row 64 channels = ONE 128-byte line.
        //  (1) per pair of column groups, v_permlane16_swap between lanes l and l + 16 (adjacent channel quads): lanes of even DPP rows end up
        //      with 8 channels of the even group, lanes of odd rows with 8 channels of the odd group -- 16 bytes in EVERY lane, and the pair's 32
        //      channels = 64 contiguous bytes per pixel row;
        //  (2) lanes l / l ^ 8 (rows r and r + 8) swap the second pair of the lower rows against the first pair of the upper rows, as the fp32
        //      stores do, so that an instruction writes rows 0-7 (then 8-15) of the fragment in full: 8 rows x 128 B.
        // Two stores per row fragment instead of four 16-rows-x-32-byte ones issued by half the lanes.
        unsigned offsP[AM];
#pragma unroll
        for (int i = 0; i < AM; ++i) offsP[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + ((lane & 15) ^ 8)] + nb - 4 * (lane >> 4));
        const unsigned q4 = (unsigned)(lane >> 4);                 // this lane's channel quad
        // after (1): even rows hold the even group's channels 16 jp*2 + 4 q .. + 7, odd rows the odd group's 16 (2 jp + 1) + 4 (q - 1) .. + 7
        const unsigned colq = odd_row ? 16u + 4u * (q4 - 1u) : 4u * q4;      // channel offset inside the pair's 32 channels
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            // (scalars, not a two-element array: `upper ? pr[0] : pr[1]` on an array makes hipcc index it in scratch memory)
            const u32x2_ ax = __builtin_amdgcn_permlane16_swap(hq[i][0].x, hq[i][1].x, false, false);
            const u32x2_ ay = __builtin_amdgcn_permlane16_swap(hq[i][0].y, hq[i][1].y, false, false);
            const u32x2_ bx = __builtin_amdgcn_permlane16_swap(hq[i][2].x, hq[i][3].x, false, false);
            const u32x2_ by = __builtin_amdgcn_permlane16_swap(hq[i][2].y, hq[i][3].y, false, false);
            // pair 0 = (ax.x, ay.x, ax.y, ay.y), pair 1 = (bx.x, by.x, bx.y, by.y): 8 consecutive channels each
            const unsigned s0 = upper ? ax.x : bx.x, s1 = upper ? ay.x : by.x, s2 = upper ? ax.y : bx.y, s3 = upper ? ay.y : by.y;      // upper sends pair 0, lower pair 1
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0x128, 0xF, 0xF, true);
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0x128, 0xF, 0xF, true);
            const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s2, 0x128, 0xF, 0xF, true);
            const unsigned r3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s3, 0x128, 0xF, 0xF, true);
            const unsigned base_own = offs[i] - 4u * q4, base_par = offsP[i];      // element offset of the fragment row's first channel (own / partner row)
            const unsigned col = (upper ? 32u : 0u) + colq;
            const uint4 da = {upper ? r0 : ax.x, upper ? r1 : ay.x, upper ? r2 : ax.y, upper ? r3 : ay.y};      // lower: own pair 0; upper: the lower row's pair 1
            const uint4 db = {upper ? bx.x : r0, upper ? by.x : r1, upper ? bx.y : r2, upper ? by.y : r3};      // lower: the upper row's pair 0; upper: own pair 1
            *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(a.y) + ((upper ? base_par : base_own) + col)) = da;      // rows 0-7
            *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(a.y) + ((upper ? base_own : base_par) + col)) = db;      // rows 8-15
        }
    }
}

// The BatchNorm-backward mode (BNB kernels, pylc_conv2d_dgrad_bn: the tile being stored is the gradient `dout` of a BatchNorm OUTPUT, so the
// sums that BatchNorm's backward needs first -- sum g and sum g xhat with g = relu'(dout), xhat = (y_bn - mean) invstd -- are taken here, from
// the registers, instead of by a read pass over (dout, y_bn) (bn.hip bn_reduce_kernel<1>).  The y_bn tile has the output's geometry (dense).
// Per element: y = c t (+ masked residual gradient), g = relu' y, sum g xhat, sum g, max |g| (the range bound of the dy to come).  Partial row
// layout = bn_reduce_kernel<1>'s: [sum g xhat | sum g].  ReLU by the 1-bit mask of bn.hip, or recomputed as y_bn scale + shift > 0 (the
// forward's own expression), or none.  EDGE: rows / channel quads past the tensor's edge are neither loaded, stored nor counted.
// Returns max |g| of this lane.
template <int AM, int PREV, bool EDGE>
__device__ __forceinline__ float pl_epilogue_bn(const GatherGemmArgs& a, f32x4v (&acc)[AM][4], const int* rowoff, float* sdst, int nb, int wave_m,
                                                int lane, float c, const float* extra, const unsigned char* amask) {
    constexpr int AT = 4, WM = 16 * AM;
    int offs[AM];                       // element offset of this lane's channel quad in row i (< 0: no such row)
#pragma unroll
    for (int i = 0; i < AM; ++i) {
        const int ro = rowoff[wave_m * WM + i * 16 + (lane & 15)];
        offs[i] = (EDGE && ro < 0) ? -(1 << 30) : ro + nb;
    }
    float gmax = 0.f;
    const bool relu = a.bn_relu != 0, bits = a.bn_mask != nullptr;
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < AT; ++j) {
        const bool cok = !EDGE || nb + j * 16 < a.N_store;          // N_store % 4 == 0: all four channels or none
        f32x4v mu = zero, is = zero, sc = zero, sf = zero;
        if (cok) {
            mu = *reinterpret_cast<const f32x4v*>(a.bn_mean + nb + j * 16);
            is = *reinterpret_cast<const f32x4v*>(a.bn_invstd + nb + j * 16);
            if (relu && !bits) { sc = *reinterpret_cast<const f32x4v*>(a.bn_scale + nb + j * 16); sf = *reinterpret_cast<const f32x4v*>(a.bn_shift + nb + j * 16); }
        }
        f32x4v prev[PREV ? AM : 1], yv[AM];
        unsigned mb[PREV == 2 ? AM : 1], ym[AM];
        bool ok[AM];
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            ok[i] = !EDGE || (cok && offs[i] >= 0);
            const unsigned e = (unsigned)(offs[i] + j * 16);
            yv[i] = ok[i] ? *reinterpret_cast<const f32x4v*>(a.bn_y + e) : zero;
            ym[i] = (relu && bits && ok[i]) ? (unsigned)a.bn_mask[e >> 3] : 0xFFu;
            if constexpr (PREV != 0) prev[i] = ok[i] ? *reinterpret_cast<const f32x4v*>(extra + e) : zero;
            if constexpr (PREV == 2) mb[i] = ok[i] ? amask[e >> 3] : 0u;
        }
        f32x4v cs = zero, css = zero;
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            const unsigned e = (unsigned)(offs[i] + j * 16);
            const unsigned sh = ((e >> 2) & 1u) * 4u;
            f32x4v v = epi_mul(acc[i][j], c);
            if constexpr (PREV == 2) {
                const int nib = (int)(mb[i] >> sh);
#pragma unroll
                for (int r = 0; r < 4; ++r) prev[i][r] = __uint_as_float(__float_as_uint(prev[i][r]) & (unsigned)__builtin_amdgcn_sbfe(nib, r, 1));
            }
            if constexpr (PREV != 0) v = epi_add(v, prev[i]);
            if (ok[i]) *reinterpret_cast<f32x4v*>(a.y + e) = v;
            int on = (int)(ym[i] >> sh);
            if (relu && !bits) {
                on = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) on |= (yv[i][r] * sc[r] + sf[r] > 0.f) ? (1 << r) : 0;
            }
            if (EDGE && !ok[i]) on = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = __uint_as_float(__float_as_uint(v[r]) & (unsigned)__builtin_amdgcn_sbfe(on, r, 1));
                const float gx = g * ((yv[i][r] - mu[r]) * is[r]);
                cs[r] = i == 0 ? gx : cs[r] + gx;          // sum g xhat (0 + x == x)
                css[r] = i == 0 ? g : css[r] + g;          // sum g
                gmax = fmaxf(gmax, fabsf(g));
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[r] = row_sum16(cs[r]); css[r] = row_sum16(css[r]); }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { sdst[(j * 16 + r) * 2] = cs[r]; sdst[(j * 16 + r) * 2 + 1] = css[r]; }
        }
        __builtin_amdgcn_sched_barrier(0);      // one column group's loads at a time (hoisting all four spills)
    }
    return gmax;
}

// The LEAN form of the fused INFERENCE epilogue (EP kernels, pylc_conv2d_fwd_bnact_ex): an edge-free tile whose result -- act(conv * scale +
// shift (+ residual)) -- leaves as a chunk-interleaved two-plane tensor (plane_stride == 32: a pixel's 32 channels = 64 B of plane 0 + 64 B of
// plane 1 = one 128-byte line), with the residual, if any, in the same format.  Same expressions and order per element as the general path
// (bit-identical; tests/test_planes_gpu.py compares the two), but
//   * no per-element edge selects,
//   * WHOLE-LINE stores: per pair of column groups (one 32-channel chunk) v_permlane16_swap between lanes l / l + 16 gives every lane 16 B of
//     one plane (8 channels), then lanes l / l ^ 8 (rows r / r + 8) swap the lower row's second group against the upper row's first, as the
//     training epilogue does: a store instruction writes 8 rows x 128 B instead of 16 rows x 2 x 32 B;
//   * the residual planes are FETCHED as whole lines by the same exchange in reverse.
// 16-byte piece p (0..7) of a chunk's line: pieces 0-3 = plane 0 channels 8 p .., pieces 4-7 = plane 1; lane quad q = lane >> 4 after the
// permlane16 exchange holds, for column group j' of the pair, piece (q & 1) * 4 + (q >> 1) + 2 j'.
// RES: 0 none, 2 two-plane interleaved residual.  Returns max |result| of this lane.
template <int AM, int RES, bool BIAS>
__device__ __forceinline__ float pl_epilogue_lean_ep(const GatherGemmArgs& a, f32x4v (&acc)[AM][4], const int* rowoff, int nchunk, int wave_m, int lane,
                                                     float c, float hscale, float res_unscale) {
    constexpr int WM = 16 * AM;
    typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const unsigned q4 = (unsigned)(lane >> 4);
    const bool upper = (lane & 8) != 0;
    // physical half offset of a row's first line of this wave's columns: element offset (a multiple of 32) x 2
    unsigned base[AM], baseP[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i) {
        base[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + (lane & 15)] + nchunk) * 2u;
        baseP[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + ((lane & 15) ^ 8)] + nchunk) * 2u;
    }
    // this lane's piece in the two store / load instructions of a (row fragment, chunk): lower lanes handle column group 0 of the pair, upper lanes group 1
    const unsigned piece = ((q4 & 1u) * 4u + (q4 >> 1) + (upper ? 2u : 0u)) * 8u;      // in halves
    const int nb = nchunk + 4 * (int)q4;
    _Float16* const yp = reinterpret_cast<_Float16*>(a.y);
    const _Float16* const rp = reinterpret_cast<const _Float16*>(a.ep_res);
    const bool relu = a.ep_relu != 0;
    float ep_max = 0.f;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        uint4 ra[RES ? AM : 1], rb[RES ? AM : 1];
        if constexpr (RES != 0) {
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                ra[i] = *reinterpret_cast<const uint4*>(rp + ((upper ? baseP[i] : base[i]) + jp * 64 + piece));      // rows 0-7 of the fragment
                rb[i] = *reinterpret_cast<const uint4*>(rp + ((upper ? base[i] : baseP[i]) + jp * 64 + piece));      // rows 8-15
            }
        }
        f32x4v esc[2], esh[2], bv[BIAS ? 2 : 1];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            esc[jj] = *reinterpret_cast<const f32x4v*>(a.ep_scale + nb + (2 * jp + jj) * 16);
            esh[jj] = *reinterpret_cast<const f32x4v*>(a.ep_shift + nb + (2 * jp + jj) * 16);
            if constexpr (BIAS) bv[jj] = *reinterpret_cast<const f32x4v*>(a.bias + nb + (2 * jp + jj) * 16);      // (the U-Net's convs carry a bias)
        }
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            uint4 qa = {0u, 0u, 0u, 0u}, qb = {0u, 0u, 0u, 0u};      // residual pieces of this lane's own row: column group 0 / 1 of the pair
            if constexpr (RES != 0) {
                const unsigned s0 = upper ? ra[i].x : rb[i].x, s1 = upper ? ra[i].y : rb[i].y, s2 = upper ? ra[i].z : rb[i].z, s3 = upper ? ra[i].w : rb[i].w;
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0x128, 0xF, 0xF, true);      // row_ror:8 = lane ^ 8
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0x128, 0xF, 0xF, true);
                const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s2, 0x128, 0xF, 0xF, true);
                const unsigned r3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s3, 0x128, 0xF, 0xF, true);
                qa = uint4{upper ? r0 : ra[i].x, upper ? r1 : ra[i].y, upper ? r2 : ra[i].z, upper ? r3 : ra[i].w};
                qb = uint4{upper ? rb[i].x : r0, upper ? rb[i].y : r1, upper ? rb[i].z : r2, upper ? rb[i].w : r3};
            }
            uint4 oa, ob;                                                // output pieces of this lane's own row: column group 0 / 1
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                f32x4v val;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r] * c;                          // (the general path's acc * unscale_a * unscale_b: powers of two)
                    if constexpr (BIAS) t += bv[jj][r];                  // (... + bias; without one the general path adds an exact 0)
                    val[r] = t * esc[jj][r] + esh[jj][r];                // BatchNorm-apply's own expression and order
                }
                if constexpr (RES != 0) {
                    const uint4 q = jj == 0 ? qa : qb;
                    const u32x2_ s0 = __builtin_amdgcn_permlane16_swap(q.x, q.z, false, false);
                    const u32x2_ s1 = __builtin_amdgcn_permlane16_swap(q.y, q.w, false, false);
                    const f16x4_ h0 = __builtin_bit_cast(f16x4_, uint2{s0.x, s1.x});
                    const f16x4_ h1 = __builtin_bit_cast(f16x4_, uint2{s0.y, s1.y});
                    const f32x4v rv = (__builtin_convertvector(h0, f32x4v) + __builtin_convertvector(h1, f32x4v) * (1.f / 2048.f)) * res_unscale;
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[r] += rv[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (relu) val[r] = fmaxf(val[r], 0.f);
                    ep_max = fmaxf(ep_max, fabsf(val[r]));
                }
                uint2 p0, p1;
                split2(val, hscale, p0, p1);
                const u32x2_ sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
                const u32x2_ sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
                // even quads: (own p0, partner's p0) -> 8 channels of plane 0;  odd quads: (partner's p1, own p1) -> plane 1 at the partner's channels
                if (jj == 0) oa = uint4{sx.x, sy.x, sx.y, sy.y}; else ob = uint4{sx.x, sy.x, sx.y, sy.y};
            }
            // lanes l / l ^ 8: the upper row gives its group-0 piece, the lower row its group-1 piece
            const unsigned s0 = upper ? oa.x : ob.x, s1 = upper ? oa.y : ob.y, s2 = upper ? oa.z : ob.z, s3 = upper ? oa.w : ob.w;
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0x128, 0xF, 0xF, true);
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0x128, 0xF, 0xF, true);
            const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s2, 0x128, 0xF, 0xF, true);
            const unsigned r3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s3, 0x128, 0xF, 0xF, true);
            const uint4 da = {upper ? r0 : oa.x, upper ? r1 : oa.y, upper ? r2 : oa.z, upper ? r3 : oa.w};      // lower: own group 0; upper: the lower row's group 1
            const uint4 db = {upper ? ob.x : r0, upper ? ob.y : r1, upper ? ob.z : r2, upper ? ob.w : r3};      // lower: the upper row's group 0; upper: own group 1
            *reinterpret_cast<uint4*>(yp + ((upper ? baseP[i] : base[i]) + jp * 64 + piece)) = da;      // rows 0-7 of the fragment, 128 B each
            *reinterpret_cast<uint4*>(yp + ((upper ? base[i] : baseP[i]) + jp * 64 + piece)) = db;      // rows 8-15
        }
        __builtin_amdgcn_sched_barrier(0);      // one chunk's residual loads at a time
    }
    return ep_max;
}

// ... and its ONE-PLANE form (precision mode 3: out_half, residual none or one plane of the same geometry).  A fragment row's 64 channels are
// one 128-byte run; stores and residual loads move whole runs by the exchanges of pl_epilogue_lean<.., HALF = true>: v_permlane16_swap merges the
// quads of lanes l / l + 16 into 8 consecutive channels (pair 0 = column groups 0-1, pair 1 = groups 2-3), lanes l / l ^ 8 swap pair 1 of the
// lower row against pair 0 of the upper row.  Per element the general path's expressions: val = act((c t) scale + shift (+ residual)),
// h = rn16(val s_y).  RES: 0 none, 1 one-plane residual.  Returns max |val| of this lane.
template <int AM, int RES, bool BIAS>
__device__ __forceinline__ float pl_epilogue_lean_ep_half(const GatherGemmArgs& a, f32x4v (&acc)[AM][4], const int* rowoff, int nchunk, int wave_m, int lane,
                                                          float c, float hscale, float res_unscale) {
    constexpr int WM = 16 * AM;
    typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const unsigned q4 = (unsigned)(lane >> 4);
    const bool upper = (lane & 8) != 0, odd_row = (q4 & 1u) != 0;
    // after the pair merge: even quads hold channels 4 q .. + 7 of the pair's first group, odd quads 16 + 4 (q - 1) .. + 7 (the second group)
    const unsigned col = (upper ? 32u : 0u) + (odd_row ? 16u + 4u * (q4 - 1u) : 4u * q4);
    unsigned base[AM], baseP[AM];                           // element offsets of the wave's first column in this lane's row / the partner row (r ^ 8)
#pragma unroll
    for (int i = 0; i < AM; ++i) {
        base[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + (lane & 15)] + nchunk);
        baseP[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + ((lane & 15) ^ 8)] + nchunk);
    }
    const int nb = nchunk + 4 * (int)q4;
    _Float16* const yp = reinterpret_cast<_Float16*>(a.y);
    const _Float16* const rp = reinterpret_cast<const _Float16*>(a.ep_res);
    const bool relu = a.ep_relu != 0;
    float ep_max = 0.f;
    f32x4v esc[4], esh[4], bv[BIAS ? 4 : 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        esc[j] = *reinterpret_cast<const f32x4v*>(a.ep_scale + nb + j * 16);
        esh[j] = *reinterpret_cast<const f32x4v*>(a.ep_shift + nb + j * 16);
        if constexpr (BIAS) bv[j] = *reinterpret_cast<const f32x4v*>(a.bias + nb + j * 16);      // (Xception's pointwise convs: the folded BatchNorm of the depthwise conv)
    }
    constexpr int IB = 2;                                   // row fragments per batch of residual loads
#pragma unroll
    for (int i0 = 0; i0 < AM; i0 += IB) {
        uint4 la[RES ? IB : 1], lb[RES ? IB : 1];
        if constexpr (RES != 0) {
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                la[ii] = *reinterpret_cast<const uint4*>(rp + ((upper ? baseP[i0 + ii] : base[i0 + ii]) + col));      // rows 0-7 of the fragment
                lb[ii] = *reinterpret_cast<const uint4*>(rp + ((upper ? base[i0 + ii] : baseP[i0 + ii]) + col));      // rows 8-15
            }
        }
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
            const int i = i0 + ii;
            uint2 rq[RES ? 4 : 1];                           // the residual's halves of this lane's four channel quads
            if constexpr (RES != 0) {
                const unsigned s0 = upper ? la[ii].x : lb[ii].x, s1 = upper ? la[ii].y : lb[ii].y, s2 = upper ? la[ii].z : lb[ii].z, s3 = upper ? la[ii].w : lb[ii].w;
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0x128, 0xF, 0xF, true);      // row_ror:8 = lane ^ 8
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0x128, 0xF, 0xF, true);
                const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s2, 0x128, 0xF, 0xF, true);
                const unsigned r3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s3, 0x128, 0xF, 0xF, true);
                // own row: pair 0 (lower lanes loaded it, upper lanes received it), pair 1 (the other way round)
                const unsigned p0x = upper ? r0 : la[ii].x, p0y = upper ? r1 : la[ii].y, p0z = upper ? r2 : la[ii].z, p0w = upper ? r3 : la[ii].w;
                const unsigned p1x = upper ? lb[ii].x : r0, p1y = upper ? lb[ii].y : r1, p1z = upper ? lb[ii].z : r2, p1w = upper ? lb[ii].w : r3;
                const u32x2_ ax = __builtin_amdgcn_permlane16_swap(p0x, p0z, false, false), ay = __builtin_amdgcn_permlane16_swap(p0y, p0w, false, false);
                const u32x2_ bx = __builtin_amdgcn_permlane16_swap(p1x, p1z, false, false), by = __builtin_amdgcn_permlane16_swap(p1y, p1w, false, false);
                rq[0] = uint2{ax.x, ay.x}; rq[1] = uint2{ax.y, ay.y}; rq[2] = uint2{bx.x, by.x}; rq[3] = uint2{bx.y, by.y};
            }
            uint2 hq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4v val;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r] * c;              // (the general path's acc * unscale_a * unscale_b: powers of two)
                    if constexpr (BIAS) t += bv[j][r];
                    val[r] = t * esc[j][r] + esh[j][r];      // BatchNorm-apply's own expression and order
                }
                if constexpr (RES != 0) {
                    const f32x4v rv = __builtin_convertvector(__builtin_bit_cast(f16x4_, rq[j]), f32x4v) * res_unscale;
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[r] += rv[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (relu) val[r] = fmaxf(val[r], 0.f);
                    ep_max = fmaxf(ep_max, fabsf(val[r]));
                }
                const f32x4v v = val * hscale;
                const f16x4_ h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                hq[j] = __builtin_bit_cast(uint2, h);
            }
            const u32x2_ ax = __builtin_amdgcn_permlane16_swap(hq[0].x, hq[1].x, false, false), ay = __builtin_amdgcn_permlane16_swap(hq[0].y, hq[1].y, false, false);
            const u32x2_ bx = __builtin_amdgcn_permlane16_swap(hq[2].x, hq[3].x, false, false), by = __builtin_amdgcn_permlane16_swap(hq[2].y, hq[3].y, false, false);
            // pair 0 = (ax.x, ay.x, ax.y, ay.y), pair 1 = (bx.x, by.x, bx.y, by.y); the upper row sends pair 0, the lower row pair 1
            const unsigned s0 = upper ? ax.x : bx.x, s1 = upper ? ay.x : by.x, s2 = upper ? ax.y : bx.y, s3 = upper ? ay.y : by.y;
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0x128, 0xF, 0xF, true);
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0x128, 0xF, 0xF, true);
            const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s2, 0x128, 0xF, 0xF, true);
            const unsigned r3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s3, 0x128, 0xF, 0xF, true);
            const uint4 da = {upper ? r0 : ax.x, upper ? r1 : ay.x, upper ? r2 : ax.y, upper ? r3 : ay.y};      // lower: own pair 0; upper: the lower row's pair 1
            const uint4 db = {upper ? bx.x : r0, upper ? by.x : r1, upper ? bx.y : r2, upper ? by.y : r3};      // lower: the upper row's pair 0; upper: own pair 1
            *reinterpret_cast<uint4*>(yp + ((upper ? baseP[i] : base[i]) + col)) = da;      // rows 0-7 of the fragment, 128 B each
            *reinterpret_cast<uint4*>(yp + ((upper ? base[i] : baseP[i]) + col)) = db;      // rows 8-15
        }
        __builtin_amdgcn_sched_barrier(0);      // one batch of residual loads at a time
    }
    return ep_max;
}

// Epilogue shared by the planes kernels (conv_igemm.hip's phased 16-byte epilogue: lookups and old values first, then arithmetic,
// then stores): fold the cross-term accumulator, undo the operand scales, bias, optional accumulation into y, BatchNorm statistics
// partials of M-tile `tile_m`.  rowoff[BM]: output element offsets of the tile's rows (-1: none); smem: free LDS for the statistics.
// AM: 16-row fragments per wave along the pixel axis -- 4 (64 x 64 wave tiles, waves 2 wide) or 2 (32 x 64 wave tiles, every wave in the
// first 64 columns: the NARROW launches for at most 64 output channels)
// Returns the number of UNCONDITIONAL vector stores this wave issued last (the lean path's output stores: nothing but the conditional
// statistics stores of the tail comes after them) -- what a persistent caller may leave in flight behind a counted vmcnt; 0 = no such promise.
template <int NTERMS, int BM, bool BNB = false, int AM = 4, bool EP = false, bool LEAN_RES = true>
__device__ __forceinline__ int pl_epilogue(const GatherGemmArgs& a, f32x4v (&acc)[AM][4], f32x4v (&acc_lo)[NTERMS == 3 ? AM : 1][NTERMS == 3 ? 4 : 1],
                                            const int* rowoff, float* smem, int tile_m, int n0, int wave_m, int wave_n, int lane, int tid, bool rows_full,
                                            const unsigned* ranges = nullptr, int stat_cols = PL_BN) {
    // ranges: {max|x| bits, max|w| bits} already in registers (the persistent kernel loads them ONCE: inside its tile loop every global load
    // of the epilogue is a wait for the next tile's operand DMA, which is older in the in-order vmcnt); nullptr = read them here
    constexpr int BN = PL_BN, WM = 16 * AM, WN = 64, AT = 4;
    typedef f32x4v f32x4v_;
    // ---- epilogue (conv_igemm.hip's phased 16-byte epilogue: lookups and old values first, then arithmetic, then stores) ----
    float* sred = smem;             // [BM / WM][BN][2]
    const bool do_stats = a.stats != nullptr;
    const unsigned ax_bits = ranges != nullptr ? ranges[0] : (a.amax_x ? *a.amax_x : 0u);
    const unsigned aw_bits = ranges != nullptr ? ranges[1] : (a.amax_w ? *a.amax_w : 0u);
    const float scale_a = a.amax_x ? pow2_scale_for(ax_bits) : 1.f;
    const float scale_b = a.amax_w ? pow2_scale_for(aw_bits) : 1.f;
    const float unscale_a = pow2_inv(scale_a), unscale_b = pow2_inv(scale_b);      // exact (bit arithmetic; the IEEE division costs ~10 instructions each)
    float hscale = 1.f;                                  // out_half / out_planes2: the output leaves as one / two fp16 planes
    if (a.out_half || (EP && a.out_planes2)) {
        float b;
        if constexpr (EP) {
            // fused inference epilogue: bound of act(conv * scale + shift + residual) from the TRUE input maximum (GatherGemmArgs::bound_x)
            b = a.out_bound_k * __uint_as_float(*(a.bound_x != nullptr ? a.bound_x : a.amax_x)) * __uint_as_float(*a.amax_w);
            if (a.ep_scale_amax != nullptr) b = b * __uint_as_float(*a.ep_scale_amax) + __uint_as_float(*a.ep_shift_amax);
            if (a.ep_res_amax != nullptr) b += __uint_as_float(*a.ep_res_amax);
        } else {
            b = a.out_bound_k * __uint_as_float(ax_bits) * __uint_as_float(aw_bits);
        }
        hscale = pow2_scale_for(__float_as_uint(b));
        if (blockIdx.x == 0 && tid == 0) *a.out_bound = __float_as_uint(b);
    }
    const float* extra = a.add_src != nullptr ? a.add_src : (a.accumulate ? a.y : ((EP && a.ep_res_fmt == 0) ? a.ep_res : nullptr));
    const bool res_planes = EP && a.ep_res != nullptr && a.ep_res_fmt != 0;      // residual as fp16 planes (y's geometry, dense)
    const float res_unscale = res_planes ? pow2_inv(pow2_scale_for(*a.ep_res_scale)) : 1.f;
    const unsigned char* amask = a.add_mask;         // (with add_src; y_pitch == N_store: element offset / 4 = the mask's vector index)
    // fold the cross-term accumulator (both paths; acc_lo is dead from here on): acc + 2^-11 lo in one rounding, as (acc + lo / 2048)
    if constexpr (NTERMS == 3) {
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int j = 0; j < AT; ++j) acc[i][j] = epi_fma(acc_lo[i][j], 1.f / 2048.f, acc[i][j]);
    }
    // the lean path (above): a tile without an edge and without a combination the general path alone knows -- decided per wave
    bool done = false;
    int tail_stores = 0;
    if constexpr (!EP && !BNB) {
        // (a bias vector holds N entries: with a bias the wave's 64 columns must lie inside N, not just inside the padded N_store; the
        //  one-plane output takes a bias only in the inference epilogue)
        const bool cols_full = n0 + wave_n * WN + WN <= (a.bias != nullptr ? a.N : a.N_store);
        if ((a.bias == nullptr || ((reinterpret_cast<uintptr_t>(a.bias) & 15) == 0 && !a.out_half)) && rows_full && cols_full && !(do_stats && extra != nullptr) && (NTERMS == 1 || !a.out_half) &&
            !(a.out_half && extra != nullptr) && !(a.dbg_flags & 8)) {
            done = true;
            tail_stores = (PYLC_EPI_FULL_LINES != 0) ? (a.out_half ? 2 * AM : 4 * AM) : 0;      // whole-line stores: two per row fragment and column-group pair (fp32) / per row fragment (one plane)
            float* const sd = smem + ((wave_m * BN) + wave_n * WN + 4 * (lane >> 4)) * 2;
            const int nb = n0 + wave_n * WN + 4 * (lane >> 4);
            const float c = unscale_a * unscale_b;
#define PYLC_LEAN(ST, PV, HF) pl_epilogue_lean<AM, ST, PV, HF>(a, acc, rowoff, sd, nb, wave_m, lane, c, hscale, extra, amask, a.bias)
            if constexpr (NTERMS == 1) {
                if (a.out_half) {
                    if (do_stats) PYLC_LEAN(1, 0, true); else PYLC_LEAN(0, 0, true);
                } else if (extra != nullptr) {
                    if (amask != nullptr) PYLC_LEAN(0, 2, false); else PYLC_LEAN(0, 1, false);
                } else if (do_stats) PYLC_LEAN(1, 0, false);
                else PYLC_LEAN(0, 0, false);
            } else {
                if (extra != nullptr) {
                    if (amask != nullptr) PYLC_LEAN(0, 2, false); else PYLC_LEAN(0, 1, false);
                } else if (do_stats) PYLC_LEAN(1, 0, false);
                else PYLC_LEAN(0, 0, false);
            }
#undef PYLC_LEAN
        }
    }
    if constexpr (EP) {
        // the lean inference epilogue (pl_epilogue_lean_ep): an edge-free tile, interleaved two-plane result, residual none or in that format
        const int nchunk = n0 + wave_n * WN;
        // (LEAN_RES false -- the 3x3 halo kernel, whose convs carry no residual in these networks: the residual form is not compiled in, it would
        //  cost that kernel 35 spilled registers)
        const bool res_ok = a.ep_res == nullptr || (LEAN_RES && a.ep_res_fmt == 2 && planes_il(a.ep_res_plane_stride));
        if (rows_full && nchunk + WN <= a.N && a.out_planes2 && !a.out_half && planes_il(a.out_plane_stride) && a.y_pitch == a.N_store && (a.bias == nullptr || a.ep_res == nullptr) &&
            a.ep_scale != nullptr && a.ep_vec_ok && extra == nullptr && res_ok &&
            !do_stats && !(a.dbg_flags & 8)) {
            done = true;
            const float c = unscale_a * unscale_b;
            float m;
            if (LEAN_RES && a.ep_res != nullptr) m = pl_epilogue_lean_ep<AM, LEAN_RES ? 2 : 0, false>(a, acc, rowoff, nchunk, wave_m, lane, c, hscale, res_unscale);
            else if (a.bias != nullptr) m = pl_epilogue_lean_ep<AM, 0, true>(a, acc, rowoff, nchunk, wave_m, lane, c, hscale, res_unscale);
            else m = pl_epilogue_lean_ep<AM, 0, false>(a, acc, rowoff, nchunk, wave_m, lane, c, hscale, res_unscale);
            if (a.ep_amax != nullptr) amax_commit(m, a.ep_amax);
        }
        if constexpr (NTERMS == 1) {
            // ... one-plane result (precision mode 3), residual none or one dense plane; 16-byte pieces need pixel rows that are multiples of 8 channels
            const bool res1_ok = a.ep_res == nullptr || (LEAN_RES && a.ep_res_fmt == 1);
            if (!done && rows_full && nchunk + WN <= a.N && a.out_half && a.y_pitch == a.N_store && (a.N_store & 7) == 0 && a.ep_scale != nullptr &&
                a.ep_vec_ok && extra == nullptr && res1_ok && !do_stats && !(a.dbg_flags & 8)) {
                done = true;
                const float c = unscale_a * unscale_b;
                float m;
                if (LEAN_RES && a.ep_res != nullptr) {
                    m = a.bias != nullptr ? pl_epilogue_lean_ep_half<AM, LEAN_RES ? 1 : 0, true>(a, acc, rowoff, nchunk, wave_m, lane, c, hscale, res_unscale)
                                          : pl_epilogue_lean_ep_half<AM, LEAN_RES ? 1 : 0, false>(a, acc, rowoff, nchunk, wave_m, lane, c, hscale, res_unscale);
                } else {
                    m = a.bias != nullptr ? pl_epilogue_lean_ep_half<AM, 0, true>(a, acc, rowoff, nchunk, wave_m, lane, c, hscale, res_unscale)
                                          : pl_epilogue_lean_ep_half<AM, 0, false>(a, acc, rowoff, nchunk, wave_m, lane, c, hscale, res_unscale);
                }
                if (a.ep_amax != nullptr) amax_commit(m, a.ep_amax);
            }
        }
    }
    constexpr bool bn = BNB;                 // (its own instantiations, gg_pl*_kernel<..., BNB = true>: the other launches keep their registers)
    float gmax = 0.f;
    if constexpr (!EP && BNB) {
        // BatchNorm-backward mode (its own instantiations; launch_gg_pl guarantees a dense fp32 output, statistics, no bias): pl_epilogue_bn
        // is the whole epilogue, with or without edges
        done = true;
        const bool cols_full = n0 + wave_n * WN + WN <= a.N_store;
        float* const sd = smem + ((wave_m * BN) + wave_n * WN + 4 * (lane >> 4)) * 2;
        const int nb = n0 + wave_n * WN + 4 * (lane >> 4);
        const float c = unscale_a * unscale_b;
        const int pv = extra == nullptr ? 0 : (amask == nullptr ? 1 : 2);
        if (rows_full && cols_full) {
            gmax = pv == 0 ? pl_epilogue_bn<AM, 0, false>(a, acc, rowoff, sd, nb, wave_m, lane, c, extra, amask)
                 : pv == 1 ? pl_epilogue_bn<AM, 1, false>(a, acc, rowoff, sd, nb, wave_m, lane, c, extra, amask)
                           : pl_epilogue_bn<AM, 2, false>(a, acc, rowoff, sd, nb, wave_m, lane, c, extra, amask);
        } else {
            gmax = pv == 0 ? pl_epilogue_bn<AM, 0, true>(a, acc, rowoff, sd, nb, wave_m, lane, c, extra, amask)
                 : pv == 1 ? pl_epilogue_bn<AM, 1, true>(a, acc, rowoff, sd, nb, wave_m, lane, c, extra, amask)
                           : pl_epilogue_bn<AM, 2, true>(a, acc, rowoff, sd, nb, wave_m, lane, c, extra, amask);
        }
    }
    if constexpr (!BNB) {
    if (!done) {
    int eoff[AM][AT];
    float bv[AT][4];
    float esc[EP ? AT : 1][4], esh[EP ? AT : 1][4];          // fused inference epilogue: eval-BatchNorm scale / shift of this lane's channels
    float ep_max = 0.f;
    {
        int offs[AM];
#pragma unroll
        for (int i = 0; i < AM; ++i) offs[i] = rowoff[wave_m * WM + i * 16 + (lane & 15)];
#pragma unroll
        for (int j = 0; j < AT; ++j) {
            const int n4 = n0 + wave_n * WN + j * 16 + 4 * (lane >> 4);
            const bool nok = n4 < a.N_store;                    // N_store % 4 == 0: all four channels or none
#pragma unroll
            for (int i = 0; i < AM; ++i) eoff[i][j] = (nok && offs[i] >= 0) ? offs[i] + n4 : -1;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[j][r] = (a.bias != nullptr && n4 + r < a.N) ? a.bias[n4 + r] : 0.f;
            if constexpr (EP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    esc[j][r] = (a.ep_scale != nullptr && n4 + r < a.N) ? a.ep_scale[n4 + r] : 1.f;
                    esh[j][r] = (a.ep_scale != nullptr && n4 + r < a.N) ? a.ep_shift[n4 + r] : 0.f;
                }
            }
        }
    }
    constexpr int PJ = 2;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][j][r] = acc[i][j][r] * unscale_a * unscale_b;
            }
    // one-plane fp16 output: the BatchNorm that follows normalises the ROUNDED tensor, so its statistics (and nothing else differs: the store
    // below reproduces the same halves exactly, hscale being a power of two) are taken from the rounded values, not from the accumulators
    if (!EP && a.out_half) {
        const float inv_h = pow2_inv(hscale);
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int j = 0; j < AT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (float)(_Float16)(acc[i][j][r] * hscale) * inv_h;
    }
    __builtin_amdgcn_sched_barrier(0);
    float* sdst = sred + ((wave_m * BN) + wave_n * WN + 4 * (lane >> 4)) * 2;
    auto finish = [&](auto has_prev, auto j0c, const f32x4v_ (&prev)[AM][PJ]) {
        constexpr int j0 = decltype(j0c)::value;
        constexpr bool phased = decltype(has_prev)::value;
        constexpr int NJ = phased ? PJ : AT;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = j0 + jj;
            float cs[4] = {0.f, 0.f, 0.f, 0.f}, css[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                const bool stored = eoff[i][j] >= 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float val = acc[i][j][r] + bv[j][r];
                    if constexpr (EP) val = val * esc[j][r] + esh[j][r];          // BatchNorm-apply's own expression and order (conv_igemm.hip's fused epilogue)
                    if constexpr (decltype(has_prev)::value) val += prev[i][jj][r];
                    if constexpr (EP) {
                        if (a.ep_relu) val = fmaxf(val, 0.f);
                        ep_max = fmaxf(ep_max, stored ? fabsf(val) : 0.f);
                    }
                    acc[i][j][r] = val;
                    const float cv = stored ? val - bv[j][r] : 0.f;      // statistics of (value - bias): conv_igemm.hip's epilogue
                    cs[r] += cv;
                    css[r] += cv * cv;
                }
            }
            if (do_stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { cs[r] = row_sum16(cs[r]); css[r] = row_sum16(css[r]); }
                if ((lane & 15) == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sdst[(j * 16 + r) * 2] = cs[r]; sdst[(j * 16 + r) * 2 + 1] = css[r]; }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // fp16 outputs: a lane holds 4 channels = 8 bytes per plane.  Lanes l and l + 16 (same pixel, the next channel quad) swap one half
        // each (v_permlane16_swap: odd DPP rows of the first operand <-> even rows of the second), so that a store carries 16 bytes:
        //   two planes: lanes of even rows write 8 channels of plane 0, lanes of odd rows the same 8 channels of plane 1;
        //   one plane : lanes of even rows write 8 channels, lanes of odd rows nothing.
        // (N_store % 8 == 0 for these formats, so a lane and its partner are stored or masked together.)
        if (a.out_half || (EP && a.out_planes2)) {
            typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            const bool odd_row = (lane >> 4) & 1;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int i = 0; i < AM; ++i) {
                    const int e = eoff[i][j0 + jj];
                    uint4 q;
                    // (two planes, chunk-interleaved output -- out_plane_stride == 32, common.h: the pair's 8 channels lie in one chunk)
                    const bool ilo = !a.out_half && planes_il(a.out_plane_stride);
                    _Float16* dst = reinterpret_cast<_Float16*>(a.y) + planes_phys(odd_row ? e - 4 : e, ilo);
                    if (a.out_half) {
                        const f32x4v_ v = acc[i][j0 + jj] * hscale;
                        const f16x4_ h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        const uint2 hu = __builtin_bit_cast(uint2, h);
                        const u32x2_ sx = __builtin_amdgcn_permlane16_swap(hu.x, hu.x, false, false);      // .y: even rows <- partner's word
                        const u32x2_ sy = __builtin_amdgcn_permlane16_swap(hu.y, hu.y, false, false);
                        q = uint4{hu.x, hu.y, sx.y, sy.y};
                        if (e >= 0 && !odd_row) *reinterpret_cast<uint4*>(dst) = q;
                    } else {
                        uint2 p0, p1;
                        split2(acc[i][j0 + jj], hscale, p0, p1);      // h0 = rn16(s v), h1 = rn16(2^11 (s v - h0)): what pylc_to_planes writes
                        const u32x2_ sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
                        const u32x2_ sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
                        // even rows: (own p0, partner's p0) -> plane 0;  odd rows: (partner's p1, own p1) -> plane 1, at the partner's channels
                        q = uint4{sx.x, sy.x, sx.y, sy.y};
                        if (odd_row) dst += a.out_plane_stride;
                        if (e >= 0) *reinterpret_cast<uint4*>(dst) = q;
                    }
                }
        } else {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int i = 0; i < AM; ++i)
                    if (eoff[i][j0 + jj] >= 0) *reinterpret_cast<f32x4v_*>(a.y + eoff[i][j0 + jj]) = acc[i][j0 + jj];
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    const f32x4v_ none[AM][PJ] = {};
    f32x4v_ prev[AM][PJ];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int jj = 0; jj < PJ; ++jj) {
                const f32x4v_ zero = {0.f, 0.f, 0.f, 0.f};
                prev[i][jj] = eoff[i][j0 + jj] >= 0 ? *reinterpret_cast<const f32x4v_*>(extra + eoff[i][j0 + jj]) : zero;
            }
        if (amask != nullptr) {
            unsigned mb[AM][PJ];
#pragma unroll
            for (int i = 0; i < AM; ++i)
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj) mb[i][jj] = eoff[i][j0 + jj] >= 0 ? amask[eoff[i][j0 + jj] >> 3] : 0u;
#pragma unroll
            for (int i = 0; i < AM; ++i)
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj) {
                    const unsigned nib = mb[i][jj] >> (((eoff[i][j0 + jj] >> 2) & 1) * 4);
                    prev[i][jj][0] = (nib & 1u) ? prev[i][jj][0] : 0.f;
                    prev[i][jj][1] = (nib & 2u) ? prev[i][jj][1] : 0.f;
                    prev[i][jj][2] = (nib & 4u) ? prev[i][jj][2] : 0.f;
                    prev[i][jj][3] = (nib & 8u) ? prev[i][jj][3] : 0.f;
                }
        }
    };
    // residual of the fused inference epilogue arriving as fp16 planes: element = (h0 + 2^-11 h1) / s
    auto fetch_res = [&](int j0) {
        typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        const _Float16* const rp = reinterpret_cast<const _Float16*>(a.ep_res);
        const bool odd_row = (lane >> 4) & 1;
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int jj = 0; jj < PJ; ++jj) {
                const int e = eoff[i][j0 + jj];
                f32x4v_ v = {0.f, 0.f, 0.f, 0.f};
                if (a.ep_res_fmt == 2) {
                    // lanes l / l + 16 (same pixel, adjacent channel quads): the even row fetches 16 bytes of plane 0, the odd row the same 8
                    // channels of plane 1, and they swap halves (as the stores above, reversed)
                    uint4 q = {0u, 0u, 0u, 0u};
                    if (e >= 0) q = *reinterpret_cast<const uint4*>(rp + planes_phys(odd_row ? e - 4 : e, planes_il(a.ep_res_plane_stride)) + (odd_row ? a.ep_res_plane_stride : 0));
                    const u32x2_ s0 = __builtin_amdgcn_permlane16_swap(q.x, q.z, false, false);      // even: q.z <- partner's q.x; odd: q.x <- partner's q.z
                    const u32x2_ s1 = __builtin_amdgcn_permlane16_swap(q.y, q.w, false, false);
                    // even rows now hold (p0 own: s0.x s1.x | p1 own: s0.y s1.y); odd rows (p0 own: s0.x s1.x | p1 own: s0.y s1.y) as well
                    const f16x4_ h0 = __builtin_bit_cast(f16x4_, uint2{s0.x, s1.x});
                    const f16x4_ h1 = __builtin_bit_cast(f16x4_, uint2{s0.y, s1.y});
                    v = (__builtin_convertvector(h0, f32x4v_) + __builtin_convertvector(h1, f32x4v_) * (1.f / 2048.f)) * res_unscale;
                    if (e < 0) v = f32x4v_{0.f, 0.f, 0.f, 0.f};
                } else if (e >= 0) {
                    const f16x4_ h0 = __builtin_bit_cast(f16x4_, *reinterpret_cast<const uint2*>(rp + e));
                    v = __builtin_convertvector(h0, f32x4v_) * res_unscale;
                }
                prev[i][jj] = v;
            }
    };
    if (res_planes) {
        fetch_res(0);
        __builtin_amdgcn_sched_barrier(0);
        finish(std::true_type{}, std::integral_constant<int, 0>{}, prev);
        __builtin_amdgcn_sched_barrier(0);
        fetch_res(2);
        __builtin_amdgcn_sched_barrier(0);
        finish(std::true_type{}, std::integral_constant<int, 2>{}, prev);
    } else if (extra == nullptr) {
        finish(std::false_type{}, std::integral_constant<int, 0>{}, none);
    } else {
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
        finish(std::true_type{}, std::integral_constant<int, 0>{}, prev);
        __builtin_amdgcn_sched_barrier(0);
        fetch(2);
        __builtin_amdgcn_sched_barrier(0);
        finish(std::true_type{}, std::integral_constant<int, 2>{}, prev);
    }
    if (EP && a.ep_amax != nullptr) amax_commit(ep_max, a.ep_amax);
    }          // general path
    }          // !BNB
    if (bn && a.bn_gmax != nullptr) amax_commit(gmax, a.bn_gmax);
    if (do_stats) {
        // (not __syncthreads(): its fence drains vmcnt -- every output store of this tile, and in the persistent kernel the next tile's
        //  operand DMA -- before the partials may be combined; only the LDS writes above have to be visible)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < stat_cols) {          // (stat_cols: the columns this block's waves cover -- 64 in gg_plhn_kernel, whose column tiles are 64 wide)
            const int n = n0 + tid;
            if (n < a.N_store) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int wm = 0; wm < BM / WM; ++wm) { sm += sred[(wm * BN + tid) * 2]; sq += sred[(wm * BN + tid) * 2 + 1]; }
                float* dst = a.stats + (size_t)tile_m * 2 * a.N_store;
                dst[n] = sm;
                dst[a.N_store + n] = sq;
            }
        }
    }
    return tail_stores;
}

// STAMPS (tools/pl_stamps.py): lane 0 of the first and the last wave of block `dbg_flags >> 16` records s_memtime at the phase
// boundaries of every K-step into LDS (dumped to a.dbg at the end); production launches use STAMPS = false.
// NARROW: launches with at most 64 output channels.  The 2-wide wave grid would leave the waves of the second column -- SIMDs 2 and 3 --
// multiplying zeros while SIMDs 0 and 1 do all the work; here every wave sits in the first 64 columns on a 32 x 64 tile (half the
// MFMAs per wave, all four SIMDs busy).  Same LDS image, DMA and reduction order per output element as the wide form: bit-identical.
// EP: the fused inference epilogue (eval BatchNorm + residual + ReLU, plane outputs; pl_epilogue<.., EP>) is compiled in only here
template <int NTERMS, int BM, bool STAMPS = false, bool BNB = false, bool NARROW = false, bool EP = false>
__global__ __launch_bounds__(BM * 2, 2) void gg_pl_kernel(const GatherGemmArgs a) {
    constexpr int BN = PL_BN, AM = NARROW ? 2 : 4, WM = 16 * AM, WN = 64, AT = 4, ROW = pl_row_bytes<NTERMS>();
    constexpr int KS = ROW / 2;                             // reduction depth of one LDS stage: 32, or 64 (two 32-deep MFMA sub-steps)
    constexpr int RPI = 1024 / ROW;                         // rows per DMA instruction (1 KB, lane-linear): 16 or 8
    constexpr int NW = BM / 32;                             // waves: 4 (2 M x 2 N) or 8 (4 M x 2 N)
    constexpr int NST = pl_stages<BM>();
    constexpr int AI = 32 / RPI;                            // pixel-row pieces per wave (32 rows): 2 or 4
    constexpr int BI = BN / (RPI * NW);                     // filter-row pieces per wave: 2 or 1 (64-byte rows), 4 or 2 (128-byte rows)
    constexpr int NPL = NTERMS == 3 ? 2 : 1;
    constexpr int NDMA = (AI + BI) * NPL;                   // DMA instructions per wave and K-step
    constexpr int STAGE = pl_stage_bytes<NTERMS, BM>();
    constexpr int OFF_B = NPL * BM * ROW;
    constexpr unsigned OOB = 0x80000000u;                   // >= num_records of every descriptor (takes_pl: buffers below 2 GiB)
    extern __shared__ __attribute__((aligned(16))) float smem[];      // ONE LDS object (a second one makes hipcc drain the DMA early)
    char* lds = reinterpret_cast<char*>(smem);
    int* rowoff = reinterpret_cast<int*>(lds + NST * STAGE);                                     // [BM] output element offsets, -1: no row
    unsigned long long* s_wave_taps = reinterpret_cast<unsigned long long*>(lds + NST * STAGE + BM * 4);    // [NW]
    unsigned long long* stamp_base = reinterpret_cast<unsigned long long*>(lds + NST * STAGE + BM * 4 + 64);  // STAMPS: [2][256]
    int n_stamp = 0;
#define PL_STAMP()                                                                                                    \
    if constexpr (STAMPS) {                                                                                           \
        if (stamper && n_stamp < 250) stamp_base[(wave != 0) * 256 + n_stamp++] = __builtin_amdgcn_s_memtime();       \
    }

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / a.tiles_n) * BM;
    const int n0 = (tile % a.tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform: LDS-DMA destinations live in M0
    const int wave_m = NARROW ? wave : wave >> 1, wave_n = NARROW ? 0 : wave & 1;
    if constexpr (BM == 128) {
        // All tiles of a launch cost the same, so the two blocks of a CU run in lockstep: both in their main loops (two waves per SIMD
        // competing for the matrix pipe), then both in their epilogues (the whole chip storing at once: on the short reductions of
        // the 1x1 convs the stores of a round are an HBM-rate burst as long as the main loop, during which no MFMA issues).  Delaying
        // the SECOND block of each CU (the one whose LDS allocation does not start at 0) by about half a tile, once, in the first
        // round, puts one block's epilogue and prologue under the other's main loop for the rest of the launch.
        if (a.stagger > 0 && (int)blockIdx.x < a.stagger_blocks) {
            if (a.dbg_flags & 32768) {
                // experiment (pylc_debug_pp_flags bit 15): de-phase CHIP HALVES instead of the two blocks of a CU -- every block of XCDs
                // 4-7 starts late, so that one half of the chip stores while the other half computes (a CU cannot overlap its own
                // stores with its own loads, but the HBM write path can be kept busy by other CUs)
                const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID
                if (xcc >= 4)
                    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(32);
            } else {
                const unsigned lds_base = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 0 << 6 | 6);     // HW_REG_LDS_ALLOC.LDS_BASE
                if (lds_base != 0)
                    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(32);                 // 32 x 64 cycles
            }
        }
    }
    // loader: LPR lanes per LDS row, RPI rows per DMA instruction; `lc` = the logical 16-byte chunk this lane fetches for its (swizzled)
    // LDS position.  64-byte rows: chunk ^ (((row >> 2) & 1) << 1).  128-byte rows: the low two chunk bits ^ ((row >> 1) & 3) -- eight
    // consecutive rows of one chunk column then cover eight different 16-byte bank groups of a 256-byte bank sweep (conflict-free
    // ds_read_b128 fragments), and the XOR stays inside a 64-byte half, i.e. inside one 32-deep sub-step.
    const int lrow = ROW == 64 ? lane >> 2 : lane >> 3;
    const int lc = ROW == 64 ? (lane & 3) ^ (((lane >> 4) & 1) << 1) : ((lane & 4) | ((lane & 3) ^ ((lane >> 4) & 3)));
    const bool ident = a.ident != 0;
    const bool stamper = STAMPS && (int)blockIdx.x == (a.dbg_flags >> 16) && lane == 0 && (wave == 0 || wave == NW - 1);
    PL_STAMP();

    // chunk-interleaved filter planes (a.w_il, PylcConvDesc.w_planes_fmt): the two planes of a 32-channel chunk are the two halves of ONE
    // 128-byte line, requested back to back below -- row and chunk offsets double, plane 1 sits 64 bytes behind plane 0
    const unsigned ilmb = (unsigned)__builtin_amdgcn_readfirstlane(a.w_il ? 2 : 1);      // provably wave-uniform: it scales the DMA's scalar offset
    // ... and the same layout for the PIXEL operand (a.x_plane_stride == 32, common.h planes_il: written so by BatchNorm / to_planes for tensors
    // of a multiple of 32 channels): pixel and chunk offsets double, plane 1 is 64 bytes behind plane 0, one descriptor spans both planes
    const bool ila = NPL == 2 && planes_il(a.x_plane_stride);
    const unsigned ilma = (unsigned)__builtin_amdgcn_readfirstlane(ila ? 2 : 1);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_planes), 0, (int)(a.w_plane_stride * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)(a.x_bytes * ilma), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.x_planes) + a.x_plane_stride * 2), 0, (int)(a.x_bytes * ilma - (ila ? 64 : 0)), 0x00020000);

    // ---- filter rows of this thread: no geometry needed, so their first DMA goes out before anything else ----
    unsigned woff_row[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int n = n0 + RPI * BI * wave + RPI * i + lrow;
        // (one-plane kernels, 128-byte LDS rows: lc counts eight 16-byte pieces = two chunks; an interleaved filter keeps a chunk's plane 0
        //  in the first 64 bytes of its line)
        const unsigned lcb = a.w_il ? ((unsigned)(lc >> 2) * 128u + (unsigned)(lc & 3) * 16u) : 16u * lc;
        woff_row[i] = n < a.N ? (unsigned)n * (unsigned)a.w_row_stride * ilmb * 2u + lcb : OOB;
    }
    const unsigned plane1_w = a.w_il ? 64u : (unsigned)(a.w_plane_stride * 2);

    // ---- pixel rows of this thread ----
    int rowh[AI], roww[AI];
    unsigned xoff[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + 32 * wave + RPI * i + lrow;
        const bool ok = m < a.M;
        if (ident) {                       // 1x1, stride 1, no padding: input pixel == output pixel, no decode
            rowh[i] = ok ? 0 : -(1 << 28);
            roww[i] = 0;
            xoff[i] = ((unsigned)m * (unsigned)a.x_pitch * ilma + 8u * lc) * 2u;
        } else {
            const int mm = ok ? m : 0;
            const int q = mm % a.Q, t = mm / a.Q;
            const int p = t % a.P, b = t / a.P;
            rowh[i] = ok ? p * a.in_sh : -(1 << 28);          // invalid rows fail every bounds check
            roww[i] = q * a.in_sw;
            xoff[i] = ((unsigned)(b * a.IH * a.IW + rowh[i] * a.IW + roww[i]) * (unsigned)a.x_pitch * ilma + 8u * lc) * 2u;      // garbage for invalid rows (masked)
        }
    }
    // output row table (read by the epilogue; the first barrier of the main loop orders it)
    if (tid < BM) {
        const int m = m0 + tid;
        long long off = -1;
        if (m < a.M) {
            if (ident) {
                off = (long long)m * a.y_pitch;
            } else {
                const int q = m % a.Q, t = m / a.Q;
                const int p = t % a.P, b = t / a.P;
                off = ((long long)(b * a.OH + p * a.out_sh + a.oh0) * a.OW + q * a.out_sw + a.ow0) * a.y_pitch;
            }
        }
        rowoff[tid] = (int)off;
    }

    const int T = a.TR * a.TS;
    const int nchunks = (a.Cin + KS - 1) / KS;
    // taps that reach at least one valid input pixel of this tile (the others are skipped): one ballot per tap, one barrier
    unsigned long long tapmask = ~0ull;
    if (T > 1) {
        unsigned long long mine = 0;
        for (int t = 0; t < T; ++t) {
            const int dh = a.dh0 + (t / a.TS) * a.dh_step, dw = a.dw0 + (t % a.TS) * a.dw_step;
            int any = 0;
#pragma unroll
            for (int i = 0; i < AI; ++i)
                any |= ((unsigned)(rowh[i] + dh) < (unsigned)a.IH) & ((unsigned)(roww[i] + dw) < (unsigned)a.IW);
            if (__builtin_amdgcn_ballot_w64(any != 0) != 0) mine |= 1ull << t;
        }
        if (lane == 0) s_wave_taps[wave] = mine;
        __syncthreads();
        unsigned long long all = 0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) all |= s_wave_taps[wv];
        tapmask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(all >> 32)) << 32) |
                  (unsigned)__builtin_amdgcn_readfirstlane((int)all);
    }
    const int ntaps = __popcll(T >= 64 ? tapmask : (tapmask & ((1ull << T) - 1)));
    const int S = ntaps * nchunks;

    f32x4v acc[AM][AT];
    f32x4v acc_lo[NTERMS == 3 ? AM : 1][NTERMS == 3 ? AT : 1];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (NTERMS == 3) acc_lo[i][j][r] = 0.f;
            }

    // reduction cursor (as conv_igemm.hip: taps innermost unless the taps are widely dilated -- dbg_flags bit 12)
    const bool tap_inner = !(a.dbg_flags & 4096);
    int ld_tap = -1, ld_tr = 0, ld_ts = -1, ld_chunk = tap_inner ? 0 : nchunks - 1;
    auto advance = [&]() {
        if (tap_inner) {
            do {
                ++ld_tap;
                if (++ld_ts == a.TS) { ld_ts = 0; ++ld_tr; }
                if (ld_tap == T) { ld_tap = 0; ld_tr = 0; ld_ts = 0; ++ld_chunk; }
            } while (!((tapmask >> ld_tap) & 1ull));
        } else if (++ld_chunk == nchunks) {
            ld_chunk = 0;
            do {
                ++ld_tap;
                if (++ld_ts == a.TS) { ld_ts = 0; ++ld_tr; }
            } while (!((tapmask >> ld_tap) & 1ull));
        }
    };
    // LDS-DMA of the next reduction tile into `stage`: per thread 2 pixel rows + 2 filter rows, NPL planes each.
    // Masked lanes (padding tap, row past M, channels past Cin, filter row past N) get an out-of-range offset: the hardware
    // writes zeros for them.
    char* const dstA = lds + (32 * wave) * ROW;              // + stage * STAGE + plane * BM * ROW + RPI i * ROW
    char* const dstB = lds + OFF_B + (RPI * BI * wave) * ROW;
    auto issue = [&](int stage) {
        advance();
        const int dh = a.dh0 + ld_tr * a.dh_step, dw = a.dw0 + ld_ts * a.dw_step;
        const int woff = a.w_off0 + ld_tr * a.w_step_r + ld_ts * a.w_step_s;
        const bool cok = ld_chunk * KS + 8 * lc < a.Cin;                 // Cin % 8 == 0; only the last chunk can be partial
        const unsigned tapdelta = (unsigned)(((dh * a.IW + dw) * a.x_pitch + ld_chunk * KS) * 2) * ilma;      // wave-uniform, may be "negative"
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((woff + ld_chunk * KS) * 2) * ilmb));      // an SGPR operand: anything less and hipcc wraps the DMA in a waterfall loop
        char* const sa = dstA + stage * STAGE;
        char* const sb = dstB + stage * STAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const bool ok = cok & ((unsigned)(rowh[i] + dh) < (unsigned)a.IH) & ((unsigned)(roww[i] + dw) < (unsigned)a.IW);      // no short-circuit: branch-free
            const unsigned vo = ok ? xoff[i] + tapdelta : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_vptr)(sa + RPI * i * ROW), 16, vo, 0, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_vptr)(sa + BM * ROW + RPI * i * ROW), 16, vo, 0, 0, 0);
        }
        // filter rows: a lane past Cin in the last chunk gets the out-of-range bit ORed onto its offset (arithmetic, not a select: hipcc turned
        // `cok ? woff_row : OOB` into a divergent branch with the DMA issued once per side)
        const unsigned oob = (unsigned)(a.Cin - 1 - (ld_chunk * KS + 8 * lc)) & OOB;      // sign bit of (Cin - 1 - k)
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const unsigned vo = woff_row[i] | oob;                    // (woff_row < 2^31, or already OOB for a row past N)
            const unsigned vo1 = (woff_row[i] + plane1_w) | oob | (woff_row[i] & OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(sb + RPI * i * ROW), 16, vo, so, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(sb + BN * ROW + RPI * i * ROW), 16, vo1, so, 0, 0);
        }
    };
    // fragment reads: lane l = row (l & 15) of a 16-row fragment, reduction elements 8 (l >> 4) .. +7 of a 32-deep (sub-)step
    const int koff = ROW == 64 ? 16 * ((lane >> 4) ^ (((lane >> 2) & 1) << 1)) : 16 * ((lane >> 4) ^ ((lane >> 1) & 3));
    const char* const ra_base = lds + (wave_m * WM + (lane & 15)) * ROW + koff;
    const char* const rb_base = lds + OFF_B + (wave_n * WN + (lane & 15)) * ROW + koff;
    auto compute = [&](int stage) {
#pragma unroll
        for (int u = 0; u < KS / 32; ++u) {                 // 128-byte rows: two 32-deep sub-steps, the second in the rows' upper 64 bytes
            const char* pa = ra_base + stage * STAGE + 64 * u;
            const char* pb = rb_base + stage * STAGE + 64 * u;
            f16x8 fb[AT][NPL];
#pragma unroll
            for (int j = 0; j < AT; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * ROW + j * 16 * ROW);
            f16x8 fa[AM][NPL];
#pragma unroll
            for (int i = 0; i < AM; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fa[i][pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * ROW + i * 16 * ROW);
#pragma unroll
            for (int i = 0; i < AM; ++i) {
#pragma unroll
                for (int j = 0; j < AT; ++j) {
                    // the filter fragment is the FIRST operand: the 16x16 result comes out transposed (lane l: pixel l & 15, channels
                    // 4 (l >> 4) .. +3) -> 16-byte epilogue stores.  Same term order as conv_igemm.hip (bit-identical sums).
                    if constexpr (NTERMS == 3) {
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][NPL - 1], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][NPL - 1], fa[i][0], acc_lo[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][0], acc[i][j], 0, 0, 0);
                }
            }
            pl_step_schedule<AM, NPL, NTERMS>();
        }
    };

    // ---- main loop, ONE barrier per K-step.  LDS-DMA data is ordered for a ds_read only by the ISSUING wave's vmcnt followed by a
    // barrier the reader has passed; hipcc does not wait for a DMA before an LDS read, so the waits are written out.
    if constexpr (NST == 2) {
        // two stages: __syncthreads() = s_waitcnt vmcnt(0) (this wave's share of tile s has landed) + s_barrier (every wave's has, and
        // every wave has finished reading the stage that tile s+1 is about to overwrite)
        if (S > 0) {
            PL_STAMP();
            issue(0);
            for (int s = 0; s < S; ++s) {
                PL_STAMP();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                PL_STAMP();
                // STAMPS builds only -- timing ablations (results are garbage): debug flag 64 = no DMA in the loop, 128 = no MFMA phase
                if (s + 1 < S && !(STAMPS && (a.dbg_flags & 64))) issue((s + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                PL_STAMP();
                if (!(STAMPS && (a.dbg_flags & 128))) compute(s & 1);
                if constexpr (STAMPS) __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // three stages: tiles s+1 and s+2 are in flight while tile s is computed.  At the top of step s this wave's DMA of tile s
        // must have landed, that of tile s+1 (NDMA instructions, issued later, in order) may still be on its way: vmcnt(NDMA).
        // Behind the barrier every wave has finished computing tile s-1, so its stage takes tile s+2.
        if (S > 0) {
            PL_STAMP();
            issue(0);
            if (S > 1) issue(1);
            int sc = 0, sn = 2;                                  // stage of the tile being computed / of the tile being requested
            // The waves of a block pass the barrier together, and a wave whose DMA issue waits for room in the CU's (saturated) load
            // queue issues no MFMAs: in lockstep the step costs DMA time PLUS MFMA time.  So the two waves of every SIMD (w, w + 4)
            // take opposite orders -- the first requests tile s+2 and then computes tile s, the second computes and then requests:
            // while one waits on the queue its partner owns the matrix pipe.  (Debug flag 32768: all waves request first.)
            const bool late = wave >= NW / 2 && !(a.dbg_flags & 32768);
            for (int s = 0; s < S; ++s) {
                PL_STAMP();
                if (s + 1 < S) {
                    static_assert(NDMA == 6, "256-row tiles: 6 DMA instructions per wave and K-step (f16x3: (2 + 1) x 2 planes; one plane, 128-byte rows: 4 + 2)");
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                PL_STAMP();
                if (!late && s + 2 < S && !(STAMPS && (a.dbg_flags & 64))) issue(sn);
                __builtin_amdgcn_sched_barrier(0);
                PL_STAMP();
                if (!(STAMPS && (a.dbg_flags & 128))) compute(sc);
                __builtin_amdgcn_sched_barrier(0);
                if (late && s + 2 < S && !(STAMPS && (a.dbg_flags & 64))) issue(sn);
                __builtin_amdgcn_sched_barrier(0);
                sc = sc == 2 ? 0 : sc + 1;
                sn = sn == 2 ? 0 : sn + 1;
            }
        }
    }
    PL_STAMP();
    __syncthreads();          // LDS stage 0 is reused for the statistics; orders the row table when S == 0

    pl_epilogue<NTERMS, BM, BNB, AM, EP>(a, acc, acc_lo, rowoff, reinterpret_cast<float*>(smem), tile / a.tiles_n, n0, wave_m, wave_n, lane, tid, m0 + BM <= a.M);
    if constexpr (STAMPS) {
        __builtin_amdgcn_sched_barrier(0);
        PL_STAMP();
        if (stamper && a.dbg != nullptr)
            for (int k = 0; k < 256; ++k) a.dbg[(wave != 0) * 256 + k] = k < n_stamp ? stamp_base[(wave != 0) * 256 + k] : 0ull;
    }
#undef PL_STAMP
}

// -------------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions (any stride, no padding) on SHORT reductions: gg_pl_kernel's 128 x 128 tile as a PERSISTENT kernel.
// -------------------------------------------------------------------------------------------------------------------------
// A K = 256 tile of gg_pl_kernel spends 2.1 k cycles in front of its loop (geometry, the first operand round trip), 16.8 k in it, 8.6 k in
// its epilogue and ~3 k leaving and being replaced (profiles/r05_stamps_1x1.txt): the two blocks of a CU are independent serial chains,
// and 45 % of each chain is not the loop.  Here a block walks a strided share of the tiles as ONE stream of K-steps:
//   * the operand DMA runs through tile boundaries -- the next tile's FIRST stage is requested at the top of this tile's last step (the
//     ordinary one-step lead) and its SECOND stage right after the last step's barrier, BEFORE the epilogue, into the stage that step freed;
//   * gfx9 counts loads, stores and LDS-DMA in ONE in-order vmcnt, so a wave cannot wait for an operand that is younger than its output stores
//     without waiting for the stores -- but both of those stages are OLDER than the stores: the next tile's first two steps wait with
//     vmcnt(<stores in flight>) and the stores have two K-steps to be acknowledged before the first younger DMA (step 2's) is waited for;
//   * no block leaves, so nothing waits for a wave's stores at s_endpgm, for a dispatch, or for a prologue's dependent global round trip.
// Same pieces, same MFMA order, same epilogue (pl_epilogue) as gg_pl_kernel<NTERMS, 128>: bit-identical results
// (tests/test_planes_gpu.py::test_persistent_1x1_kernel_is_bit_identical).  T == 1 only: a tap mask would need a block barrier per tile.
// The lean epilogue of an edge-free fp32-output tile WITHOUT statistics (plain dgrad; dgrad + residual-gradient source, masked or not) in the
// order the persistent kernel needs (same expressions per element as pl_epilogue_lean<4, 0, PREV, false>: bit-identical):
//   (1) row offsets from LDS, (2) EVERY load of the residual-gradient source and its mask bits, (3) `between()` = the caller's operand DMA for
//   the next tile, (4) the output stores.
// Why the order: a wait for a load's data is a wait for everything older in gfx9's one in-order vmcnt.  With the loads in front of the DMA hipcc's
// counted waits for them leave the DMA in flight; behind it -- or, as in pl_epilogue_lean, the second half behind the first stores -- they
// would wait for the DMA and for those stores.  acc holds the FOLDED accumulators.
template <int PREV, typename Between, typename Stamp>
__device__ __forceinline__ void plp_epilogue_lean(const GatherGemmArgs& a, f32x4v (&acc)[4][4], const int* rowoff, int n0, int wave_m, int wave_n, int lane, float c,
                                                  const float* extra, const unsigned char* amask, Between&& between, Stamp&& stamp) {
    constexpr int AM = 4, AT = 4, WM = 64, WN = 64;
    const int nb = n0 + wave_n * WN + 4 * (lane >> 4);
    unsigned offs[AM], offsP[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i) {
        offs[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + (lane & 15)] + nb);
        offsP[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + ((lane & 15) ^ 8)] + nb);
    }
    const bool upper = (lane & 8) != 0;
    f32x4v la[PREV ? 2 : 1][PREV ? AM : 1], lb[PREV ? 2 : 1][PREV ? AM : 1];
    uint2 mw[PREV == 2 ? AM : 1];            // the 64 mask bits of this lane's pixel row and the wave's 64 channels: ONE 8-byte load per row (the bytes (offs + 16 j) >> 3, j = 0..3, lie in one aligned 8-byte word: dense rows of a multiple of 128 channels)
    if constexpr (PREV != 0) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const unsigned col = (upper ? 2 * jp + 1 : 2 * jp) * 16;
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                la[jp][i] = *reinterpret_cast<const f32x4v*>(extra + ((upper ? offsP[i] : offs[i]) + col));      // rows 0-7 of the fragment
                lb[jp][i] = *reinterpret_cast<const f32x4v*>(extra + ((upper ? offs[i] : offsP[i]) + col));      // rows 8-15
            }
        }
        if constexpr (PREV == 2) {
#pragma unroll
            for (int i = 0; i < AM; ++i) mw[i] = *reinterpret_cast<const uint2*>(amask + (((offs[i] - 4u * (unsigned)(lane >> 4)) >> 3)));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    between();
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    f32x4v hold[AM];
#pragma unroll
    for (int j = 0; j < AT; ++j) {
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            f32x4v v = epi_mul(acc[i][j], c);                       // (fma(t, c, 0) == t c: a power of two, exact)
            if constexpr (PREV != 0) {
                f32x4v prev;
                if ((j & 1) == 0) {
                    // the whole-line loads' swap (pl_epilogue_lean): lanes l / l ^ 8 hand each other the column group that belongs to the other
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float send = upper ? la[j >> 1][i][r] : lb[j >> 1][i][r];
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xF, 0xF, true));
                        prev[r] = upper ? recv : la[j >> 1][i][r];          // this lane's even group
                        la[j >> 1][i][r] = upper ? lb[j >> 1][i][r] : recv;  // ... and its odd group, kept for the next column group
                    }
                } else {
                    prev = la[j >> 1][i];
                }
                if constexpr (PREV == 2) {
                    // element (row i, channel 16 j + 4 (lane >> 4) + r of the wave's 64) = bit 16 j + 4 (lane >> 4) + r of the row's word: the same bit
                    // pl_epilogue_lean takes from byte (offs + 16 j) >> 3, nibble ((offs + 16 j) >> 2) & 1
                    const int nib = (int)(((j < 2) ? mw[i].x : mw[i].y) >> (16u * (unsigned)(j & 1) + 4u * (unsigned)(lane >> 4)));
#pragma unroll
                    for (int r = 0; r < 4; ++r) prev[r] = __uint_as_float(__float_as_uint(prev[r]) & (unsigned)__builtin_amdgcn_sbfe(nib, r, 1));
                }
                v = epi_add(v, prev);
            }
            if ((j & 1) == 0) {
                hold[i] = v;
            } else {
                f32x4v recv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float send = upper ? hold[i][r] : v[r];
                    recv[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xF, 0xF, true));
                }
                const unsigned col = (upper ? j : j - 1) * 16;
                f32x4v da, db;
#pragma unroll
                for (int r = 0; r < 4; ++r) { da[r] = upper ? recv[r] : hold[i][r]; db[r] = upper ? v[r] : recv[r]; }
                *reinterpret_cast<f32x4v*>(a.y + ((upper ? offsP[i] : offs[i]) + col)) = da;      // rows 0-7 of the fragment, 128 B each
                *reinterpret_cast<f32x4v*>(a.y + ((upper ? offs[i] : offsP[i]) + col)) = db;      // rows 8-15
            }
        }
    }
}

// The statistics epilogue of the persistent kernel in gg_pl_kernel's own order -- per column group: statistics, then (at every second group) the
// whole-line stores -- with the operand DMA of the next tile issued FIRST.  What makes that order possible behind an LDS-DMA in flight: the
// partials go through LDS by INLINE ASM (ds_write_b32 / ds_read_b32), which hipcc's wait-count pass does not see -- a C++ access to LDS there
// gets a vmcnt wait for the DMA and, being in program order behind the first stores, for those stores too (plp_epilogue_lean above avoids
// that by taking everything it reads before the DMA and all stores after it; a statistics-first form of this epilogue was 1-2 % slower on
// the forward 1x1 class: profiles/r06_plp_stats_order_ab.txt).  Same expressions and
// order of additions: bit-identical.
template <typename Between, typename Stamp>
__device__ __forceinline__ void plp_epilogue_stats(const GatherGemmArgs& a, f32x4v (&acc)[4][4], const int* rowoff, float* sred, int tile_m, int n0,
                                                   int wave_m, int wave_n, int lane, int tid, float c, Between&& between, Stamp&& stamp) {
    constexpr int AM = 4, AT = 4, WM = 64, WN = 64, BN = PL_BN, BM = 128;
    const int nb = n0 + wave_n * WN + 4 * (lane >> 4);
    unsigned offs[AM], offsP[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i) {
        offs[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + (lane & 15)] + nb);
        offsP[i] = (unsigned)(rowoff[wave_m * WM + i * 16 + ((lane & 15) ^ 8)] + nb);
    }
    const bool upper = (lane & 8) != 0;
    // LDS byte addresses (inline asm takes the 32-bit address): this lane's partial slots, and the combine's
    const unsigned sdst_at = (unsigned)(uintptr_t)(lds_vptr)(sred + ((wave_m * BN) + wave_n * WN + 4 * (lane >> 4)) * 2);
    const unsigned comb_at = (unsigned)(uintptr_t)(lds_vptr)(sred + (tid & (BN - 1)) * 2);
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    between();
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    f32x4v hold[AM];
#pragma unroll
    for (int j = 0; j < AT; ++j) {
        f32x4v cs = {0.f, 0.f, 0.f, 0.f}, css = cs;
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            const f32x4v t = acc[i][j];
            cs = i == 0 ? t : epi_add(cs, t);
            css = i == 0 ? epi_sq(t) : epi_add(css, epi_sq(t));
            const f32x4v v = epi_mul(t, c);
            if ((j & 1) == 0) {
                hold[i] = v;
            } else {
                f32x4v recv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float send = upper ? hold[i][r] : v[r];
                    recv[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xF, 0xF, true));
                }
                const unsigned col = (upper ? j : j - 1) * 16;
                f32x4v da, db;
#pragma unroll
                for (int r = 0; r < 4; ++r) { da[r] = upper ? recv[r] : hold[i][r]; db[r] = upper ? v[r] : recv[r]; }
                *reinterpret_cast<f32x4v*>(a.y + ((upper ? offsP[i] : offs[i]) + col)) = da;
                *reinterpret_cast<f32x4v*>(a.y + ((upper ? offs[i] : offsP[i]) + col)) = db;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[r] = row_sum16(cs[r]); css[r] = row_sum16(css[r]); }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s0 = cs[r] * c, s1 = css[r] * c * c;
                asm volatile("ds_write_b32 %0, %1" :: "v"(sdst_at + (unsigned)((j * 16 + r) * 8)), "v"(s0) : "memory");
                asm volatile("ds_write_b32 %0, %1" :: "v"(sdst_at + (unsigned)((j * 16 + r) * 8 + 4)), "v"(s1) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < BN) {
        const int n = n0 + tid;
        if (n < a.N_store) {
            float p00, p01, p10, p11;          // [wave_m 0 | 1][sum | sumsq] of this column
            asm volatile("ds_read_b32 %0, %1" : "=v"(p00) : "v"(comb_at) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(p01) : "v"(comb_at + 4u) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(p10) : "v"(comb_at + (unsigned)(BN * 8)) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(p11) : "v"(comb_at + (unsigned)(BN * 8 + 4)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the asm reads are not tied to the wait by data flow: keep the additions behind it)
            asm volatile("" : "+v"(p00), "+v"(p01), "+v"(p10), "+v"(p11));
            float sm = 0.f, sq = 0.f;
            sm += p00; sq += p01;
            sm += p10; sq += p11;
            float* dst = a.stats + (size_t)tile_m * 2 * a.N_store;
            dst[n] = sm;
            dst[a.N_store + n] = sq;
        }
    }
    static_assert(BM / WM == 2, "two row halves per statistics column");
}

template <int NTERMS>
constexpr int plp_lds_bytes() { return 2 * pl_stage_bytes<NTERMS, 128>() + 4 * 128 * 4 + 2 * PL_BN * 2 * 4 + 64; }

// s_waitcnt vmcnt(n) with the other counters left alone, as the BUILTIN (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt [6:4],
// lgkmcnt [11:8]): hipcc's wait-count pass reads it, so it knows what is still in flight -- behind an inline-asm wait it assumed the LDS-DMA
// of the previous step pending and put its own vmcnt(0) in front of the next LDS read
#define PLP_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
__device__ __forceinline__ void plp_wait_vm(int n) {
    // (the immediate must be a constant; every value pl_epilogue can return for AM = 4)
    if (n >= 16) PLP_VMCNT(16);
    else if (n >= 8) PLP_VMCNT(8);
    else PLP_VMCNT(0);
}

// EPK: the ONE lean epilogue this instantiation holds -- 0 plain, 1 + BatchNorm statistics, 2 + residual-gradient source, 3 + ... under the 1-bit
// mask -- chosen by the host (it is a property of the launch).  One kernel with all four was measured first: hipcc's wait-count pass merges
// the pending-load state of the paths that load a residual into the paths that load nothing, and protected registers with vmcnt waits
// between the DMA and the stores of EVERY path and in front of the next tile's first writes (the stores were waited for after all).
// STAMPS (tools/plp_stamps.py): lane 0 of waves 0 and 3 of block `dbg_flags >> 16` keeps s_memtime stamps in LDS -- written by inline asm, which
// hipcc's wait-count pass does not see: a C++ store to LDS behind an LDS-DMA gets a vmcnt wait of its own -- and dumps them to a.dbg at the end.
template <int NTERMS, int EPK, bool STAMPS = false>
__global__ __launch_bounds__(256, 2) void gg_plp_kernel(const GatherGemmArgs a) {
    constexpr int BM = 128, BN = PL_BN, AM = 4, WM = 64, WN = 64, AT = 4, ROW = pl_row_bytes<NTERMS>();
    constexpr int KS = ROW / 2, RPI = 1024 / ROW, NW = 4, AI = 32 / RPI, BI = BN / (RPI * NW), NPL = NTERMS == 3 ? 2 : 1;
    constexpr int STAGE = pl_stage_bytes<NTERMS, BM>();
    constexpr int OFF_B = NPL * BM * ROW;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // ONE LDS object (a second one makes hipcc drain the DMA early)
    char* lds = reinterpret_cast<char*>(smem);
    // [4][BM] output row tables, by tile sequence number & 3: with two-step tiles the table of tile k + 1 is written (when the last DMA of tile k
    // goes out) BEFORE the epilogue of tile k - 1 has read its own
    int* const rowoff4 = reinterpret_cast<int*>(lds + 2 * STAGE);
    float* const sred = reinterpret_cast<float*>(lds + 2 * STAGE + 4 * BM * 4);                // [BM / WM][BN][2] statistics partials (its own room: the stages never rest)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const bool stamper = STAMPS && (int)blockIdx.x == (a.dbg_flags >> 16) && lane == 0 && (wave == 0 || wave == 3);
    unsigned stamp_at = (unsigned)(uintptr_t)(lds_vptr)(lds + plp_lds_bytes<NTERMS>() + (wave != 0) * 2048);      // LDS byte address of the next stamp
    int n_stamp = 0;
    auto stamp = [&]() {
        if constexpr (STAMPS) {
            if (stamper && n_stamp < 250) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                asm volatile("ds_write_b64 %0, %1" :: "v"(stamp_at), "v"(t) : "memory");
                stamp_at += 8;
                ++n_stamp;
            }
        }
    };
    stamp();
    if (a.stagger > 0) {                                            // de-phase the two blocks of a CU once (gg_pl_kernel)
        const unsigned lds_base = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 0 << 6 | 6);
        if (lds_base != 0)
            for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    const int lrow = ROW == 64 ? lane >> 2 : lane >> 3;
    const int lc = ROW == 64 ? (lane & 3) ^ (((lane >> 4) & 1) << 1) : ((lane & 4) | ((lane & 3) ^ ((lane >> 4) & 3)));
    const unsigned ilmb = (unsigned)__builtin_amdgcn_readfirstlane(a.w_il ? 2 : 1);
    const bool ila = NPL == 2 && planes_il(a.x_plane_stride);
    const unsigned ilma = (unsigned)__builtin_amdgcn_readfirstlane(ila ? 2 : 1);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_planes), 0, (int)(a.w_plane_stride * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)(a.x_bytes * ilma), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.x_planes) + a.x_plane_stride * 2), 0, (int)(a.x_bytes * ilma - (ila ? 64 : 0)), 0x00020000);
    const unsigned plane1_w = a.w_il ? 64u : (unsigned)(a.w_plane_stride * 2);
    const int S = (a.Cin + KS - 1) / KS;                             // K-steps per tile (>= 2: launch_gg_pl)
    const int n_tiles = a.n_tiles, G = (int)gridDim.x;

    // ---- the LOAD side: operand rows of the tile whose DMA is being issued (runs up to two steps ahead of the tile being multiplied).
    //      launch_gg_pl sends only launches without an edge here (M % 128 == 0, N_store % 128 == 0) whose input pixel IS the output pixel ----
    unsigned woff_row[BI], xoff[AI];
    int ld_v = (int)blockIdx.x, ld_k = 0, ld_chunk = 0;              // virtual block id, sequence number and K-chunk of the next DMA
    auto setup_load = [&]() {
        const int tile = xcd_remap(ld_v, n_tiles);
        const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int n = n0 + RPI * BI * wave + RPI * i + lrow;
            const unsigned lcb = a.w_il ? ((unsigned)(lc >> 2) * 128u + (unsigned)(lc & 3) * 16u) : 16u * lc;
            woff_row[i] = n < a.N ? (unsigned)n * (unsigned)a.w_row_stride * ilmb * 2u + lcb : OOB;      // (N may be below N_store: pad channels fetch zeros)
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = m0 + 32 * wave + RPI * i + lrow;
            xoff[i] = ((unsigned)m * (unsigned)a.x_pitch * ilma + 8u * lc) * 2u;
        }
        if (tid < BM) rowoff4[(ld_k & 3) * BM + tid] = (m0 + tid) * a.y_pitch;      // the tile's output row table (read by ITS epilogue, at least two barriers from here)
    };
    char* const dstA = lds + (32 * wave) * ROW;
    char* const dstB = lds + OFF_B + (RPI * BI * wave) * ROW;
    // the next DMA of the stream into `stage`; false once the block's share of the tiles is exhausted
    auto issue = [&](int stage) -> bool {
        if (ld_v >= n_tiles) return false;
        const bool cok = ld_chunk * KS + 8 * lc < a.Cin;
        const unsigned tapdelta = (unsigned)(ld_chunk * KS * 2) * ilma;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((a.w_off0 + ld_chunk * KS) * 2) * ilmb));
        char* const sa = dstA + stage * STAGE;
        char* const sb = dstB + stage * STAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const unsigned vo = cok ? xoff[i] + tapdelta : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_vptr)(sa + RPI * i * ROW), 16, vo, 0, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_vptr)(sa + BM * ROW + RPI * i * ROW), 16, vo, 0, 0, 0);
        }
        const unsigned oob = (unsigned)(a.Cin - 1 - (ld_chunk * KS + 8 * lc)) & OOB;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const unsigned vo = woff_row[i] | oob;
            const unsigned vo1 = (woff_row[i] + plane1_w) | oob | (woff_row[i] & OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(sb + RPI * i * ROW), 16, vo, so, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(sb + BN * ROW + RPI * i * ROW), 16, vo1, so, 0, 0);
        }
        if (++ld_chunk == S) {             // the stream moves on to the block's next tile
            ld_chunk = 0;
            ld_v += G;
            ++ld_k;
            if (ld_v < n_tiles) setup_load();
        }
        return true;
    };
    const int koff = ROW == 64 ? 16 * ((lane >> 4) ^ (((lane >> 2) & 1) << 1)) : 16 * ((lane >> 4) ^ ((lane >> 1) & 3));
    const char* const ra_base = lds + (wave_m * WM + (lane & 15)) * ROW + koff;
    const char* const rb_base = lds + OFF_B + (wave_n * WN + (lane & 15)) * ROW + koff;

    f32x4v acc[AM][AT];
    f32x4v acc_lo[NTERMS == 3 ? AM : 1][NTERMS == 3 ? AT : 1];
    auto compute = [&](int stage) {
#pragma unroll
        for (int u = 0; u < KS / 32; ++u) {
            const char* pa = ra_base + stage * STAGE + 64 * u;
            const char* pb = rb_base + stage * STAGE + 64 * u;
            f16x8 fb[AT][NPL];
#pragma unroll
            for (int j = 0; j < AT; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * ROW + j * 16 * ROW);
            f16x8 fa[AM][NPL];
#pragma unroll
            for (int i = 0; i < AM; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fa[i][pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * ROW + i * 16 * ROW);
#pragma unroll
            for (int i = 0; i < AM; ++i) {
#pragma unroll
                for (int j = 0; j < AT; ++j) {
                    if constexpr (NTERMS == 3) {
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][NPL - 1], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][NPL - 1], fa[i][0], acc_lo[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][0], acc[i][j], 0, 0, 0);
                }
            }
            pl_step_schedule<AM, NPL, NTERMS>();
        }
    };

    const unsigned ranges[2] = {a.amax_x ? *a.amax_x : 0u, a.amax_w ? *a.amax_w : 0u};      // once: inside the tile loop every global load is a wait for the next tile's operand DMA
    const float c_unscale = pow2_inv(a.amax_x ? pow2_scale_for(ranges[0]) : 1.f) * pow2_inv(a.amax_w ? pow2_scale_for(ranges[1]) : 1.f);
    setup_load();
    issue(0);
    // ONE loop over the block's K-steps (tile boundaries are a branch inside it: a nested loop gets its first iteration peeled, and the
    // peeled copy came with a compiler-made vmcnt(0) between its LDS reads -- a wait for the previous tile's stores in every tile).
    // State that crosses a boundary:
    //   soft_waits  steps from now whose operands are OLDER than the stores in flight (requested before the previous epilogue)
    //   skip_issue  steps from now that request nothing (their successor's operands are already on their way)
    const int my_tiles = (n_tiles - 1 - (int)blockIdx.x) / G + 1;
    const int total = my_tiles * S;
    int stores_in_flight = 0, soft_waits = 0, skip_issue = 0;
    int v = (int)blockIdx.x, k = 0, s = 0;
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int j = 0; j < AT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[i][j][r] = 0.f;
                    if constexpr (NTERMS == 3) acc_lo[i][j][r] = 0.f;
                }
    };
    zero_acc();
#pragma clang loop unroll(disable)
    for (int g = 0; g < total; ++g) {       // global K-step of this block: its operands live in stage g & 1
        // this wave's share of step g must have landed
        stamp();
        if (soft_waits > 0) { plp_wait_vm(stores_in_flight); --soft_waits; }
        else PLP_VMCNT(0);
        if constexpr (STAMPS) { __builtin_amdgcn_sched_barrier(0); stamp(); }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        if (skip_issue > 0) --skip_issue;
        else issue((g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        compute(g & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (++s == S) {
            // ---- tile boundary.  Every wave is done reading this step's stage once it is past the barrier: the stage takes the stream's next
            //      DMA -- the next tile's SECOND step (its first went out at the top of this step) -- before the epilogue puts its stores into
            //      the counter
            s = 0;
            stamp();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp();
            const int tile = xcd_remap(v, n_tiles);
            const int n0 = (tile % a.tiles_n) * BN;
            bool pre2 = false;
            if constexpr (NTERMS == 3) {
#pragma unroll
                for (int i = 0; i < AM; ++i)
#pragma unroll
                    for (int j = 0; j < AT; ++j) acc[i][j] = epi_fma(acc_lo[i][j], 1.f / 2048.f, acc[i][j]);
            }
            auto between = [&]() { pre2 = issue(g & 1); };
            if constexpr (EPK == 1)
                plp_epilogue_stats(a, acc, rowoff4 + (k & 3) * BM, sred, tile / a.tiles_n, n0, wave_m, wave_n, lane, tid, c_unscale, between, stamp);
            else
                plp_epilogue_lean<EPK >= 2 ? EPK - 1 : 0>(a, acc, rowoff4 + (k & 3) * BM, n0, wave_m, wave_n, lane, c_unscale, a.add_src != nullptr ? a.add_src : a.y,
                                                          a.add_mask, between, stamp);
            stamp();
            // (true already -- the 16 stores are this wave's youngest operations -- but said to hipcc's wait-count pass in its own terms)
            PLP_VMCNT(16);
            stores_in_flight = 16;
            __builtin_amdgcn_sched_barrier(0);
            zero_acc();
            v += G;
            ++k;
            soft_waits = (pre2 && !(a.dbg_flags & 2097152)) ? 2 : 0;      // (bit 21: wait for everything at every step -- A/B)
            skip_issue = pre2 ? 1 : 0;
        }
    }
    if constexpr (STAMPS) {
        stamp();
        PLP_VMCNT(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (stamper && a.dbg != nullptr) {
            const unsigned long long* sb = reinterpret_cast<const unsigned long long*>(lds + plp_lds_bytes<NTERMS>() + (wave != 0) * 2048);
            for (int q = 0; q < 256; ++q) a.dbg[(wave != 0) * 256 + q] = q < n_stamp ? sb[q] : 0ull;
        }
    }
}

template __global__ void gg_plp_kernel<3, 0>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<3, 0, true>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<3, 1, true>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<3, 1>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<3, 2>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<3, 3>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<1, 0>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<1, 1>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<1, 2>(const GatherGemmArgs);
template __global__ void gg_plp_kernel<1, 3>(const GatherGemmArgs);

// -------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / dilation 1 with the input HALO kept in LDS.
// -------------------------------------------------------------------------------------------------------------------------
// gg_pl_kernel re-fetches a pixel-row tile for each of the nine taps: per channel chunk 9 x (32 + 16) KB of operand DMA for a
// 256 x 128 tile, and the L2 -> LDS path is what bounds that kernel (header).  Here an M-tile is a 16 x 16 PATCH of output pixels of
// one image; per 32-channel chunk the 18 x 18 input halo of the patch is brought to LDS ONCE (324 rows of 64 bytes per plane, 24
// DMA pieces) and the nine taps read their fragments from it at shifted rows: pixel (py, px), tap (r, s) -> halo row
// (py + r') * 18 + px + s'.  A-operand DMA per chunk drops from 288 KB to 48 KB; with the filter tiles (16 KB per tap) the kernel moves
// 21 KB per K-step instead of 48.  The chunk swizzle ((row >> 2) & 1) << 1 depends on the row's position within its group of
// eight only, so 16 consecutive halo rows starting ANYWHERE read conflict-free (the four rows of a bank-quarter residue get
// four different positions).  Padding pixels are DMA'd as zeros (out-of-range offsets), so no tap is ever skipped or masked.
//
// Block: 8 waves (4 patch-row groups x 2 channel halves, 64 x 64 outputs each as in gg_pl_kernel), one per CU.  LDS: two halo
// buffers of 48 KB (chunk c+1 arrives while chunk c is computed) + a ring of three filter stages of 16 KB = 144 KB.  One raw
// barrier per K-step; DMA order per wave: [filter tile s+2] at every step, [halo c+1] at the first tap of chunk c; the counted
// vmcnt at the top of a step leaves exactly the younger requests in flight.
// Conditions (launch_gg_pl): 3 x 3 taps with unit steps (forward pad 1 / valid, and the stride-1 dgrad of either), output height
// and width multiples of 16, output geometry == pixel geometry.
constexpr int PLH_HW = 18;                                   // halo width / height (16 + 2)
constexpr int PLH_PIECES = 24;                               // 16-row DMA pieces per plane (21 cover the 324 halo rows; 3 per wave)
constexpr int PLH_HROWS = PLH_PIECES * 16;                   // 384
template <int NTERMS>
constexpr int plh_halo_bytes() { return (NTERMS == 3 ? 2 : 1) * PLH_HROWS * PL_ROW; }
template <int NTERMS>
constexpr int plh_bstage_bytes() { return (NTERMS == 3 ? 2 : 1) * PL_BN * PL_ROW; }
template <int NTERMS>
constexpr int plh_lds_bytes() { return 2 * plh_halo_bytes<NTERMS>() + 3 * plh_bstage_bytes<NTERMS>() + 256 * 4 + 64; }

template <int NTERMS, bool BNB = false, bool EP = false>
__global__ __launch_bounds__(512, 2) void gg_plh_kernel(const GatherGemmArgs a) {
    constexpr int BM = 256, BN = PL_BN, WN = 64, AT = 4, ROW = PL_ROW;
    constexpr int NPL = NTERMS == 3 ? 2 : 1;
    constexpr int HALO = plh_halo_bytes<NTERMS>(), HPL = PLH_HROWS * ROW;          // one halo buffer / one plane of it
    constexpr int BST = plh_bstage_bytes<NTERMS>();
    constexpr int NB = NPL, NH = 3 * NPL;                     // DMA instructions per wave: filter tile / halo
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    char* const ldsB = lds + 2 * HALO;
    int* rowoff = reinterpret_cast<int*>(lds + 2 * HALO + 3 * BST);

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / a.tiles_n;
    const int n0 = (tile % a.tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int lc = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    // patch position: tm -> (image, patch row, patch column); the last patch row / column may overhang the output (ragged sizes: the
    // overhanging pixels are computed from whatever the halo holds and neither stored nor counted -- rowoff = -1)
    const int pw = (a.Q + 15) >> 4, ph = (a.P + 15) >> 4;
    const int tx = tm % pw, t1 = tm / pw;
    const int ty = t1 % ph, b = t1 / ph;
    const int y0 = ty * 16, x0 = tx * 16;
    const int hmin_h = a.dh_step > 0 ? a.dh0 : a.dh0 + 2 * a.dh_step;          // smallest tap offset = origin of the halo
    const int hmin_w = a.dw_step > 0 ? a.dw0 : a.dw0 + 2 * a.dw_step;

    const unsigned ilmb = (unsigned)__builtin_amdgcn_readfirstlane(a.w_il ? 2 : 1);      // chunk-interleaved filter planes, as in gg_pl_kernel (wave-uniform: scales the DMA's scalar offset)
    const bool ila = NPL == 2 && planes_il(a.x_plane_stride);                             // ... and chunk-interleaved pixel planes
    const unsigned ilma = (unsigned)__builtin_amdgcn_readfirstlane(ila ? 2 : 1);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_planes), 0, (int)(a.w_plane_stride * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)(a.x_bytes * ilma), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.x_planes) + a.x_plane_stride * 2), 0, (int)(a.x_bytes * ilma - (ila ? 64 : 0)), 0x00020000);

    // filter row of this thread (one 16-row piece per wave)
    const int nrow = n0 + 16 * wave + (lane >> 2);
    const unsigned woff_row = nrow < a.N ? ((unsigned)nrow * (unsigned)a.w_row_stride * ilmb + 8u * lc) * 2u : OOB;
    const unsigned plane1_w = a.w_il ? 64u : (unsigned)(a.w_plane_stride * 2);
    // halo rows of this thread: pieces wave, wave + 8, wave + 16 -> rows 16 g + (lane >> 2); input pixel or padding (zeros)
    unsigned hoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int h = 16 * (wave + 8 * j) + (lane >> 2);
        const int hy = h / PLH_HW, hx = h - hy * PLH_HW;
        const int yy = y0 + hmin_h + hy, xx = x0 + hmin_w + hx;
        const bool ok = (h < PLH_HW * PLH_HW) & ((unsigned)yy < (unsigned)a.IH) & ((unsigned)xx < (unsigned)a.IW);
        hoff[j] = ok ? ((unsigned)((b * a.IH + yy) * a.IW + xx) * (unsigned)a.x_pitch * ilma + 8u * lc) * 2u : OOB;
    }
    // output rows: patch pixel (py, px) = (row >> 4, row & 15)
    if (tid < BM) {
        const int py = tid >> 4, px = tid & 15;
        rowoff[tid] = (y0 + py < a.P && x0 + px < a.Q) ? (int)(((long long)(b * a.OH + y0 + py) * a.OW + x0 + px) * a.y_pitch) : -1;
    }

    const int nchunks = (a.Cin + BK - 1) / BK;
    const int S = 9 * nchunks;
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

    auto issue_halo = [&](int c) {                            // chunk c -> halo buffer c & 1
        char* const dst = lds + (c & 1) * HALO + (16 * wave) * ROW;
        const unsigned oob = (unsigned)(a.Cin - 1 - (c * BK + 8 * lc)) & OOB;
        const unsigned cb = (unsigned)(c * BK * 2) * ilma;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned vo = (hoff[j] + cb) | oob | (hoff[j] & OOB);      // (hoff < 2^31 - the chunk offsets, or exactly OOB for padding / rows past the halo)
            char* const d = dst + (128 * j) * ROW;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_vptr)d, 16, vo, 0, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_vptr)(d + HPL), 16, vo, 0, 0, 0);
        }
    };
    auto issue_b = [&](int s, int slot) {                      // filter tile of step s = (chunk, tap) -> ring slot
        const int c = s / 9, t = s - 9 * c;
        const int tr = t / 3, ts = t - 3 * tr;
        const unsigned oob = (unsigned)(a.Cin - 1 - (c * BK + 8 * lc)) & OOB;      // lanes past Cin (last chunk): out-of-range bit, by arithmetic (gg_pl_kernel)
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((a.w_off0 + tr * a.w_step_r + ts * a.w_step_s + c * BK) * 2) * ilmb));
        const unsigned vo = woff_row | oob;
        const unsigned vo1 = (woff_row + plane1_w) | oob | (woff_row & OOB);
        char* const d = ldsB + slot * BST + (16 * wave) * ROW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)d, 16, vo, so, 0, 0);
        if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(d + BN * ROW), 16, vo1, so, 0, 0);
    };
    const int l15 = lane & 15, kq = lane >> 4;
    const int koffB = 16 * (kq ^ (((lane >> 2) & 1) << 1));
    const char* const rb_base = ldsB + (wave_n * WN + l15) * ROW + koffB;
    auto compute = [&](int c, int t, int slot) {
        const int tr = t / 3, ts = t - 3 * tr;
        const int oh = a.dh0 + tr * a.dh_step - hmin_h, ow = a.dw0 + ts * a.dw_step - hmin_w;      // 0 .. 2
        const char* const ha = lds + (c & 1) * HALO;
        const char* pb = rb_base + slot * BST;
        f16x8 fb[AT][NPL];
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * ROW + j * 16 * ROW);
        const int hr0 = (wave_m * 4 + oh) * PLH_HW + ow + l15;                 // halo row of this lane's pixel in patch row 4 wave_m
        f16x8 fa[AT][NPL];
#pragma unroll
        for (int i = 0; i < AT; ++i) {
            const int hr = hr0 + i * PLH_HW;
            const char* pa = ha + hr * ROW + 16 * (kq ^ (((hr >> 2) & 1) << 1));
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[i][pl] = *reinterpret_cast<const f16x8*>(pa + pl * HPL);
        }
#pragma unroll
        for (int i = 0; i < AT; ++i) {
#pragma unroll
            for (int j = 0; j < AT; ++j) {
                if constexpr (NTERMS == 3) {
                    acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][NPL - 1], acc_lo[i][j], 0, 0, 0);
                    acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][NPL - 1], fa[i][0], acc_lo[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][0], acc[i][j], 0, 0, 0);
            }
        }
        pl_step_schedule<AT, NPL, NTERMS>();
    };

    // prologue: halo 0, filter tiles 0 and 1
    issue_halo(0);
    issue_b(0, 0);
    issue_b(1, 1);
    int c = 0, t = 0, slot = 0, nslot = 2;
    for (int s = 0; s < S; ++s) {
        // younger than filter tile s (and, at a chunk's first tap, than its halo) and allowed to stay in flight: filter tile s+1,
        // plus the halo of the next chunk if it was requested at one of the last two steps (first tap of this chunk = step s - t)
        const bool more = c + 1 < nchunks;
        const int young = (s + 1 < S ? NB : 0) + ((more && (t == 1 || t == 2)) ? NH : 0);
        if (young == NB + NH) {
            if constexpr (NPL == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (young == NB) {
            if constexpr (NPL == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < S) issue_b(s + 2, nslot);
        if (t == 0 && more) issue_halo(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(c, t, slot);
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == 2 ? 0 : slot + 1;
        nslot = nslot == 2 ? 0 : nslot + 1;
        if (++t == 9) { t = 0; ++c; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // LDS is reused for the statistics; orders the row table
    pl_epilogue<NTERMS, BM, BNB, 4, EP, false>(a, acc, acc_lo, rowoff, reinterpret_cast<float*>(smem), tm, n0, wave_m, wave_n, lane, tid, y0 + 16 <= a.P && x0 + 16 <= a.Q);
}

template __global__ void gg_plh_kernel<3>(const GatherGemmArgs);
template __global__ void gg_plh_kernel<1>(const GatherGemmArgs);

template __global__ void gg_pl_kernel<3, 128>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 256>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 128>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 256>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 128, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 256, true>(const GatherGemmArgs);
#ifdef PYLC_EXPERIMENTAL
template __global__ void gg_pl_kernel<3, 128, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 256, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 128, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 256, false, true>(const GatherGemmArgs);
#endif
template __global__ void gg_pl_kernel<3, 128, false, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 256, false, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 128, false, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 256, false, false, true>(const GatherGemmArgs);
#ifdef PYLC_EXPERIMENTAL
template __global__ void gg_plh_kernel<3, true>(const GatherGemmArgs);
template __global__ void gg_plh_kernel<1, true>(const GatherGemmArgs);
#endif
template __global__ void gg_pl_kernel<3, 128, false, false, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 256, false, false, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 128, false, false, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 256, false, false, false, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 128, false, false, true, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<3, 256, false, false, true, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 128, false, false, true, true>(const GatherGemmArgs);
template __global__ void gg_pl_kernel<1, 256, false, false, true, true>(const GatherGemmArgs);
template __global__ void gg_plh_kernel<3, false, true>(const GatherGemmArgs);
template __global__ void gg_plh_kernel<1, false, true>(const GatherGemmArgs);


// -------------------------------------------------------------------------------------------------------------------------
// The halo kernel for at most 64 output channels: 64 x 64 wave tiles, four waves, TWO blocks per CU.
// -------------------------------------------------------------------------------------------------------------------------
// A 64-channel 3x3 layer (the U-Net's first levels, unet.py:107-126; ResNet layer1's conv2) has one column tile, and the per-tap form
// (gg_pl_kernel<.., NARROW>; a 512 x 64 tile of eight 64 x 64 wave tiles was 4-9 % ahead of it and went the same way) pays the L2 -> LDS path for it: 64-byte pieces of the pixel tile arrive at ~30 B / cycle / CU, nine
// times per channel chunk (72 KB per 1536 MFMA cycles of a SIMD: 2.4 k cycles).  Round 3 / round 5 tried gg_plh_kernel's wave grid folded into
// the first 64 columns (eight waves of 32 x 64 on one 16 x 16 patch, one block per CU): half-length steps under the same barrier and issue costs, neutral.
// Here a block is FOUR waves, each with the full 64 x 64 wave tile (four patch rows x 64 channels: 0.5 fragment reads per MFMA group), on one
// 16 x 16 patch; ONE halo buffer (48 KB) + a ring of three 64-row filter stages (8 KB each) = 73 KB, so two blocks share a CU and the wait
// for a chunk's halo (nothing can be requested into the single buffer before the previous chunk's last tap has been read) is covered by the
// other block (a start delay for the CU's second block, as gg_pl_kernel<.., 128>'s stagger, measured no better than none).
// Same LDS image, K order and epilogue as gg_plh_kernel: y bit-identical; statistics partials per patch (a.halo_tiles_m rows).
template <int NTERMS>
constexpr int plhn_bstage_bytes() { return (NTERMS == 3 ? 2 : 1) * 64 * PL_ROW; }
template <int NTERMS>
constexpr int plhn_lds_bytes() { return plh_halo_bytes<NTERMS>() + 3 * plhn_bstage_bytes<NTERMS>() + 256 * 4 + 64; }

template <int NTERMS, bool EP = false>
__global__ __launch_bounds__(256, 2) void gg_plhn_kernel(const GatherGemmArgs a) {
    constexpr int BM = 256, BNN = 64, AT = 4, ROW = PL_ROW;
    constexpr int NPL = NTERMS == 3 ? 2 : 1;
    constexpr int HALO = plh_halo_bytes<NTERMS>(), HPL = PLH_HROWS * ROW;
    constexpr int BST = plhn_bstage_bytes<NTERMS>();
    constexpr int NB = NPL;                                   // filter-tile DMA instructions per wave and step
    constexpr int HP = PLH_PIECES / 4;                        // halo pieces per wave: 6
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    char* const ldsB = lds + HALO;
    int* rowoff = reinterpret_cast<int*>(lds + HALO + 3 * BST);

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / a.tiles_n;
    const int n0 = (tile % a.tiles_n) * BNN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.stagger > 0 && (int)blockIdx.x < a.stagger_blocks) {
        const unsigned lds_base = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 0 << 6 | 6);     // HW_REG_LDS_ALLOC.LDS_BASE: the CU's second block
        if (lds_base != 0)
            for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    const int lc = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const int pw = (a.Q + 15) >> 4, ph = (a.P + 15) >> 4;
    const int tx = tm % pw, t1 = tm / pw;
    const int ty = t1 % ph, b = t1 / ph;
    const int y0 = ty * 16, x0 = tx * 16;
    const int hmin_h = a.dh_step > 0 ? a.dh0 : a.dh0 + 2 * a.dh_step;
    const int hmin_w = a.dw_step > 0 ? a.dw0 : a.dw0 + 2 * a.dw_step;

    const unsigned ilmb = (unsigned)__builtin_amdgcn_readfirstlane(a.w_il ? 2 : 1);
    const bool ila = NPL == 2 && planes_il(a.x_plane_stride);
    const unsigned ilma = (unsigned)__builtin_amdgcn_readfirstlane(ila ? 2 : 1);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_planes), 0, (int)(a.w_plane_stride * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)(a.x_bytes * ilma), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.x_planes) + a.x_plane_stride * 2), 0, (int)(a.x_bytes * ilma - (ila ? 64 : 0)), 0x00020000);

    // filter row of this thread (one 16-row piece per wave: 64 rows)
    const int nrow = n0 + 16 * wave + (lane >> 2);
    const unsigned woff_row = nrow < a.N ? ((unsigned)nrow * (unsigned)a.w_row_stride * ilmb + 8u * lc) * 2u : OOB;
    const unsigned plane1_w = a.w_il ? 64u : (unsigned)(a.w_plane_stride * 2);
    // halo rows of this thread: pieces wave, wave + 4, ..., wave + 20
    unsigned hoff[HP];
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        const int h = 16 * (wave + 4 * j) + (lane >> 2);
        const int hy = h / PLH_HW, hx = h - hy * PLH_HW;
        const int yy = y0 + hmin_h + hy, xx = x0 + hmin_w + hx;
        const bool ok = (h < PLH_HW * PLH_HW) & ((unsigned)yy < (unsigned)a.IH) & ((unsigned)xx < (unsigned)a.IW);
        hoff[j] = ok ? ((unsigned)((b * a.IH + yy) * a.IW + xx) * (unsigned)a.x_pitch * ilma + 8u * lc) * 2u : OOB;
    }
    {
        const int py = tid >> 4, px = tid & 15;
        rowoff[tid] = (y0 + py < a.P && x0 + px < a.Q) ? (int)(((long long)(b * a.OH + y0 + py) * a.OW + x0 + px) * a.y_pitch) : -1;
    }

    const int nchunks = (a.Cin + BK - 1) / BK;
    const int S = 9 * nchunks;
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

    auto issue_halo = [&](int c) {
        char* const dst = lds + (16 * wave) * ROW;
        const unsigned oob = (unsigned)(a.Cin - 1 - (c * BK + 8 * lc)) & OOB;
        const unsigned cb = (unsigned)(c * BK * 2) * ilma;
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const unsigned vo = (hoff[j] + cb) | oob | (hoff[j] & OOB);
            char* const d = dst + (64 * j) * ROW;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_vptr)d, 16, vo, 0, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_vptr)(d + HPL), 16, vo, 0, 0, 0);
        }
    };
    auto issue_b = [&](int s, int slot) {
        const int c = s / 9, t = s - 9 * c;
        const int tr = t / 3, ts = t - 3 * tr;
        const unsigned oob = (unsigned)(a.Cin - 1 - (c * BK + 8 * lc)) & OOB;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((a.w_off0 + tr * a.w_step_r + ts * a.w_step_s + c * BK) * 2) * ilmb));
        const unsigned vo = woff_row | oob;
        const unsigned vo1 = (woff_row + plane1_w) | oob | (woff_row & OOB);
        char* const d = ldsB + slot * BST + (16 * wave) * ROW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)d, 16, vo, so, 0, 0);
        if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(d + BNN * ROW), 16, vo1, so, 0, 0);
    };
    const int l15 = lane & 15, kq = lane >> 4;
    const int koffB = 16 * (kq ^ (((lane >> 2) & 1) << 1));
    const char* const rb_base = ldsB + l15 * ROW + koffB;
    auto compute = [&](int t, int slot) {
        const int tr = t / 3, ts = t - 3 * tr;
        const int oh = a.dh0 + tr * a.dh_step - hmin_h, ow = a.dw0 + ts * a.dw_step - hmin_w;      // 0 .. 2
        const char* pb = rb_base + slot * BST;
        f16x8 fb[AT][NPL];
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BNN * ROW + j * 16 * ROW);
        const int hr0 = (wave * 4 + oh) * PLH_HW + ow + l15;
        f16x8 fa[AT][NPL];                          // (all rows' fragments named, so that pl_step_schedule can place their reads)
#pragma unroll
        for (int i = 0; i < AT; ++i) {
            const int hr = hr0 + i * PLH_HW;
            const char* pa = lds + hr * ROW + 16 * (kq ^ (((hr >> 2) & 1) << 1));
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[i][pl] = *reinterpret_cast<const f16x8*>(pa + pl * HPL);
        }
#pragma unroll
        for (int i = 0; i < AT; ++i) {
#pragma unroll
            for (int j = 0; j < AT; ++j) {
                if constexpr (NTERMS == 3) {
                    acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][NPL - 1], acc_lo[i][j], 0, 0, 0);
                    acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][NPL - 1], fa[i][0], acc_lo[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][0], acc[i][j], 0, 0, 0);
            }
        }
        pl_step_schedule<AT, NPL, NTERMS>();
    };

    issue_halo(0);
    issue_b(0, 0);
    issue_b(1, 1);
    int c = 0, t = 0, slot = 0, nslot = 2;
    for (int s = 0; s < S; ++s) {
        if (t == 0 && c > 0) {
            // chunk boundary: behind this barrier every wave has read the old halo for the last time; the new one is requested, awaited, published
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue_halo(c);
            if (s + 2 < S) {                                  // (a chunk has nine steps: true unless this is the last chunk's ... never at t == 0; kept for symmetry)
                issue_b(s + 2, nslot);
                // the halo is older than filter tile s+2: that one may stay in flight
                if constexpr (NPL == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        } else {
            // filter tile s (and at step 0 the first halo) has landed; tile s+1 (NB instructions, younger) may still be on its way
            if (s + 1 < S) {
                if constexpr (NPL == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < S) issue_b(s + 2, nslot);
        }
        __builtin_amdgcn_sched_barrier(0);
        compute(t, slot);
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == 2 ? 0 : slot + 1;
        nslot = nslot == 2 ? 0 : nslot + 1;
        if (++t == 9) { t = 0; ++c; }
    }
    static_assert(NB == NPL, "counted waits above: one filter DMA instruction per plane, wave and step");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // LDS is reused for the statistics; orders the row table
    pl_epilogue<NTERMS, BM, false, 4, EP, false>(a, acc, acc_lo, rowoff, reinterpret_cast<float*>(smem), tm, n0, wave, 0, lane, tid, y0 + 16 <= a.P && x0 + 16 <= a.Q, nullptr, BNN);
}

template __global__ void gg_plhn_kernel<3>(const GatherGemmArgs);
template __global__ void gg_plhn_kernel<1>(const GatherGemmArgs);
template __global__ void gg_plhn_kernel<3, true>(const GatherGemmArgs);      // fused inference epilogue (as gg_plh_kernel<.., EP>)
template __global__ void gg_plhn_kernel<1, true>(const GatherGemmArgs);

// geometry / size conditions on top of: A operand given as planes, prepared filter planes present

int g_stagger = -1;        // < 0: the launch heuristic; >= 0: forced start delay in 2048-cycle units (pylc_debug_stagger)
extern "C" int pylc_debug_stagger(int units) { g_stagger = units; return PYLC_OK; }

// geometry / size conditions on top of: A operand given as planes, prepared filter planes present
bool takes_pl(const GatherGemmArgs& a) {
    const bool aligned = a.y_pitch % 4 == 0 && a.N_store % 4 == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(a.x_planes) & 15) == 0 && (a.x_plane_stride % 8) == 0 && a.x_pitch % 8 == 0;
    const bool ila = a.nterms == 3 && planes_il(a.x_plane_stride);      // chunk-interleaved pixel planes: one descriptor spans both planes
    if (ila && (a.Cin % 32 != 0 || a.x_pitch != a.Cin || a.x_bytes >= (1ll << 30))) return false;
    return a.x_planes != nullptr && a.w_planes != nullptr && aligned && a.Cin % 8 == 0 && a.x_bytes > 0 && a.x_bytes < (1ll << 31) &&
           a.w_plane_stride * 4 < (1ll << 31);
}

template <int NTERMS, int BM>
static void launch_pl(const GatherGemmArgs& a, unsigned grid, hipStream_t st) {
    constexpr int lds_bytes = pl_lds_bytes<NTERMS, BM>();
    if (NTERMS == 3 && (a.dbg != nullptr || (a.dbg_flags & (64 | 128)))) {      // stamped / ablation build (tools/pl_stamps.py, pl_ablate.py)
        hipLaunchKernelGGL((gg_pl_kernel<3, BM, true>), dim3(grid), dim3(BM * 2), lds_bytes + 4096, st, a);
        return;
    }
    const bool ep = a.ep_scale != nullptr || a.ep_amax != nullptr || a.out_planes2 || a.ep_res != nullptr;      // fused inference epilogue
    if (ep && a.N_store <= 64 && !(g_pp_flags & 65536)) hipLaunchKernelGGL((gg_pl_kernel<NTERMS, BM, false, false, true, true>), dim3(grid), dim3(BM * 2), lds_bytes, st, a);
    else if (ep) hipLaunchKernelGGL((gg_pl_kernel<NTERMS, BM, false, false, false, true>), dim3(grid), dim3(BM * 2), lds_bytes, st, a);
#ifdef PYLC_EXPERIMENTAL
    else if (a.bn_y != nullptr) hipLaunchKernelGGL((gg_pl_kernel<NTERMS, BM, false, true>), dim3(grid), dim3(BM * 2), lds_bytes, st, a);
#endif
    else if (a.N_store <= 64 && !(g_pp_flags & 65536)) hipLaunchKernelGGL((gg_pl_kernel<NTERMS, BM, false, false, true>), dim3(grid), dim3(BM * 2), lds_bytes, st, a);
    else hipLaunchKernelGGL((gg_pl_kernel<NTERMS, BM>), dim3(grid), dim3(BM * 2), lds_bytes, st, a);
}

int launch_gg_pl(GatherGemmArgs& a, hipStream_t st) {
    PYLC_REQUIRE(takes_pl(a), "conv (fp16-plane operands): needs prepared filter planes, Cin %% 8 == 0, pitches %% 8 == 0, 16-byte aligned "
                              "buffers below 2 GiB per plane");
    if (a.dh_step > 2 || a.dh_step < -2 || a.dw_step > 2 || a.dw_step < -2) a.dbg_flags |= 4096;      // as launch_gg_pp
    a.ident = a.TR == 1 && a.TS == 1 && a.in_sh == 1 && a.in_sw == 1 && a.dh0 == 0 && a.dw0 == 0 && a.IH == a.P && a.IW == a.Q &&
              a.out_sh == 1 && a.out_sw == 1 && a.oh0 == 0 && a.ow0 == 0 && a.OH == a.P && a.OW == a.Q;
    PYLC_REQUIRE(!(a.out_half || a.out_planes2) ||
                     (a.out_bound != nullptr && a.amax_x && a.amax_w && !a.accumulate && a.add_src == nullptr && a.bn_y == nullptr &&
                      (a.bias == nullptr || a.ep_scale != nullptr) && a.y_pitch == a.N_store && a.N_store % 8 == 0),
                 "conv with an fp16-plane output: needs out_bound, the operand ranges and a dense y of a multiple of 8 channels; no accumulation, residual-gradient source or BatchNorm sums "
                 "(a bias only with the fused inference epilogue)");
    // (fetch_res pairs lanes l / l + 16 -- adjacent channel quads -- through permlane16_swap and reads 16 bytes = 8 channels per pair: with
    //  N_store % 8 == 4 the last quad would have no partner and its load would run past the row)
    PYLC_REQUIRE(a.ep_res == nullptr || a.ep_res_fmt == 0 || (a.ep_res_scale != nullptr && a.y_pitch == a.N_store && a.N_store % 8 == 0),
                 "conv with a fused fp16-plane residual: needs the residual's scale bound and a dense y of a multiple of 8 channels");
    PYLC_REQUIRE(!(a.ep_scale != nullptr || a.out_planes2 || a.ep_res != nullptr) || (a.stats == nullptr && !a.accumulate && a.add_src == nullptr && a.bn_y == nullptr),
                 "conv with the fused inference epilogue: no statistics, accumulation or backward fusions");
    PYLC_REQUIRE(a.add_src == nullptr || (a.y_pitch == a.N_store && a.N_store % 8 == 0 && !a.accumulate),
                 "conv dgrad with a masked residual source needs a dense output (pitch == channels, channels %% 8 == 0) and no accumulation");
    PYLC_REQUIRE(a.bn_y == nullptr || (a.y_pitch == a.N_store && a.stats != nullptr && a.bn_mean && a.bn_invstd && (!a.bn_relu || a.bn_mask || (a.bn_scale && a.bn_shift))),
                 "conv dgrad with BatchNorm-backward sums needs a dense output, a partials buffer, mean / invstd and a mask source");
#ifndef PYLC_EXPERIMENTAL
    PYLC_REQUIRE(a.bn_y == nullptr, "conv dgrad with BatchNorm-backward sums: this library was built without EXPERIMENTAL=1");
#endif
    // Tile height.  256 rows: 25 % fewer operand bytes per MFMA and a two-step DMA lead, but one block per CU (nothing hides a
    // tile's prologue / epilogue) -- for long reductions on grids that still fill the chip.  128 rows: two blocks per CU.
    // 3x3 / unit steps / 16-aligned output: the halo kernel (A-operand DMA once per channel chunk instead of once per tap)
    // (ragged sizes -- the U-Net's valid convs: 508, 250, 121 ... -- take it while the overhang of the last patches costs less than the
    // kernel gains: at most 1/8 more patch area than pixels)
    const long long halo_tiles_m = (long long)(a.M / (a.P * a.Q)) * cdiv(a.P, 16) * cdiv(a.Q, 16);
    const bool halo_geom = a.TR == 3 && a.TS == 3 && a.in_sh == 1 && a.in_sw == 1 && (a.dh_step == 1 || a.dh_step == -1) &&
                           a.dw_step == a.dh_step && halo_tiles_m * 256 * 8 <= (long long)a.M * 9 && !(g_pp_flags & 262144 && (a.P % 16 || a.Q % 16)) &&
                           a.out_sh == 1 && a.out_sw == 1 && a.oh0 == 0 &&
                           a.ow0 == 0 && a.OH == a.P && a.OW == a.Q && a.w_step_s * 3 == a.w_step_r;
    const bool narrow_first = a.N_store <= 64 && !(g_pp_flags & (65536 | 131072)) && a.bn_y == nullptr;      // <= 64 output channels: the 32 x 64-wave-tile form of gg_pl_kernel
    const bool ep_fused = a.ep_scale != nullptr || a.ep_amax != nullptr || a.out_planes2 || a.ep_res != nullptr;      // fused inference epilogue
    // <= 64 output channels: the four-wave halo kernel, two blocks per CU (pylc_debug_pp_flags bit 24: off)
    // ... and wider launches as 64-wide column tiles of it when that makes at least one full round of blocks (two per CU): since the K-step's
    // instruction order is given to the scheduler (pl_step_schedule) the one-round 256 -> 256 @32^2 x 32 launches are ahead on it too (89 vs 93 us;
    // in the step +0.7 %).  pylc_debug_pp_flags bit 27: never (gg_plh_kernel for every wide launch); bit 26: only from two rounds on (the rule before)
    const bool wide_too = a.bn_y == nullptr && !(g_pp_flags & 134217728) && halo_tiles_m * cdiv(a.N_store, 64) >= ((g_pp_flags & 67108864) ? 4 : 2) * kNumCU;
    if (halo_geom && (narrow_first || wide_too) && !(g_pp_flags & (16384 | 16777216)) && a.dbg == nullptr && !(a.dbg_flags & (64 | 128)) && halo_tiles_m * cdiv(a.N_store, 64) >= 2 * kNumCU) {
        a.tile_bm = 256;
        a.tiles_n = cdiv(a.N_store, 64);
        a.halo_tiles_m = (int)halo_tiles_m;
        a.n_tiles = (int)(halo_tiles_m * a.tiles_n);
        a.stagger = g_stagger >= 0 ? g_stagger : 0;          // (A/B knob pylc_debug_stagger, 2048-cycle units: 0 measured best -- profiles/r06_narrow_halo_ab.txt)
        a.stagger_blocks = 2 * kNumCU;
        if (ep_fused) {
            if (a.nterms == 1) hipLaunchKernelGGL((gg_plhn_kernel<1, true>), dim3((unsigned)a.n_tiles), dim3(256), plhn_lds_bytes<1>(), st, a);
            else hipLaunchKernelGGL((gg_plhn_kernel<3, true>), dim3((unsigned)a.n_tiles), dim3(256), plhn_lds_bytes<3>(), st, a);
        } else if (a.nterms == 1) hipLaunchKernelGGL((gg_plhn_kernel<1>), dim3((unsigned)a.n_tiles), dim3(256), plhn_lds_bytes<1>(), st, a);
        else hipLaunchKernelGGL((gg_plhn_kernel<3>), dim3((unsigned)a.n_tiles), dim3(256), plhn_lds_bytes<3>(), st, a);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    if (halo_geom && !narrow_first && !(g_pp_flags & 16384) && halo_tiles_m * cdiv(a.N_store, PL_BN) >= kNumCU / 2) {
        a.tile_bm = 256;
        a.tiles_n = cdiv(a.N_store, PL_BN);
        a.halo_tiles_m = (int)halo_tiles_m;
        const long long n_tiles = halo_tiles_m * a.tiles_n;
        a.n_tiles = (int)n_tiles;
        if (a.ep_scale != nullptr || a.ep_amax != nullptr || a.out_planes2 || a.ep_res != nullptr) {
            if (a.nterms == 1) hipLaunchKernelGGL((gg_plh_kernel<1, false, true>), dim3((unsigned)n_tiles), dim3(512), plh_lds_bytes<1>(), st, a);
            else hipLaunchKernelGGL((gg_plh_kernel<3, false, true>), dim3((unsigned)n_tiles), dim3(512), plh_lds_bytes<3>(), st, a);
#ifdef PYLC_EXPERIMENTAL
        } else if (a.bn_y != nullptr) {
            if (a.nterms == 1) hipLaunchKernelGGL((gg_plh_kernel<1, true>), dim3((unsigned)n_tiles), dim3(512), plh_lds_bytes<1>(), st, a);
            else hipLaunchKernelGGL((gg_plh_kernel<3, true>), dim3((unsigned)n_tiles), dim3(512), plh_lds_bytes<3>(), st, a);
#endif
        } else if (a.nterms == 1) hipLaunchKernelGGL((gg_plh_kernel<1>), dim3((unsigned)n_tiles), dim3(512), plh_lds_bytes<1>(), st, a);
        else hipLaunchKernelGGL((gg_plh_kernel<3>), dim3((unsigned)n_tiles), dim3(512), plh_lds_bytes<3>(), st, a);
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    const long long ksteps = (long long)a.TR * a.TS * cdiv(a.Cin, BK);
    const long long tiles256 = (long long)cdiv(a.M, 256) * cdiv(a.N_store, PL_BN);
    bool big = ksteps >= 24 && tiles256 >= kNumCU;
    if (g_pp_flags & 2048) big = false;               // A/B knobs (pylc_debug_pp_flags)
    // multi-tap launches (dilated / strided 3x3, ragged 3x3 the halo kernels do not take): two 128-row blocks per CU, each the other's cover, are ahead of the
    // one 256-row block whatever the reduction length (R101 step +0.5-0.9 %: profiles/r06_tile128_multitap_ab.txt; bit 23: the old choice)
    if (a.TR * a.TS > 1 && !(g_pp_flags & 8388608)) big = false;
    if (g_pp_flags & 8192) big = true;
    const int bm = big ? 256 : 128;
    a.tile_bm = bm;
    a.tiles_n = cdiv(a.N_store, PL_BN);
    const long long n_tiles = (long long)cdiv(a.M, bm) * a.tiles_n;
    PYLC_REQUIRE(n_tiles > 0 && n_tiles < (1ll << 31), "conv grid out of range");
    a.n_tiles = (int)n_tiles;
    a.stagger = 0;
    a.stagger_blocks = 2 * kNumCU;
    if (!big && n_tiles >= 3 * kNumCU) {             // at least two rounds of blocks: the one-off delay pays back
        // half a tile of this launch: ~1200 cycles per K-step of a block that shares its SIMDs + half the epilogue / prologue
        const long long cycles = (ksteps * 1200 + 8000) / 2;
        a.stagger = (int)(cycles / 2048);
        if (g_stagger >= 0) a.stagger = g_stagger;   // A/B knob (pylc_debug_stagger)
    }
    // 1x1 convs on 128-row tiles with more tiles than the chip holds blocks: the persistent form (gg_plp_kernel) -- plain launches only
    // (no narrow / inference / BatchNorm-backward / stamped variant), no padding (the one tap of every valid row lies inside the input),
    // at least two K-steps per tile.  pylc_debug_pp_flags bit 19 (524288): the per-tile kernel (A/B, bit-identity reference)
    const bool ep_any = a.ep_scale != nullptr || a.ep_amax != nullptr || a.out_planes2 || a.ep_res != nullptr;
    const int ks_tile = a.nterms == 1 ? 64 : 32;
    const float* extra_src = a.add_src != nullptr ? a.add_src : (a.accumulate ? a.y : nullptr);
    if (!big && !(g_pp_flags & 524288) && a.ident && n_tiles > 2 * kNumCU && !ep_any && a.bn_y == nullptr && (a.dbg == nullptr || a.nterms == 3) &&
        !(a.dbg_flags & (8 | 64 | 128)) && cdiv(a.Cin, ks_tile) >= 2 &&
        // every tile is a lean tile: no edge, fp32 output, no bias, statistics XOR a residual source
        a.M % 128 == 0 && a.N_store % PL_BN == 0 && a.bias == nullptr && !a.out_half && !(a.stats != nullptr && extra_src != nullptr) && PYLC_EPI_FULL_LINES != 0 &&
        (long long)a.M * a.y_pitch < (1ll << 31) && (a.add_mask == nullptr || (a.y_pitch == a.N_store && (reinterpret_cast<uintptr_t>(a.add_mask) & 7) == 0))) {
        const unsigned grid = (g_pp_flags & 1048576) ? (unsigned)n_tiles : ((g_pp_flags & 4194304) ? 3 * kNumCU : 2 * kNumCU);      // (bit 20: one tile per block -- the loop and epilogue code alone; bit 22: 768 blocks)
        const int epk = extra_src != nullptr ? (a.add_mask != nullptr ? 3 : 2) : (a.stats != nullptr ? 1 : 0);
#define PYLC_PLP(NT, E) hipLaunchKernelGGL((gg_plp_kernel<NT, E>), dim3(grid), dim3(256), plp_lds_bytes<NT>(), st, a)
        if (a.dbg != nullptr && epk <= 1) {          // stamped build (tools/plp_stamps.py)
            if (epk == 0) hipLaunchKernelGGL((gg_plp_kernel<3, 0, true>), dim3(grid), dim3(256), plp_lds_bytes<3>() + 4096, st, a);
            else hipLaunchKernelGGL((gg_plp_kernel<3, 1, true>), dim3(grid), dim3(256), plp_lds_bytes<3>() + 4096, st, a);
            PYLC_LAUNCH_CHECK();
            return PYLC_OK;
        }
        if (a.nterms == 1) { if (epk == 0) PYLC_PLP(1, 0); else if (epk == 1) PYLC_PLP(1, 1); else if (epk == 2) PYLC_PLP(1, 2); else PYLC_PLP(1, 3); }
        else { if (epk == 0) PYLC_PLP(3, 0); else if (epk == 1) PYLC_PLP(3, 1); else if (epk == 2) PYLC_PLP(3, 2); else PYLC_PLP(3, 3); }
#undef PYLC_PLP
        PYLC_LAUNCH_CHECK();
        return PYLC_OK;
    }
    if (a.nterms == 1) {
        if (big) launch_pl<1, 256>(a, (unsigned)n_tiles, st); else launch_pl<1, 128>(a, (unsigned)n_tiles, st);
    } else {
        if (big) launch_pl<3, 256>(a, (unsigned)n_tiles, st); else launch_pl<3, 128>(a, (unsigned)n_tiles, st);
    }
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

template <typename K>
static hipError_t opt_in(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int conv_pl_init() {
    PYLC_HIP(opt_in(gg_pl_kernel<3, 128>, pl_lds_bytes<3, 128>()));
    PYLC_HIP(opt_in(gg_pl_kernel<3, 256>, pl_lds_bytes<3, 256>()));
    PYLC_HIP(opt_in(gg_pl_kernel<1, 128>, pl_lds_bytes<1, 128>()));
    PYLC_HIP(opt_in(gg_pl_kernel<1, 256>, pl_lds_bytes<1, 256>()));
    PYLC_HIP(opt_in(gg_plh_kernel<3>, plh_lds_bytes<3>()));
    PYLC_HIP(opt_in(gg_plh_kernel<1>, plh_lds_bytes<1>()));
    PYLC_HIP(opt_in(gg_plhn_kernel<3>, plhn_lds_bytes<3>()));
    PYLC_HIP(opt_in(gg_plhn_kernel<1>, plhn_lds_bytes<1>()));
    PYLC_HIP(opt_in((gg_plhn_kernel<3, true>), plhn_lds_bytes<3>()));
    PYLC_HIP(opt_in((gg_plhn_kernel<1, true>), plhn_lds_bytes<1>()));
    PYLC_HIP(opt_in((gg_plp_kernel<3, 0>), plp_lds_bytes<3>()));
    PYLC_HIP(opt_in((gg_plp_kernel<3, 0, true>), plp_lds_bytes<3>() + 4096));
    PYLC_HIP(opt_in((gg_plp_kernel<3, 1, true>), plp_lds_bytes<3>() + 4096));
    PYLC_HIP(opt_in((gg_plp_kernel<3, 1>), plp_lds_bytes<3>()));
    PYLC_HIP(opt_in((gg_plp_kernel<3, 2>), plp_lds_bytes<3>()));
    PYLC_HIP(opt_in((gg_plp_kernel<3, 3>), plp_lds_bytes<3>()));
    PYLC_HIP(opt_in((gg_plp_kernel<1, 0>), plp_lds_bytes<1>()));
    PYLC_HIP(opt_in((gg_plp_kernel<1, 1>), plp_lds_bytes<1>()));
    PYLC_HIP(opt_in((gg_plp_kernel<1, 2>), plp_lds_bytes<1>()));
    PYLC_HIP(opt_in((gg_plp_kernel<1, 3>), plp_lds_bytes<1>()));
#ifdef PYLC_EXPERIMENTAL
    PYLC_HIP(opt_in((gg_pl_kernel<3, 128, false, true>), pl_lds_bytes<3, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<3, 256, false, true>), pl_lds_bytes<3, 256>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 128, false, true>), pl_lds_bytes<1, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 256, false, true>), pl_lds_bytes<1, 256>()));
#endif
    PYLC_HIP(opt_in((gg_pl_kernel<3, 128, false, false, true>), pl_lds_bytes<3, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<3, 256, false, false, true>), pl_lds_bytes<3, 256>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 128, false, false, true>), pl_lds_bytes<1, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 256, false, false, true>), pl_lds_bytes<1, 256>()));
#ifdef PYLC_EXPERIMENTAL
    PYLC_HIP(opt_in((gg_plh_kernel<3, true>), plh_lds_bytes<3>()));
    PYLC_HIP(opt_in((gg_plh_kernel<1, true>), plh_lds_bytes<1>()));
#endif
    PYLC_HIP(opt_in((gg_pl_kernel<3, 128, false, false, false, true>), pl_lds_bytes<3, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<3, 256, false, false, false, true>), pl_lds_bytes<3, 256>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 128, false, false, false, true>), pl_lds_bytes<1, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 256, false, false, false, true>), pl_lds_bytes<1, 256>()));
    PYLC_HIP(opt_in((gg_pl_kernel<3, 128, false, false, true, true>), pl_lds_bytes<3, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<3, 256, false, false, true, true>), pl_lds_bytes<3, 256>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 128, false, false, true, true>), pl_lds_bytes<1, 128>()));
    PYLC_HIP(opt_in((gg_pl_kernel<1, 256, false, false, true, true>), pl_lds_bytes<1, 256>()));
    PYLC_HIP(opt_in((gg_plh_kernel<3, false, true>), plh_lds_bytes<3>()));
    PYLC_HIP(opt_in((gg_plh_kernel<1, false, true>), plh_lds_bytes<1>()));
    PYLC_HIP(opt_in(gg_pl_kernel<3, 128, true>, pl_lds_bytes<3, 128>() + 4096));
    PYLC_HIP(opt_in(gg_pl_kernel<3, 256, true>, pl_lds_bytes<3, 256>() + 4096));

    return PYLC_OK;
}

}  // namespace pylc
