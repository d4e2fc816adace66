// wgrad with BOTH operands given as fp16 planes:  dW[n][tap][c] = sum_m dY[m][n] * X[gather(m, tap)][c].
//
// Same tiling, LDS image (pixel-major planes, fragments by the hardware transpose read ds_read_b64_tr_b16), MFMA order and split-K
// slabs as conv_igemm.hip's wgrad_split_kernel -- so the result is bit-identical to it -- but the operand tiles are 16-byte copies
// global -> register -> LDS: no vector-ALU split of fp32 values (the 128 x 128 configuration spent 144 vector instructions per 24 MFMAs
// on it), half the load instructions (8 halves per lane and plane instead of 4 floats).  dY comes from pylc_bn_bwd_apply_ex, X from
// pylc_bn_apply_ex / pylc_to_planes.  Register staging rather than LDS-DMA: the rows are padded (conflict-free transpose reads), and
// an LDS-DMA instruction writes 1 KB contiguously.
//
// Geometry paths as wgrad_split_kernel: FAST 1 (OW % 32 == 0: every 32-pixel reduction tile lies in one output row, wave-uniform
// gather state) and FAST 2 (any OW >= 16: per-row pixel coordinates as counters).  NTERMS 3 = f16x3, 1 = plane 0 only (mode 3).
//
// Measured and dropped (round 2): the 128 x 128 tile as ONE 8-wave block per CU with 32 x 64 outputs per wave (64 accumulator
// registers, 120 registers per wave, LDS reservation so that a second block cannot join): half of every SIMD's register file stays
// free and the BatchNorm kernels of the main stream do run next to it (bn_bwd_apply 17.5 -> 12.6 ms per step), but the wgrad itself
// takes 33 instead of 20 ms on the side queue and overlaps -- and slows -- the matrix-bound dgrads (3x3 dgrad 307 -> 219 TFLOP/s):
// 360 vs 371 tiles/s.
#include "conv_common.h"

namespace pylc {

// ACC1 (NTERMS 3 only): ONE accumulator set instead of two -- the plane-1 fragments are multiplied by 2^-11 in registers (8 v_pk_mul_f16
// per K-step next to 24 MFMAs; gfx950's MFMA keeps fp16 subnormals) and all three terms add into the same fp32 accumulator.  Same
// terms, fp32-grade result, NOT bit-identical to the two-accumulator form (the small terms are rounded into the large sum as they
// arrive).  What it buys is registers: 64 instead of 128 accumulators put the kernel under 128 VGPRs, so that with two blocks per CU
// (an LDS reservation keeps a third and fourth out) half of every SIMD's register file stays free for the BatchNorm kernels of the
// main stream -- with the efficient 64 x 64 wave tiles, unlike the 8-wave experiment above.
// M16 (round 5; 128 x 128 block tiles, 64 x 64 wave tiles): the products are v_mfma_f32_16x16x32_f16 -- one instruction per term and 32-pixel
// K-step for each of the wave's 4 x 4 tiles of 16 x 16 -- instead of v_mfma_f32_32x32x16_f16: same FLOPs per matrix-pipe cycle, but the
// 16 x 16 x 32 form draws less power (bare loops on random data sustain 2005 vs 1690 TFLOP/s under the board's cap,
// tools/micro/mfma_shapes.hip; conv_pl.hip made the same move in round 1).  The LDS image is then unpadded 256-byte pixel rows with the
// 16-byte chunk index XORed by ((row & 3) << 2) | ((row >> 2) & 3): conflict-free for the ds_write_b128 stores and for the transposed
// reads of a 16 x 16 x 32 operand (a 32-lane half reads two 4-pixel blocks 8 pixels apart in the same 16 channels), 32 KB instead of 40.
// Same products, summed 32 instead of 16 pixels per instruction: fp32-rounding-level differences from the 32 x 32 x 16 form.
//
// DMA (M16 only): the operand tiles go global -> LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane, 1 KB = four 256-byte pixel rows per
// wave instruction) into TWO LDS stages of 32 KB -- the swizzle is applied on the GLOBAL side (lane = physical chunk, it fetches logical chunk
// lane ^ swz(row)), which the padded rows of the 32x32x16 image did not allow.  No staging registers, no ds_write_b128 (the LDS's slowest
// instruction: ~13 cycles per wave, 8 per thread and K-step), one barrier per K-step; tile s + 1 is in flight while tile s is multiplied.
typedef __attribute__((address_space(3))) void* wg_lds_vptr;
template <int BN, int BC, int WN, int WC, int NTERMS, int FAST, bool ACC1 = false, int SETS = 1, bool M16 = false, bool DMA = false>
__global__ __launch_bounds__(256, ACC1 ? 4 : 2) void wgrad_pl_kernel(const WgradArgs a) {
    static_assert(!DMA || (M16 && SETS == 1), "DMA: the 16x16x32 image only");
    constexpr int NPL = NTERMS == 3 ? 2 : 1;
    constexpr int WAVES_C = BC / WC;
    constexpr int NT = WN / 32, CT = WC / 32;
    static_assert(!M16 || (BN == 128 && BC == 128 && WN == 64 && WC == 64 && !ACC1), "M16: 128 x 128 block tiles, 64 x 64 wave tiles, two accumulator sets");
    constexpr int VA = BN / 8, RA = 256 / VA, IA = 32 / RA > 0 ? 32 / RA : 1;      // dy tile: 8 halves per lane, RA rows per pass
    constexpr int VB = BC / 8, RB = 256 / VB, IB = 32 / RB > 0 ? 32 / RB : 1;
    constexpr bool A_ALL = RA <= 32, B_ALL = RB <= 32;                             // else only threads with row < 32 take part (BN = 32)
    constexpr int ROWA = M16 ? 256 : wg_rowb(BN), ROWB = M16 ? 256 : wg_rowb(BC);
    constexpr int PLA = 32 * ROWA, PLB = 32 * ROWB;          // bytes per plane
    static_assert((BN / WN) * (BC / WC) == 4, "4 waves per block");
    static_assert(FAST == 1 || FAST == 2, "buffer-load geometry paths only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sA = reinterpret_cast<char*>(smem);               // NPL planes of dy
    char* sB = sA + NPL * PLA;                               // NPL planes of gathered x
    constexpr int STG = NPL * (PLA + PLB);                   // DMA: bytes of one LDS stage (two of them)

    const int T = a.TR * a.TS;
    // Rasterisation: taps fastest, then (channel tile, cout tile), split slowest, every XCD a contiguous range.  The blocks resident on an
    // XCD at one time are then the nine taps of a few (channel tile, cout tile) pairs of the SAME pixel chunk: a chunk of x / dy is
    // fetched from HBM once and served to its other readers by that XCD's 4 MB L2.  Measured per launch (tools/wgrad_traffic.py,
    // profiles/r04_wgrad_traffic.txt; L2-miss bytes / operand bytes): 256 -> 256 3x3 @128^2 x 32: 4.02 x with taps slowest and 1170-step
    // blocks, 1.86 x with taps fastest and blocks capped at 256 K-steps (plan_wgrad); 512 -> 512 @32^2: 4.97 -> 3.13; 256 -> 256 @32^2: 1.66 -> 1.36.
    // (debug flags, pylc_debug_wgrad_flags: 1 = plain blockIdx order, 2 = split index fastest -- the sharers far apart --, 4 = the
    // round-3 order, taps slowest)
    int id = (a.dbg_flags & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int tc, tn, tap, split;
    if (a.dbg_flags & 2) {
        split = id % a.splits; id /= a.splits;
        tc = id % a.tiles_c; id /= a.tiles_c;
        tn = id % a.tiles_n; id /= a.tiles_n;
        tap = id;
    } else if (a.dbg_flags & 4) {
        tc = id % a.tiles_c; id /= a.tiles_c;
        tn = id % a.tiles_n; id /= a.tiles_n;
        tap = id % T; id /= T;
        split = id;
    } else if (a.Cin > a.N && !(a.dbg_flags & 8)) {
        // more input than output channels: x is the larger operand, so the blocks that share an x slice (same channel tile, all cout
        // tiles, all taps) are the adjacent ones (debug flag 8: channel tiles inner regardless)
        tap = id % T; id /= T;
        tn = id % a.tiles_n; id /= a.tiles_n;
        tc = id % a.tiles_c; id /= a.tiles_c;
        split = id;
    } else {
        tap = id % T; id /= T;
        tc = id % a.tiles_c; id /= a.tiles_c;
        tn = id % a.tiles_n; id /= a.tiles_n;
        split = id;
    }
    const int n0 = tn * BN, c0 = tc * BC;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const int S = (m_end - m_begin + 31) / 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave / WAVES_C, wave_c = wave % WAVES_C;
    const int va = tid % VA, pra = tid / VA;
    const int vb = tid % VB, prb = tid / VB;
    // DMA: a lane IS a physical 16-byte chunk of its LDS row, so it fetches the logical chunk that the swizzle puts there
    const int sw_row = ((pra & 3) << 2) | ((pra >> 2) & 3);      // swz(row): rows pra + 16 i share it (M16: VA == VB == 16)
    const int va_g = DMA ? (va ^ sw_row) : va, vb_g = DMA ? (vb ^ sw_row) : vb;
    const int dh = a.dh0 + (tap / a.TS) * a.dh_step, dw = a.dw0 + (tap % a.TS) * a.dw_step;
    const bool b_col_ok = c0 + 8 * vb_g < a.Cin && (B_ALL || prb < 32);          // Cin % 8 == 0
    const bool a_col_ok = n0 + 8 * va_g < a.N_ld && (A_ALL || pra < 32);         // N_ld % 8 == 0 (launch_wg_pl)

    f32x16 acc[M16 ? 1 : NT][M16 ? 1 : CT];
    constexpr bool TWO_ACC = NTERMS == 3 && !ACC1;
    f32x16 acc_lo[TWO_ACC && !M16 ? NT : 1][TWO_ACC && !M16 ? CT : 1];
    f32x4 acc16[M16 ? 4 : 1][M16 ? 4 : 1], acc16_lo[M16 && TWO_ACC ? 4 : 1][M16 && TWO_ACC ? 4 : 1];      // M16: 4 x 4 tiles of 16 x 16
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (TWO_ACC) acc16_lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    } else {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (TWO_ACC) acc_lo[i][j][r] = 0.f;
            }
    }
    const float scale_a = pow2_scale_for(*a.amax_dy), scale_b = pow2_scale_for(*a.amax_x);

    constexpr unsigned OOB = 0xFFFFFFF0u;                 // >= num_records: the load returns zeros
    // chunk-interleaved operands (plane_stride == 32, common.h): pixel offsets double, a column c sits at halves (c >> 5) * 64 + (c & 31) of its
    // row, plane 1 is 64 bytes behind plane 0 and one descriptor spans both planes
    const bool il_dy = NPL == 2 && planes_il(a.dy_plane_stride), il_x = NPL == 2 && planes_il(a.x_plane_stride);
    const unsigned mdy = (unsigned)__builtin_amdgcn_readfirstlane(il_dy ? 2 : 1), mx = (unsigned)__builtin_amdgcn_readfirstlane(il_x ? 2 : 1);
    auto colh = [](unsigned c, bool il) { return il ? ((c >> 5) << 6) + (c & 31u) : c; };      // column -> halves within the (doubled) row
    const __amdgpu_buffer_rsrc_t rdy0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy_planes), 0, (int)((unsigned)a.dy_bytes * mdy), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.dy_planes) + a.dy_plane_stride * 2), 0, (int)((unsigned)a.dy_bytes * mdy - (il_dy ? 64u : 0u)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)((unsigned)a.x_bytes * mx), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.x_planes) + a.x_plane_stride * 2), 0, (int)((unsigned)a.x_bytes * mx - (il_x ? 64u : 0u)), 0x00020000);

    // Operand staging registers, TWO sets (round 5): the loads of tile s + 2 are in flight while tile s is computed.  With one set the loads
    // of tile s + 1 had one K-step (~3 k cycles with the CU's other block) to come back from L2 / HBM and the step waited for them: PMC
    // showed the matrix pipe busy 0.43 of the kernel's cycles with two 1536-cycle MFMA phases per SIMD and step.
    constexpr int NSET = SETS;
    uint4 ra[NSET][IA][NPL], rb[NSET][IB][NPL];      // (DMA: unused, optimised away)
    int f_mb = m_begin, f_q0 = 0, f_p = 0, f_b = 0;
    unsigned f_va[IA], f_tx[IB];
    int f_wc[IB];
    int g_q[FAST == 2 ? IB : 1], g_p[FAST == 2 ? IB : 1], g_b[FAST == 2 ? IB : 1];      // FAST 2: per-row pixel coordinates
    if constexpr (FAST == 2) {
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int m = m_begin + prb + RB * i;
            g_q[i] = m % a.Q;
            const int t = m / a.Q;
            g_p[i] = t % a.P;
            g_b[i] = t / a.P;
        }
    }
    {
        f_q0 = m_begin % a.Q;
        const int t = m_begin / a.Q;
        f_p = t % a.P;
        f_b = t / a.P;
#pragma unroll
        for (int i = 0; i < IA; ++i) f_va[i] = a_col_ok ? ((unsigned)(pra + RA * i) * (unsigned)a.dy_pitch * mdy + colh((unsigned)(n0 + 8 * va_g), il_dy)) * 2u : OOB;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            f_tx[i] = ((unsigned)((prb + RB * i) * a.in_sw) * (unsigned)a.x_pitch * mx + colh((unsigned)(c0 + 8 * vb_g), il_x)) * 2u;
            f_wc[i] = (prb + RB * i) * a.in_sw + dw;
        }
    }
    // DMA: `set` is the LDS stage; a wave's instruction i covers pixel rows 4 wave + 16 i .. + 3 (lane l: row + (l >> 4), physical chunk l & 15)
    char* dma_a = nullptr; char* dma_b = nullptr;
    if constexpr (DMA) { dma_a = sA + (4 * wave) * 256; dma_b = sB + (4 * wave) * 256; }
    auto ldp = [&](const __amdgpu_buffer_rsrc_t& r0, const __amdgpu_buffer_rsrc_t& r1, unsigned voff, unsigned soff, uint4 (&dst)[NPL], char* lds0, int plb) {
        if constexpr (DMA) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (wg_lds_vptr)lds0, 16, voff, soff, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (wg_lds_vptr)(lds0 + plb), 16, voff, soff, 0, 0);
        } else {
            dst[0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r0, voff, soff, 0));
            if constexpr (NPL == 2) dst[1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, voff, soff, 0));
        }
    };
    auto load_tile = [&](auto setc) {
        constexpr int set = DMA ? 0 : decltype(setc)::value;
        constexpr int stg = DMA ? decltype(setc)::value * STG : 0;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)f_mb * (unsigned)a.dy_pitch * 2u * mdy));
        if constexpr (FAST == 2) {
            const unsigned colb = colh((unsigned)(c0 + 8 * vb_g), il_x);
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int hi = g_p[i] * a.in_sh + dh, wi = g_q[i] * a.in_sw + dw;
                const bool ok = b_col_ok & (f_mb + prb + RB * i < m_end) & ((unsigned)hi < (unsigned)a.IH) & ((unsigned)wi < (unsigned)a.IW);
                const unsigned off = (((unsigned)(g_b[i] * a.IH + hi) * (unsigned)a.IW + (unsigned)wi) * (unsigned)a.x_pitch * mx + colb) * 2u;
                ldp(rx0, rx1, ok ? off : OOB, 0u, rb[set][i], dma_b + stg + RB * i * 256, PLB);
                g_q[i] += 32;
                while (g_q[i] >= a.Q) { g_q[i] -= a.Q; ++g_p[i]; }
                while (g_p[i] >= a.P) { g_p[i] -= a.P; ++g_b[i]; }
            }
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const bool ok = f_mb + pra + RA * i < m_end;
                ldp(rdy0, rdy1, ok ? f_va[i] : OOB, soff, ra[set][i], dma_a + stg + RA * i * 256, PLA);
            }
            f_mb += 32;
            return;
        }
        // FAST 1: tile = 32 consecutive output pixels of row (f_b, f_p) starting at column f_q0
        const int hi = f_p * a.in_sh + dh;
        const bool row_ok = b_col_ok && (unsigned)hi < (unsigned)a.IH;
        const unsigned delta = (unsigned)((((f_b * a.IH + hi) * a.IW + f_q0 * a.in_sw + dw) * a.x_pitch) * 2) * mx;
        const int wq = f_q0 * a.in_sw;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const bool ok = row_ok & ((unsigned)(f_wc[i] + wq) < (unsigned)a.IW);
            ldp(rx0, rx1, ok ? f_tx[i] + delta : OOB, 0u, rb[set][i], dma_b + stg + RB * i * 256, PLB);
        }
#pragma unroll
        for (int i = 0; i < IA; ++i) ldp(rdy0, rdy1, f_va[i], soff, ra[set][i], dma_a + stg + RA * i * 256, PLA);
        f_mb += 32;
        f_q0 += 32;
        if (f_q0 == a.Q) {
            f_q0 = 0;
            if (++f_p == a.P) { f_p = 0; ++f_b; }
        }
    };
    auto store_tile = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
        if constexpr (M16) {
            // swizzled 256-byte rows: chunk va of row r at 16 * (va ^ swz(r)); r = pra + 16 i leaves swz unchanged (RA == RB == 16)
            const int sa = ((pra & 3) << 2) | ((pra >> 2) & 3);
#pragma unroll
            for (int i = 0; i < IA; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(sA + pl * PLA + (pra + RA * i) * 256 + 16 * (va ^ sa)) = ra[set][i][pl];
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(sB + pl * PLB + (prb + RB * i) * 256 + 16 * (vb ^ sa)) = rb[set][i][pl];
            return;
        }
        if (A_ALL || pra < 32) {
#pragma unroll
            for (int i = 0; i < IA; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(sA + pl * PLA + (pra + RA * i) * ROWA + 16 * va) = ra[set][i][pl];
        }
        if (B_ALL || prb < 32) {
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(sB + pl * PLB + (prb + RB * i) * ROWB + 16 * vb) = rb[set][i][pl];
        }
    };
    // transpose-read addressing (as wgrad_split_kernel): group g = lane>>4 covers channels 16*(g&1).., reduction half h = g>>1
    const int g = lane >> 4, h = g >> 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const char* fa_base = sA + (8 * h + q4) * ROWA + 2 * (wave_n * WN + 16 * (g & 1) + 4 * p4);
    const char* fb_base = sB + (8 * h + q4) * ROWB + 2 * (wave_c * WC + 16 * (g & 1) + 4 * p4);

    // M16 transposed reads: lane group g = the operand's k-group (pixels 8g .. 8g + 7: rows 8g + q by the first read, 8g + 4 + q by the
    // second), lane 4q + p of the group supplies row q, channels 4p .. 4p + 3 of the 16-channel tile: chunk (tile base / 8) + (p >> 1),
    // byte 8 (p & 1) within it.  swz(8g + q) = (q << 2) | (2g & 3), swz(8g + 4 + q) = that | 1: the second read sits 1024 bytes on with
    // chunk bit 0 flipped; tile t of the wave XORs 32 t into the address (chunk bits 1-2; the wave's base occupies bit 3 only).
    const int swz_lo = (q4 << 2) | ((2 * g) & 3);
    const int m16_a0 = (8 * g + q4) * 256 + 16 * ((wave_n * 8 + (p4 >> 1)) ^ swz_lo) + 8 * (p4 & 1);
    const int m16_b0 = (8 * g + q4) * 256 + 16 * ((wave_c * 8 + (p4 >> 1)) ^ swz_lo) + 8 * (p4 & 1);
    auto tr16 = [&](const char* base, int off0, int t) {
        const int o = off0 ^ (32 * t);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + o));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + ((o + 1024) ^ 16)));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto compute_tile16 = [&](auto stgc) {
        constexpr int so = decltype(stgc)::value * STG;      // DMA: the LDS stage; 0 otherwise
        // the dy fragments of the wave's four 16-row tiles stay; the x fragments come one 16-channel tile at a time (40 fragment registers
        // live instead of 64); within a tile the terms run term-major, four independent products between two that share an accumulator
        f16x8 fa[4][NPL];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[t][pl] = tr16(sA + so + pl * PLA, m16_a0, t);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f16x8 fb[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[pl] = tr16(sB + so + pl * PLB, m16_b0, j);
            if constexpr (NTERMS == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][NPL - 1], fb[0], acc16_lo[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[NPL - 1], acc16_lo[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[0], acc16[i][j], 0, 0, 0);
        }
    };
    // DMA form of a K-step: ALL fragment reads of tile s first, then the DMA of tile s + 1, then the 48 products.  hipcc orders an LDS read
    // behind every LDS-DMA issued before it in program order (s_waitcnt vmcnt(0): it cannot tell the stages apart, seen in the listing), so
    // the DMA goes after the step's last LDS read; the next read is behind the next barrier, where the tile is waited for anyway.
    [[maybe_unused]] auto step_dma = [&](auto stgc, auto issue) {      // (the round-5 order, kept for reference: reads, DMA, products)
        constexpr int so = decltype(stgc)::value * STG;
        f16x8 fa[4][NPL], fb[4][NPL];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                fa[t][pl] = tr16(sA + so + pl * PLA, m16_a0, t);
                fb[t][pl] = tr16(sB + so + pl * PLB, m16_b0, t);
            }
        __builtin_amdgcn_sched_barrier(0);
        issue();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (NTERMS == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][NPL - 1], fb[j][0], acc16_lo[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[j][NPL - 1], acc16_lo[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[j][0], acc16[i][j], 0, 0, 0);
        }
    };
    // The same K-step with the DMA of tile s + 1 issued FIRST (round 6).  What kept it behind the reads was hipcc, not the hardware: the
    // transposed reads are inline asm here (ds_read_b64_tr_b16), which the wait-count pass does not see, so nothing puts a vmcnt(0) between the
    // DMA and them; their arrival is waited for by hand (lgkmcnt(0)) and the fragments are tied to that wait.  The tile then has the step's
    // reads AND its 48 products to land in.  Same reads, same products, same order: bit-identical.
    // addresses: tile t of an operand at base + (off0 ^ 32 t), its second half 1024 bytes on with chunk bit 0 flipped = base + ((off0 ^ 32 t) ^ 16)
    // + 1024; plane, stage and the 1024 ride in the instruction's offset field, so 16 address registers serve all 64 reads of both stages
    unsigned ad_a[4][2], ad_b[4][2];
    {
        const unsigned la0 = (unsigned)(uintptr_t)(wg_lds_vptr)sA, lb0 = (unsigned)(uintptr_t)(wg_lds_vptr)sB;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ad_a[t][0] = la0 + (unsigned)(m16_a0 ^ (32 * t)); ad_a[t][1] = la0 + (unsigned)((m16_a0 ^ (32 * t)) ^ 16);
            ad_b[t][0] = lb0 + (unsigned)(m16_b0 ^ (32 * t)); ad_b[t][1] = lb0 + (unsigned)((m16_b0 ^ (32 * t)) ^ 16);
        }
    }
    auto tr16_asm = [&](const unsigned (&ad)[2], auto offc) {
        constexpr int off = decltype(offc)::value;
        unsigned long long lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad[0]), "n"(off) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad[1]), "n"(off + 1024) : "memory");
        return ulonglong2{lo, hi};
    };
    auto step_dma_early = [&](auto stgc, auto issue) {
        constexpr int so = decltype(stgc)::value * STG;
        __builtin_amdgcn_sched_barrier(0);
        issue();
        __builtin_amdgcn_sched_barrier(0);
        ulonglong2 ua[4][NPL], ub[4][NPL];
        auto rd = [&](auto tc, auto plc) {
            constexpr int t = decltype(tc)::value, pl = decltype(plc)::value;
            ua[t][pl] = tr16_asm(ad_a[t], std::integral_constant<int, so + pl * PLA>{});
            ub[t][pl] = tr16_asm(ad_b[t], std::integral_constant<int, so + pl * PLB>{});
        };
        typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
        typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
        rd(I0{}, I0{}); if constexpr (NPL == 2) rd(I0{}, I1{});
        rd(I1{}, I0{}); if constexpr (NPL == 2) rd(I1{}, I1{});
        rd(I2{}, I0{}); if constexpr (NPL == 2) rd(I2{}, I1{});
        rd(I3{}, I0{}); if constexpr (NPL == 2) rd(I3{}, I1{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (the asm reads are not tied to the wait by data flow: every fragment passes through an empty asm behind it)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(ua[t][pl].x), "+v"(ua[t][pl].y), "+v"(ub[t][pl].x), "+v"(ub[t][pl].y));
        __builtin_amdgcn_sched_barrier(0);
        f16x8 fa[4][NPL], fb[4][NPL];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) { fa[t][pl] = __builtin_bit_cast(f16x8, ua[t][pl]); fb[t][pl] = __builtin_bit_cast(f16x8, ub[t][pl]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (NTERMS == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][NPL - 1], fb[j][0], acc16_lo[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[j][NPL - 1], acc16_lo[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[j][0], acc16[i][j], 0, 0, 0);
        }
    };
    auto compute_tile = [&]() {
        if constexpr (M16) { compute_tile16(std::integral_constant<int, 0>{}); return; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[NT][NPL], fb[CT][NPL];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fa[i][pl] = tr_frag(fa_base + pl * PLA + 16 * ks * ROWA + 64 * i, ROWA);
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = tr_frag(fb_base + pl * PLB + 16 * ks * ROWB + 64 * j, ROWB);
            if constexpr (NTERMS == 3 && ACC1) {
                // cross terms first, each term over all four accumulators (a dependent MFMA is four instructions away)
                const _Float16 k11 = (_Float16)(1.f / 2048.f);
                f16x8 a1s[NT], b1s[CT];
#pragma unroll
                for (int i = 0; i < NT; ++i) a1s[i] = __builtin_bit_cast(f16x8, fa[i][1]) * k11;
#pragma unroll
                for (int j = 0; j < CT; ++j) b1s[j] = __builtin_bit_cast(f16x8, fb[j][1]) * k11;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1s[i], __builtin_bit_cast(f16x8, fb[j][0]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][0]), b1s[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][0]), __builtin_bit_cast(f16x8, fb[j][0]),
                                                                           acc[i][j], 0, 0, 0);
            } else
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const f16x8 a0 = __builtin_bit_cast(f16x8, fa[i][0]), b0 = __builtin_bit_cast(f16x8, fb[j][0]);
                    if constexpr (NTERMS == 3) {
                        const f16x8 a1 = __builtin_bit_cast(f16x8, fa[i][NPL - 1]), b1 = __builtin_bit_cast(f16x8, fb[j][NPL - 1]);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc_lo[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[i][j], 0, 0, 0);
                }
        }
    };
    typedef std::integral_constant<int, 0> Set0;
    typedef std::integral_constant<int, NSET - 1> Set1;
    // One K-step: hand the staged tile to LDS, refill its register set with tile s + NSET, multiply.  The refill is UNCONDITIONAL -- past the
    // end of the block's reduction it fetches rows nobody stores (zeros beyond the buffers' ends; at most two extra tiles per block) -- so that
    // the loop body is straight-line and hipcc's counted vmcnt before the LDS writes leaves exactly the OTHER set's eight loads in flight (with
    // a conditional refill it merges the two control-flow states and drains everything: seen in the listing).
    auto step = [&](auto setc, int s) {
        if (s > 0) __syncthreads();
        store_tile(setc);
        __syncthreads();
        if (NSET == 2 || s + 1 < S) load_tile(setc);
        compute_tile();
    };
    if constexpr (DMA) {
        // two LDS stages, ONE barrier per K-step: __syncthreads() = s_waitcnt vmcnt(0) (this wave's share of tile s has landed) + s_barrier
        // (every wave's has, and every wave is done reading the stage tile s + 1 is about to overwrite).  Every DMA issued is waited for by a
        // later barrier: none is in flight when the block ends (its LDS may belong to the next block by then).
        typedef std::integral_constant<int, 1> Stg1;
        if (S > 0) {
            load_tile(Set0{});
            for (int s = 0; s < S; s += 2) {
                __syncthreads();
                step_dma_early(Set0{}, [&] { if (s + 1 < S) load_tile(Stg1{}); });
                if (s + 1 < S) {
                    __syncthreads();
                    step_dma_early(Stg1{}, [&] { if (s + 2 < S) load_tile(Set0{}); });
                }
            }
        }
    } else
    if (S > 0) {
        int s = 0;
        load_tile(Set0{});
        if constexpr (NSET == 2) {
            load_tile(Set1{});
            for (; s + 1 < S; s += 2) {
                step(Set0{}, s);
                step(Set1{}, s + 1);
            }
            if (s < S) step(Set0{}, s);
        } else {
            for (; s < S; ++s) step(Set0{}, s);
        }
    }

    const float unscale_a = 1.f / scale_a, unscale_b = 1.f / scale_b;
    float* out = a.out + (size_t)split * a.slab_stride;
    const int col_base = tap * a.Cin;
    if constexpr (M16) {
        // 16 x 16 accumulator: lane l holds rows 4 (l >> 4) + r (cout), column l & 15 (channel)
        const float us = unscale_a * unscale_b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + wave_c * WC + j * 16 + (lane & 15);
            if (c >= a.Cin) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wave_n * WN + i * 16 + 4 * (lane >> 4) + r;
                    float val = acc16[i][j][r];
                    if constexpr (TWO_ACC) val = (val + acc16_lo[i][j][r] * (1.f / 2048.f)) * unscale_a * unscale_b;
                    else val = val * unscale_a * unscale_b;
                    (void)us;
                    if (n < a.N) out[(size_t)n * a.out_row_stride + col_base + c] = val;
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int c = c0 + wave_c * WC + j * 32 + (lane & 31);
        if (c >= a.Cin) continue;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wave_n * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float val = acc[i][j][r];
                if constexpr (TWO_ACC) val = (val + acc_lo[i][j][r] * (1.f / 2048.f)) * unscale_a * unscale_b;
                else val = val * unscale_a * unscale_b;
                if (n < a.N) out[(size_t)n * a.out_row_stride + col_base + c] = val;
            }
        }
    }
}

template <int BN, int BC, int NTERMS>
constexpr size_t wgpl_smem() { return (size_t)(NTERMS == 3 ? 2 : 1) * 32 * (wg_rowb(BN) + wg_rowb(BC)); }

// Measured (tools/acc1_ab.sh): the BatchNorm kernels do speed up next to it, but the matrix-bound dgrads of the main stream, which now
// share their SIMDs with wgrad waves instead of alternating with whole wgrad blocks, lose far more (3x3 dgrad 319 -> 204 TFLOP/s, 1x1
// 159 -> 98): 352-354 vs 376 tiles/s; raising the conv kernels' wave priority (s_setprio 3) does not change that.  OFF by default.
int g_wg_flags = 0;          // rasterisation experiments (WgradArgs::dbg_flags)
extern "C" int pylc_debug_wgrad_flags(int flags) { g_wg_flags = flags; return PYLC_OK; }
// Measured (profiles/r05_ab_runs.txt): two sets take the isolated wgrad from 20.1 to 19.7 ms per step and, inside the step, finish the side
// queue early enough that the dgrads beside it gain 1.7 ms (dgrad1x1 143 -> 160, dgrad3x3 300 -> 316 TFLOP/s) -- which the BatchNorm passes,
// now sharing HBM with a hungrier wgrad, give back (31.8 -> 33.6 ms): the step is unchanged with policy 1 (405.5 vs 405.0 tiles/s) and 0.4 %
// slower with policy 2 (the 1x1 wgrads move as many bytes per FLOP as the BatchNorm passes they run beside).  Default 1.
// Measured (profiles/r05_wgrad_m16.txt): serial, on the step's own tensors, the 16 x 16 x 32 form is +1-2 % on the 1x1 filters and -1..-8 % on
// the multi-tap ones (which give up their second staging set); INSIDE the step it is +0.4 % (416.4 vs 414.9 tiles/s, three interleaved rounds)
// when every f16x3 wgrad takes it and neutral for the 1x1 filters alone -- the cheaper matrix instructions leave power to the dgrads beside them.
int g_wg_m16 = 2;         // 128 x 128 f16x3 wgrad on 16 x 16 x 32 MFMAs: 0 never, 1 single-tap filters, 2 always (pylc_debug_wgrad_m16)
extern "C" int pylc_debug_wgrad_m16(int on) { g_wg_m16 = on; return PYLC_OK; }
int g_wg_dma = 1;         // 1: the 16x16x32 form takes its tiles by LDS-DMA into two LDS stages (pylc_debug_wgrad_dma)
extern "C" int pylc_debug_wgrad_dma(int on) { g_wg_dma = on; return PYLC_OK; }
int g_wg_sets = 1;        // staging register sets policy (pylc_debug_wgrad_sets)
extern "C" int pylc_debug_wgrad_sets(int mode) { g_wg_sets = mode; return PYLC_OK; }
int g_wg_acc1 = 0;        // 1: the 128 x 128 f16x3 wgrad runs its one-accumulator, <= 128-register form (pylc_debug_wgrad_acc1)
extern "C" int pylc_debug_wgrad_acc1(int on) { g_wg_acc1 = on; return PYLC_OK; }
constexpr size_t kAcc1LdsReserve = 72 * 1024;      // two blocks per CU, not four: the other half of the register file is for other kernels

template <int BN, int BC, int WN, int WC>
static int launch_cfg(const WgradArgs& a, long long grid, hipStream_t st) {
    const int fast = a.Q % 32 == 0 ? 1 : 2;
    const dim3 g((unsigned)grid), b(256);
    if constexpr (BN == 128 && BC == 128) {
        if (a.nterms == 3 && g_wg_acc1) {
            if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 1, true>), g, b, kAcc1LdsReserve, st, a);
            else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 2, true>), g, b, kAcc1LdsReserve, st, a);
            PYLC_LAUNCH_CHECK();
            return PYLC_OK;
        }
    }
    // two staging sets (loads two tiles ahead): pylc_debug_wgrad_sets -- 0 never, 1 multi-tap filters only, 2 always
    const bool two = g_wg_sets == 2 || (g_wg_sets == 1 && a.TR * a.TS > 1);
    if constexpr (BN == 128 && BC == 128) {
        if (a.nterms == 1 && g_wg_m16 && g_wg_dma) {
            // one plane (precision mode 3): register-staged, this form is 5-28 % slower than the 32x32x16 one (profiles/r05_wgrad_m16.txt);
            // with LDS-DMA it is the faster one
            constexpr size_t lds = 2 * 2 * 32 * 256;
            if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 1, 1, false, 1, true, true>), g, b, lds, st, a);
            else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 1, 2, false, 1, true, true>), g, b, lds, st, a);
            PYLC_LAUNCH_CHECK();
            return PYLC_OK;
        }
        if (a.nterms == 3 && (g_wg_m16 == 2 || (g_wg_m16 == 1 && a.TR * a.TS == 1))) {
            constexpr size_t lds = 4 * 32 * 256;
            if (g_wg_dma) {
                if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 1, false, 1, true, true>), g, b, 2 * lds, st, a);
                else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 2, false, 1, true, true>), g, b, 2 * lds, st, a);
                PYLC_LAUNCH_CHECK();
                return PYLC_OK;
            }
            // (two staging sets do not fit beside the 40 fragment registers of this form: 256 VGPRs and spills -- one set)
            if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 1, false, 1, true>), g, b, lds, st, a);
            else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 2, false, 1, true>), g, b, lds, st, a);
            PYLC_LAUNCH_CHECK();
            return PYLC_OK;
        }
    }
    if (a.nterms == 1) {
        constexpr size_t lds = wgpl_smem<BN, BC, 1>();
        if (two) {
            if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 1, 1, false, 2>), g, b, lds, st, a);
            else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 1, 2, false, 2>), g, b, lds, st, a);
        } else if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 1, 1>), g, b, lds, st, a);
        else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 1, 2>), g, b, lds, st, a);
    } else {
        constexpr size_t lds = wgpl_smem<BN, BC, 3>();
        if (two) {
            if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 1, false, 2>), g, b, lds, st, a);
            else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 2, false, 2>), g, b, lds, st, a);
        } else if (fast == 1) hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 1>), g, b, lds, st, a);
        else hipLaunchKernelGGL((wgrad_pl_kernel<BN, BC, WN, WC, 3, 2>), g, b, lds, st, a);
    }
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

int launch_wg_pl(WgradArgs& a, int cfg, long long grid, hipStream_t st) {
    a.dbg_flags = g_wg_flags;
    PYLC_REQUIRE(grid > 0 && grid < (1ll << 31), "wgrad grid out of range");
    PYLC_REQUIRE(a.x_planes && a.dy_planes && a.amax_dy && a.amax_x, "wgrad (fp16-plane operands): null planes / ranges");
    PYLC_REQUIRE(a.Q >= 16 && a.Cin % 8 == 0 && a.N_ld % 8 == 0 && a.x_pitch % 8 == 0 && a.dy_pitch % 8 == 0,
                 "wgrad (fp16-plane operands): needs OW >= 16 and channel counts / pitches that are multiples of 8");
    PYLC_REQUIRE(a.x_bytes < 0xFFFFFFF0ll && a.dy_bytes < 0xFFFFFFF0ll, "wgrad (fp16-plane operands): a plane must stay below 4 GiB");
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 64, 64>(a, grid, st);
        case 1: return launch_cfg<64, 64, 32, 32>(a, grid, st);
        default: return launch_cfg<32, 128, 32, 32>(a, grid, st);
    }
}

template <typename K>
static hipError_t opt_in(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int wgrad_pl_init() {
#define PYLC_OPT(BN, BC, WN, WC)                                                                      \
    PYLC_HIP(opt_in(wgrad_pl_kernel<BN, BC, WN, WC, 3, 1>, wgpl_smem<BN, BC, 3>()));                  \
    PYLC_HIP(opt_in(wgrad_pl_kernel<BN, BC, WN, WC, 3, 2>, wgpl_smem<BN, BC, 3>()));                  \
    PYLC_HIP(opt_in(wgrad_pl_kernel<BN, BC, WN, WC, 1, 1>, wgpl_smem<BN, BC, 1>()));                  \
    PYLC_HIP(opt_in(wgrad_pl_kernel<BN, BC, WN, WC, 1, 2>, wgpl_smem<BN, BC, 1>()));                  \
    PYLC_HIP(opt_in((wgrad_pl_kernel<BN, BC, WN, WC, 3, 1, false, 2>), wgpl_smem<BN, BC, 3>()));      \
    PYLC_HIP(opt_in((wgrad_pl_kernel<BN, BC, WN, WC, 3, 2, false, 2>), wgpl_smem<BN, BC, 3>()));      \
    PYLC_HIP(opt_in((wgrad_pl_kernel<BN, BC, WN, WC, 1, 1, false, 2>), wgpl_smem<BN, BC, 1>()));      \
    PYLC_HIP(opt_in((wgrad_pl_kernel<BN, BC, WN, WC, 1, 2, false, 2>), wgpl_smem<BN, BC, 1>()));
    PYLC_OPT(128, 128, 64, 64)
    PYLC_HIP(opt_in((wgrad_pl_kernel<128, 128, 64, 64, 3, 1, false, 1, true, true>), 2 * 4 * 32 * 256));
    PYLC_HIP(opt_in((wgrad_pl_kernel<128, 128, 64, 64, 3, 2, false, 1, true, true>), 2 * 4 * 32 * 256));
    PYLC_HIP(opt_in(wgrad_pl_kernel<128, 128, 64, 64, 3, 1, true>, kAcc1LdsReserve));
    PYLC_HIP(opt_in(wgrad_pl_kernel<128, 128, 64, 64, 3, 2, true>, kAcc1LdsReserve));
    PYLC_OPT(64, 64, 32, 32)
    PYLC_OPT(32, 128, 32, 32)
#undef PYLC_OPT
    return PYLC_OK;
}

}  // namespace pylc
