// 1x1 / stride-1 convolution (forward, and the dgrads that are one) on fp16-plane operands: a PERSISTENT variant of conv_pl.hip's
// 128 x 128 kernel whose output tile leaves under the NEXT tile's main loop.
//
// Why: on the short reductions of the 1x1 convs (K = Cin / 32 = 2 .. 64 steps) a tile of gg_pl_kernel spends as long in its prologue and
// epilogue as in its main loop, and the epilogue is a burst -- every resident block stores its 64 KB at the same time, the stores leave at
// HBM write rate, and a CU cannot overlap them with its own operand loads (tools/pl_ablate.py on 256 -> 1024 @32^2: 77 us, of which 40 us
// remain with the DMA and the MFMAs removed; tools/pl_stagger_ab.py: de-phasing the blocks of a CU does not help).  Here a block walks a
// strided share of the tiles as ONE stream of K-steps: when a tile's last step is done its accumulators are folded into a second
// register set (`pend`, 64 VGPRs) and the block goes straight on with the next tile, whose first operand tile is already in LDS; the
// sixteen 16-byte stores per lane of the finished tile are issued SPT at a time inside the following K-steps.  The counted vmcnt at the
// top of a step lets exactly those stores stay in flight (gfx9 counts loads and stores in one in-order counter), so a step never waits for
// a store, and the write traffic is spread evenly over the kernel.  The BatchNorm statistics of the finished tile are reduced at fold
// time and combined behind the next step's barrier.
//
// Same pieces, same MFMA order, same fold and statistics arithmetic as gg_pl_kernel: results are bit-identical to it.
// No bias, no accumulation into y, no fused inference epilogue (those launches stay on gg_pl_kernel); Cin >= 64 (two K-steps).
#include "conv_common.h"

namespace pylc {

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int P1_BM = 128, P1_BN = 128, P1_ROW = 64;
template <int NTERMS>
constexpr int p1_stage_bytes() { return (NTERMS == 3 ? 2 : 1) * (P1_BM + P1_BN) * P1_ROW; }
template <int NTERMS>
constexpr int p1_lds_bytes() { return 2 * p1_stage_bytes<NTERMS>() + 2 * P1_BN * 2 * 4 + 64; }

template <int NTERMS, int SPT>
__global__ __launch_bounds__(256, 2) void gg_p1_kernel(const GatherGemmArgs a) {
    constexpr int BM = P1_BM, BN = P1_BN, WM = 64, WN = 64, AT = 4, ROW = P1_ROW, BI = 2;
    constexpr int NPL = NTERMS == 3 ? 2 : 1;
    constexpr int STAGE = p1_stage_bytes<NTERMS>();
    constexpr int OFF_B = NPL * BM * ROW;
    constexpr unsigned OOB = 0x80000000u;                   // >= num_records of every descriptor (takes_pl: buffers below 2 GiB)
    static_assert(SPT == 1 || SPT == 2 || SPT == 4 || SPT == 8, "stores per step");
    extern __shared__ __attribute__((aligned(16))) float smem[];      // ONE LDS object (a second one makes hipcc drain the DMA early)
    char* lds = reinterpret_cast<char*>(smem);
    float* sred = reinterpret_cast<float*>(lds + 2 * STAGE);          // [BM / WM][BN][2] statistics partials of the folded tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // provably wave-uniform: LDS-DMA destinations live in M0
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int lrow = lane >> 2;                                       // loader: 4 lanes per 64-byte row, 16 rows per DMA instruction
    const int lc = (lane & 3) ^ (((lane >> 4) & 1) << 1);             // logical chunk this lane fetches for its LDS position

    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_planes), 0, (int)(a.w_plane_stride * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(a.x_planes) + a.x_plane_stride * 2), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((long long)a.M * a.y_pitch * 4), 0x00020000);      // < 2 GiB: takes_p1
    const unsigned plane1_w = (unsigned)(a.w_plane_stride * 2);
    const int S = (a.Cin + BK - 1) / BK;                              // K-steps per tile (>= 2: launch_gg_p1)
    const int n_tiles = a.n_tiles;
    const int stride_v = (int)gridDim.x;

    // ---- load cursor: the (tile, chunk) whose operand tiles the next issue() requests; runs one K-step ahead of the compute cursor ----
    int ld_v = (int)blockIdx.x, ld_chunk = 0;
    unsigned xoff[2], woff_row[BI];
    auto set_load_tile = [&](int v) {
        const int tile = xcd_remap(v, n_tiles);
        const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 32 * wave + 16 * i + lrow;
            const unsigned off = ((unsigned)m * (unsigned)a.x_pitch + 8u * lc) * 2u;
            xoff[i] = m < a.M ? off : OOB;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int n = n0 + 16 * BI * wave + 16 * i + lrow;
            const unsigned off = ((unsigned)n * (unsigned)a.w_row_stride + 8u * lc) * 2u;
            woff_row[i] = n < a.N ? off : OOB;
        }
    };
    char* const dstA = lds + (32 * wave) * ROW;              // + stage * STAGE + plane * BM * ROW + 16 i * ROW
    char* const dstB = lds + OFF_B + (16 * BI * wave) * ROW;
    // LDS-DMA of the load cursor's reduction tile into `stage` (2 pixel-row pieces + 2 filter-row pieces per thread, NPL planes each;
    // masked lanes fetch zeros through an out-of-range offset), then the cursor moves on -- into the block's next tile after S chunks
    auto issue = [&](int stage) {
        const bool cok = ld_chunk * BK + 8 * lc < a.Cin;                 // Cin % 8 == 0; only the last chunk can be partial
        const unsigned cdelta = (unsigned)(ld_chunk * BK * 2);
        const unsigned so = (unsigned)((a.w_off0 + ld_chunk * BK) * 2);
        char* const sa = dstA + stage * STAGE;
        char* const sb = dstB + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = cok & (xoff[i] != OOB);
            const unsigned vo = ok ? xoff[i] + cdelta : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_vptr)(sa + 16 * i * ROW), 16, vo, 0, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_vptr)(sa + BM * ROW + 16 * i * ROW), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const bool ok = cok & (woff_row[i] != OOB);
            const unsigned vo = ok ? woff_row[i] : OOB;
            const unsigned vo1 = ok ? woff_row[i] + plane1_w : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(sb + 16 * i * ROW), 16, vo, so, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(sb + BN * ROW + 16 * i * ROW), 16, vo1, so, 0, 0);
        }
        if (++ld_chunk == S) {
            ld_chunk = 0;
            ld_v += stride_v;
            if (ld_v < n_tiles) set_load_tile(ld_v);
        }
    };

    // fragment reads: lane l = row (l & 15) of a 16-row fragment, reduction elements 8 (l >> 4) .. +7 of the 32-deep step
    const int koff = 16 * ((lane >> 4) ^ (((lane >> 2) & 1) << 1));
    const char* const ra_base = lds + (wave_m * WM + (lane & 15)) * ROW + koff;
    const char* const rb_base = lds + OFF_B + (wave_n * WN + (lane & 15)) * ROW + koff;

    f32x4v acc[AT][AT];
    f32x4v acc_lo[NTERMS == 3 ? AT : 1][NTERMS == 3 ? AT : 1];
    f32x4v pend[AT][AT];                                     // the finished tile, folded, on its way to memory
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][j][r] = 0.f;
                pend[i][j][r] = 0.f;
                if constexpr (NTERMS == 3) acc_lo[i][j][r] = 0.f;
            }
    int pend_m0 = 0, pend_n0 = 0, pend_tile_m = 0;
    int pq = 16;                                             // vectors of `pend` already stored (16 = nothing pending)
    bool stats_ready = false;                                // sred holds the partials of `pend`'s tile, not yet combined
    const bool do_stats = a.stats != nullptr;
    const float scale_a = a.amax_x ? pow2_scale_for(*a.amax_x) : 1.f;
    const float scale_b = a.amax_w ? pow2_scale_for(*a.amax_w) : 1.f;
    const float unscale_a = 1.f / scale_a, unscale_b = 1.f / scale_b;

    // one vector of `pend`.  A raw buffer store, UNCONDITIONAL as an instruction: rows / channel quads outside the tensor get an
    // out-of-range offset and the hardware drops them (N_store % 4 == 0: all four channels or none).  A branch around a fully masked
    // store would make the number of stores per step data dependent, and the counted vmcnt at the top of the next step would then let
    // operand DMAs instead of stores stay in flight.
#define P1_STORE(Q)                                                                                                        \
    {                                                                                                                      \
        constexpr int i_ = (Q) / 4, j_ = (Q) % 4;                                                                          \
        const int row_ = pend_m0 + wave_m * WM + i_ * 16 + (lane & 15);                                                    \
        const int n4_ = pend_n0 + wave_n * WN + j_ * 16 + 4 * (lane >> 4);                                                 \
        const unsigned off_ = ((unsigned)row_ * (unsigned)a.y_pitch + (unsigned)n4_) * 4u;                                 \
        const unsigned vo_ = (row_ < a.M && n4_ < a.N_store) ? off_ : OOB;                                                 \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, pend[i_][j_]), ry, (int)vo_, 0, 0);             \
    }
#define P1_STORE_GROUP(G)                                                                                                  \
    case G:                                                                                                                \
        if constexpr ((G) * SPT < 16) {                                                                                    \
            P1_STORE((G) * SPT)                                                                                            \
            if constexpr (SPT >= 2) P1_STORE((G) * SPT + 1)                                                                \
            if constexpr (SPT >= 4) { P1_STORE((G) * SPT + 2) P1_STORE((G) * SPT + 3) }                                    \
            if constexpr (SPT >= 8) { P1_STORE((G) * SPT + 4) P1_STORE((G) * SPT + 5) P1_STORE((G) * SPT + 6) P1_STORE((G) * SPT + 7) } \
        }                                                                                                                  \
        break;
    // (the switches below pick the group pq / SPT of the pending tile: compile-time register indices inside each case)

    auto combine_stats = [&]() {
        if (tid < BN) {
            const int n = pend_n0 + tid;
            if (n < a.N_store) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int wm = 0; wm < BM / WM; ++wm) { sm += sred[(wm * BN + tid) * 2]; sq += sred[(wm * BN + tid) * 2 + 1]; }
                float* dst = a.stats + (size_t)pend_tile_m * 2 * a.N_store;
                dst[n] = sm;
                dst[a.N_store + n] = sq;
            }
        }
    };

    if ((int)blockIdx.x >= n_tiles) return;
    set_load_tile(ld_v);
    issue(0);
    int stage = 0;
    bool stored_prev = false;
    for (int cv = (int)blockIdx.x; cv < n_tiles; cv += stride_v) {
        for (int s = 0; s < S; ++s) {
            // this wave's DMA of the tile about to be computed was issued one step ago, BEFORE that step's stores: leaving exactly
            // those stores in flight means the DMA has landed (in-order vmcnt); the barrier extends that to every wave's share and
            // says that every wave has finished reading the stage the next request overwrites
            if (stored_prev) {
                if constexpr (SPT == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if constexpr (SPT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if constexpr (SPT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (ld_v < n_tiles) issue(stage ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if (stats_ready) {                              // written at the fold, before the barrier just passed
                combine_stats();
                stats_ready = false;
            }
            // ---- compute this step ----
            {
                const char* pa = ra_base + stage * STAGE;
                const char* pb = rb_base + stage * STAGE;
                f16x8 fb[AT][NPL];
#pragma unroll
                for (int j = 0; j < AT; ++j)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * ROW + j * 16 * ROW);
#pragma unroll
                for (int i = 0; i < AT; ++i) {
                    f16x8 fa[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) fa[pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * ROW + i * 16 * ROW);
#pragma unroll
                    for (int j = 0; j < AT; ++j) {
                        if constexpr (NTERMS == 3) {
                            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[1], acc_lo[i][j], 0, 0, 0);
                            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][1], fa[0], acc_lo[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[0], acc[i][j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- SPT stores of the previous tile ----
            stored_prev = pq < 16;
            if (stored_prev) {
                switch (pq / SPT) {
                    P1_STORE_GROUP(0)
                    P1_STORE_GROUP(1)
                    P1_STORE_GROUP(2)
                    P1_STORE_GROUP(3)
                    P1_STORE_GROUP(4)
                    P1_STORE_GROUP(5)
                    P1_STORE_GROUP(6)
                    P1_STORE_GROUP(7)
                    P1_STORE_GROUP(8)
                    P1_STORE_GROUP(9)
                    P1_STORE_GROUP(10)
                    P1_STORE_GROUP(11)
                    P1_STORE_GROUP(12)
                    P1_STORE_GROUP(13)
                    P1_STORE_GROUP(14)
                    P1_STORE_GROUP(15)
                    default: break;
                }
                pq += SPT;
            }
            __builtin_amdgcn_sched_barrier(0);
            stage ^= 1;
        }
        // ---- the tile is complete: fold it into `pend` (gg_pl_kernel's epilogue arithmetic), take its statistics, start the next ----
        const int tile = xcd_remap(cv, n_tiles);
        pend_tile_m = tile / a.tiles_n;
        pend_m0 = pend_tile_m * BM;
        pend_n0 = (tile % a.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int j = 0; j < AT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (NTERMS == 3) pend[i][j][r] = (acc[i][j][r] + acc_lo[i][j][r] * (1.f / 2048.f)) * unscale_a * unscale_b;
                    else pend[i][j][r] = acc[i][j][r] * unscale_a * unscale_b;
                    acc[i][j][r] = 0.f;
                    if constexpr (NTERMS == 3) acc_lo[i][j][r] = 0.f;
                }
        pq = 0;
        if (do_stats) {
            float* sdst = sred + ((wave_m * BN) + wave_n * WN + 4 * (lane >> 4)) * 2;
#pragma unroll
            for (int j = 0; j < AT; ++j) {
                const int n4 = pend_n0 + wave_n * WN + j * 16 + 4 * (lane >> 4);
                float cs[4] = {0.f, 0.f, 0.f, 0.f}, css[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < AT; ++i) {
                    const int row = pend_m0 + wave_m * WM + i * 16 + (lane & 15);
                    const bool stored = row < a.M && n4 < a.N_store;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float cv_ = stored ? pend[i][j][r] : 0.f;
                        cs[r] += cv_;
                        css[r] += cv_ * cv_;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { cs[r] = row_sum16(cs[r]); css[r] = row_sum16(css[r]); }
                if ((lane & 15) == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sdst[(j * 16 + r) * 2] = cs[r]; sdst[(j * 16 + r) * 2 + 1] = css[r]; }
                }
            }
            stats_ready = true;
        }
    }
    // ---- drain: what is left of the last tile ----
    if (stats_ready) {
        __syncthreads();
        combine_stats();
    }
    while (pq < 16) {
        switch (pq / SPT) {
            P1_STORE_GROUP(0)
            P1_STORE_GROUP(1)
            P1_STORE_GROUP(2)
            P1_STORE_GROUP(3)
            P1_STORE_GROUP(4)
            P1_STORE_GROUP(5)
            P1_STORE_GROUP(6)
            P1_STORE_GROUP(7)
            P1_STORE_GROUP(8)
            P1_STORE_GROUP(9)
            P1_STORE_GROUP(10)
            P1_STORE_GROUP(11)
            P1_STORE_GROUP(12)
            P1_STORE_GROUP(13)
            P1_STORE_GROUP(14)
            P1_STORE_GROUP(15)
            default: break;
        }
        pq += SPT;
    }
#undef P1_STORE_GROUP
#undef P1_STORE
}

// Measured (tools/p1_ab.py, tools/p1_bench_ab.sh; profiles/r02_p1_ab.txt): bit-identical to gg_pl_kernel, 0-9 % faster per launch in
// isolation (256->1024 @32^2 83 -> 73-84 us, 512->2048 269 -> 250, 64->256 @128^2 195 -> 177), +3 % on the step's forward 1x1 class, and
// NO change of the step time (372.3 / 371.2 vs 371.7 / 371.3 tiles/s).  The in-order vmcnt is why it is not more: a store issued in step
// s has to be complete by the top of step s + 2 (the DMA issued after it is waited for there), so the stores are spread over a few
// steps but still throttle the stream whenever HBM write latency exceeds a K-step.  OFF by default; pylc_debug_p1(1) / PYLC_P1=1.
int g_p1 = 0;
extern "C" int pylc_debug_p1(int on) { g_p1 = on; return PYLC_OK; }

bool takes_p1(const GatherGemmArgs& a) {
    const long long tiles = (long long)cdiv(a.M, P1_BM) * cdiv(a.N_store, P1_BN);
    return g_p1 && !a.w_il && !planes_il(a.x_plane_stride) && a.ident && !a.accumulate && a.bias == nullptr && a.ep_scale == nullptr && a.ep_amax == nullptr && a.ep_res == nullptr && !a.out_planes2 && a.dbg == nullptr && a.Cin >= 2 * BK &&
           tiles >= 2 * kNumCU &&      // at least one full round of resident blocks, or there is no next tile to hide a store behind
           (long long)a.M * a.y_pitch * 4 < (1ll << 31);      // y addressed through a 32-bit buffer offset
}

template <int NTERMS>
static void launch_p1(const GatherGemmArgs& a, unsigned grid, int spt, hipStream_t st) {
    constexpr int lds_bytes = p1_lds_bytes<NTERMS>();
    if (spt == 4) hipLaunchKernelGGL((gg_p1_kernel<NTERMS, 4>), dim3(grid), dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL((gg_p1_kernel<NTERMS, 8>), dim3(grid), dim3(256), lds_bytes, st, a);
}

int launch_gg_p1(GatherGemmArgs& a, hipStream_t st) {
    PYLC_REQUIRE(takes_pl(a) && takes_p1(a), "conv (persistent 1x1 kernel): not a plain 1x1 / stride-1 launch on fp16-plane operands");
    a.tile_bm = P1_BM;
    a.tiles_n = cdiv(a.N_store, P1_BN);
    const long long n_tiles = (long long)cdiv(a.M, P1_BM) * a.tiles_n;
    PYLC_REQUIRE(n_tiles > 0 && n_tiles < (1ll << 31), "conv grid out of range");
    a.n_tiles = (int)n_tiles;
    const int S = cdiv(a.Cin, BK);
    // 16 stores per lane and tile, all issued within the next tile's first steps: 4 per step (8 when a tile has only 2-3 steps).  (1 or 2
    // per step would spread them further, but those instantiations spill: the switch over 16 / 8 store groups costs registers.)
    const int spt = S >= 4 ? 4 : 8;
    const unsigned grid = (unsigned)(n_tiles < 2 * kNumCU ? n_tiles : 2 * kNumCU);
    if (a.nterms == 1) launch_p1<1>(a, grid, spt, st); else launch_p1<3>(a, grid, spt, st);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

template <typename K>
static hipError_t opt_in_p1(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int conv_p1_init() {
    PYLC_HIP(opt_in_p1(gg_p1_kernel<3, 4>, p1_lds_bytes<3>()));
    PYLC_HIP(opt_in_p1(gg_p1_kernel<3, 8>, p1_lds_bytes<3>()));
    PYLC_HIP(opt_in_p1(gg_p1_kernel<1, 4>, p1_lds_bytes<1>()));
    PYLC_HIP(opt_in_p1(gg_p1_kernel<1, 8>, p1_lds_bytes<1>()));
    return PYLC_OK;
}

template __global__ void gg_p1_kernel<3, 4>(const GatherGemmArgs);
template __global__ void gg_p1_kernel<3, 8>(const GatherGemmArgs);
template __global__ void gg_p1_kernel<1, 4>(const GatherGemmArgs);
template __global__ void gg_p1_kernel<1, 8>(const GatherGemmArgs);

}  // namespace pylc
