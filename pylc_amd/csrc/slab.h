// Row-slab walk shared by the HBM-bound [M rows][C channels] kernels (bn.hip, dwconv.hip).
#pragma once
#include "common.h"
#include <cstdlib>

namespace pylc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxSlabs = 2048;    // workspace bound; the launch limit is slab_limit()
constexpr int kDefaultSlabs = 1024;   // dwconv.hip's strip kernels
// Blocks per launch of the row-slab kernels (BatchNorm, ReLU, dropout).  With one row per iteration 1024 (4 blocks per CU) was
// best (512 cost 10-25 %); with the batched row walks (walk_rows) a wave keeps 8-12 loads in flight and 768 = three blocks per CU
// wins inside the training step: 344-345 vs 336-337 tiles/s on one box (640: 342, 896: 340, 1536 / 2048: 335).
// Round 5, one queue (wgrad no longer beside these kernels): 512 = two blocks per CU -- R101 +0.1 .. +0.35 % by box, U-Net +0.6 %, configs[4]
// +1.1 %; 256 / 384 / 640 lose 0.5-1 %, 1024 and 1536 2.3-2.8 % (profiles/r05_ab_runs.txt).
// PYLC_MAX_SLABS overrides for A/B runs.
constexpr int kRowSlabs = 512;
inline int slab_limit() {
    static const int v = [] {
        const char* e = getenv("PYLC_MAX_SLABS");
        const int n = e ? atoi(e) : kRowSlabs;
        return n < 64 ? 64 : (n > kMaxSlabs ? kMaxSlabs : n);
    }();
    return v;
}

struct Slab {
    int CV;             // float4 vectors per row
    int cols;           // vector columns handled per pass = min(CV, 256)
    int RL;             // row lanes = 256 / cols
    int rows_per_slab;  // multiple of RL
    int nslab;
    long long M;
};

inline Slab make_slab(long long M, int C) {
    Slab g;
    g.M = M;
    g.CV = C / 4;
    g.cols = g.CV < 256 ? g.CV : 256;
    g.RL = 256 / g.cols;
    long long rps = cdiv<long long>(M, slab_limit());
    const long long min_rows = (long long)g.RL * 8;
    if (rps < min_rows) rps = min_rows;
    rps = cdiv<long long>(rps, g.RL) * g.RL;
    g.rows_per_slab = (int)rps;
    g.nslab = (int)cdiv<long long>(M, rps);
    return g;
}

// A thread's rows r0, r0 + RL, ... are walked kRowBatch at a time: `load(u, row)` for every row of the batch first, then
// `math(u, row, valid)` for every row (unconditional arithmetic on the loaded values), then `store(u, row)` for the valid ones.
// Why three passes: gfx9 counts vector loads and stores in one in-order counter (vmcnt), and once a store is in flight a wait
// for an older load can only be vmcnt(0).  A row-at-a-time loop (load, wait, store, next load, wait ...) therefore has ONE
// 16-byte load in flight per wave and waits for its own previous store every iteration; here a wave has kRowBatch x (loads per
// row) in flight and meets its stores once per batch.  The tail batch clamps its row indices, so the loads stay unconditional
// (straight-line code); rows are used in ascending order, so per-thread sums are bit-identical to the row-at-a-time order.
constexpr int kRowBatch = 4;
// `batch_end()` runs once per batch after its math pass (the reductions fold their fp32 batch sums into fp64 accumulators there).
template <int NB = kRowBatch, class L, class M, class S, class E>
__device__ __forceinline__ void walk_rows(long long r0, long long r_end, int RL, L load, M math, S store, E batch_end) {
    for (long long r = r0; r < r_end; r += (long long)NB * RL) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const long long rr = r + (long long)u * RL;
            load(u, rr < r_end ? rr : r_end - 1);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const long long rr = r + (long long)u * RL;
            math(u, rr, rr < r_end);
        }
        batch_end();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const long long rr = r + (long long)u * RL;
            if (rr < r_end) store(u, rr);
        }
    }
}

template <int NB = kRowBatch, class L, class M, class S>
__device__ __forceinline__ void walk_rows(long long r0, long long r_end, int RL, L load, M math, S store) {
    walk_rows<NB>(r0, r_end, RL, load, math, store, [] {});
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- one-plane fp16 tensors (precision mode 3: include/pylc_hip.h "fp16 planes" with nplanes = 1) as operands of the HBM-bound kernels ----
// element = rn16(s x), s = pow2_scale_for(bound) from the tensor's range bound; 8 bytes per float4 channel vector.  ldq / stq address
// in ELEMENTS from the tensor's base, so the index math of a kernel is the same for both formats.
__device__ __forceinline__ f32x4 half4_to_f32(uint2 u) {
    typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
    const f16x4_ h = __builtin_bit_cast(f16x4_, u);
    const f32x4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return v;
}
__device__ __forceinline__ uint2 f32_to_half4(f32x4 v) {
    typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
    const f16x4_ h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    return __builtin_bit_cast(uint2, h);
}
template <bool HALF>
__device__ __forceinline__ f32x4 ldq(const void* base, size_t elem, float inv_scale) {
    if constexpr (HALF) return half4_to_f32(*reinterpret_cast<const uint2*>(static_cast<const _Float16*>(base) + elem)) * inv_scale;
    else return ld4(static_cast<const float*>(base) + elem);
}
template <bool HALF>
__device__ __forceinline__ void stq(void* base, size_t elem, f32x4 v, float scale) {
    if constexpr (HALF) *reinterpret_cast<uint2*>(static_cast<_Float16*>(base) + elem) = f32_to_half4(v * scale);
    else st4(static_cast<float*>(base) + elem, v);
}
// the power of two that maps a tensor bounded by `bound_bits` (float bits) into [2^14, 2^15) (conv_common.h pow2_scale_for)
__device__ __forceinline__ float half_scale_for(unsigned bound_bits) {
    int e = (int)((bound_bits >> 23) & 0xFFu);
    int se = 127 + 14 - (e - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}

}  // namespace pylc
