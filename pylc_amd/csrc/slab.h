// Row-slab walk shared by the HBM-bound [M rows][C channels] kernels (bn.hip, dwconv.hip).
#pragma once
#include "common.h"

namespace pylc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxSlabs = 1024;    // measured: 512 slabs (2 blocks/CU) costs the HBM-bound BN kernels 10-25 %

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
    long long rps = cdiv<long long>(M, kMaxSlabs);
    const long long min_rows = (long long)g.RL * 8;
    if (rps < min_rows) rps = min_rows;
    rps = cdiv<long long>(rps, g.RL) * g.RL;
    g.rows_per_slab = (int)rps;
    g.nslab = (int)cdiv<long long>(M, rps);
    return g;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

}  // namespace pylc
