// Issue cost (cycles per wave64 instruction) of the VALU forms used by the f16x3 operand split, measured (a) by one wave
// alone on its SIMD and (b) by a wave whose SIMD partner issues v_mfma_f32_32x32x16_f16 back to back (the ping-pong
// kernel's regime).  Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__device__ __forceinline__ void body(float& a0, float& a1, float& a2, float& a3, unsigned& b0, unsigned& b1, unsigned& b2, unsigned& b3, float s) {
    // 64 x 4 = 256 independent-ish instructions
    if (KIND == 0) asm volatile(REP64("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
    if (KIND == 1) asm volatile(REP64("v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
    if (KIND == 2) asm volatile(REP64("v_cvt_pk_f16_f32 %4, %0, %1\n v_cvt_pk_f16_f32 %5, %1, %2\n v_cvt_pk_f16_f32 %6, %2, %3\n v_cvt_pk_f16_f32 %7, %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
    if (KIND == 3) asm volatile(REP64("v_cvt_pkrtz_f16_f32 %4, %0, %1\n v_cvt_pkrtz_f16_f32 %5, %1, %2\n v_cvt_pkrtz_f16_f32 %6, %2, %3\n v_cvt_pkrtz_f16_f32 %7, %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
    if (KIND == 4) asm volatile(REP64("v_fma_mix_f32 %0, %4, %8, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %5, %8, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %6, %8, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %7, %8, %3 op_sel_hi:[1,0,0]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
    if (KIND == 5) asm volatile(REP64("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %5\n v_cvt_f32_f16 %2, %6\n v_cvt_f32_f16 %3, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
    if (KIND == 6) asm volatile(REP64("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
    if (KIND == 7) asm volatile(REP64("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));
}

// blockDim 64: the wave alone.  blockDim 512: waves 4-7 measure, waves 0-3 (their SIMD partners) issue MFMAs for the whole time.
template <int KIND>
__global__ void bench(unsigned long long* out, float seed, int mfma_iters) {
    const int wave = threadIdx.x >> 6;
    if (blockDim.x == 512 && wave < 4) {
        f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        f16x8 fa = {(_Float16)seed, 0, 0, 0, 0, 0, 0, 0}, fb = fa;
        for (int it = 0; it < mfma_iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc3, 0, 0, 0);
        }
        if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.f) out[63] = 1;
        return;
    }
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, s = 1.5f;
    unsigned b0 = 1, b1 = 2, b2 = 3, b3 = 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 8; ++it) body<KIND>(a0, a1, a2, a3, b0, b1, b2, b3, s);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[KIND * 8 + (wave & 3)] = t1 - t0;
    if (a0 + a1 + a2 + a3 == 12345.f && b0 + b1 + b2 + b3 == 7) out[62] = 0;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64 * 8);
    const char* names[8] = {"v_mul_f32", "v_and_b32", "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32", "v_fma_mix_f32", "v_cvt_f32_f16", "v_fma_f32", "v_sub_f32"};
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, 64 * 8);
        const int threads = mode ? 512 : 64;
#define RUN(K) hipLaunchKernelGGL(bench<K>, dim3(1), dim3(threads), 0, 0, d, 1.0f, 4000); hipDeviceSynchronize();
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
        unsigned long long h[64];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf(mode ? "beside a SIMD partner issuing MFMAs back to back:\n" : "one wave alone on its SIMD:\n");
        for (int k = 0; k < 8; ++k) printf("  %-22s %.2f cycles per instruction\n", names[k], (double)h[k * 8] / (8.0 * 256.0));
    }
    return 0;
}
