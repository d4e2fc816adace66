// Sustained rate of the two fp16 MFMA shapes with every CU busy (operands in registers, random data): is 16x16x32 cheaper
// per FLOP than 32x32x16 under the board power cap?   hipcc -O3 --offload-arch=gfx950 mfma_shapes.hip -o /tmp/ms && /tmp/ms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const _Float16* in, float* out, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(in + ((threadIdx.x * 8 + i) * 8) % 4096);
        b[i] = *reinterpret_cast<const f16x8*>(in + ((threadIdx.x * 8 + i + 4) * 8) % 4096);
    }
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[3], acc[3], 0, 0, 0);
            }
        for (int j = 0; j < 4; ++j) sum += acc[j][0] + acc[j][7];
    } else {
        f32x4 acc[8] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[q & 3], acc[q], 0, 0, 0);
            }
        for (int j = 0; j < 8; ++j) sum += acc[j][0] + acc[j][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main() {
    _Float16* in; float* out;
    hipMalloc(&in, 4096 * 2 + 64); hipMalloc(&out, 2048 * 512 * 4);
    _Float16 h[4096 + 32];
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int shape : {32, 16}) {
            const int iters = 40000, blocks = 512;          // 2 blocks of 8 waves per CU
            // FLOPs per wave per iteration: 32x32x16: 16 MFMAs x 32768; 16x16x32: 32 MFMAs x 16384 -> same
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 8 * iters * 16 * 32768.0;
            printf("mfma %dx%dx%d f16: %.1f ms  %.0f TFLOP/s executed\n", shape, shape, shape == 32 ? 16 : 32, ms, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
