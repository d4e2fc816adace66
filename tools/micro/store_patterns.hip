// Cost of the conv epilogue's store patterns: every block (512 threads = 8 waves, one block per CU like the ping-pong conv
// kernel) writes a 256 x 128 fp32 output tile (each wave a 64 x 64 sub-tile) of a [M][pitch] matrix, timed per block with
// s_memtime from the first store to `s_waitcnt vmcnt(0)`.  Patterns (what one wave-level store instruction covers):
//   0  dword  per lane,  4 pixel rows x  64 B   (16x16x32 accumulator layout: lane = channel, 4 rows per lane)
//   1  16 B   per lane, 16 pixel rows x  64 B   (transposed accumulators: lane = pixel row, 4 consecutive channels)
//   2  16 B   per lane,  4 pixel rows x 256 B   (the wave's 64-channel strip of a row in one piece: needs an LDS transpose)
//   3  16 B   per lane,  2 pixel rows x 512 B   (whole 128-channel tile rows: two waves' strips written by one)
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/store_patterns.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(512) void bench(float* y, int pitch, int tiles_n, int iters, unsigned long long* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const f32x4 v4 = {1.f * tid, 2.f, 3.f, 4.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int tile = blockIdx.x + it * gridDim.x;
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 128;
        if (P == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wave_m * 64 + i * 16 + 4 * (lane >> 4) + r, n = n0 + wave_n * 64 + j * 16 + (lane & 15);
                        y[(size_t)m * pitch + n] = v4.x;
                    }
        } else if (P == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wave_m * 64 + i * 16 + (lane & 15), n = n0 + wave_n * 64 + j * 16 + 4 * (lane >> 4);
                    *reinterpret_cast<f32x4*>(y + (size_t)m * pitch + n) = v4;
                }
        } else if (P == 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = m0 + wave_m * 64 + k * 4 + (lane >> 4), n = n0 + wave_n * 64 + 4 * (lane & 15);
                *reinterpret_cast<f32x4*>(y + (size_t)m * pitch + n) = v4;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = m0 + wave * 32 + k * 2 + (lane >> 5), n = n0 + 4 * (lane & 31);
                *reinterpret_cast<f32x4*>(y + (size_t)m * pitch + n) = v4;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int P>
static void run(float* y, int pitch, int tiles_n, int grid, int iters, unsigned long long* dout) {
    std::vector<unsigned long long> h(grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<P>, dim3(grid), dim3(512), 0, 0, y, pitch, tiles_n, iters, dout);      // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<P>, dim3(grid), dim3(512), 0, 0, y, pitch, tiles_n, iters, dout);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dout, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double bytes = (double)grid * iters * 256 * 128 * 4;
    printf("  pattern %d  grid %3d  tiles/block %2d : median %7llu ticks per block = %6.0f per tile | kernel %7.1f us = %5.2f TB/s\n", P, grid, iters,
           h[grid / 2], (double)h[grid / 2] / iters, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int pitch = 1024, tiles_n = pitch / 128, M = 256 * 1024;      // 1 GiB matrix: 8192 tiles
    float* y;
    unsigned long long* dout;
    hipMalloc(&y, (size_t)M * pitch * 4);
    hipMalloc(&dout, 256 * sizeof(unsigned long long));
    printf("ticks = s_memtime (100 MHz on gfx950 unless it follows the shader clock: compare with the kernel time)\n");
    for (int grid : {16, 256})
        for (int iters : {1, 4}) {
            run<0>(y, pitch, tiles_n, grid, iters, dout);
            run<1>(y, pitch, tiles_n, grid, iters, dout);
            run<2>(y, pitch, tiles_n, grid, iters, dout);
            run<3>(y, pitch, tiles_n, grid, iters, dout);
        }
    return 0;
}
