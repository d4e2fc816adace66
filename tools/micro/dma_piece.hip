// L2 -> LDS rate of LDS-DMA (buffer_load_dwordx4 ... lds) as a function of the PIECE a row contributes to one wave-instruction.
// The conv kernels' LDS image has 64-byte rows (32 halves of one K-step): a wave-instruction copies 16 rows x 64 B, i.e. HALF of a
// 128-byte cache line per row; the other half of the line belongs to the next K-step and is requested again one step later.
// Question: does a K-step built from 8 rows x 128 B pieces (whole lines, 64-deep steps) stream faster from L2?
//   mode 0: 16 rows x  64 B per instruction, step k reads bytes [64 k, 64 k + 64) of every row     (today's kernels)
//   mode 1:  8 rows x 128 B per instruction, step k reads bytes [128 k, 128 k + 128) of every row  (whole lines)
//   mode 2: 16 rows x  64 B, but the two halves of a line are requested back to back by the same wave (k, k + 1 in one step)
// Every block (256 threads, 2 per CU like gg_pl_kernel<3,128>) streams `rows` x `pitch` bytes of its own 128-row panel (L2-resident
// after the first pass; panels of 8 neighbouring blocks coincide like the N-tiles of a conv) `iters` times.  No MFMAs, no LDS reads.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/dma_piece.hip -o /tmp/dp && /tmp/dp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_vptr;

template <int MODE>
__global__ __launch_bounds__(256, 2) void bench(const char* x, int pitch, int panels, int iters, unsigned long long bytes_per_panel) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks b, b + 8, ... share an XCD (round-robin dispatch): 8 of them share a panel (the N-tiles of one M-tile), panels / 8 panels per XCD
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int panel = xcd * (panels >> 3) + ((idx >> 3) % (panels >> 3));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x) + (size_t)panel * bytes_per_panel, 0, (int)bytes_per_panel, 0x00020000);
    const int steps = pitch / (MODE == 1 ? 128 : 64);
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < steps; k += (MODE == 2 ? 2 : 1)) {
            char* dst = lds + ((k & 1) * 32768) + wave * 8192;
            if (MODE == 0 || MODE == 2) {
                // 128 rows x 64 B x 2 "planes" (two buffers' worth) = 16 KB per block and step: 4 waves x 4 instructions x 1 KB
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned row = 32 * wave + 16 * (i & 1) + (lane >> 2);
                    const unsigned off = row * pitch + k * 64 + (lane & 3) * 16 + (i >> 1) * (pitch * 128);       // second "plane": the panel's other half
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(dst + i * 1024), 16, off, 0, 0, 0);
                    if (MODE == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(dst + 4096 + i * 1024), 16, off + 64, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned row = 32 * wave + 8 * (i & 3) + (lane >> 3);
                    const unsigned off = row * pitch + k * 128 + (lane & 7) * 16 + (i >> 2) * (pitch * 128);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(dst + i * 1024), 16, off, 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
static void run(const char* x, int pitch, int panels, int iters, size_t bytes_per_panel, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(bench<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 512;
    hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 65536, 0, x, pitch, panels, 2, (unsigned long long)bytes_per_panel);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 65536, 0, x, pitch, panels, iters, (unsigned long long)bytes_per_panel);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * iters * 2.0 * 128 * pitch;           // useful bytes copied to LDS
    printf("pitch %5d B  %-34s %8.1f us  %6.2f TB/s useful (%.1f B/clk/CU at 2.1 GHz)\n", pitch, name, 1e3 * ms, bytes / ms / 1e9,
           bytes / ms / 1e9 * 1e12 / 256 / 2.1e9 / 1e3);
}

int main() {
    for (int pitch : {512, 2048}) {
        const int panels = pitch == 512 ? 64 : 32;           // 1 MB / 2 MB of panels per XCD: L2-resident                         // 256 / 1024 channels of fp16
        const size_t bytes_per_panel = (size_t)256 * pitch;       // 128 rows x 2 halves
        char* x;
        hipMalloc(&x, panels * bytes_per_panel);
        hipMemset(x, 1, panels * bytes_per_panel);
        const int iters = pitch == 512 ? 64 : 16;
        for (int rep = 0; rep < 2; ++rep) {
            run<0>(x, pitch, panels, iters, bytes_per_panel, "16 rows x 64 B per instruction");
            run<1>(x, pitch, panels, iters, bytes_per_panel, "8 rows x 128 B per instruction");
            run<2>(x, pitch, panels, iters, bytes_per_panel, "16 rows x 64 B, halves back to back");
        }
        hipFree(x);
    }
    return 0;
}
