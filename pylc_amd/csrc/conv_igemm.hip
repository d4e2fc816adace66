// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   forward / dgrad : "gather GEMM"  Y[m][n] = sum_{tap, c} X[gather(m, tap)][c] * Wt[n][tap][c]
//                     m = output pixel (b,p,q), n = output channel.  The same kernel serves the forward
//                     pass (taps = kernel window), stride-1 dgrad (flipped taps, CRSK weights) and
//                     stride-2 dgrad (one launch per output parity class with that class's tap subset),
//                     because the tap set is an arithmetic progression described by scalars.
//   wgrad           : dW[n][tap][c] = sum_m dY[m][n] * X[gather(m, tap)][c], split-K over pixels with
//                     deterministic slab reduction.
//
// Data layout: activations NHWC with a channel pitch (so producers can write straight into concat
// buffers), weights KRSC.  Tiles are staged global -> VGPR -> LDS (register staging lets rows be padded to
// 36 floats, which makes the ds_read_b128 fragment reads bank-conflict free) and double-buffered: the
// global loads of step s+1 are in flight while the 64 MFMAs of step s issue.
//
// Reference call sites replaced: nn.Conv2d in models/backbone/resnet.py:21-26,72,92,
// models/modules/aspp.py:18,64,67, models/decoder.py:27-38, models/architectures/unet.py:78,112,116,137,
// models/backbone/xception.py:32,48,122,126 and their autograd backward.
#include "conv_common.h"

namespace pylc {

// PREC 0: v_mfma_f32_32x32x2_f32 (bit-exact fp32 fmaf chain, 157 TFLOP/s peak).
// PREC 1: "bf16x6" -- every fp32 operand is split EXACTLY into three bf16 pieces (x = x0 + x1 + x2, 8 mantissa bits
//         each, by truncation) while it is staged into LDS, and each product a*b is evaluated as the six leading
//         cross terms a2b0 + a0b2 + a1b1 + a1b0 + a0b1 + a0b0 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//         The dropped terms are <= 2^-23 |a*b|, i.e. the result is as accurate as the fp32 chain (measured: mean
//         error 1.2e-8 vs 1.9e-8 of sum|a*b| at K = 2304), but the matrix pipe runs 16x faster per instruction, so
//         6 terms cost 6/16 of the fp32-MFMA time.
template <int BM, int BN, int WM, int WN, bool CIN4, int PREC>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, 2) void gather_gemm_kernel(const GatherGemmArgs a) {
    constexpr int WAVES_N = BN / WN;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int NTHR = (BM / WM) * (BN / WN) * 64;      // 256 (4 waves) or 512 (8 waves, two per SIMD)
    constexpr int RPI = NTHR / 8;                          // tile rows staged per loader iteration
    constexpr int A_IT = BM / RPI, B_IT = BN / RPI;
    static_assert(NTHR == 256 || NTHR == 512, "4 or 8 waves per block");
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must divide evenly over the loader");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                    // [2][BM][LDT]
    float* sB = smem + 2 * BM * LDT;     // [2][BN][LDT]

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / a.tiles_n) * BM;
    const int n0 = (tile % a.tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int v = tid & 7, r0 = tid >> 3;

    // ---- per-row gather bases (fixed for the whole K loop) ----
    int rowh[A_IT], roww[A_IT], rowpix[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + RPI * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int q = mm % a.Q, t = mm / a.Q;
        const int p = t % a.P, b = t / a.P;
        rowh[i] = ok ? p * a.in_sh : -(1 << 28);     // invalid rows fail every bounds check
        roww[i] = q * a.in_sw;
        rowpix[i] = b * a.IH * a.IW;
    }

    const int T = a.TR * a.TS;
    const int nchunks = CIN4 ? (T + 7) / 8 : (a.Cin + BK - 1) / BK;
    // ---- tap skipping: a tap whose window is out of bounds for every row of this tile is never loaded ----
    unsigned long long tapmask = ~0ull;
    if (!CIN4 && T > 1) {
        tapmask = 0;
        for (int t = 0; t < T; ++t) {
            const int dh = a.dh0 + (t / a.TS) * a.dh_step, dw = a.dw0 + (t % a.TS) * a.dw_step;
            int any = 0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                any |= ((unsigned)(rowh[i] + dh) < (unsigned)a.IH) & ((unsigned)(roww[i] + dw) < (unsigned)a.IW);
            if (__syncthreads_or(any)) tapmask |= 1ull << t;
        }
    }
    const int ntaps = CIN4 ? 1 : __popcll(T >= 64 ? tapmask : (tapmask & ((1ull << T) - 1)));
    const int S = ntaps * nchunks;

    f32x16 acc[MT][NT];
    f32x16 acc_lo[PREC == 2 ? MT : 1][PREC == 2 ? NT : 1];      // PREC 2: the two cross terms, carried at 2^11 x their value
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (PREC == 2) acc_lo[i][j][r] = 0.f;
            }

    f32x4 ra[A_IT], rb[B_IT];
    float scale_a = 1.f, scale_b = 1.f;                              // PREC 2: power-of-two operand scales
    f32x4 ra2[PREC != 0 ? A_IT : 1], rb2[PREC != 0 ? B_IT : 1];     // second in-flight tile (bf16x6: prefetch distance 2)
    int ld_tap = -1, ld_chunk = nchunks - 1;     // position of the NEXT tile to load (advanced before use)

    auto advance = [&]() {
        if (++ld_chunk == nchunks) {
            ld_chunk = 0;
            if (!CIN4) {
                do { ++ld_tap; } while (!((tapmask >> ld_tap) & 1ull));
            }
        }
    };
    auto load_tile_into = [&](f32x4* ra, f32x4* rb) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (CIN4) {
            const int tap = ld_chunk * 8 + v;
            const bool tok = tap < T;
            const int dh = a.dh0 + (tap / a.TS) * a.dh_step, dw = a.dw0 + (tap % a.TS) * a.dw_step;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int hi = rowh[i] + dh, wi = roww[i] + dw;
                const bool ok = tok && (unsigned)hi < (unsigned)a.IH && (unsigned)wi < (unsigned)a.IW;
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + (size_t)(rowpix[i] + hi * a.IW + wi) * a.x_pitch) : zero;
            }
            const int kk = ld_chunk * 32 + 4 * v;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int n = n0 + r0 + RPI * i;
                const bool ok = tok && n < a.N;
                rb[i] = ok ? *reinterpret_cast<const f32x4*>(a.w + (size_t)n * a.w_row_stride + a.w_off0 + kk) : zero;
            }
        } else {
            const int tr = ld_tap / a.TS, ts = ld_tap % a.TS;
            const int dh = a.dh0 + tr * a.dh_step, dw = a.dw0 + ts * a.dw_step;
            const int woff = a.w_off0 + tr * a.w_step_r + ts * a.w_step_s;
            const int c = ld_chunk * BK + 4 * v;
            const bool cok = c < a.Cin;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int hi = rowh[i] + dh, wi = roww[i] + dw;
                const bool ok = cok && (unsigned)hi < (unsigned)a.IH && (unsigned)wi < (unsigned)a.IW;
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + (size_t)(rowpix[i] + hi * a.IW + wi) * a.x_pitch + c) : zero;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int n = n0 + r0 + RPI * i;
                const bool ok = cok && n < a.N;
                rb[i] = ok ? *reinterpret_cast<const f32x4*>(a.w + (size_t)n * a.w_row_stride + woff + c) : zero;
            }
        }
    };
    auto load_tile = [&]() { load_tile_into(ra, rb); };

    if constexpr (PREC == 0) {
        auto store_tile = [&](int buf) {
            float* pa = sA + buf * BM * LDT + r0 * LDT + 4 * v;
            float* pb = sB + buf * BN * LDT + r0 * LDT + 4 * v;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) *reinterpret_cast<f32x4*>(pa + RPI * i * LDT) = ra[i];
#pragma unroll
            for (int i = 0; i < B_IT; ++i) *reinterpret_cast<f32x4*>(pb + RPI * i * LDT) = rb[i];
        };
        if (S > 0) {
            advance();
            load_tile();
            store_tile(0);
            __syncthreads();
            for (int s = 0; s < S; ++s) {
                const int buf = s & 1;
                if (s + 1 < S) {
                    advance();
                    load_tile();            // global loads in flight during the MFMAs below
                }
                // lane l supplies A[row = l&31][k = 8g + 4(l>>5) + j] to MFMA (g, j): one ds_read_b128 feeds 4 MFMAs
                const float* pa = sA + buf * BM * LDT + (wave_m * WM + (lane & 31)) * LDT + 4 * (lane >> 5);
                const float* pb = sB + buf * BN * LDT + (wave_n * WN + (lane & 31)) * LDT + 4 * (lane >> 5);
#pragma unroll
                for (int g = 0; g < BK / 8; ++g) {
                    f32x4 fa[MT], fb[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(pa + i * 32 * LDT + 8 * g);
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(pb + j * 32 * LDT + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
                }
                if (s + 1 < S) store_tile(buf ^ 1);
                __syncthreads();
            }
        }
    } else {
        // single LDS stage of NPL 16-bit planes per operand; rows padded to 80 B -> conflict-free ds_read_b128
        constexpr int NPL = PREC == 2 ? 2 : 3;
        char* lds = reinterpret_cast<char*>(smem);
        char* pA0 = lds;                               // plane p of A at pA0 + p * BM * LDB
        char* pB0 = lds + NPL * BM * LDB;              // plane p of B at pB0 + p * BN * LDB
        if constexpr (PREC == 2) {
            scale_a = a.amax_x ? pow2_scale_for(*a.amax_x) : 1.f;
            scale_b = a.amax_w ? pow2_scale_for(*a.amax_w) : 1.f;
        }
        auto store_tile_from = [&](const f32x4* xa, const f32x4* xb) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                char* d = pA0 + (r0 + RPI * i) * LDB + 8 * v;
                uint2 q0, q1, q2;
                if constexpr (PREC == 2) {
                    split2(xa[i], scale_a, q0, q1);
                } else {
                    split3(xa[i], q0, q1, q2);
                    *reinterpret_cast<uint2*>(d + 2 * BM * LDB) = q2;
                }
                *reinterpret_cast<uint2*>(d) = q0;
                *reinterpret_cast<uint2*>(d + BM * LDB) = q1;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                char* d = pB0 + (r0 + RPI * i) * LDB + 8 * v;
                uint2 q0, q1, q2;
                if constexpr (PREC == 2) {
                    split2(xb[i], scale_b, q0, q1);
                } else {
                    split3(xb[i], q0, q1, q2);
                    *reinterpret_cast<uint2*>(d + 2 * BN * LDB) = q2;
                }
                *reinterpret_cast<uint2*>(d) = q0;
                *reinterpret_cast<uint2*>(d + BN * LDB) = q1;
            }
        };
        const char* ra_base = pA0 + (wave_m * WM + (lane & 31)) * LDB + 16 * (lane >> 5);
        const char* rb_base = pB0 + (wave_n * WN + (lane & 31)) * LDB + 16 * (lane >> 5);
        auto compute = [&]() {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[MT][NPL], fb[NT][NPL];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        fa[i][pl] = *reinterpret_cast<const bf16x8*>(ra_base + pl * BM * LDB + i * 32 * LDB + 32 * ks);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        fb[j][pl] = *reinterpret_cast<const bf16x8*>(rb_base + pl * BN * LDB + j * 32 * LDB + 32 * ks);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (PREC == 2) {
                            const f16x8 a0 = __builtin_bit_cast(f16x8, fa[i][0]), a1 = __builtin_bit_cast(f16x8, fa[i][1]);
                            const f16x8 b0 = __builtin_bit_cast(f16x8, fb[j][0]), b1 = __builtin_bit_cast(f16x8, fb[j][1]);
                            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc_lo[i][j], 0, 0, 0);
                            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc_lo[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[i][j], 0, 0, 0);
                        } else {
                            // smallest terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][NPL - 1], fb[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NPL - 1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                        }
                    }
            }
        };
        if (S > 0) {
            // two register sets: the global loads of tile s+2 are issued before the MFMAs of tile s, so every load has
            // a full step (>= 1536 matrix-pipe cycles plus the partner block's step) to land before it is consumed
            constexpr bool ONE_SET = PREC == 2 && BM == 256 && BN == 64;    // 128 accumulator registers + 8 A vectors per set: one set fits
            advance();
            load_tile_into(ra, rb);
            if constexpr (ONE_SET) {
                for (int s = 0; s < S; ++s) {
                    if (s > 0) __syncthreads();
                    store_tile_from(ra, rb);
                    __syncthreads();
                    if (s + 1 < S) { advance(); load_tile_into(ra, rb); }
                    compute();
                }
            } else {
            if (S > 1) { advance(); load_tile_into(ra2, rb2); }
            for (int s = 0; s < S; s += 2) {
                if (s > 0) __syncthreads();      // every wave has finished reading the previous tile
                store_tile_from(ra, rb);
                __syncthreads();
                if (s + 2 < S) { advance(); load_tile_into(ra, rb); }
                compute();
                if (s + 1 < S) {
                    __syncthreads();
                    store_tile_from(ra2, rb2);
                    __syncthreads();
                    if (s + 3 < S) { advance(); load_tile_into(ra2, rb2); }
                    compute();
                }
            }
            }
            __syncthreads();                     // LDS is reused for the epilogue's row table
        }
    }

    // ---- epilogue: per-row output offsets through LDS, then 128-B coalesced row segments per half-wave ----
    int* rowoff = reinterpret_cast<int*>(smem);            // element offsets of the tile's output rows (< 2^31: check_desc), -1 = no row
    if (tid < BM) {
        const int m = m0 + tid;
        long long off = -1;
        if (m < a.M) {
            const int q = m % a.Q, t = m / a.Q;
            const int p = t % a.P, b = t / a.P;
            off = ((long long)(b * a.OH + p * a.out_sh + a.oh0) * a.OW + q * a.out_sw + a.ow0) * a.y_pitch;
        }
        rowoff[tid] = (int)off;
    }
    __syncthreads();
    // BatchNorm statistics ride along: each lane owns one output column of the wave tile, so the column sums of the
    // values just computed cost 2 FMAs per element here instead of a separate full read of y
    float* sred = reinterpret_cast<float*>(smem) + 1024;          // [BM/WM][BN][2], past the row table
    float ep_max = 0.f;
    const bool do_stats = a.stats != nullptr;
    const float unscale_a = 1.f / scale_a, unscale_b = 1.f / scale_b;     // exact (powers of two); applied one after the other
    // Per-column vectors of ALL column tiles are fetched and waited for here, before the first store: gfx9 counts loads and
    // stores in one in-order counter, so a wait for them inside the conditional store blocks below would be repeated in every
    // block and, behind a store in flight, could only be vmcnt(0) -- every store would wait for the previous one to reach memory.
    const bool ep = a.ep_scale != nullptr;
    float bvs[NT], escs[NT], eshs[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wave_n * WN + j * 32 + (lane & 31);
        bvs[j] = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
        escs[j] = (ep && n < a.N) ? a.ep_scale[n] : 0.f;
        eshs[j] = (ep && n < a.N) ? a.ep_shift[n] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(bvs[j]), "v"(escs[j]), "v"(eshs[j]));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wave_n * WN + j * 32 + (lane & 31);
        const bool nok = n < a.N_store;
        const float bv = bvs[j], esc = escs[j], esh = eshs[j];
        float cs = 0.f, css = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int offs[16];
            float prev[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave_m * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                offs[r] = rowoff[row];
            }
            const float* extra = a.accumulate ? a.y : a.ep_res;      // old output (accumulate) or the residual (fused epilogue)
            if (extra != nullptr) {  // all 16 values first (independent loads), then add and store
#pragma unroll
                for (int r = 0; r < 16; ++r) prev[r] = (nok && offs[r] >= 0) ? extra[offs[r] + n] : 0.f;
            }
            // arithmetic for all 16 values first, unconditionally (this is where the old values are waited for), then stores that
            // depend on ALU results only -- see the note on vmcnt above
            float vals[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float val = acc[i][j][r];
                if constexpr (PREC == 2) val = (val + acc_lo[i][j][r] * (1.f / 2048.f)) * unscale_a * unscale_b;
                val += bv;
                if (ep) val = val * esc + esh;                       // BatchNorm-apply's own expression and order
                if (extra != nullptr) val += prev[r];
                if (a.ep_relu) val = fmaxf(val, 0.f);
                vals[r] = val;
                const bool counted = nok && offs[r] >= 0;            // rows / channels outside the tensor do not count
                // statistics are taken of (stored value - bias): a conv whose bias dwarfs its spread (U-Net's first layer on the
                // 1/255-scaled input: |mean| / sigma ~ 20) would otherwise lose its variance in sum(y^2)/n - mean^2; the finalize adds
                // the bias back to the mean (pylc_bn_finalize*_ex `shift`).  bv == 0 leaves every bit as it was.
                const float cv = counted ? val - bv : 0.f;
                cs += cv;
                css += cv * cv;
                ep_max = fmaxf(ep_max, counted ? fabsf(val) : 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (nok && offs[r] >= 0) a.y[offs[r] + n] = vals[r];
        }
        if (do_stats) {
            cs += __shfl_xor(cs, 32, 64);                           // lanes l and l+32 hold the same column
            css += __shfl_xor(css, 32, 64);
            if (lane < 32) {
                float* d = sred + ((wave_m * BN) + wave_n * WN + j * 32 + lane) * 2;
                d[0] = cs;
                d[1] = css;
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        if (tid < BN) {
            const int n = n0 + tid;
            if (n < a.N_store) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int wm = 0; wm < BM / WM; ++wm) { sm += sred[(wm * BN + tid) * 2]; sq += sred[(wm * BN + tid) * 2 + 1]; }
                float* dst = a.stats + (size_t)(tile / a.tiles_n) * 2 * a.N_store;
                dst[n] = sm;
                dst[a.N_store + n] = sq;
            }
        }
    }
    if (a.ep_amax != nullptr) amax_commit(ep_max, a.ep_amax);
}

// -------------------------------------------------------------------------------------------------
// f16x3 gather-GEMM, 256x128 tile, ping-pong schedule
// -------------------------------------------------------------------------------------------------
// Same arithmetic and epilogue as gather_gemm_kernel<256,128,64,64,false,2>; different main loop.  A 512-thread block
// puts two waves on each SIMD (waves w and w+4).  The two halves of the block run the identical per-step work -- (C) 16
// fragment reads + 24 MFMAs on the current LDS stage, (S) split + store of their share of the next tile into the other
// stage -- half a step apart: while waves 0-3 are in C (matrix pipe), their SIMD partners 4-7 are in S (VALU, LDS
// stores, global loads), then they swap.  The matrix pipe and the VALU of every SIMD are busy in the same interval
// instead of one after the other (MI355X_MICROARCH.md, "Two waves per SIMD").  LDS stores are 16 bytes per lane (8
// reduction elements of one plane): wide stores keep their rate with one storing wave per SIMD.
__device__ __attribute__((aligned(16))) float g_zero_page[8];      // zero-initialised: where masked-off lanes load from

constexpr int PP_BM = 256, PP_BN = 128;
constexpr int PP_STAGE = 2 * (PP_BM + PP_BN) * LDB;      // bytes per LDS stage (two fp16 planes per operand), 80-byte padded rows
constexpr int PP_STAGE_SWZ = 2 * (PP_BM + PP_BN) * 64;   // ... with the unpadded swizzled rows: three stages fit

__device__ __forceinline__ void split2x8(const f32x4 lo, const f32x4 hi, float s, uint4& p0, uint4& p1) {
    uint2 a0, a1, b0, b1;
    split2(lo, s, a0, a1);
    split2(hi, s, b0, b1);
    p0 = make_uint4(a0.x, a0.y, b0.x, b0.y);
    p1 = make_uint4(a1.x, a1.y, b1.x, b1.y);
}

// plane 0 only: rn16(s v) of eight values (the first term of split2: the one-plane arithmetic of precision mode 3)
__device__ __forceinline__ uint4 plane0x8(const f32x4 lo, const f32x4 hi, float s) {
    typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
    const f32x4 a = lo * s, b = hi * s;
    const f16x8_ h = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
    return __builtin_bit_cast(uint4, h);
}

// M16 (with SWZ): the products run on v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16 -- same FLOPs per
// matrix-pipe cycle, same LDS traffic, but 16 % more sustained throughput at the board's power cap (bare loops on random
// data: 2005 vs 1690 TFLOP/s, tools/micro/mfma_shapes.hip).  Wave tile 64x64 = 4x4 tiles of 16x16.
// EP: the fused inference epilogue (ep_scale / ep_shift / ep_res / ep_relu / ep_amax) is compiled in only for EP = true; the
// training kernels do not carry it (its mere presence cost 1 % of the training step).
// ONE (inference in precision mode 3, with EP and M16): the operands are rounded to ONE fp16 plane -- rn16(s x), the first term of the split --
// and each product is one MFMA: a third of the matrix work, half of the split work and of the LDS traffic (plane 1 of a stage stays unused).
template <bool STAMPS, bool BPL, bool SWZ, bool M16, bool S3 = SWZ, bool EP = false, bool ONE = false>
__global__ __launch_bounds__(512, 2) void gather_gemm_pp_kernel(const GatherGemmArgs a) {
    static_assert(!ONE || (M16 && EP), "the one-plane form exists for the fused-inference instantiation only");
    static_assert(!S3 || SWZ, "three stages need the unpadded rows");
    static_assert(!M16 || SWZ, "the 16x16x32 variant uses the unpadded swizzled LDS rows");
    constexpr int BM = PP_BM, BN = PP_BN, WM = 64, WN = 64, MT = 2, NT = 2, WAVES_N = 2;
    // LDS rows: SWZ = unpadded 64-byte rows with the 16-byte chunk index XORed by bits 2-3 of the row (conflict-free
    // ds_read_b128 fragments AND ds_write_b128 stores under the 16-lane / 8-lane group rules); !SWZ = rows padded to 80 bytes
    constexpr int LDB = SWZ ? 64 : pylc::LDB;
    constexpr int PP_STAGE = 2 * (BM + BN) * LDB;
    // STAMPS: waves 0 and 4 of block 0 record s_memtime at every segment boundary into LDS (dumped to a.dbg at the end)
    int n_stamp = 0;
#define PP_STAMP()                                                                                                     \
    if constexpr (STAMPS) {                                                                                            \
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && n_stamp < 256) {                                            \
            reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(smem) + (S3 ? 3 : 2) * PP_STAGE)[(threadIdx.x >> 8) * 256 + n_stamp] = \
                __builtin_amdgcn_s_memtime();                                                                          \
            ++n_stamp;                                                                                                 \
        }                                                                                                              \
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    // whole-tile phases (slots 248..255 of each group's stamp row): block start, main loop start, main loop end, fold done, stores
    // issued; 253..255 split the prologue: geometry done, first operands landed, first LDS stores done
#define PP_STAMP_AT(slot)                                                                                              \
    if constexpr (STAMPS) {                                                                                            \
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0)                                                               \
            reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(smem) + (S3 ? 3 : 2) * PP_STAGE)[(threadIdx.x >> 8) * 256 + (slot)] = \
                __builtin_amdgcn_s_memtime();                                                                          \
    }
    PP_STAMP_AT(248);

    // Tile loop: normally one tile per block (grid = n_tiles); with fewer blocks than tiles each block walks a strided
    // share (persistent mode, see launch_gg_pp).
    for (int t_ = blockIdx.x; t_ < a.n_tiles; t_ += gridDim.x) {
    const int tile = xcd_remap(t_, a.n_tiles);
    const int m0 = (tile / a.tiles_n) * BM;
    const int n0 = (tile % a.tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int grp = wave >> 2;                    // 0: waves 0-3, 1: their SIMD partners 4-7
    const int tg = tid & 255, v = tg & 3, lr = tg >> 2;      // loader: 4 lanes x 8 elements per 32-deep row, 64 rows per pass
    const int arow0 = 128 * grp + lr;             // this thread stages A rows arow0, arow0 + 64 and B row brow
    const int brow = 64 * grp + lr;

    int rowh[2], roww[2], rowpix[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + arow0 + 64 * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int q = mm % a.Q, t = mm / a.Q;
        const int p = t % a.P, b = t / a.P;
        rowh[i] = ok ? p * a.in_sh : -(1 << 28);
        roww[i] = q * a.in_sw;
        rowpix[i] = b * a.IH * a.IW;
    }
    const int T = a.TR * a.TS;
    const int nchunks = (a.Cin + BK - 1) / BK;
    // Taps that reach at least one valid input pixel of this tile (the others are skipped): a ballot per tap gives the wave's
    // mask, the 8 wave masks meet in LDS behind ONE barrier.  (A __syncthreads_or per tap -- two barriers and an LDS reduction
    // each -- made the prologue of every 3x3 tile 10 k cycles longer than a 1x1 tile's: tools/pp_stamps.py.)
    unsigned long long tapmask = ~0ull;
    if (T > 1) {
        __shared__ unsigned long long s_wave_taps[8];
        unsigned long long mine = 0;
        for (int t = 0; t < T; ++t) {
            const int dh = a.dh0 + (t / a.TS) * a.dh_step, dw = a.dw0 + (t % a.TS) * a.dw_step;
            int any = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                any |= ((unsigned)(rowh[i] + dh) < (unsigned)a.IH) & ((unsigned)(roww[i] + dw) < (unsigned)a.IW);
            if (__builtin_amdgcn_ballot_w64(any != 0) != 0) mine |= 1ull << t;
        }
        if (lane == 0) s_wave_taps[wave] = mine;
        __syncthreads();
        unsigned long long all = 0;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) all |= s_wave_taps[wv];
        tapmask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(all >> 32)) << 32) |
                  (unsigned)__builtin_amdgcn_readfirstlane((int)all);
    }
    const int ntaps = __popcll(T >= 64 ? tapmask : (tapmask & ((1ull << T) - 1)));
    const int S = ntaps * nchunks;

    constexpr int AT = M16 ? 4 : 2, AR = M16 ? 4 : 16;      // accumulator tiles per wave-tile side, registers per tile
    using acc_t = typename std::conditional<M16, f32x4v, f32x16>::type;
    acc_t acc[AT][AT], acc_lo[ONE ? 1 : AT][ONE ? 1 : AT];
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (!ONE) acc_lo[i][j][r] = 0.f;
            }
    const float scale_a = a.amax_x ? pow2_scale_for(*a.amax_x) : 1.f;
    const float scale_b = a.amax_w ? pow2_scale_for(*a.amax_w) : 1.f;

    struct Regs { f32x4 a[2][2]; f32x4 b[2]; };
    Regs R0, R1;
    // reduction position of the next tile to load: tap (ld_tr, ld_ts) -- kept as counters, no division in the loop -- and chunk
    // Order: taps innermost -- the nine taps of a 3x3 window re-read the SAME 32 channels of overlapping pixel rows (40 KB per
    // block and channel chunk) back to back, so eight of the nine reads hit in the cache hierarchy; with channel chunks
    // innermost a tap's re-read came a whole input tile (256 KB per block, 8 MB per XCD against 4 MB of L2) later and went out
    // to the fabric again.  (dbg_flags bit 12 (4096): the old chunk-inner order, for A/B runs.)
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
    // Operands are fetched with raw buffer loads: a lane that is masked off (padding tap, row past M, channel past Cin,
    // filter row past N, tile past the end of the reduction) gets an offset beyond the buffer and the hardware returns
    // zeros.  No branches around loads (the loop body stays straight-line, so the compiler waits with a counted vmcnt for
    // exactly the register set it consumes while the other set stays in flight), no 64-bit address arithmetic, and the
    // per-row part of the offset is computed once.
    constexpr unsigned OOB = 0x80000000u;                 // >= num_records (launch_gg_pp only takes buffers below 2 GiB)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    // BPL: the filter comes pre-split (two fp16 planes of 2 bytes per element, prepared once per optimiser step), so the
    // B tile is copied to LDS without any arithmetic; otherwise fp32 filters are split here like the activations
    const __amdgpu_buffer_rsrc_t rw = BPL
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_planes), 0, (int)(a.w_plane_stride * 4), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    unsigned xoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)       // byte offset of (row's pixel at tap offset (0,0), channel 8v); garbage for invalid rows (masked by rowh)
        xoff[i] = ((unsigned)(rowpix[i] + rowh[i] * a.IW + roww[i]) * (unsigned)a.x_pitch + 8u * v) * 4u;
    const int bn = n0 + brow;
    // (chunk-interleaved filter planes, a.w_il: row and chunk offsets double, plane 1 sits 64 bytes behind plane 0)
    const unsigned wil = (unsigned)__builtin_amdgcn_readfirstlane((BPL && a.w_il) ? 2 : 1);      // (wave-uniform: it scales the loads' scalar offset)
    const unsigned woff_row = bn < a.N ? ((unsigned)bn * (unsigned)a.w_row_stride * wil + 8u * v) * (BPL ? 2u : 4u) : OOB;
    const unsigned plane1 = (BPL && a.w_il) ? 64u : (unsigned)(a.w_plane_stride * 2);
    auto ldx = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, f32x4& lo, f32x4& hi) {
        lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
        hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u, soff, 0));
    };
    auto load = [&](Regs& R, bool valid) {
        if (valid) advance();
        const int tr = ld_tr, ts = ld_ts;
        const int dh = a.dh0 + tr * a.dh_step, dw = a.dw0 + ts * a.dw_step;
        const int woff = a.w_off0 + tr * a.w_step_r + ts * a.w_step_s;
        const bool cok = valid && ld_chunk * BK + 8 * v < a.Cin;                  // Cin % 8 == 0 on this path
        const unsigned tapdelta = (unsigned)(((dh * a.IW + dw) * a.x_pitch + ld_chunk * BK) * 4);     // wave-uniform, may be "negative"
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = rowh[i] + dh, wi = roww[i] + dw;
            const bool ok = cok && (unsigned)hi < (unsigned)a.IH && (unsigned)wi < (unsigned)a.IW;
            ldx(rx, ok ? xoff[i] + tapdelta : OOB, 0u, R.a[i][0], R.a[i][1]);
        }
        if constexpr (BPL) {       // 8 halves of plane 0 and of plane 1
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((woff + ld_chunk * BK) * 2) * wil));
            R.b[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, cok ? woff_row : OOB, so, 0));
            if constexpr (!ONE) R.b[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, cok ? woff_row + plane1 : OOB, so, 0));
        } else {
            ldx(rw, cok ? woff_row : OOB, (unsigned)((woff + ld_chunk * BK) * 4), R.b[0], R.b[1]);
        }
    };
    // stage layout: A plane 0 [256][LDB] | A plane 1 | B plane 0 [128][LDB] | B plane 1
    // chunk swizzle of a row: conflict-free for the fragment pattern in use (32-row fragments: bits 2-3 of the row;
    // 16-row fragments of the 16x16x32 MFMA: bit 2 of the row into bit 1 of the chunk -- found by exhaustive search over
    // the XOR-linear maps against the ds_read_b128 lane groups)
    auto swz = [](int row) { return M16 ? (((row >> 2) & 1) << 1) : ((row >> 2) & 3); };
    char* st_a = lds + arow0 * LDB + 16 * (SWZ ? (v ^ swz(arow0)) : v);
    char* st_b = lds + 2 * BM * LDB + brow * LDB + 16 * (SWZ ? (v ^ swz(brow)) : v);
    auto store = [&](int stage, const Regs& R) {
        char* base_a = st_a + stage * PP_STAGE;
        if constexpr (STAMPS) {
            if (a.dbg_flags & 16) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); PP_STAMP(); }     // fine timeline: loads landed
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint4 p0, p1;
            if constexpr (ONE) {
                p0 = plane0x8(R.a[i][0], R.a[i][1], scale_a);
                *reinterpret_cast<uint4*>(base_a + 64 * i * LDB) = p0;
            } else {
                split2x8(R.a[i][0], R.a[i][1], scale_a, p0, p1);
                *reinterpret_cast<uint4*>(base_a + 64 * i * LDB) = p0;
                *reinterpret_cast<uint4*>(base_a + 64 * i * LDB + BM * LDB) = p1;
            }
            if constexpr (STAMPS) {
                if (a.dbg_flags & 16) { __builtin_amdgcn_sched_barrier(0); PP_STAMP(); }               // item i split + stored (stamp drains LDS)
            }
        }
        uint4 p0, p1;
        if constexpr (BPL) {
            p0 = __builtin_bit_cast(uint4, R.b[0]);
            if constexpr (!ONE) p1 = __builtin_bit_cast(uint4, R.b[1]);
        } else if constexpr (ONE) {
            p0 = plane0x8(R.b[0], R.b[1], scale_b);
        } else {
            split2x8(R.b[0], R.b[1], scale_b, p0, p1);
        }
        char* base_b = st_b + stage * PP_STAGE;
        *reinterpret_cast<uint4*>(base_b) = p0;
        if constexpr (!ONE) *reinterpret_cast<uint4*>(base_b + BN * LDB) = p1;
    };
    const char* ra_base = lds + (wave_m * WM + (lane & (M16 ? 15 : 31))) * LDB;
    const char* rb_base = lds + 2 * BM * LDB + (wave_n * WN + (lane & (M16 ? 15 : 31))) * LDB;
    const int sw_r = SWZ ? swz(lane & 31) : 0;           // rows of one lane differ by multiples of 16 / 32: same swizzle
    const int koff[2] = {M16 ? 16 * ((lane >> 4) ^ sw_r) : 16 * ((lane >> 5) ^ sw_r), 16 * (((lane >> 5) + 2) ^ sw_r)};
    auto compute = [&](int stage) {
        const char* pa = ra_base + stage * PP_STAGE;
        const char* pb = rb_base + stage * PP_STAGE;
        if constexpr (M16) {
            // lane l: row (l & 15) of a 16-row fragment, reduction elements 8 (l >> 4) .. +7 of the 32-deep step
            constexpr int NPLF = ONE ? 1 : 2;
            f16x8 fb[4][NPLF];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pl = 0; pl < NPLF; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * LDB + j * 16 * LDB + koff[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x8 fa[NPLF];
#pragma unroll
                for (int pl = 0; pl < NPLF; ++pl) fa[pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * LDB + i * 16 * LDB + koff[0]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // the filter fragment is the FIRST operand: the 16x16 result comes out transposed (lane l: pixel l & 15, channels
                    // 4 (l >> 4) .. +3), which is what lets the epilogue store 16 bytes per lane
                    if constexpr (!ONE) {
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[NPLF - 1], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][NPLF - 1], fa[0], acc_lo[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[0], acc[i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 fa[MT][2], fb[NT][2];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        fa[i][pl] = *reinterpret_cast<const f16x8*>(pa + pl * BM * LDB + i * 32 * LDB + koff[ks]);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        fb[j][pl] = *reinterpret_cast<const f16x8*>(pb + pl * BN * LDB + j * 32 * LDB + koff[ks]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], acc_lo[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        }
    };

    if (S > 0) {
        // Tiles t >= S are zero tiles (all lanes read the zero page): an odd S runs one harmless extra step and the
        // steady-state loop needs no tail conditions.
#define PP_SYNC()                                \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __syncthreads();                         \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
        if constexpr (S3) {
            // THREE LDS stages (48 KB each with the unpadded rows), ONE barrier per K-step.  (Two swizzled stages = 96 KB would let
            // a wgrad block share the CU; measured inside the full step: 326 vs 329 tiles/s, so three it is.)  Step s: every wave computes on
            // the stage of tile s and stores its share of tile s+2 into the stage that tile s-1 vacated (last read before the
            // previous barrier); tile s+1 was completed before that barrier too.  The two halves of the block do the two
            // segments in opposite order, so on every SIMD one wave is in its MFMA segment while its partner splits / stores /
            // loads -- with no barrier in the middle of the step for either to wait at.
            PP_STAMP_AT(253);                                 // geometry and tap mask done
            load(R0, true);                                   // tile 0
            load(R1, 1 < S);                                  // tile 1
            if constexpr (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            PP_STAMP_AT(254);                                 // first operand round trip
            store(0, R0);
            store(1, R1);
            if constexpr (STAMPS) __builtin_amdgcn_sched_barrier(0);
            PP_STAMP_AT(255);                                 // split + LDS stores of two operand tiles
            load(R0, 2 < S);                                  // tile 2
            load(R1, 3 < S);                                  // tile 3
            __syncthreads();
            PP_STAMP_AT(249);
            int sc = 0;                                       // stage of the tile being computed
            auto plus1 = [](int x) { return x == 2 ? 0 : x + 1; };
            auto plus2 = [](int x) { return x == 0 ? 2 : x - 1; };
            if (grp == 0) {
                for (int s = 0; s < S; s += 2) {
                    PP_STAMP();
                    compute(sc);
                    __builtin_amdgcn_sched_barrier(0);
                    PP_STAMP();
                    store(plus2(sc), R0);                     // tile s+2
                    load(R0, s + 4 < S);
                    PP_STAMP();
                    PP_SYNC();
                    sc = plus1(sc);
                    PP_STAMP();
                    compute(sc);
                    __builtin_amdgcn_sched_barrier(0);
                    PP_STAMP();
                    store(plus2(sc), R1);                     // tile s+3
                    load(R1, s + 5 < S);
                    PP_STAMP();
                    PP_SYNC();
                    sc = plus1(sc);
                }
            } else {
                for (int s = 0; s < S; s += 2) {
                    PP_STAMP();
                    store(plus2(sc), R0);
                    load(R0, s + 4 < S);
                    __builtin_amdgcn_sched_barrier(0);
                    PP_STAMP();
                    compute(sc);
                    PP_STAMP();
                    PP_SYNC();
                    sc = plus1(sc);
                    PP_STAMP();
                    store(plus2(sc), R1);
                    load(R1, s + 5 < S);
                    __builtin_amdgcn_sched_barrier(0);
                    PP_STAMP();
                    compute(sc);
                    PP_STAMP();
                    PP_SYNC();
                    sc = plus1(sc);
                }
            }
        } else {
        load(R0, true);                                   // tile 0
        load(R1, 1 < S);                                  // tile 1
        store(0, R0);
        load(R0, 2 < S);                                  // tile 2
        __syncthreads();
        // hipcc is free to sink MFMAs below an s_barrier (nothing orders them against it), which would smear each wave's
        // compute segment into its own store segment and undo the ping-pong: pin every segment boundary
        // Each half alternates a compute segment (fragment reads + 24 MFMAs on the current stage) and a store segment (split +
        // LDS stores of the next tile, then the buffer loads of the tile after next into the register set just freed).
        // Invariant at the top of pair s: tile s complete in stage 0, stage 1 free, tile s+1 in R1, tile s+2 in flight into R0.
        // (Measured alternatives: loads issued inside the compute segments, s_setprio around the MFMA cluster or around the
        // store segment -- all slower or neutral.)
        if (grp == 0) {
            for (int s = 0; s < S; s += 2) {
                PP_STAMP();
                compute(0);
                PP_STAMP();
                PP_SYNC();
                PP_STAMP();
                store(1, R1);
                load(R1, s + 3 < S);
                PP_STAMP();
                PP_SYNC();
                PP_STAMP();
                compute(1);
                PP_STAMP();
                PP_SYNC();
                PP_STAMP();
                store(0, R0);
                load(R0, s + 4 < S);
                PP_STAMP();
                PP_SYNC();
            }
        } else {
            for (int s = 0; s < S; s += 2) {
                PP_STAMP();
                store(1, R1);
                load(R1, s + 3 < S);
                PP_STAMP();
                PP_SYNC();
                PP_STAMP();
                compute(0);
                PP_STAMP();
                PP_SYNC();
                PP_STAMP();
                store(0, R0);
                load(R0, s + 4 < S);
                PP_STAMP();
                PP_SYNC();
                PP_STAMP();
                compute(1);
                PP_STAMP();
                PP_SYNC();
            }
        }
        }
#undef PP_SYNC
        PP_STAMP_AT(250);
    }

    // ---- epilogue (as gather_gemm_kernel) ----
    int* rowoff = reinterpret_cast<int*>(smem);            // element offsets of the tile's output rows (< 2^31: check_desc), -1 = no row
    if (tid < BM) {
        const int m = m0 + tid;
        long long off = -1;
        if (m < a.M) {
            const int q = m % a.Q, t = m / a.Q;
            const int p = t % a.P, b = t / a.P;
            off = ((long long)(b * a.OH + p * a.out_sh + a.oh0) * a.OW + q * a.out_sw + a.ow0) * a.y_pitch;
        }
        rowoff[tid] = (int)off;
    }
    __syncthreads();
    float* sred = reinterpret_cast<float*>(smem) + 1024;
    float ep_max = 0.f;
    const bool do_stats = a.stats != nullptr;
    const float unscale_a = 1.f / scale_a, unscale_b = 1.f / scale_b;
    // fold the cross-term accumulator in first: the store loops below then hold 128, not 256, accumulator registers
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < AT; ++j)
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                if constexpr (ONE) acc[i][j][r] = acc[i][j][r] * unscale_a * unscale_b;
                else acc[i][j][r] = (acc[i][j][r] + acc_lo[i][j][r] * (1.f / 2048.f)) * unscale_a * unscale_b;
            }
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP_AT(251);
    if constexpr (M16) {
    // 16x16 tiles, transposed (see compute()): lane l holds pixel row l & 15 of every 16-row tile and the four CONSECUTIVE
    // channels 4 (l >> 4) .. +3 of every 16-channel tile -> one 16-byte store per tile and lane.
    // The epilogue runs in two phases because gfx9 counts loads AND stores in one in-order counter (vmcnt): any load issued after
    // a store -- an old value to accumulate into, a bias, even the reload of a spilled register -- can only be waited for by
    // waiting for that store to reach memory first.  Interleaved with the stores (as they were: per 16 x 16 tile), such waits cost
    // 14-16 k cycles per output tile, as much as the 8 K-steps of a short-K 1x1 conv (tools/pp_stamps.py; the stores alone take
    // 2-4 k: tools/micro/store_patterns.hip).  Phase 1: every row lookup, element offset, bias / scale vector and old value;
    // phase 2: arithmetic and stores only.
    const float* extra = a.accumulate ? a.y : (EP ? a.ep_res : nullptr);
    const bool ep = EP && a.ep_scale != nullptr;
    int eoff[AT][AT];                                            // element offset of (pixel row of tile i, channel quad of tile j), -1: not stored
    float bv[AT][4], esc[EP ? AT : 1][4], esh[EP ? AT : 1][4];
    {
        int offs[AT];
#pragma unroll
        for (int i = 0; i < AT; ++i) offs[i] = rowoff[wave_m * WM + i * 16 + (lane & 15)];
#pragma unroll
        for (int j = 0; j < AT; ++j) {
            const int n4 = n0 + wave_n * WN + j * 16 + 4 * (lane >> 4);
            const bool nok = n4 < a.N_store;                    // N_store % 4 == 0: all four channels or none
#pragma unroll
            for (int i = 0; i < AT; ++i) eoff[i][j] = (nok && offs[i] >= 0) ? offs[i] + n4 : -1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bv[j][r] = (a.bias != nullptr && n4 + r < a.N) ? a.bias[n4 + r] : 0.f;
                if constexpr (EP) {
                    esc[j][r] = (ep && n4 + r < a.N) ? a.ep_scale[n4 + r] : 0.f;
                    esh[j][r] = (ep && n4 + r < a.N) ? a.ep_shift[n4 + r] : 0.f;
                }
            }
        }
    }
    float* sdst = sred + ((wave_m * BN) + wave_n * WN + 4 * (lane >> 4)) * 2;
    // phase 2 for the channel tiles [j0, j0 + NJ): (a) ALL the arithmetic, unconditionally and in place -- this is where every
    // loaded value (bias, scale, old value) is waited for; (b) the stores, which then depend on ALU results only.  (With the
    // arithmetic inside the per-tile `if (stored)` blocks the compiler has to re-wait for the conditionally loaded registers in
    // every block, and once a store is in flight such a wait can only be vmcnt(0): each store waited for the previous one to
    // reach memory -- 16 x 800 cycles per output tile.)
    constexpr int PJ = EP ? 1 : 2;          // channel tiles per batch of old values (the fused inference epilogue carries 32 more registers)
    auto finish = [&](auto has_prev, auto j0c, const f32x4v (&prev)[AT][PJ]) {
        constexpr int j0 = decltype(j0c)::value;
        constexpr int NJ = decltype(has_prev)::value ? PJ : AT;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = j0 + jj;
            float cs[4] = {0.f, 0.f, 0.f, 0.f}, css[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < AT; ++i) {
                const bool stored = eoff[i][j] >= 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float val = acc[i][j][r] + bv[j][r];
                    if constexpr (EP) {
                        if (ep) val = val * esc[j][r] + esh[j][r];       // BatchNorm-apply's own expression and order
                    }
                    if constexpr (decltype(has_prev)::value) val += prev[i][jj][r];
                    if (EP && a.ep_relu) val = fmaxf(val, 0.f);
                    acc[i][j][r] = val;
                    const float cv = stored ? val - bv[j][r] : 0.f;      // rows / channels outside the tensor do not count; statistics of
                    cs[r] += cv;                                         // (value - bias), see gather_gemm_kernel's epilogue
                    css[r] += cv * cv;
                    if (EP) ep_max = fmaxf(ep_max, stored ? fabsf(val) : 0.f);
                }
            }
            if (do_stats) {                                         // the 16 lanes of a DPP row hold the 16 pixel rows of one channel quad
#pragma unroll
                for (int r = 0; r < 4; ++r) { cs[r] = row_sum16(cs[r]); css[r] = row_sum16(css[r]); }
                if ((lane & 15) == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sdst[(j * 16 + r) * 2] = cs[r]; sdst[(j * 16 + r) * 2 + 1] = css[r]; }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int i = 0; i < AT; ++i)
                if (eoff[i][j0 + jj] >= 0) *reinterpret_cast<f32x4v*>(a.y + eoff[i][j0 + jj]) = acc[i][j0 + jj];
    };
    __builtin_amdgcn_sched_barrier(0);                          // nothing of phase 1 may sink below a store
    if (extra == nullptr) {
        const f32x4v none[AT][PJ] = {};
        finish(std::false_type{}, std::integral_constant<int, 0>{}, none);
    } else {
        // accumulate / fused residual: the old values of PJ channel tiles at a time (16 registers each) -- one wait behind
        // stores per batch instead of one per 16 x 16 tile
        f32x4v prev[AT][PJ];
        auto fetch = [&](int j0) {
#pragma unroll
            for (int i = 0; i < AT; ++i)
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj) {
                    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
                    prev[i][jj] = eoff[i][j0 + jj] >= 0 ? *reinterpret_cast<const f32x4v*>(extra + eoff[i][j0 + jj]) : zero;
                }
        };
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
        finish(std::true_type{}, std::integral_constant<int, 0>{}, prev);
        if constexpr (PJ == 1) {
            __builtin_amdgcn_sched_barrier(0);
            fetch(1);
            __builtin_amdgcn_sched_barrier(0);
            finish(std::true_type{}, std::integral_constant<int, 1>{}, prev);
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(2);
        __builtin_amdgcn_sched_barrier(0);
        finish(std::true_type{}, std::integral_constant<int, 2>{}, prev);
        if constexpr (PJ == 1) {
            __builtin_amdgcn_sched_barrier(0);
            fetch(3);
            __builtin_amdgcn_sched_barrier(0);
            finish(std::true_type{}, std::integral_constant<int, 3>{}, prev);
        }
    }
    } else {
    // accumulator layout: 32x32 tiles -- lane l holds column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    constexpr int TS = 32;
#pragma unroll
    for (int j = 0; j < AT; ++j) {
        const int n = n0 + wave_n * WN + j * TS + (lane & (TS - 1));
        const bool nok = n < a.N_store;
        const float bv = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
        const bool ep = EP && a.ep_scale != nullptr;
        const float esc = (ep && n < a.N) ? a.ep_scale[n] : 0.f, esh = (ep && n < a.N) ? a.ep_shift[n] : 0.f;
        float cs = 0.f, css = 0.f;
#pragma unroll
        for (int i = 0; i < AT; ++i) {
            int offs[AR];
            float prev[AR];
#pragma unroll
            for (int r = 0; r < AR; ++r) offs[r] = rowoff[wave_m * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
            // accumulate / fused residual: fetch all values first (independent loads in flight), then add and store --
            // interleaved load/store pairs serialise because the compiler cannot prove the rows distinct
            const float* extra = a.accumulate ? a.y : (EP ? a.ep_res : nullptr);
            if (extra != nullptr) {
#pragma unroll
                for (int r = 0; r < AR; ++r) prev[r] = (nok && offs[r] >= 0) ? extra[offs[r] + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                if (nok && offs[r] >= 0) {
                    float val = acc[i][j][r] + bv;
                    if (ep) val = val * esc + esh;                   // BatchNorm-apply's own expression and order
                    if (extra != nullptr) val += prev[r];
                    if (EP && a.ep_relu) val = fmaxf(val, 0.f);
                    a.y[offs[r] + n] = val;
                    cs += val - bv;                                  // statistics of (value - bias), see gather_gemm_kernel's epilogue
                    css += (val - bv) * (val - bv);
                    if (EP) ep_max = fmaxf(ep_max, fabsf(val));
                }
            }
        }
        if (do_stats) {
            cs += __shfl_xor(cs, 32, 64);                           // lanes l and l+32 hold the same column
            css += __shfl_xor(css, 32, 64);
            if (lane < TS) {
                float* d = sred + ((wave_m * BN) + wave_n * WN + j * TS + lane) * 2;
                d[0] = cs;
                d[1] = css;
            }
        }
    }
    }
    if (do_stats) {
        __syncthreads();
        if (tid < BN) {
            const int n = n0 + tid;
            if (n < a.N_store) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int wm = 0; wm < BM / WM; ++wm) { sm += sred[(wm * BN + tid) * 2]; sq += sred[(wm * BN + tid) * 2 + 1]; }
                float* dst = a.stats + (size_t)(tile / a.tiles_n) * 2 * a.N_store;
                dst[n] = sm;
                dst[a.N_store + n] = sq;
            }
        }
    }
    if (EP && a.ep_amax != nullptr) amax_commit(ep_max, a.ep_amax);
    if constexpr (STAMPS) __builtin_amdgcn_sched_barrier(0);
    PP_STAMP_AT(252);
    if constexpr (STAMPS) {
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && a.dbg != nullptr)
            for (int k = 0; k < 256; ++k)
                a.dbg[(threadIdx.x >> 8) * 256 + k] = (k < n_stamp || k >= 248)
                    ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(smem) + (S3 ? 3 : 2) * PP_STAGE)[(threadIdx.x >> 8) * 256 + k] : 0ull;
    }
    __syncthreads();          // the next tile reuses the LDS stages and the epilogue tables
    }
#undef PP_STAMP
#undef PP_STAMP_AT
}

// -------------------------------------------------------------------------------------------------
// wgrad
// -------------------------------------------------------------------------------------------------

template <int BN, int BC, int WN, int WC, bool CIN4>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int WAVES_C = BC / WC;
    constexpr int NT = WN / 32, CT = WC / 32;
    constexpr int VA = BN / 4, RA = 256 / VA, IA = 32 / RA;      // dy tile loader geometry
    constexpr int VB = BC / 4, RB = 256 / VB, IB = 32 / RB;      // x tile loader geometry
    static_assert((BN / WN) * (BC / WC) == 4, "4 waves per block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                  // [2][32][BN]  dy tile, pixel-major
    float* sB = smem + 2 * 32 * BN;    // [2][32][BC]  gathered x tile, pixel-major

    const int T = a.TR * a.TS;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = id % a.tiles_c; id /= a.tiles_c;
    const int tn = id % a.tiles_n; id /= a.tiles_n;
    int tap = 0;
    if (!CIN4) { tap = id % T; id /= T; }
    const int split = id;
    const int n0 = tn * BN, c0 = tc * BC;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const int S = (m_end - m_begin + 31) / 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave / WAVES_C, wave_c = wave % WAVES_C;
    const int va = tid % VA, pra = tid / VA;
    const int vb = tid % VB, prb = tid / VB;

    int dh = 0, dw = 0;
    if (!CIN4) {
        dh = a.dh0 + (tap / a.TS) * a.dh_step;
        dw = a.dw0 + (tap % a.TS) * a.dw_step;
    } else {
        const int t = c0 / 4 + vb;           // CIN4: tile columns are (tap, 4 channels)
        dh = a.dh0 + (t / a.TS) * a.dh_step;
        dw = a.dw0 + (t % a.TS) * a.dw_step;
    }
    const bool b_col_ok = CIN4 ? (c0 / 4 + vb < T) : (c0 + 4 * vb < a.Cin);
    const bool a_col_ok = n0 + 4 * va < a.N_ld;

    f32x16 acc[NT][CT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[IA], rb[IB];
    auto load_tile = [&](int s) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const int mb = m_begin + s * 32;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int m = mb + pra + RA * i;
            const bool ok = a_col_ok && m < m_end;
            ra[i] = ok ? *reinterpret_cast<const f32x4*>(a.dy + (size_t)m * a.dy_pitch + n0 + 4 * va) : zero;
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int m = mb + prb + RB * i;
            bool ok = b_col_ok && m < m_end;
            const int mm = ok ? m : 0;
            const int q = mm % a.Q, t = mm / a.Q;
            const int p = t % a.P, b = t / a.P;
            const int hi = p * a.in_sh + dh, wi = q * a.in_sw + dw;
            ok = ok && (unsigned)hi < (unsigned)a.IH && (unsigned)wi < (unsigned)a.IW;
            const size_t off = (size_t)((b * a.IH + hi) * a.IW + wi) * a.x_pitch + (CIN4 ? 0 : c0 + 4 * vb);
            rb[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + off) : zero;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < IA; ++i) *reinterpret_cast<f32x4*>(sA + buf * 32 * BN + (pra + RA * i) * BN + 4 * va) = ra[i];
#pragma unroll
        for (int i = 0; i < IB; ++i) *reinterpret_cast<f32x4*>(sB + buf * 32 * BC + (prb + RB * i) * BC + 4 * vb) = rb[i];
    };

    if (S > 0) {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int s = 0; s < S; ++s) {
            const int buf = s & 1;
            if (s + 1 < S) load_tile(s + 1);
            // A[i = cout][k = pixel], B[k = pixel][j = cin]: lane l reads pixel row k = 2*ks + (l>>5), column l&31
            const float* pa = sA + buf * 32 * BN + (lane >> 5) * BN + wave_n * WN + (lane & 31);
            const float* pb = sB + buf * 32 * BC + (lane >> 5) * BC + wave_c * WC + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                float fa[NT], fb[CT];
#pragma unroll
                for (int i = 0; i < NT; ++i) fa[i] = pa[2 * ks * BN + 32 * i];
#pragma unroll
                for (int j = 0; j < CT; ++j) fb[j] = pb[2 * ks * BC + 32 * j];
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            if (s + 1 < S) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    float* out = a.out + (size_t)split * a.slab_stride;
    const int col_limit = CIN4 ? T * 4 : a.Cin;
    const int col_base = CIN4 ? 0 : tap * a.Cin;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int c = c0 + wave_c * WC + j * 32 + (lane & 31);
        if (c >= col_limit) continue;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wave_n * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < a.N) out[(size_t)n * a.out_row_stride + col_base + c] = acc[i][j][r];
            }
        }
    }
}

// wgrad on the bf16x6 arithmetic (see gather_gemm_kernel PREC 1).  Both operands are pixel-major in memory while the
// MFMA wants 8 consecutive reduction indices (pixels) per lane, so the bf16 planes are stored pixel-major in LDS
// exactly as they are loaded ([32 pixels][channels], rows padded so that 4 consecutive rows fall in different
// bank quarters) and the fragments are fetched with the hardware transpose read ds_read_b64_tr_b16: per 16-lane
// group, lane 4q+p addresses row q / columns 4p..4p+3 and lane i receives column i of the 4 rows.
// FAST (chosen by launch_wg when OW % 32 == 0, not the thin-input mode, both tensors below 4 GiB): every 32-pixel reduction
// tile then lies inside one output row, so the whole gather geometry of a tile is wave-uniform (scalar unit, updated by
// counters) and the per-lane work per row is one add, one compare and one select; operands come through raw buffer loads
// (out-of-range lanes read zeros, no branches, no 64-bit address arithmetic).  The general path recomputes (b, p, q) per
// row with integer divisions every step, which cost more vector issue slots than the operand split itself.
// FAST 2 (any OW >= 16): the same buffer-load scheme with per-row pixel coordinates kept as counters in every lane
// (advance by 32 pixels per step with at most two row wraps) -- no divisions either, a few more vector instructions.
template <int BN, int BC, int WN, int WC, bool CIN4, int PREC, int FAST>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(const WgradArgs a) {
    constexpr int NPL = PREC == 2 ? 2 : 3;
    constexpr int WAVES_C = BC / WC;
    constexpr int NT = WN / 32, CT = WC / 32;
    constexpr int VA = BN / 4, RA = 256 / VA, IA = 32 / RA;
    constexpr int VB = BC / 4, RB = 256 / VB, IB = 32 / RB;
    constexpr int ROWA = wg_rowb(BN), ROWB = wg_rowb(BC);
    constexpr int PLA = 32 * ROWA, PLB = 32 * ROWB;          // bytes per plane
    static_assert((BN / WN) * (BC / WC) == 4, "4 waves per block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sA = reinterpret_cast<char*>(smem);               // NPL planes of dy
    char* sB = sA + NPL * PLA;                               // NPL planes of gathered x

    const int T = a.TR * a.TS;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = id % a.tiles_c; id /= a.tiles_c;
    const int tn = id % a.tiles_n; id /= a.tiles_n;
    int tap = 0;
    if (!CIN4) { tap = id % T; id /= T; }
    const int split = id;
    const int n0 = tn * BN, c0 = tc * BC;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const int S = (m_end - m_begin + 31) / 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave / WAVES_C, wave_c = wave % WAVES_C;
    const int va = tid % VA, pra = tid / VA;
    const int vb = tid % VB, prb = tid / VB;

    int dh = 0, dw = 0;
    if (!CIN4) {
        dh = a.dh0 + (tap / a.TS) * a.dh_step;
        dw = a.dw0 + (tap % a.TS) * a.dw_step;
    } else {
        const int t = c0 / 4 + vb;
        dh = a.dh0 + (t / a.TS) * a.dh_step;
        dw = a.dw0 + (t % a.TS) * a.dw_step;
    }
    const bool b_col_ok = CIN4 ? (c0 / 4 + vb < T) : (c0 + 4 * vb < a.Cin);
    const bool a_col_ok = n0 + 4 * va < a.N_ld;

    f32x16 acc[NT][CT];
    f32x16 acc_lo[PREC == 2 ? NT : 1][PREC == 2 ? CT : 1];     // f16x3: the two cross terms, carried at 2^11 x their value
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (PREC == 2) acc_lo[i][j][r] = 0.f;
            }
    float scale_a = 1.f, scale_b = 1.f;
    if constexpr (PREC == 2) {
        scale_a = a.amax_dy ? pow2_scale_for(*a.amax_dy) : 1.f;
        scale_b = a.amax_x ? pow2_scale_for(*a.amax_x) : 1.f;
    }

    f32x4 ra[IA], rb[IB];
    // FAST-path state: tile position as wave-uniform counters, per-thread constant offsets
    constexpr unsigned OOB = 0xFFFFFFF0u;                 // >= num_records: the load returns zeros
    int f_mb = m_begin, f_q0 = 0, f_p = 0, f_b = 0;
    unsigned f_va[FAST ? IA : 1], f_tx[FAST ? IB : 1];
    int f_wc[FAST ? IB : 1];
    int g_q[FAST == 2 ? IB : 1], g_p[FAST == 2 ? IB : 1], g_b[FAST == 2 ? IB : 1];      // FAST 2: per-row pixel coordinates
    __amdgpu_buffer_rsrc_t rdy, rx;
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
    if constexpr (FAST != 0) {
        rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)(unsigned)a.dy_bytes, 0x00020000);
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(unsigned)a.x_bytes, 0x00020000);
        f_q0 = m_begin % a.Q;
        const int t = m_begin / a.Q;
        f_p = t % a.P;
        f_b = t / a.P;
#pragma unroll
        for (int i = 0; i < IA; ++i) f_va[i] = a_col_ok ? ((unsigned)(pra + RA * i) * (unsigned)a.dy_pitch + n0 + 4 * va) * 4u : OOB;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            f_tx[i] = ((unsigned)((prb + RB * i) * a.in_sw) * (unsigned)a.x_pitch + c0 + 4 * vb) * 4u;
            f_wc[i] = (prb + RB * i) * a.in_sw + dw;
        }
    }
    auto load_tile = [&](int s) {
        if constexpr (FAST == 2) {
            const unsigned colb = (unsigned)(c0 + 4 * vb);
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int hi = g_p[i] * a.in_sh + dh, wi = g_q[i] * a.in_sw + dw;
                const bool ok = b_col_ok && f_mb + prb + RB * i < m_end && (unsigned)hi < (unsigned)a.IH && (unsigned)wi < (unsigned)a.IW;
                const unsigned off = (((unsigned)(g_b[i] * a.IH + hi) * (unsigned)a.IW + (unsigned)wi) * (unsigned)a.x_pitch + colb) * 4u;
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : OOB, 0, 0));
                g_q[i] += 32;
                while (g_q[i] >= a.Q) { g_q[i] -= a.Q; ++g_p[i]; }
                while (g_p[i] >= a.P) { g_p[i] -= a.P; ++g_b[i]; }
            }
            const unsigned soff = (unsigned)f_mb * (unsigned)a.dy_pitch * 4u;
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const bool ok = f_mb + pra + RA * i < m_end;
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? f_va[i] : OOB, soff, 0));
            }
            f_mb += 32;
            return;
        }
        if constexpr (FAST == 1) {
            // tile = 32 consecutive output pixels of row (f_b, f_p) starting at column f_q0
            const int hi = f_p * a.in_sh + dh;
            const bool row_ok = b_col_ok && (unsigned)hi < (unsigned)a.IH;
            const unsigned delta = (unsigned)((((f_b * a.IH + hi) * a.IW + f_q0 * a.in_sw + dw) * a.x_pitch) * 4);
            const int wq = f_q0 * a.in_sw;
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const bool ok = row_ok && (unsigned)(f_wc[i] + wq) < (unsigned)a.IW;
                // delta already contains dw: the thread constant is the row's pixel step only
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? f_tx[i] + delta : OOB, 0, 0));
            }
            const unsigned soff = (unsigned)f_mb * (unsigned)a.dy_pitch * 4u;
#pragma unroll
            for (int i = 0; i < IA; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, f_va[i], soff, 0));
            f_mb += 32;
            f_q0 += 32;
            if (f_q0 == a.Q) {
                f_q0 = 0;
                if (++f_p == a.P) { f_p = 0; ++f_b; }
            }
            return;
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const int mb = m_begin + s * 32;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int m = mb + pra + RA * i;
            const bool ok = a_col_ok && m < m_end;
            ra[i] = ok ? *reinterpret_cast<const f32x4*>(a.dy + (size_t)m * a.dy_pitch + n0 + 4 * va) : zero;
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int m = mb + prb + RB * i;
            bool ok = b_col_ok && m < m_end;
            const int mm = ok ? m : 0;
            const int q = mm % a.Q, t = mm / a.Q;
            const int p = t % a.P, b = t / a.P;
            const int hi = p * a.in_sh + dh, wi = q * a.in_sw + dw;
            ok = ok && (unsigned)hi < (unsigned)a.IH && (unsigned)wi < (unsigned)a.IW;
            const size_t off = (size_t)((b * a.IH + hi) * a.IW + wi) * a.x_pitch + (CIN4 ? 0 : c0 + 4 * vb);
            rb[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + off) : zero;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            char* d = sA + (pra + RA * i) * ROWA + 8 * va;
            uint2 q0, q1, q2;
            if constexpr (PREC == 2) {
                split2(ra[i], scale_a, q0, q1);
            } else {
                split3(ra[i], q0, q1, q2);
                *reinterpret_cast<uint2*>(d + 2 * PLA) = q2;
            }
            *reinterpret_cast<uint2*>(d) = q0;
            *reinterpret_cast<uint2*>(d + PLA) = q1;
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            char* d = sB + (prb + RB * i) * ROWB + 8 * vb;
            uint2 q0, q1, q2;
            if constexpr (PREC == 2) {
                split2(rb[i], scale_b, q0, q1);
            } else {
                split3(rb[i], q0, q1, q2);
                *reinterpret_cast<uint2*>(d + 2 * PLB) = q2;
            }
            *reinterpret_cast<uint2*>(d) = q0;
            *reinterpret_cast<uint2*>(d + PLB) = q1;
        }
    };
    // transpose-read addressing: group g = lane>>4 covers channels 16*(g&1).., reduction half h = g>>1
    const int g = lane >> 4, h = g >> 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const char* fa_base = sA + (8 * h + q4) * ROWA + 2 * (wave_n * WN + 16 * (g & 1) + 4 * p4);
    const char* fb_base = sB + (8 * h + q4) * ROWB + 2 * (wave_c * WC + 16 * (g & 1) + 4 * p4);

    if (S > 0) {
        load_tile(0);
        for (int s = 0; s < S; ++s) {
            if (s > 0) __syncthreads();
            store_tile();
            __syncthreads();
            if (s + 1 < S) load_tile(s + 1);
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
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j) {
                        if constexpr (PREC == 2) {
                            const f16x8 a0 = __builtin_bit_cast(f16x8, fa[i][0]), a1 = __builtin_bit_cast(f16x8, fa[i][1]);
                            const f16x8 b0 = __builtin_bit_cast(f16x8, fb[j][0]), b1 = __builtin_bit_cast(f16x8, fb[j][1]);
                            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc_lo[i][j], 0, 0, 0);
                            acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc_lo[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][NPL - 1], fb[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NPL - 1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                        }
                    }
            }
        }
    }

    const float unscale_a = 1.f / scale_a, unscale_b = 1.f / scale_b;
    float* out = a.out + (size_t)split * a.slab_stride;
    const int col_limit = CIN4 ? T * 4 : a.Cin;
    const int col_base = CIN4 ? 0 : tap * a.Cin;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int c = c0 + wave_c * WC + j * 32 + (lane & 31);
        if (c >= col_limit) continue;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wave_n * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float val = acc[i][j][r];
                if constexpr (PREC == 2) val = (val + acc_lo[i][j][r] * (1.f / 2048.f)) * unscale_a * unscale_b;
                if (n < a.N) out[(size_t)n * a.out_row_stride + col_base + c] = val;
            }
        }
    }
}

// dw[i] = sum_s slab[s][i] in a fixed order (deterministic).  A block covers 32 float4 columns x 8 split lanes: lane j adds
// slabs j, j+8, j+16, ... (independent loads in flight instead of one dependent chain of up to 256), then the 8 partial sums
// are combined in lane order through LDS.  The association is fixed by (splits), not by timing.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, long long n4,
                                                            int splits, long long slab_stride) {
    __shared__ f32x4 red[8][32];
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (long long base = (long long)blockIdx.x * 32; base < n4; base += (long long)gridDim.x * 32) {
        const long long i = base + col;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < n4)
            for (int k = sl; k < splits; k += 4 * 8) {          // four slabs in flight per thread (past-the-end ones re-load slab k and add 0); add order unchanged
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(slabs + (size_t)(k + 8 * u < splits ? k + 8 * u : k) * slab_stride + 4 * i);
#pragma unroll
                for (int u = 0; u < 4; ++u) { if (k + 8 * u < splits) s += v[u]; }
            }
        red[sl][col] = s;
        __syncthreads();
        if (sl == 0 && i < n4) {
#pragma unroll
            for (int j = 1; j < 8; ++j) s += red[j][col];
            *reinterpret_cast<f32x4*>(dw + 4 * i) = s;
        }
        __syncthreads();
    }
}

// [K][RS][C] -> [C][RS][Kp] with Kp = roundup4(K), zero-filled pad (dgrad reads Kp-wide vectors).
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int RS, int C, int Kp) {
    __shared__ float tile[32][33];
    const int rs = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        tile[i][tx] = (k < K && c < C) ? w[((size_t)k * RS + rs) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (c < C && k < Kp) wt[((size_t)c * RS + rs) * Kp + k] = tile[tx][i];
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int PREC>
constexpr size_t gg_smem() {
    return PREC == 0 ? (size_t)2 * (BM + BN) * LDT * sizeof(float) : (size_t)(PREC == 2 ? 2 : 3) * (BM + BN) * LDB;
}
template <int BN, int BC>
constexpr size_t wg_smem() { return (size_t)2 * 32 * (BN + BC) * sizeof(float); }
template <int BN, int BC, int PREC>
constexpr size_t wg16_smem() { return (size_t)(PREC == 2 ? 2 : 3) * 32 * (wg_rowb(BN) + wg_rowb(BC)); }

int g_big_tile = 2;      // 0: 128x128 tiles only, 1: 256x128 8-wave tile (lock-step schedule), 2: 256x128 ping-pong kernel for f16x3
int g_pp_flags = 0;
int g_wgrad_fast = 1;           // A/B knob (pylc_debug_pp_flags bit 8 clears it)
unsigned long long* g_pp_stamps = nullptr;      // debug: see pylc_debug_pp_stamps
int g_conv_precision = 2;      // 0 = fp32 MFMA, 1 = bf16x6, 2 = f16x3 (default); see pylc_set_conv_precision

static thread_local int g_last_bm = 128;      // M-tile height of the most recent gather-GEMM launch on this thread

template <int BM, int BN, int WM, int WN, bool CIN4, int PREC>
static int launch_gg(GatherGemmArgs& a, hipStream_t st) {
    PYLC_REQUIRE(a.w != nullptr, "conv: this geometry reads the fp32 filter, but only prepared planes were given");
    g_last_bm = BM;
    const int tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.N_store, BN);
    const long long grid = (long long)tiles_m * a.tiles_n;
    PYLC_REQUIRE(grid > 0 && grid < (1ll << 31), "conv grid out of range");
    const size_t lds = gg_smem<BM, BN, PREC>();
    hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, WM, WN, CIN4, PREC>), dim3((unsigned)grid), dim3((BM / WM) * (BN / WN) * 64), lds, st, a);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

// geometry / size conditions of the ping-pong kernel (on top of: f16x3 mode, stored N > 64, >= 192 tiles of 256x128)
static bool takes_pp(const GatherGemmArgs& a) {
    // (16-byte epilogue stores: pitches and stored widths are multiples of 4 floats by check_desc; the bases must be aligned too)
    const bool aligned = a.y_pitch % 4 == 0 && a.N_store % 4 == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(a.ep_res) & 15) == 0;
    return g_big_tile == 2 && aligned && a.Cin % 8 == 0 && a.x_bytes > 0 && a.x_bytes < (1ll << 31) && a.w_bytes > 0 && a.w_bytes < (1ll << 31);
}

static int launch_gg_pp(GatherGemmArgs& a, hipStream_t st) {
    g_last_bm = PP_BM;
    const int tiles_m = cdiv(a.M, PP_BM);
    a.tiles_n = cdiv(a.N_store, PP_BN);
    const long long n_tiles = (long long)tiles_m * a.tiles_n;
    PYLC_REQUIRE(n_tiles > 0 && n_tiles < (1ll << 31), "conv grid out of range");
    a.n_tiles = (int)n_tiles;
    // Reduction order: taps innermost when neighbouring taps read overlapping pixels (dilation <= 2: 30 % less fetch traffic
    // per launch over the training step, time-neutral); widely dilated taps (ASPP) share nothing, there the channel chunks stay
    // innermost (measured 3 % faster on the d = 12 shape)
    if (a.dh_step > 2 || a.dh_step < -2 || a.dw_step > 2 || a.dw_step < -2) a.dbg_flags |= 4096;
    // One block per tile.  The kernel can also run as persistent blocks (fewer blocks than tiles: each walks a strided
    // share), which is +0-6 % on short-K shapes in isolation -- but a static share per block is fragile when the wgrad
    // stream holds some CUs: late-starting blocks then finish their whole share late (measured: a 317 -> 177 tiles/s outlier).
    const long long grid = (g_pp_flags & 64) ? (n_tiles < kNumCU ? n_tiles : kNumCU) : n_tiles;
    const bool swz = !(g_pp_flags & 128);        // bit 7 (128): padded 80-byte LDS rows, two stages (A/B)
    const bool m16 = swz && !(g_pp_flags & 256); // bit 8 (256): 32x32x16 instead of 16x16x32 MFMAs (A/B)
    const bool bpl = a.w_planes != nullptr;
    const dim3 g((unsigned)grid), b(512);
#define PYLC_PP(ST, BP, SW, M) hipLaunchKernelGGL((gather_gemm_pp_kernel<ST, BP, SW, M>), g, b, ((SW) ? 3 * PP_STAGE_SWZ : 2 * PP_STAGE) + ((ST) ? 4096 : 0), st, a)
    if (a.ep_scale != nullptr) {                 // fused inference epilogue: its own instantiations of the default variant
        if (g_conv_precision == 3) {             // precision mode 3: one fp16 plane per operand, one MFMA per product
            if (bpl) hipLaunchKernelGGL((gather_gemm_pp_kernel<false, true, true, true, true, true, true>), g, b, 3 * PP_STAGE_SWZ, st, a);
            else hipLaunchKernelGGL((gather_gemm_pp_kernel<false, false, true, true, true, true, true>), g, b, 3 * PP_STAGE_SWZ, st, a);
        } else if (bpl) hipLaunchKernelGGL((gather_gemm_pp_kernel<false, true, true, true, true, true>), g, b, 3 * PP_STAGE_SWZ, st, a);
        else hipLaunchKernelGGL((gather_gemm_pp_kernel<false, false, true, true, true, true>), g, b, 3 * PP_STAGE_SWZ, st, a);
    } else if (a.dbg != nullptr) {               // stamped builds (tools/pp_stamps.py)
        if (bpl && m16) PYLC_PP(true, true, true, true);
        else if (bpl && swz) PYLC_PP(true, true, true, false);
        else if (bpl) PYLC_PP(true, true, false, false);
        else PYLC_PP(true, false, false, false);
    } else if (bpl) {
        if (m16) PYLC_PP(false, true, true, true);
        else if (swz) PYLC_PP(false, true, true, false);
        else PYLC_PP(false, true, false, false);
    } else {
        if (m16) PYLC_PP(false, false, true, true);
        else if (swz) PYLC_PP(false, false, true, false);
        else PYLC_PP(false, false, false, false);
    }
#undef PYLC_PP
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

template <int PREC>
static int dispatch_gg_p(GatherGemmArgs& a, bool cin4, hipStream_t st) {
    if (cin4) {
        if (a.N_store <= 32) return launch_gg<128, 32, 32, 32, true, PREC>(a, st);
        if (a.N_store <= 64) return launch_gg<256, 64, 64, 64, true, PREC>(a, st);
        return launch_gg<128, 128, 64, 64, true, PREC>(a, st);
    }
    if (a.N_store <= 32) return launch_gg<128, 32, 32, 32, false, PREC>(a, st);
    if (a.N_store <= 64) return launch_gg<256, 64, 64, 64, false, PREC>(a, st);
    if (PREC != 0 && g_big_tile && ((long long)cdiv(a.M, 256) * cdiv(a.N_store, 128) >= 192 || (g_pp_flags & 1024))) {
        if (PREC == 2 && takes_pp(a)) return launch_gg_pp(a, st);
        return launch_gg<256, 128, 64, 64, false, PREC>(a, st);      // 8 waves: halves LDS-write bytes per MFMA
    }
    return launch_gg<128, 128, 64, 64, false, PREC>(a, st);
}

static int dispatch_gg(GatherGemmArgs& a, bool cin4, hipStream_t st) {
    if (a.x_planes != nullptr) {       // A operand pre-split by its producer: conv_pl.hip (no fp32 view of x exists)
        PYLC_REQUIRE(g_conv_precision >= 2 && !cin4, "conv: fp16-plane operands need precision mode 2 or 3 and a dense (not thin-input) geometry");
        a.nterms = g_conv_precision == 3 ? 1 : 3;
        return launch_gg_pl(a, st);
    }
    // mode 3 (plain fp16 operands) exists only for plane operands; fp32 operands (stem, ranges unknown to the producer) run f16x3
    return g_conv_precision == 0 ? dispatch_gg_p<0>(a, cin4, st) : g_conv_precision >= 2 ? dispatch_gg_p<2>(a, cin4, st) : dispatch_gg_p<1>(a, cin4, st);
}

template <typename K>
static hipError_t opt_in_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int conv_init() {
    if (int rc = conv_pl_init()) return rc;
    if (int rc = conv_stem_init()) return rc;
    if (int rc = wgrad_pl_init()) return rc;
#define PYLC_OPT_GG(BM, BN, WM, WN)                                                                           \
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<BM, BN, WM, WN, false, 0>, gg_smem<BM, BN, 0>()));                 \
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<BM, BN, WM, WN, true, 0>, gg_smem<BM, BN, 0>()));                  \
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<BM, BN, WM, WN, false, 1>, gg_smem<BM, BN, 1>()));                 \
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<BM, BN, WM, WN, true, 1>, gg_smem<BM, BN, 1>()));                  \
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<BM, BN, WM, WN, false, 2>, gg_smem<BM, BN, 2>()));                 \
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<BM, BN, WM, WN, true, 2>, gg_smem<BM, BN, 2>()));
    PYLC_OPT_GG(128, 128, 64, 64)
    PYLC_OPT_GG(256, 64, 64, 64)
    PYLC_OPT_GG(128, 32, 32, 32)
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<256, 128, 64, 64, false, 1>, gg_smem<256, 128, 1>()));
    PYLC_HIP(opt_in_lds(gather_gemm_kernel<256, 128, 64, 64, false, 2>, gg_smem<256, 128, 2>()));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, false, false, false>), 2 * PP_STAGE));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, true, false, false>), 2 * PP_STAGE));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, false, true, true>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, false, true, true, true, true>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, true, true, true, true, true>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, false, true, true, true, true, true>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, true, true, true, true, true, true>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, true, true, true>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, true, true, false>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<false, false, true, false>), 3 * PP_STAGE_SWZ));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<true, false, false, false>), 2 * PP_STAGE + 4096));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<true, true, false, false>), 2 * PP_STAGE + 4096));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<true, true, true, false>), 3 * PP_STAGE_SWZ + 4096));
    PYLC_HIP(opt_in_lds((gather_gemm_pp_kernel<true, true, true, true>), 3 * PP_STAGE_SWZ + 4096));
#undef PYLC_OPT_GG
    PYLC_HIP(opt_in_lds(wgrad_kernel<128, 128, 64, 64, false>, wg_smem<128, 128>()));
    PYLC_HIP(opt_in_lds(wgrad_kernel<64, 64, 32, 32, false>, wg_smem<64, 64>()));
    PYLC_HIP(opt_in_lds(wgrad_kernel<32, 128, 32, 32, false>, wg_smem<32, 128>()));
    PYLC_HIP(opt_in_lds(wgrad_kernel<64, 64, 32, 32, true>, wg_smem<64, 64>()));
#define PYLC_OPT_WG(P)                                                                                                \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<128, 128, 64, 64, false, P, 0>, wg16_smem<128, 128, P>()));              \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<64, 64, 32, 32, false, P, 0>, wg16_smem<64, 64, P>()));                  \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<32, 128, 32, 32, false, P, 0>, wg16_smem<32, 128, P>()));                \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<128, 128, 64, 64, false, P, 1>, wg16_smem<128, 128, P>()));              \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<64, 64, 32, 32, false, P, 1>, wg16_smem<64, 64, P>()));                  \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<32, 128, 32, 32, false, P, 1>, wg16_smem<32, 128, P>()));                \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<128, 128, 64, 64, false, P, 2>, wg16_smem<128, 128, P>()));              \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<64, 64, 32, 32, false, P, 2>, wg16_smem<64, 64, P>()));                  \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<32, 128, 32, 32, false, P, 2>, wg16_smem<32, 128, P>()));                \
    PYLC_HIP(opt_in_lds(wgrad_split_kernel<64, 64, 32, 32, true, P, 0>, wg16_smem<64, 64, P>()));
    PYLC_OPT_WG(1)
    PYLC_OPT_WG(2)
#undef PYLC_OPT_WG
    return PYLC_OK;
}

static int check_desc(const PylcConvDesc* d) {
    PYLC_REQUIRE(d != nullptr, "null conv descriptor");
    PYLC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "non-positive conv dims");
    PYLC_REQUIRE(d->R > 0 && d->S > 0 && d->R * d->S <= 64, "kernel window %dx%d unsupported (<= 64 taps)", d->R, d->S);
    PYLC_REQUIRE(d->stride == 1 || d->stride == 2, "stride %d unsupported (1 or 2)", d->stride);
    PYLC_REQUIRE(d->dil >= 1 && d->pad >= 0, "bad dilation/padding");
    PYLC_REQUIRE(d->Cin % 4 == 0, "Cin=%d must be a multiple of 4 (pad images with pylc_image_pack)", d->Cin);
    PYLC_REQUIRE(d->x_pitch >= d->Cin && d->x_pitch % 4 == 0, "x_pitch=%d invalid for Cin=%d", d->x_pitch, d->Cin);
    PYLC_REQUIRE(d->y_pitch >= d->Cout && d->y_pitch % 4 == 0, "y_pitch=%d invalid for Cout=%d", d->y_pitch, d->Cout);
    const int oh = (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
    const int ow = (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
    PYLC_REQUIRE(oh == d->OH && ow == d->OW && oh > 0 && ow > 0, "OH/OW (%d,%d) inconsistent with geometry (%d,%d)", d->OH, d->OW, oh, ow);
    PYLC_REQUIRE((long long)d->B * d->H * d->W * d->x_pitch < (1ll << 31) && (long long)d->B * d->OH * d->OW * d->y_pitch < (1ll << 31),
                 "tensor exceeds 2^31 elements");
    return PYLC_OK;
}

static inline int roundup4(int v) { return (v + 3) & ~3; }

}  // namespace pylc

using namespace pylc;

extern "C" int pylc_set_conv_precision(int mode) {
    PYLC_REQUIRE(mode >= 0 && mode <= 3, "conv precision mode must be 0 (fp32 MFMA), 1 (bf16x6), 2 (f16x3) or 3 (fp16 operands, fp32 accumulation)");
    g_conv_precision = mode;
    return PYLC_OK;
}

extern "C" int pylc_get_conv_precision(void) { return g_conv_precision; }

// profiling aid (tools/pp_stamps.py): the next forward convs that take the ping-pong kernel record per-segment clock
// stamps of block 0 (waves 0 and 4) into `buf` (2 x 256 uint64, device memory); NULL switches it off
extern "C" int pylc_debug_pp_stamps(unsigned long long* buf) { g_pp_stamps = buf; return PYLC_OK; }
extern "C" int pylc_debug_pp_flags(int flags) { g_pp_flags = flags; g_wgrad_fast = !(flags & 8); return PYLC_OK; }

// tuning knob (tools/conv_bench.py): allow / forbid the 256x128 8-wave tile
extern "C" int pylc_debug_set_big_tile(int on) { g_big_tile = on; return PYLC_OK; }

namespace {
struct FwdEpilogue {
    const float* scale; const float* shift; const float* res; unsigned* amax; int relu;
    const PylcFwdEp* ex = nullptr;          // pylc_conv2d_fwd_bnact_ex: plane residual / plane output / true input range
};
}
static int conv2d_fwd_impl(const PylcConvDesc* d, const float* x, const float* w, const float* bias, float* y, float* stats, int* stats_rows,
                           void* stream, const FwdEpilogue* ep = nullptr);

extern "C" int pylc_conv2d_fwd(const PylcConvDesc* d, const float* x, const float* w, const float* bias, float* y, void* stream) {
    return conv2d_fwd_impl(d, x, w, bias, y, nullptr, nullptr, stream);
}

// Inference: y = act(conv(x, w) * scale + shift (+ residual)) in the conv epilogue -- eval-mode BatchNorm (scale / shift
// from pylc_bn_eval_coeffs), the residual add and the ReLU without a separate pass over y.  Same expression and order as
// pylc_bn_apply, so the result is bit-identical to conv followed by bn_apply.
extern "C" int pylc_conv2d_fwd_bnact(const PylcConvDesc* d, const float* x, const float* w, const float* bias, const float* scale,
                                     const float* shift, const float* residual, int relu, float* y, unsigned int* amax_out, void* stream) {
    PYLC_REQUIRE(scale && shift, "conv2d_fwd_bnact: null scale / shift");
    const FwdEpilogue ep{scale, shift, residual, amax_out, relu};
    return conv2d_fwd_impl(d, x, w, bias, y, nullptr, nullptr, stream, &ep);
}

// ... with fp16-plane tensors on either side (PylcFwdEp): x as planes (d->x_fmt = 1), the residual as fp32 or planes, y as fp32
// (d->out_fmt = 0), one fp16 plane (1, precision mode 3) or two planes (2, f16x3).  A plane output is scaled with the bound the kernel forms
// from device scalars -- Cin R S x TRUE max|x| x max|w| x max|scale| + max|shift| + max|residual| -- and writes to *d->out_bound; the true
// maximum of what it stores is max-accumulated into ep->amax_out (zero-initialised by the caller) and is what the NEXT layer's bound
// starts from, so that the looseness of the bounds does not compound through an eval-mode network, which has no batch statistics to
// re-anchor them (DESIGN.md "inference on plane tensors").
extern "C" int pylc_conv2d_fwd_bnact_ex(const PylcConvDesc* d, const void* x, const float* w, const float* bias, const PylcFwdEp* ep, void* y,
                                        void* stream) {
    PYLC_REQUIRE(ep != nullptr && ep->scale && ep->shift, "conv2d_fwd_bnact_ex: null epilogue / scale / shift");
    PYLC_REQUIRE(d->x_fmt == 1, "conv2d_fwd_bnact_ex: x must be an fp16-plane tensor (pylc_conv2d_fwd_bnact takes fp32 tensors)");
    PYLC_REQUIRE(ep->res_fmt >= 0 && ep->res_fmt <= 2 && (ep->res_fmt == 0 || (ep->residual && ep->res_scale_bound)),
                 "conv2d_fwd_bnact_ex: res_fmt 1 / 2 needs the residual planes and the bound they were scaled with");
    PYLC_REQUIRE(d->out_fmt == 0 || (d->out_bound && ep->scale_amax && ep->shift_amax && (ep->residual == nullptr || ep->res_amax)),
                 "conv2d_fwd_bnact_ex: a plane output needs out_bound, max|scale|, max|shift| and (with a residual) max|residual|");
    const FwdEpilogue fe{ep->scale, ep->shift, ep->res_fmt == 0 ? static_cast<const float*>(ep->residual) : nullptr, ep->amax_out, ep->relu, ep};
    return conv2d_fwd_impl(d, static_cast<const float*>(x), w, bias, static_cast<float*>(y), nullptr, nullptr, stream, &fe);
}

extern "C" size_t pylc_conv2d_fwd_stats_floats(const PylcConvDesc* d) {
    if (check_desc(d)) return 0;
    const long long M = (long long)d->B * d->OH * d->OW;
    return (size_t)cdiv<long long>(M, 128) * 2 * (size_t)roundup4(d->Cout);      // smallest M tile is 128 rows
}

extern "C" int pylc_conv2d_fwd_stats(const PylcConvDesc* d, const float* x, const float* w, const float* bias, float* y, float* stats_partial,
                                     int* stats_rows, void* stream) {
    PYLC_REQUIRE(stats_partial && stats_rows, "conv2d_fwd_stats: null statistics buffer");
    return conv2d_fwd_impl(d, x, w, bias, y, stats_partial, stats_rows, stream);
}

static int conv2d_fwd_impl(const PylcConvDesc* d, const float* x, const float* w, const float* bias, float* y, float* stats, int* stats_rows,
                           void* stream, const FwdEpilogue* ep) {
    if (int rc = check_desc(d)) return rc;
    PYLC_REQUIRE(x && y && (w || (d->x_fmt == 1 && d->w_planes)), "null pointer");
    GatherGemmArgs a{};
    if (ep != nullptr) {
        a.ep_scale = ep->scale; a.ep_shift = ep->shift; a.ep_res = ep->res; a.ep_amax = ep->amax; a.ep_relu = ep->relu;
        // (decided here, not in the kernel: the same test on the kernel's side cost the R101 inference 2 %, tools/eval_ab.py same box)
        a.ep_vec_ok = ((reinterpret_cast<uintptr_t>(ep->scale) | reinterpret_cast<uintptr_t>(ep->shift) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0;
    }
    a.stats = stats;
    a.dbg = g_pp_stamps;
    a.dbg_flags = g_pp_flags;
    PYLC_REQUIRE(g_conv_precision < 2 || (d->x_amax && d->w_amax), "f16x3 / fp16 mode: conv2d_fwd needs x_amax and w_amax in the descriptor");
    a.amax_x = d->x_amax; a.amax_w = d->w_amax;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.x_bytes = (((long long)d->B * d->H * d->W - 1) * d->x_pitch + d->Cin) * 4;
    a.w_bytes = (long long)d->Cout * d->R * d->S * d->Cin * 4;
    a.w_planes = d->w_planes; a.w_plane_stride = (long long)d->Cout * d->R * d->S * d->Cin;
    a.w_il = (d->w_planes != nullptr && (d->w_planes_fmt & 1)) ? 1 : 0;
    if (d->x_fmt == 1) {                     // x points at plane 0 of an fp16-plane tensor
        a.x_planes = x; a.x = nullptr;
        a.x_plane_stride = planes_stride_rule((long long)d->B * d->H * d->W, d->Cin, d->x_pitch, (g_conv_precision == 3 ? 1 : 2));
        a.x_bytes /= 2;                      // bytes of ONE plane
    }
    a.P = d->OH; a.Q = d->OW; a.M = d->B * d->OH * d->OW;
    a.IH = d->H; a.IW = d->W; a.Cin = d->Cin; a.x_pitch = d->x_pitch;
    a.in_sh = a.in_sw = d->stride;
    a.TR = d->R; a.TS = d->S;
    a.dh0 = -d->pad; a.dh_step = d->dil; a.dw0 = -d->pad; a.dw_step = d->dil;
    a.w_off0 = 0; a.w_step_r = d->S * d->Cin; a.w_step_s = d->Cin; a.w_row_stride = d->R * d->S * d->Cin;
    a.N = d->Cout; a.N_store = roundup4(d->Cout) <= d->y_pitch ? roundup4(d->Cout) : d->Cout;
    a.OH = d->OH; a.OW = d->OW; a.out_sh = a.out_sw = 1; a.oh0 = a.ow0 = 0; a.y_pitch = d->y_pitch;
    a.accumulate = 0;
    if (ep != nullptr && ep->ex != nullptr) {
        const PylcFwdEp* ex = ep->ex;
        a.bound_x = ex->x_true_amax;
        a.ep_scale_amax = ex->scale_amax; a.ep_shift_amax = ex->shift_amax;
        if (ex->residual != nullptr) {
            a.ep_res = static_cast<const float*>(ex->residual);
            a.ep_res_fmt = ex->res_fmt; a.ep_res_scale = ex->res_scale_bound; a.ep_res_amax = ex->res_amax;
            a.ep_res_plane_stride = planes_stride_rule((long long)d->B * d->OH * d->OW, d->Cout, d->Cout, ex->res_fmt == 2 ? 2 : 1);
        }
        if (d->out_fmt != 0) {
            PYLC_REQUIRE((d->out_fmt == 1 || d->out_fmt == 2) && d->y_pitch == d->Cout && d->Cout % 4 == 0,
                         "conv2d_fwd_bnact_ex: a plane output (out_fmt 1 / 2) needs a dense y with Cout %% 4 == 0");
            a.out_half = d->out_fmt == 1; a.out_planes2 = d->out_fmt == 2;
            a.out_plane_stride = planes_stride_rule((long long)d->B * d->OH * d->OW, d->Cout, d->y_pitch, d->out_fmt == 2 ? 2 : 1);
            a.out_bound_k = (float)(d->Cin * d->R * d->S); a.out_bound = d->out_bound;
        }
    } else if (d->out_fmt == 1) {
        PYLC_REQUIRE(d->x_fmt == 1 && d->out_bound && bias == nullptr && ep == nullptr && d->y_pitch == d->Cout && d->Cout % 4 == 0,
                     "conv2d_fwd: a one-plane fp16 output needs fp16-plane input, out_bound, no bias / fused epilogue and a dense y");
        a.out_half = 1; a.out_bound_k = (float)(d->Cin * d->R * d->S); a.out_bound = d->out_bound;
    }
    const bool cin4 = d->Cin == 4 && d->R * d->S > 1;
    if (cin4 && ep == nullptr && takes_stem(a)) {
        if (int rc = launch_stem_fwd(a, as_stream(stream))) return rc;
    } else if (int rc = dispatch_gg(a, cin4, as_stream(stream))) return rc;
    if (stats_rows) *stats_rows = a.halo_tiles_m > 0 ? a.halo_tiles_m : cdiv(a.M, a.x_planes != nullptr ? a.tile_bm : (a.tiles_n > 0 ? g_last_bm : 128));
    return PYLC_OK;
}

// 1 when pylc_conv2d_dgrad(d, ...) will read the fp32 transposed filter (w_crsk), 0 when the prepared planes in
// d->w_planes_t are all it touches (f16x3 mode, stride 1, a geometry that dispatches to the ping-pong kernel)
extern "C" int pylc_conv2d_dgrad_needs_f32_weights(const PylcConvDesc* d) {
    if (check_desc(d)) return 1;
    if (g_conv_precision < 2 || d->w_planes_t == nullptr) return 1;
    if (d->dy_fmt == 1) return 0;            // plane operands: conv_pl.hip reads prepared planes only (any stride)
    if (d->stride != 1) return 1;
    const int Kp = roundup4(d->Cout);
    GatherGemmArgs a{};
    a.Cin = Kp;
    a.x_bytes = (((long long)d->B * d->OH * d->OW - 1) * d->y_pitch + Kp) * 4;
    a.w_bytes = (long long)d->Cin * d->R * d->S * Kp * 4;
    const long long M = (long long)d->B * d->H * d->W;
    const bool big = d->Cin > 64 && g_big_tile && cdiv<long long>(M, 256) * cdiv(d->Cin, 128) >= 192;
    return (big && takes_pp(a)) ? 0 : 1;
}

extern "C" int pylc_conv2d_dgrad_add(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                                     const float* add_src, const void* add_mask, void* stream);
extern "C" int pylc_conv2d_dgrad(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate, void* stream) {
    return pylc_conv2d_dgrad_add(d, dy, w_crsk, dx, accumulate, nullptr, nullptr, stream);
}

// one implementation behind pylc_conv2d_dgrad, _dgrad_add and (EXPERIMENTAL builds) _dgrad_bn
struct BnBackArgs { const float* y; const float* mean; const float* invstd; const float* scale; const float* shift; const void* relu_mask; int relu; unsigned int* g_amax; };
static int conv2d_dgrad_impl(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                             const float* add_src, const void* add_mask, const BnBackArgs* bn, float* sums_partial, int* sums_rows, void* stream);

extern "C" int pylc_conv2d_dgrad_add(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                                     const float* add_src, const void* add_mask, void* stream) {
    return conv2d_dgrad_impl(d, dy, w_crsk, dx, accumulate, add_src, add_mask, nullptr, nullptr, nullptr, stream);
}

#ifdef PYLC_EXPERIMENTAL
extern "C" size_t pylc_conv2d_dgrad_bn_floats(const PylcConvDesc* d) {
    if (check_desc(d)) return 0;
    const long long M = (long long)d->B * d->H * d->W;
    return (size_t)cdiv<long long>(M, 128) * 2 * (size_t)d->Cin;          // smallest M tile is 128 rows
}

extern "C" int pylc_conv2d_dgrad_bn(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                                    const float* add_src, const void* add_mask, const PylcBnBack* bn, float* sums_partial, int* sums_rows,
                                    void* stream) {
    if (bn == nullptr) return conv2d_dgrad_impl(d, dy, w_crsk, dx, accumulate, add_src, add_mask, nullptr, sums_partial, sums_rows, stream);
    const BnBackArgs b{bn->y, bn->mean, bn->invstd, bn->scale, bn->shift, bn->relu_mask, bn->relu, bn->g_amax};
    return conv2d_dgrad_impl(d, dy, w_crsk, dx, accumulate, add_src, add_mask, &b, sums_partial, sums_rows, stream);
}
#endif

static int conv2d_dgrad_impl(const PylcConvDesc* d, const float* dy, const float* w_crsk, float* dx, int accumulate,
                             const float* add_src, const void* add_mask, const BnBackArgs* bn, float* sums_partial, int* sums_rows, void* stream) {
    if (int rc = check_desc(d)) return rc;
    PYLC_REQUIRE(bn == nullptr || (d->dy_fmt == 1 && d->stride == 1 && d->x_pitch == d->Cin && d->Cin % 8 == 0 && sums_partial && sums_rows &&
                                   bn->y && bn->mean && bn->invstd && (!bn->relu || bn->relu_mask || (bn->scale && bn->shift))),
                 "conv2d_dgrad_bn: the BatchNorm-backward sums need fp16-plane dy, stride 1, a dense dx with Cin %% 8 == 0, a partials buffer, "
                 "y / mean / invstd and (with a ReLU) scale + shift or the 1-bit mask");
    PYLC_REQUIRE(dy && dx, "null pointer");
    PYLC_REQUIRE(add_src == nullptr || (d->dy_fmt == 1 && d->stride == 1 && !accumulate && d->x_pitch == d->Cin),
                 "conv2d_dgrad_add: the masked residual source needs fp16-plane dy, stride 1, a dense dx and accumulate == 0");
    PYLC_REQUIRE(add_mask == nullptr || add_src != nullptr, "conv2d_dgrad_add: add_mask without add_src");
    PYLC_REQUIRE(w_crsk || (d->w_planes_t && !pylc_conv2d_dgrad_needs_f32_weights(d)),
                 "conv2d_dgrad: this geometry needs the fp32 transposed filter (pylc_conv2d_dgrad_needs_f32_weights)");
    hipStream_t st = as_stream(stream);
    const int Kp = roundup4(d->Cout);       // reduction runs over output channels, padded to 4 (zero weights / zero dy)
    PYLC_REQUIRE(Kp <= d->y_pitch, "dy pitch %d must cover roundup4(Cout)=%d", d->y_pitch, Kp);
    GatherGemmArgs a{};
    a.dbg_flags = g_pp_flags;
    PYLC_REQUIRE(g_conv_precision < 2 || (d->dy_amax && d->w_amax), "f16x3 / fp16 mode: conv2d_dgrad needs dy_amax and w_amax in the descriptor");
    a.amax_x = d->dy_amax; a.amax_w = d->w_amax;
    a.x = dy; a.w = w_crsk; a.bias = nullptr; a.y = dx;
    a.x_bytes = (((long long)d->B * d->OH * d->OW - 1) * d->y_pitch + Kp) * 4;
    a.w_bytes = (long long)d->Cin * d->R * d->S * Kp * 4;
    a.w_planes = d->w_planes_t; a.w_plane_stride = (long long)d->Cin * d->R * d->S * Kp;
    a.w_il = (d->w_planes_t != nullptr && (d->w_planes_fmt & 2)) ? 1 : 0;
    if (d->dy_fmt == 1) {
        a.x_planes = dy; a.x = nullptr;
        a.x_plane_stride = planes_stride_rule((long long)d->B * d->OH * d->OW, Kp, d->y_pitch, (g_conv_precision == 3 ? 1 : 2));
        a.x_bytes /= 2;
    }
    a.IH = d->OH; a.IW = d->OW; a.Cin = Kp; a.x_pitch = d->y_pitch;
    a.N = d->Cin; a.N_store = d->Cin;
    a.OH = d->H; a.OW = d->W; a.y_pitch = d->x_pitch;
    a.w_row_stride = d->R * d->S * Kp;
    a.accumulate = accumulate;
    a.add_src = add_src;
    a.add_mask = static_cast<const unsigned char*>(add_mask);
    if (d->out_fmt == 1) {
        PYLC_REQUIRE(d->dy_fmt == 1 && d->stride == 1 && d->out_bound && !accumulate && add_src == nullptr && bn == nullptr && d->x_pitch == d->Cin && d->Cin % 4 == 0,
                     "conv2d_dgrad: a one-plane fp16 output needs fp16-plane dy, stride 1, out_bound, a dense dx, no accumulation / residual source / BatchNorm sums");
        a.out_half = 1; a.out_bound_k = (float)(Kp * d->R * d->S); a.out_bound = d->out_bound;
    }
    if (bn != nullptr) {
        a.bn_y = bn->y; a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bn_scale = bn->scale; a.bn_shift = bn->shift;
        a.bn_mask = static_cast<const unsigned char*>(bn->relu_mask); a.bn_relu = bn->relu; a.bn_gmax = bn->g_amax;
        a.stats = sums_partial;
    }
    if (d->stride == 1) {
        a.P = d->H; a.Q = d->W; a.M = d->B * d->H * d->W;
        a.in_sh = a.in_sw = 1;
        a.TR = d->R; a.TS = d->S;
        a.dh0 = d->pad; a.dh_step = -d->dil; a.dw0 = d->pad; a.dw_step = -d->dil;     // ho = hi + pad - r*dil
        a.w_off0 = 0; a.w_step_r = d->S * Kp; a.w_step_s = Kp;
        a.out_sh = a.out_sw = 1; a.oh0 = a.ow0 = 0;
        if (int rc = dispatch_gg(a, false, st)) return rc;
        if (sums_rows) *sums_rows = a.halo_tiles_m > 0 ? a.halo_tiles_m : cdiv(a.M, a.tile_bm);
        return PYLC_OK;
    }
    // stride 2: dx pixels of parity class (ph, pw) receive only taps with (ph + pad - r*dil) even.
    // Each class is a dense gather-GEMM over its own tap progression; classes with no taps are zero.
    bool need_zero = false;
    int cnt_r[2] = {0, 0}, first_r[2] = {-1, -1}, cnt_s[2] = {0, 0}, first_s[2] = {-1, -1};
    for (int ph = 0; ph < 2; ++ph) {
        for (int r = 0; r < d->R; ++r)
            if (((ph + d->pad - r * d->dil) & 1) == 0) { if (first_r[ph] < 0) first_r[ph] = r; ++cnt_r[ph]; }
        for (int s = 0; s < d->S; ++s)
            if (((ph + d->pad - s * d->dil) & 1) == 0) { if (first_s[ph] < 0) first_s[ph] = s; ++cnt_s[ph]; }
    }
    const int step = (d->dil & 1) ? 2 : 1;      // valid taps are every 2nd (odd dilation) or all/none (even dilation)
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw)
            if (cnt_r[ph] == 0 || cnt_s[pw] == 0) need_zero = true;
    PYLC_REQUIRE(!(need_zero && !accumulate) || d->x_pitch == d->Cin, "strided dgrad with empty parity classes needs a dense dx");
    if (need_zero && !accumulate)
        PYLC_HIP(hipMemsetAsync(dx, 0, (size_t)d->B * d->H * d->W * d->x_pitch * sizeof(float), st));
    for (int ph = 0; ph < 2; ++ph) {
        for (int pw = 0; pw < 2; ++pw) {
            if (cnt_r[ph] == 0 || cnt_s[pw] == 0) continue;
            GatherGemmArgs c = a;
            c.P = (d->H - ph + 1) / 2; c.Q = (d->W - pw + 1) / 2;
            if (c.P <= 0 || c.Q <= 0) continue;
            c.M = d->B * c.P * c.Q;
            c.in_sh = c.in_sw = 1;
            c.TR = cnt_r[ph]; c.TS = cnt_s[pw];
            // ho = (2*hq + ph + pad - r*dil)/2 = hq + (ph + pad - r*dil)/2, r = first + step*i
            c.dh0 = (ph + d->pad - first_r[ph] * d->dil) / 2; c.dh_step = -(step * d->dil) / 2;
            c.dw0 = (pw + d->pad - first_s[pw] * d->dil) / 2; c.dw_step = -(step * d->dil) / 2;
            c.w_off0 = (first_r[ph] * d->S + first_s[pw]) * Kp; c.w_step_r = step * d->S * Kp; c.w_step_s = step * Kp;
            c.out_sh = c.out_sw = 2; c.oh0 = ph; c.ow0 = pw;
            if (int rc = dispatch_gg(c, false, st)) return rc;
        }
    }
    return PYLC_OK;
}

namespace {
struct WgradPlan { int cfg; int tiles_n, tiles_c, splits, m_per_split; long long slab; bool cin4; };
// cfg 0: 128x128, 1: 64x64, 2: 32(cout)x128(cin), 3: cin4 64x64
int g_wg_max_steps = 256;       // longest reduction (K-steps of 32 pixels) of one block of a multi-tap wgrad (pylc_debug_wgrad_max_steps; 0: no cap)
extern "C" int pylc_debug_wgrad_max_steps(int steps) { g_wg_max_steps = steps; return PYLC_OK; }

WgradPlan plan_wgrad(const PylcConvDesc* d) {
    WgradPlan p{};
    const int T = d->R * d->S;
    const long long M = (long long)d->B * d->OH * d->OW;
    p.cin4 = d->Cin == 4 && T > 1;
    int bn, bc;
    if (p.cin4) { p.cfg = 3; bn = 64; bc = 64; }
    else if (d->Cout <= 32) { p.cfg = 2; bn = 32; bc = 128; }
    else if (d->Cout <= 64 || d->Cin <= 64) { p.cfg = 1; bn = 64; bc = 64; }
    else { p.cfg = 0; bn = 128; bc = 128; }
    p.tiles_n = cdiv(d->Cout, bn);
    p.tiles_c = p.cin4 ? cdiv(T * 4, bc) : cdiv(d->Cin, bc);
    const long long tiles = (long long)p.tiles_n * p.tiles_c * (p.cin4 ? 1 : T);
    // A fixed number of blocks is resident at once (LDS-limited) and all blocks of a launch do equal work: size the split
    // so that tiles * splits fills whole rounds of resident blocks (792 blocks on 512 slots run as 512 + 280 = 2 rounds
    // at 77 %), with at least 16 K-steps (512 pixels) per block.
    const long long slots = (p.cfg == 0 ? 2 : 4) * kNumCU;      // resident blocks: 61 KB LDS (128x128 tile) vs 37 KB
    long long max_splits = cdiv<long long>(M, 512);
    if (max_splits > 256) max_splits = 256;
    if (max_splits < 1) max_splits = 1;
    // Multi-tap filters: the 9 x tiles_n x tiles_c blocks of a split re-read the same pixel rows and only the XCD's 4 MB L2 makes that
    // one HBM fetch -- which holds while the blocks stay within a few tens of K-steps of each other.  Blocks that run for a thousand
    // K-steps drift apart (measured, tools/wgrad_traffic.py: the decoder's 256 -> 256 3x3 at 128^2 x 32 with 14 splits of 1170 steps
    // fetched 4.3 GB for 1.07 GB of operands); blocks of at most g_wg_max_steps K-steps start together round after round and stay together.
    long long min_splits = 1;
    if (T > 1 && !p.cin4 && g_wg_max_steps > 0) {
        min_splits = cdiv<long long>(M, 32ll * g_wg_max_steps);
        if (min_splits > max_splits) min_splits = max_splits;
    }
    long long s = min_splits;
    double best = -1.0;
    for (long long c = min_splits; c <= max_splits; ++c) {
        const long long blocks = tiles * c;
        const double eff = (double)blocks / (double)(cdiv<long long>(blocks, slots) * slots);
        if (eff > best + 0.04) { best = eff; s = c; }      // prefer the smallest split within 4 % of the best fill
    }
    long long mps = cdiv<long long>(cdiv<long long>(M, s), 32) * 32;
    s = cdiv<long long>(M, mps);
    p.splits = (int)s; p.m_per_split = (int)mps;
    p.slab = (long long)d->Cout * T * d->Cin;
    return p;
}
}  // namespace

extern "C" size_t pylc_conv2d_wgrad_workspace(const PylcConvDesc* d) {
    if (check_desc(d)) return 0;
    WgradPlan p = plan_wgrad(d);
    return p.splits > 1 ? (size_t)p.splits * p.slab * sizeof(float) : 0;
}

template <int BN, int BC, int WN, int WC, bool CIN4>
static int launch_wg(WgradArgs& a, long long grid, hipStream_t st) {
    PYLC_REQUIRE(grid > 0 && grid < (1ll << 31), "wgrad grid out of range");
    const bool buf_ok = !CIN4 && g_wgrad_fast && a.x_bytes < 0xFFFFFFF0ll && a.dy_bytes < 0xFFFFFFF0ll;
    const int fast = !buf_ok ? 0 : (a.Q % 32 == 0 ? 1 : (a.Q >= 16 ? 2 : 0));
#define PYLC_LAUNCH_WG(P)                                                                                                             \
    {                                                                                                                                  \
        const size_t lds16 = wg16_smem<BN, BC, P>();                                                                                   \
        if (fast == 1) hipLaunchKernelGGL((wgrad_split_kernel<BN, BC, WN, WC, CIN4, P, CIN4 ? 0 : 1>), dim3((unsigned)grid), dim3(256), lds16, st, a); \
        else if (fast == 2) hipLaunchKernelGGL((wgrad_split_kernel<BN, BC, WN, WC, CIN4, P, CIN4 ? 0 : 2>), dim3((unsigned)grid), dim3(256), lds16, st, a); \
        else hipLaunchKernelGGL((wgrad_split_kernel<BN, BC, WN, WC, CIN4, P, 0>), dim3((unsigned)grid), dim3(256), lds16, st, a);      \
        PYLC_LAUNCH_CHECK();                                                                                                           \
        return PYLC_OK;                                                                                                                \
    }
    if (g_conv_precision == 1) PYLC_LAUNCH_WG(1)
    if (g_conv_precision >= 2) PYLC_LAUNCH_WG(2)
#undef PYLC_LAUNCH_WG
    const size_t lds = wg_smem<BN, BC>();
    hipLaunchKernelGGL((wgrad_kernel<BN, BC, WN, WC, CIN4>), dim3((unsigned)grid), dim3(256), lds, st, a);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

// The split-K slab sums of MANY wgrads in one launch (pylc_splitk_reduce_batch): tile t of the launch is 32 float4 columns of the entry whose
// tile range [tile_prefix[e], tile_prefix[e + 1]) holds t; per column the arithmetic is splitk_reduce_kernel's (same association: bit-identical).
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const PylcSlabSum* __restrict__ tab, const long long* __restrict__ tile_prefix, int n,
                                                                  long long total_tiles) {
    __shared__ f32x4 red[8][32];
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (long long t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        int lo = 0, hi = n;                     // largest e with tile_prefix[e] <= t
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (tile_prefix[mid] <= t) lo = mid; else hi = mid;
        }
        const PylcSlabSum e = tab[lo];
        const long long i = (t - tile_prefix[lo]) * 32 + col;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < e.n4)
            for (int k = sl; k < e.splits; k += 4 * 8) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(e.slabs + (size_t)(k + 8 * u < e.splits ? k + 8 * u : k) * e.slab_stride + 4 * i);
#pragma unroll
                for (int u = 0; u < 4; ++u) { if (k + 8 * u < e.splits) s += v[u]; }
            }
        red[sl][col] = s;
        __syncthreads();
        if (sl == 0 && i < e.n4) {
#pragma unroll
            for (int j = 1; j < 8; ++j) s += red[j][col];
            *reinterpret_cast<f32x4*>(e.dw + 4 * i) = s;
        }
        __syncthreads();
    }
}

extern "C" int pylc_splitk_reduce_batch(const PylcSlabSum* table_dev, const long long* tile_prefix_dev, int n, long long total_tiles, void* stream) {
    PYLC_REQUIRE(table_dev && tile_prefix_dev && n > 0 && total_tiles > 0, "splitk_reduce_batch: bad arguments");
    const int blocks = (int)(total_tiles < 4096 ? total_tiles : 4096);
    hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), table_dev, tile_prefix_dev, n, total_tiles);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}

static int wgrad_impl(const PylcConvDesc* d, const float* x, const float* dy, float* dw, float* dbias,
                      void* workspace, size_t workspace_bytes, PylcSlabSum* pending, void* stream);

extern "C" int pylc_conv2d_wgrad(const PylcConvDesc* d, const float* x, const float* dy, float* dw, float* dbias,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_impl(d, x, dy, dw, dbias, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int pylc_conv2d_wgrad_slabs(const PylcConvDesc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                                       PylcSlabSum* pending, void* stream) {
    PYLC_REQUIRE(pending != nullptr, "conv2d_wgrad_slabs: null `pending`");
    return wgrad_impl(d, x, dy, dw, nullptr, workspace, workspace_bytes, pending, stream);
}

static int wgrad_impl(const PylcConvDesc* d, const float* x, const float* dy, float* dw, float* dbias,
                      void* workspace, size_t workspace_bytes, PylcSlabSum* pending, void* stream) {
    if (int rc = check_desc(d)) return rc;
    PYLC_REQUIRE(x && dy && dw, "null pointer");
    PYLC_REQUIRE(dbias == nullptr, "dbias: use pylc_bn_stats on dy (column sums)");
    PYLC_REQUIRE(d->x_fmt == d->dy_fmt, "conv2d_wgrad: x and dy must be in the same format (both fp32 or both fp16 planes; pylc_to_planes converts)");
    hipStream_t st = as_stream(stream);
    const WgradPlan p = plan_wgrad(d);
    const size_t need = p.splits > 1 ? (size_t)p.splits * p.slab * sizeof(float) : 0;
    if (need > workspace_bytes || (need && !workspace))
        return fail(PYLC_ERR_WORKSPACE, "wgrad workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    PYLC_REQUIRE(p.slab % 4 == 0, "dw size must be a multiple of 4");
    const int T = d->R * d->S;
    WgradArgs a{};
    a.x = x; a.dy = dy; a.out = p.splits > 1 ? static_cast<float*>(workspace) : dw;
    a.P = d->OH; a.Q = d->OW; a.M = d->B * d->OH * d->OW;
    a.IH = d->H; a.IW = d->W; a.Cin = d->Cin; a.x_pitch = d->x_pitch; a.in_sh = a.in_sw = d->stride;
    a.TR = d->R; a.TS = d->S; a.dh0 = -d->pad; a.dh_step = d->dil; a.dw0 = -d->pad; a.dw_step = d->dil;
    a.N = d->Cout; a.N_ld = roundup4(d->Cout); a.dy_pitch = d->y_pitch;
    PYLC_REQUIRE(a.N_ld <= d->y_pitch, "dy pitch must cover roundup4(Cout)");
    a.out_row_stride = T * d->Cin;
    a.tiles_n = p.tiles_n; a.tiles_c = p.tiles_c; a.splits = p.splits; a.m_per_split = p.m_per_split;
    a.slab_stride = p.slab;
    PYLC_REQUIRE(g_conv_precision < 2 || (d->dy_amax && d->x_amax), "f16x3 / fp16 mode: conv2d_wgrad needs dy_amax and x_amax in the descriptor");
    a.x_bytes = (((long long)d->B * d->H * d->W - 1) * d->x_pitch + d->Cin) * 4;
    a.dy_bytes = (((long long)d->B * d->OH * d->OW - 1) * d->y_pitch + a.N_ld) * 4;
    a.amax_dy = d->dy_amax; a.amax_x = d->x_amax;
    const long long grid = (long long)p.tiles_n * p.tiles_c * (p.cin4 ? 1 : T) * p.splits;
    int rc;
    if (d->x_fmt == 1) {           // both operands pre-split by their producers: wgrad_pl.hip
        PYLC_REQUIRE(g_conv_precision >= 2 && !p.cin4 && d->Cout % 8 == 0, "conv2d_wgrad: fp16-plane operands need precision mode 2 or 3, "
                     "a dense geometry and Cout %% 8 == 0");
        a.x_planes = x; a.dy_planes = dy; a.x = nullptr; a.dy = nullptr;
        a.x_plane_stride = planes_stride_rule((long long)d->B * d->H * d->W, d->Cin, d->x_pitch, (g_conv_precision == 3 ? 1 : 2));
        a.dy_plane_stride = planes_stride_rule((long long)d->B * d->OH * d->OW, d->Cout, d->y_pitch, (g_conv_precision == 3 ? 1 : 2));
        a.N_ld = d->Cout;
        a.x_bytes /= 2; a.dy_bytes = (((long long)d->B * d->OH * d->OW - 1) * d->y_pitch + a.N_ld) * 2;
        a.nterms = g_conv_precision == 3 ? 1 : 3;
        rc = launch_wg_pl(a, p.cfg, grid, st);
    } else
    switch (p.cfg) {
        case 0: rc = launch_wg<128, 128, 64, 64, false>(a, grid, st); break;
        case 1: rc = launch_wg<64, 64, 32, 32, false>(a, grid, st); break;
        case 2: rc = launch_wg<32, 128, 32, 32, false>(a, grid, st); break;
        default: rc = launch_wg<64, 64, 32, 32, true>(a, grid, st); break;
    }
    if (rc) return rc;
    if (pending != nullptr) {          // the caller sums the slabs later (pylc_splitk_reduce_batch); splits 0 = dw is complete
        *pending = PylcSlabSum{p.splits > 1 ? static_cast<const float*>(workspace) : nullptr, dw, p.slab / 4, p.slab, p.splits > 1 ? p.splits : 0, 0};
        return PYLC_OK;
    }
    if (p.splits > 1) {
        const long long n4 = p.slab / 4;
        const int blocks = (int)(cdiv<long long>(n4, 32) < 4096 ? cdiv<long long>(n4, 32) : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(workspace), dw, n4,
                           p.splits, p.slab);
        PYLC_LAUNCH_CHECK();
    }
    return PYLC_OK;
}

extern "C" int pylc_weight_transpose(const float* w, float* wt, int K, int RS, int C, void* stream) {
    PYLC_REQUIRE(w && wt && K > 0 && RS > 0 && C > 0, "bad weight_transpose arguments");
    const int Kp = roundup4(K);
    dim3 grid(cdiv(C, 32), cdiv(Kp, 32), RS);
    hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(256), 0, as_stream(stream), w, wt, K, RS, C, Kp);
    PYLC_LAUNCH_CHECK();
    return PYLC_OK;
}
