// Probe (not product code): the 256 x 256 x 64 tile of gemm256 partitioned as FOUR waves of 128 x 128 (one per SIMD, 256 accumulator
// registers each) instead of eight of 128 x 64 -- does the main loop get closer to the MFMA rate when every fragment read from LDS
// feeds eight MFMAs instead of four and no second wave competes for the SIMD's issue slot?  NT form, bf16, plain stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fbk_fairseq_st_amd/csrc -I include -o tools/_bin/gemm4w_probe tools/gemm4w_probe.hip
//   tools/_bin/gemm4w_probe [M N K reps]
// LDS: two K-tile buffers of 64 KiB = A-h0 | A-h1 | B-h0 | B-h1, a half = 128 rows x 128 B (64 k), chunks XOR-swizzled with row & 7 on
// the DMA's source side; wave (wr, wc) reads A-h[wr] and B-h[wc].  Per K-tile and wave: 128 MFMAs in two k-halves of 64 (8 x 8
// tiles), 32 ds_read_b128, 16 LDS-DMA instructions, ONE barrier:
//     P0  MFMA k-half 0 | reads k-half 1 of this K-tile                       ; wait reads + DMA(t+1) ; barrier
//     P1  MFMA k-half 1 | DMA K-tile t+2 into this buffer ; reads k-half 0 of K-tile t+1 from the other buffer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#include "common.hpp"

typedef __attribute__((address_space(3))) void lds_void;
#ifndef X_MASK
#define X_MASK 0        // bit 0: no barriers, bit 1: no DMA in the loop, bit 2: no fragment reads in the loop (timing only, wrong results)
#endif

constexpr int HALF = 16384, BUFB = 65536, BK = 64;

__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C,
                                                        int M, int N, int K, int lda, int ldb, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = N / 256;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (tile / tiles_n) * 256, col0 = (tile % tiles_n) * 256;
    const int nk = K / BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1, r16 = lane & 15, q = lane >> 4;

    // staging: wave w fills pieces 4w .. 4w+3 (1 KiB = 8 rows) of each of the four halves
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, (int)((size_t)M * lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B), 0, (int)((size_t)N * ldb * 2), 0x00020000);
    const int prow = lane >> 3, pos = lane & 7;
    uint32_t voA[2], voB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        voA[h] = (uint32_t)(((size_t)(row0 + 128 * h + 32 * wave + prow) * lda + ((pos ^ prow) << 3)) * 2);
        voB[h] = (uint32_t)(((size_t)(col0 + 128 * h + 32 * wave + prow) * ldb + ((pos ^ prow) << 3)) * 2);
    }
    const uint32_t pstepA = 8u * lda * 2u, pstepB = 8u * ldb * 2u;
    auto stage = [&](int t, int which) {      // which: 0..15 = (half-kind, piece); one DMA instruction
        char* base = smem + (t & 1) * BUFB + wave * 4096;
        const uint32_t ko = (uint32_t)t * (BK * 2);
        const int kind = which >> 2, i = which & 3;           // kind 0 A-h0, 1 A-h1, 2 B-h0, 3 B-h1
        char* dst = base + kind * HALF + i * 1024;
        if (kind < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)dst, 16, voA[kind], ko + i * pstepA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)dst, 16, voB[kind - 2], ko + i * pstepB, 0, 0);
    };

    f32x4 acc[8][8];
    u32x4 fa[2][8], fb[2][8];                    // [k-half set][tile]
#ifdef VALU_PER_ROW
    float dummy[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) dummy[v] = (float)(lane + v);
#endif
    const int fr_off = r16 * 128;
    const int swz[2] = {((0 + q) ^ (r16 & 7)) << 4, ((4 + q) ^ (r16 & 7)) << 4};
    auto readA = [&](int t, int s, int i) {
        fa[s][i] = *reinterpret_cast<const u32x4*>(smem + (t & 1) * BUFB + wr * HALF + i * 2048 + fr_off + swz[s]);
    };
    auto readB = [&](int t, int s, int j) {
        fb[s][j] = *reinterpret_cast<const u32x4*>(smem + (t & 1) * BUFB + 2 * HALF + wc * HALF + j * 2048 + fr_off + swz[s]);
    };
#define BAR() do { __builtin_amdgcn_sched_barrier(0); if (!(X_MASK & 1)) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

    // prologue
#pragma unroll
    for (int w = 0; w < 16; ++w) stage(0, w);
#pragma unroll
    for (int w = 0; w < 16; ++w) stage(1, w);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) { readA(0, 0, i); readB(0, 0, i); }
    __builtin_amdgcn_s_waitcnt(0xC07F);

    auto ktile = [&](int t, auto first_tag, auto last_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // ---- P0: k-half 0 (set 0) | reads of k-half 1 into set 1
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!(X_MASK & 4)) { readA(t, 1, i); readB(t, 1, i); }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = mma16<bf16>(fb[0][j], fa[0][i], FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j]);
#ifdef VALU_PER_ROW
            // stand-in for an epilogue step hidden under the MFMAs: VALU_PER_ROW fmas on 8 independent chains per 8-MFMA row
#pragma unroll
            for (int v = 0; v < VALU_PER_ROW; ++v) dummy[v & 7] = __builtin_fmaf(dummy[v & 7], 1.0001f, 0.5f);
#endif
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);       // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 DS read
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0), in the form hipcc's counter model sees
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // K-tile t+1 (issued in P1 of K-tile t-1) has landed
        BAR();
        // ---- P1: k-half 1 (set 1) | DMA of K-tile t+2 into this buffer, reads of K-tile t+1's k-half 0 into set 0
        // past the last K-tile the DMA re-stages K-tile nk-1 into the buffer just read and the reads take the other buffer as it is:
        // in bounds, never used (a persistent version stages the next tile's first K-tiles there) -- P1 stays one basic block
        const int t2 = min(t + 2, nk - 1);
#pragma unroll
        for (int g = 0; g < 16; ++g) {                               // source order = the order asked of the scheduler: DMA, read, 4 MFMAs
            const int i = g >> 1, j0 = 4 * (g & 1);
            if (!(X_MASK & 2)) {
                char* base = smem + (t & 1) * BUFB + wave * 4096;
                const uint32_t ko = (uint32_t)t2 * (BK * 2);
                const int kind = g >> 2, pi = g & 3;
                char* dst = base + kind * HALF + pi * 1024;
                if (kind < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)dst, 16, voA[kind], ko + pi * pstepA, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)dst, 16, voB[kind - 2], ko + pi * pstepB, 0, 0);
            }
            if (!(X_MASK & 4)) { if (g & 1) readB(t + 1, 0, i); else readA(t + 1, 0, i); }
#pragma unroll
            for (int j = j0; j < j0 + 4; ++j) acc[i][j] = mma16<bf16>(fb[1][j], fa[1][i], acc[i][j]);
#ifdef VALU_PER_ROW
#pragma unroll
            for (int v = 0; v < VALU_PER_ROW / 2; ++v) dummy[v & 7] = __builtin_fmaf(dummy[v & 7], 1.0001f, 0.5f);
#endif
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // 1 VMEM read (the DMA)
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);       // 4 MFMA
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
    };
    ktile(0, std::true_type{}, std::false_type{});
    for (int t = 1; t < nk; ++t) ktile(t, std::false_type{}, std::false_type{});

#ifdef VALU_PER_ROW
    { float sd = 0.f;
#pragma unroll
      for (int v = 0; v < 8; ++v) sd += dummy[v];
      if (sd == 123.456f) C[0] = from_f32<bf16>(sd); }
#endif
    // plain epilogue: lane holds C[row0 + 128 wr + 16 i + r16][col0 + 128 wc + 16 j + 4 q .. + 3]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16 o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = from_f32<bf16>(acc[i][j][e]);
            *reinterpret_cast<u32x2*>(C + (size_t)(row0 + 128 * wr + 16 * i + r16) * ldc + col0 + 128 * wc + 16 * j + 4 * q) = *reinterpret_cast<const u32x2*>(o);
        }
}


typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- the same tile, LDS image, DMA and barrier structure with v_mfma_f32_32x32x16_bf16: 4 x 4 tiles of 32 x 32 per wave, four k-steps of
// 16 per K-tile (16 MFMAs each), the fragments of the next k-step in flight under the MFMAs of the current one.
//     steps 0-2  MFMA k-step s | reads k-step s+1 of this K-tile          ; after step 2: own reads returned, own DMA(t+1) landed ; barrier
//     step 3     MFMA k-step 3 | DMA K-tile t+2 into this buffer ; reads k-step 0 of K-tile t+1 from the other buffer
__global__ __launch_bounds__(256, 1) void gemm4w32_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C,
                                                          int M, int N, int K, int lda, int ldb, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = N / 256;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (tile / tiles_n) * 256, col0 = (tile % tiles_n) * 256;
    const int nk = K / BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1, r32 = lane & 31, hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, (int)((size_t)M * lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B), 0, (int)((size_t)N * ldb * 2), 0x00020000);
    const int prow = lane >> 3, pos = lane & 7;
    uint32_t voA[2], voB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        voA[h] = (uint32_t)(((size_t)(row0 + 128 * h + 32 * wave + prow) * lda + ((pos ^ prow) << 3)) * 2);
        voB[h] = (uint32_t)(((size_t)(col0 + 128 * h + 32 * wave + prow) * ldb + ((pos ^ prow) << 3)) * 2);
    }
    const uint32_t pstepA = 8u * lda * 2u, pstepB = 8u * ldb * 2u;
    auto stage = [&](int t, int which) {
        char* base = smem + (t & 1) * BUFB + wave * 4096;
        const uint32_t ko = (uint32_t)t * (BK * 2);
        const int kind = which >> 2, i = which & 3;
        char* dst = base + kind * HALF + i * 1024;
        if (kind < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)dst, 16, voA[kind], ko + i * pstepA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)dst, 16, voB[kind - 2], ko + i * pstepB, 0, 0);
    };
    f32x16 acc[4][4];
    u32x4 fa[2][4], fb[2][4];                    // [set][tile]: k-step s uses set s & 1
    const int fr_off = r32 * 128;
    int swz[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) swz[s] = ((2 * s + hi) ^ (r32 & 7)) << 4;
    auto readA = [&](int t, int s, int i) { fa[s & 1][i] = *reinterpret_cast<const u32x4*>(smem + (t & 1) * BUFB + wr * HALF + i * 4096 + fr_off + swz[s]); };
    auto readB = [&](int t, int s, int j) { fb[s & 1][j] = *reinterpret_cast<const u32x4*>(smem + (t & 1) * BUFB + 2 * HALF + wc * HALF + j * 4096 + fr_off + swz[s]); };
#pragma unroll
    for (int w = 0; w < 16; ++w) stage(0, w);
#pragma unroll
    for (int w = 0; w < 16; ++w) stage(1, w);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) { readA(0, 0, i); readB(0, 0, i); }
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int t = 0; t < nk; ++t) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {                                // group g: one read, two MFMAs
                if (!(X_MASK & 4)) { if (g & 1) readB(t, s + 1, g >> 1); else readA(t, s + 1, g >> 1); }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int i = g >> 1, j = 2 * (g & 1) + m;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[s & 1][j]), __builtin_bit_cast(bf16x8, fa[s & 1][i]), acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BAR();
        const int t2 = min(t + 2, nk - 1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {                                    // step 3: two DMAs, one read, two MFMAs per group
            if (!(X_MASK & 2)) {
                char* base = smem + (t & 1) * BUFB + wave * 4096;
                const uint32_t ko = (uint32_t)t2 * (BK * 2);
#pragma unroll
                for (int w = 2 * g; w < 2 * g + 2; ++w) {
                    const int kind = w >> 2, pi = w & 3;
                    char* dst = base + kind * HALF + pi * 1024;
                    if (kind < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)dst, 16, voA[kind], ko + pi * pstepA, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)dst, 16, voB[kind - 2], ko + pi * pstepB, 0, 0);
                }
            }
            if (!(X_MASK & 4)) { if (g & 1) readB(t + 1, 0, g >> 1); else readA(t + 1, 0, g >> 1); }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int i = g >> 1, j = 2 * (g & 1) + m;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[1][j]), __builtin_bit_cast(bf16x8, fa[1][i]), acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    // lane holds, for tile (i, j): output row 32 i + r32, columns 32 j + 8 g + 4 hi + 0..3 (g = 0..3) as acc[i][j][4 g + 0..3]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16 o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = from_f32<bf16>(acc[i][j][4 * g + e]);
                *reinterpret_cast<u32x2*>(C + (size_t)(row0 + 128 * wr + 32 * i + r32) * ldc + col0 + 128 * wc + 32 * j + 8 * g + 4 * hi) = *reinterpret_cast<const u32x2*>(o);
            }
}

// MFMA-only loops of the two bf16 shapes with the same flops per wave and K-tile (128 x 128 x 64): which one does the chip clock higher?
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void mfma_only_kernel(float* out, int nk) {
    const int lane = threadIdx.x & 63;
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i}; b[i] = u32x4{0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u + i, 0x3f803f80u}; }
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nk; ++t) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = mma16<bf16>(b[j], a[i], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int t = 0; t < nk; ++t) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j + 4 * (s & 1)]), __builtin_bit_cast(bf16x8, a[i + 4 * (s >> 1)]), acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][15];
    }
    if (sum == 123.456f) out[threadIdx.x] = sum;
}

__global__ void ref_kernel(const bf16* A, const bf16* B, float* R, int N, int K, int lda, int ldb, const int* rows, int nrows) {
    const int r = rows[blockIdx.x];
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += to_f32(A[(size_t)r * lda + k]) * to_f32(B[(size_t)n * ldb + k]);
        R[(size_t)blockIdx.x * N + n] = s;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 24064, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 2048;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    if (M % 256 || N % 256 || K % 64 || K < 128) { printf("M, N multiples of 256, K of 64\n"); return 1; }
    std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd();
    bf16 *A, *B, *C; float* R; int* rows;
    CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, (size_t)M * N * 2));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUFB));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUFB));
    const bool k32 = getenv("SHAPE32") != nullptr;
    const int tiles = (M / 256) * (N / 256);
    hipLaunchKernelGGL(k32 ? gemm4w32_kernel : gemm4w_kernel, dim3(tiles), dim3(256), 2 * BUFB, 0, A, B, C, M, N, K, K, K, N);
    CK(hipDeviceSynchronize());
    // check 64 rows spread over the tiles
    const int nr = 64; std::vector<int> hr(nr);
    for (int i = 0; i < nr; ++i) hr[i] = (int)(((long)i * 7919 * 131) % M);
    CK(hipMalloc(&rows, nr * 4)); CK(hipMalloc(&R, (size_t)nr * N * 4));
    CK(hipMemcpy(rows, hr.data(), nr * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3(nr), dim3(256), 0, 0, A, B, R, N, K, K, K, rows, nr);
    std::vector<float> href((size_t)nr * N); std::vector<uint16_t> hc((size_t)M * N);
    CK(hipMemcpy(href.data(), R, href.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int i = 0; i < nr; ++i)
        for (int n = 0; n < N; ++n) {
            uint32_t u = (uint32_t)hc[(size_t)hr[i] * N + n] << 16; float c; memcpy(&c, &u, 4);
            worst = fmax(worst, fabs(c - href[(size_t)i * N + n])); scale = fmax(scale, fabs(href[(size_t)i * N + n]));
        }
    printf("check: worst |C - ref| = %.4g on values up to %.4g (%s)\n", worst, scale, worst <= scale * 0.01 + 1e-3 ? "ok" : (X_MASK ? "expected: X_MASK" : "WRONG"));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k32 ? gemm4w32_kernel : gemm4w_kernel, dim3(tiles), dim3(256), 2 * BUFB, 0, A, B, C, M, N, K, K, K, N);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k32 ? gemm4w32_kernel : gemm4w_kernel, dim3(tiles), dim3(256), 2 * BUFB, 0, A, B, C, M, N, K, K, K, N);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / us * 1e-6;
    printf("gemm4w %s X=%d  %d x %d x %d: %.1f us  %.0f TFLOP/s  (%d tiles; per tile at that duration: %.0f%% of one CU's MFMA peak)\n", k32 ? "32x32x16" : "16x16x32", X_MASK, M, N, K, us, tf, tiles,
           100.0 * (2.0 * 256 * 256 * K / (us * 1e-6)) / (2.5e15 / 256));
    {   // MFMA-only shape comparison: 256 workgroups x 4 waves, nk K-tiles of 128 x 128 x 64 per wave
        const int nkk = K / 64 * ((M / 256) * (N / 256) + 255) / 256;          // the K-tiles one CU would run for this product
        float* dummy; CK(hipMalloc(&dummy, 4096));
        for (int shape = 16; shape <= 32; shape += 16) {
            auto launch = [&]() {
                if (shape == 16) hipLaunchKernelGGL(mfma_only_kernel<16>, dim3(256), dim3(256), 0, 0, dummy, nkk);
                else hipLaunchKernelGGL(mfma_only_kernel<32>, dim3(256), dim3(256), 0, 0, dummy, nkk);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us2 = ms * 1e3 / reps, tf2 = 2.0 * 256 * 256 * 64 * (double)nkk * 256 / us2 * 1e-6;
            printf("MFMA only, %s, 256 CUs x %d K-tiles: %.1f us  %.0f TFLOP/s = %.3f of 2,500\n", shape == 16 ? "16x16x32" : "32x32x16", nkk, us2, tf2, tf2 / 2500.0);
        }
    }
    return 0;
}
