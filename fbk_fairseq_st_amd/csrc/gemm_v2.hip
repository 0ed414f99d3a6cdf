// GEMM v2: the forward/NT shape  C[M,N] = epi(A[M,K] . W[N,K]^T)  with bf16 operands, for the big projections.
//
// Why a second kernel: PMC + ablations on gemm_fast_kernel (profiles/README.md) show the k-loop limited by the LDS
// store side of register staging (8 x ds_write_b128 per thread per k-tile cost more LDS-issue cycles than the 16
// fragment reads) and by one-deep cover of the global-load latency.  Here
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no VGPRs, no ds_write), 1 KiB per wave-instruction;
//     the XOR swizzle that keeps ds_read_b128 conflict-free is applied on the per-lane SOURCE address because the LDS
//     destination of the DMA is lane-linear (cdna_hip_programming.md, rule 21);
//   * a 3-stage LDS ring keeps two k-tiles in flight; a counted s_waitcnt vmcnt(6) + raw s_barrier per k-tile
//     (never vmcnt(0) in the loop, never __syncthreads());
//   * tile 256 x 128 with 8 waves (4 x 2, 64 x 64 each): one workgroup per CU, two waves per SIMD.
// Epilogue = the shared LDS-staged vector epilogue (gemm_epilogue.hpp).
#include "common.hpp"
#include "gemm_epilogue.hpp"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

namespace {
constexpr int V2_BM = 256, V2_BN = 128, V2_BK = 64, V2_NTH = 512;
constexpr int V2_STAGE = (V2_BM + V2_BN) * 128;          // 48 KiB per stage
constexpr int V2_NSTAGE = 3;
constexpr int V2_LOADS = 6;                               // DMA instructions per wave per k-tile (4 for A, 2 for B)
}

template <typename TO>
__global__ __launch_bounds__(512, 2) void gemm_v2_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + V2_BN - 1) / V2_BN, tiles_m = (p.M + V2_BM - 1) / V2_BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = wg / tiles_n, tn = wg % tiles_n;
    const int row0 = tm * V2_BM, col0 = tn * V2_BN;
    const int nk = p.K / V2_BK;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably wave-uniform (LDS DMA base -> M0)
    const int wr = wave >> 1, wc = wave & 1, r16 = lane & 15, q = lane >> 4;

    // per-lane source pointers of the 6 DMA pieces (8 rows x 128 B each): row = 8*piece + lane/8, the 16-byte chunk
    // fetched by lane l is (l&7) ^ (l>>3) so that LDS position l holds chunk (l&7)^(row&7) of its row
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    const int sub = lane >> 3, chunk = (lane & 7) ^ sub;
    const bf16* ga[4];
    const bf16* gb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = min(row0 + 8 * (wave * 4 + i) + sub, p.M - 1);
        ga[i] = A + (size_t)r * p.lda + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = min(col0 + 8 * (wave * 2 + i) + sub, p.N - 1);
        gb[i] = B + (size_t)r * p.ldb + chunk * 8;
    }
    auto issue = [&](int stage) {
        char* sA = smem + stage * V2_STAGE + (wave * 4) * 1024;
        char* sB = smem + stage * V2_STAGE + V2_BM * 128 + (wave * 2) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void*)ga[i], (lds_void*)(sA + i * 1024), 16, 0, 0);
            ga[i] += V2_BK;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void*)gb[i], (lds_void*)(sB + i * 1024), 16, 0, 0);
            gb[i] += V2_BK;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (nk > 1) { issue(1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int st = 0;                                            // ring position of k-tile t
    for (int t = 0; t < nk; ++t) {
        if (t + 2 < nk) issue(st >= 1 ? st - 1 : 2);      // (st + 2) % 3: that stage was last read in compute(t-1)
        const char* la = smem + st * V2_STAGE + (wr * 64) * 128;
        const char* lb = smem + st * V2_STAGE + V2_BM * 128 + (wc * 64) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fa[4], fb[4];
            const int sw = ((4 * s + q) ^ (r16 & 7)) << 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(la + (16 * i + r16) * 128 + sw);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const u32x4*>(lb + (16 * j + r16) * 128 + sw);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma16<bf16>(fb[j], fa[i], acc[i][j]);
        }
        // k-tile t+1 must have landed (all but the newest 6 DMA of this wave), my LDS reads must have returned
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        st = (st == 2) ? 0 : st + 1;
    }
    gemm_epilogue<TO, V2_BM, V2_BN, 4, 4, V2_NTH>(p, acc, smem, row0, col0, wr * 64, wc * 64, q, r16);
}

// returns 1 if launched, 0 if the shape is not eligible, < 0 on error
int s2t_gemm_v2_try(const GemmArgs& a, int out_dtype, hipStream_t st) {
    // measured (tools/gemm_ab.py, bench.py A/B): on the K = 512..2048 shapes of this model the ring kernel ties the
    // 2-deep register-prefetch kernel (both are bound by the epilogue + per-tile prologue, not the k-loop), so it is
    // opt-in until the epilogue overlaps the next tile; set S2T_GEMM_V2=1 to use it
    static const bool on = getenv("S2T_GEMM_V2") != nullptr;
    if (!on) return 0;
    if (a.mapA || a.mapB || a.mapC || a.splitk > 1) return 0;
    if (a.K % V2_BK || a.K < V2_BK || (a.lda % 8) || (a.ldb % 8) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return 0;
    if (a.M < 2048 || a.N < 128) return 0;                 // small problems: the 128/64-wide kernels fill the chip better
    const int tiles = ((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN);
    size_t lds = (size_t)V2_NSTAGE * V2_STAGE;
    const size_t epi = (size_t)V2_BM * (V2_BN * 4 + 16);
    if (epi > lds) lds = epi;
    if (out_dtype == S2T_BF16) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v2_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL(gemm_v2_kernel<bf16>, dim3(tiles), dim3(V2_NTH), lds, st, a);
    } else return 0;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : S2T_EHIP(e);
}
