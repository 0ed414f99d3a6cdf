// Probe of an ACTIVATION-STATIONARY bf16 GEMM for the K = 512 products (Y = X . W^T + b, X [M][512], W [N][512]):
//   * a workgroup = 4 waves, ONE per SIMD (512 registers per lane); a wave keeps 64 rows of X -- all 512 of their k -- in 256
//     accumulator-file registers a[0:255] as v_mfma_f32_32x32x16_bf16 B-operand fragments, loaded once;
//   * the weights stream through a four-slot LDS ring by LDS-DMA, one slot = 32 weight rows x 1,024 B (a 32-column block of the output);
//     a W fragment (one ds_read_b128) feeds two MFMAs (the wave's two 32-row tiles);
//   * per column block a wave runs ONE chain of 32 MFMAs per 32 x 32 accumulator (16 registers): no tile boundary -- the previous
//     block's accumulators (a second set) are converted, lane-swapped and stored in the gaps of the running chain, so the output
//     leaves as a steady trickle instead of a burst;
//   * one s_barrier per column block (2,048 MFMA cycles), DMA two to three blocks ahead, counted vmcnt.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fbk_fairseq_st_amd/csrc tools/lab/ars_probe.hip -o tools/_bin/ars_probe
// Run:   tools/_bin/ars_probe [M N parts reps]
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
#include <type_traits>

typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

static constexpr int KD = 512;             // the reduction length this kernel is built for
static constexpr int SLOT = 32768;         // one ring slot: 32 weight rows x 1,024 B
static constexpr int RING = 4 * SLOT;
static constexpr int BIAS_OFF = RING;      // f32 bias of this workgroup's column range (<= 2,048 columns)
static constexpr int LDS_BYTES = RING + 32768;      // 160 KiB: the ring + the bias of up to 8,192 columns

template <int... I, typename F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

struct ArsSeg { int rb, cb0, cb1, pad; };     // one piece of a workgroup's work: row block rb (256 rows), column blocks [cb0, cb1) of 32
static constexpr int MAXSEG = 3;
struct ArsArgs { const bf16* X; const bf16* W; bf16* C; const float* bias; int M, N, ldx, ldw, ldc; const ArsSeg* segs; unsigned long long* dbg; };
#ifdef ARS_STAMPS
#define ARS_RT(I_) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 16 + (I_)] = t_; } while (0)
#define ARS_T(I_) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 16 + (I_)] = t_; } while (0)
#else
#define ARS_T(I_)
#define ARS_RT(I_)
#endif

// the activation fragments live in a[0:255] by convention: the compiler is told they are clobbered (so it allocates them and
// keeps nothing of its own there); every MFMA names its fragment by number
#define A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define ARS_RESERVE() asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A8(1), A8(2), A8(3), A8(4), A8(5), A8(6), \
    A8(7), A8(8), A8(9), A8(10), A8(11), A8(12), A8(13), A8(14), A8(15), A8(16), A8(17), A8(18), A8(19), A8(20), A8(21), A8(22), A8(23), \
    A8(24), "a250", "a251", "a252", "a253", "a254", "a255")

template <int LO> __device__ __forceinline__ void mfma_acc(f32x16& acc, const u32x4& w) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(w), "n"(LO), "n"(LO + 3));
}
template <int LO> __device__ __forceinline__ void mfma_first(f32x16& acc, const u32x4& w, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %4" : "=&v"(acc) : "v"(w), "n"(LO), "n"(LO + 3), "v"(c));
}
template <int LO> __device__ __forceinline__ void mfma16_acc(f32x4& acc, const u32x4& w) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(w), "n"(LO), "n"(LO + 3));
}
template <int LO> __device__ __forceinline__ void mfma16_first(f32x4& acc, const u32x4& w, const f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%2:%3], %4" : "=&v"(acc) : "v"(w), "n"(LO), "n"(LO + 3), "v"(c));
}
__device__ __forceinline__ void swap16(uint32_t& x, uint32_t& y) {       // x's odd 16-lane rows <-> y's even ones
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0]; y = r[1];
}
template <int OFF> __device__ __forceinline__ void lds_read16(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int LO, int OFF> __device__ __forceinline__ void lds_read16_acc(uint32_t addr) {
    asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" :: "v"(addr), "n"(LO), "n"(LO + 3), "n"(OFF));
}
__device__ __forceinline__ void swap32(uint32_t& x, uint32_t& y) {       // x's upper 32 lanes <-> y's lower 32 lanes
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    const bf16x2_t v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
#define ARS_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef ARS_X
#define ARS_X 0      // timing-only twins (WRONG results): 1 no DMA in the loop, 2 no barrier in the loop, 4 no epilogue pieces, 8 no bias reads
#endif
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void gemm_ars_kernel(ArsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ARS_RESERVE();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const uint32_t s0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.X), 0, (int)((size_t)p.M * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (int)((size_t)p.N * p.ldw * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((size_t)p.M * p.ldc * 2), 0x00020000);
    ARS_T(0); ARS_RT(12);

    // ---- the whole bias -> LDS, once, by LDS-DMA (1 KiB = 256 floats per wave-instruction; columns past N read as zeros through the
    // buffer's bounds check): older than every activation slab, so landed when the first activation load is, and published by
    // the barrier that follows it.  No bias: zeros.
    if (p.bias) {
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.N * 4, 0x00020000);
        for (int i = wave; i * 256 < p.N; i += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)(smem + BIAS_OFF + i * 1024), 16, (uint32_t)(lane * 16), (uint32_t)(i * 1024), 0, 0);
    } else {
        for (int i = threadIdx.x; i * 4 < p.N; i += 256) *reinterpret_cast<u32x4*>(smem + BIAS_OFF + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- constants of the weight stream.  Slot image: [32 columns][1,024 B], chunk c of column n at chunk c ^ (n & 15).
    uint32_t vw[8];                                                // per-lane source offsets of this wave's eight rows of a block
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int col = 8 * wave + i;
        vw[i] = (uint32_t)((size_t)col * p.ldw * 2 + ((lane ^ (col & 15)) << 4));
    }
    const uint32_t wstep = 32u * (uint32_t)p.ldw * 2u;             // bytes from one column block to the next
    // fragment addresses in slot 0: column n = lane & 31, k-step s = 8 a + b: byte (32 b + 16 h) ^ ((n & 15) << 4) of 256-byte group a
    uint32_t va[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) va[b] = s0 + m * 1024 + ((32 * b + 16 * h) ^ ((m & 15) << 4));
    const uint32_t vbias = s0 + BIAS_OFF + 16 * h;                 // + 128 cb: this lane's four floats of each 8-column group
    uint32_t ar[4];                                                // activation staging: fragment addresses in this wave's quarter
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) ar[sp] = s0 + wave * SLOT + m * 128 + ((((2 * sp + h) ^ ((m >> 1) & 7))) << 4);

    u32x4 wf[4];
    f32x16 acc[2][2];                                              // [set][row tile]
    f32x16 bv;                                                     // the block's bias in accumulator layout (the chains' C operand)
    uint32_t pk[2][8];
    u32x4 bt[4];
#pragma unroll
    for (int q_ = 0; q_ < 2; ++q_)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q_][t][e] = 0.f;

    for (int sg = 0; sg < MAXSEG; ++sg) {
        const ArsSeg seg = p.segs[blockIdx.x * MAXSEG + sg];
        const int cb0 = __builtin_amdgcn_readfirstlane(seg.cb0), nb = __builtin_amdgcn_readfirstlane(seg.cb1) - cb0;
        if (nb <= 0) break;
        const int R0 = __builtin_amdgcn_readfirstlane(seg.rb) * 256 + wave * 64;
        if (sg > 0) { asm volatile("s_barrier" ::: "memory"); ARS_SB(); }     // every wave has left the ring

        // ---- this wave's 64 rows of X -> a[0:255], through its own quarter of the idle ring: eight 64-k slabs, four in flight.
        // Slab image: [64 rows][128 B], 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 of 32-row
        // fragments); the DMA writes 1 KiB = 8 rows linearly, so the permutation sits on the source address.
        {
            uint32_t vx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * i + (lane >> 3), g = (row >> 1) & 7;
                const int gr = min(R0 + row, p.M - 1);
                vx[i] = (uint32_t)(((size_t)gr * p.ldx + 8 * ((lane & 7) ^ g)) * 2);
            }
            auto slab = [&](int u) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void*)(smem + wave * SLOT + (u & 3) * 8192 + i * 1024), 16, vx[i],
                                                             (uint32_t)(128 * u), 0, 0);
            };
            slab(0); slab(1); slab(2); slab(3);
            static_for<8>([&](auto u_) {
                constexpr int u = decltype(u_)::value;
                ARS_SB();
                wait_vm<8 * (u <= 4 ? 3 : 7 - u)>();               // all but the slabs behind this one have landed
                ARS_SB();
                static_for<2>([&](auto t_) {
                    constexpr int t = decltype(t_)::value;
                    static_for<4>([&](auto sp_) {
                        constexpr int sp = decltype(sp_)::value;
                        lds_read16_acc<128 * t + 16 * u + 4 * sp, (u & 3) * 8192 + t * 4096>(ar[sp]);
                    });
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ARS_SB();
                if constexpr (u + 4 < 8) slab(u + 4);
            });
        }
        ARS_SB();
        ARS_T(1 + 5 * sg);
        asm volatile("s_barrier" ::: "memory");                   // every wave is done with its staging quarter: the ring is free
        ARS_SB();

        // output offsets of this lane's rows (tile t: row R0 + 32 t + m), its 16 bytes start at column 8 h of a 16-column half block
        uint32_t vo[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = R0 + 32 * t + m;
            vo[t] = row < p.M ? (uint32_t)(((size_t)row * p.ldc + 8 * h) * 2) : 0xFFFFFFF0u;
        }
        auto dma = [&](int j, int i) {                             // row i of this wave's share of block j (clamped past the end)
            const int jj = min(j, nb - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void*)(smem + (j & 3) * SLOT + (8 * wave + i) * 1024), 16, vw[i],
                                                     (uint32_t)(cb0 + jj) * wstep, 0, 0);
        };
        auto bias_issue = [&](int j) {                             // bias of block j: four ds_read_b128 (in the wave's LDS queue)
            const uint32_t a = vbias + 128u * (uint32_t)(cb0 + j);
            lds_read16<0>(bt[0], a); lds_read16<32>(bt[1], a); lds_read16<64>(bt[2], a); lds_read16<96>(bt[3], a);
        };
        auto bias_collect = [&]() {                                // once they have returned: -> bv
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const uint32_t u_ = bt[g][e]; bv[4 * g + e] = __uint_as_float(u_); }   // (bit_cast of a vector ELEMENT reads element 0)
        };
        // epilogue pieces of the accumulator set Q (block jq): convert, swap halves between lanes l and l + 32, store
        auto cvt = [&](int Q, int t, int i) {
            pk[t][2 * i] = pack2(acc[Q][t][4 * i], acc[Q][t][4 * i + 1]);
            pk[t][2 * i + 1] = pack2(acc[Q][t][4 * i + 2], acc[Q][t][4 * i + 3]);
        };
        auto swp = [&](int t, int g2) { swap32(pk[t][4 * g2], pk[t][4 * g2 + 2]); swap32(pk[t][4 * g2 + 1], pk[t][4 * g2 + 3]); };
        auto store = [&](int t, int g2, int jq, uint32_t vofs) {
            const u32x4 d = {pk[t][4 * g2], pk[t][4 * g2 + 1], pk[t][4 * g2 + 2], pk[t][4 * g2 + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(d, rC, vofs, (uint32_t)(((cb0 + jq) * 32 + 16 * g2) * 2), 0);
            asm volatile("s_nop 4" :: "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
        };

        // ---- prologue of the stream: blocks 0, 1 and the first three rows of block 2
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(0, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(1, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma(2, i);
        bias_issue(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ARS_SB();
        bias_collect();
        wait_vm<11>();
        asm volatile("s_barrier" ::: "memory");
        ARS_SB();
        ARS_T(2 + 5 * sg);
        lds_read16<0>(wf[0], va[0]); lds_read16<0>(wf[1], va[1]); lds_read16<0>(wf[2], va[2]);
        ARS_SB();

        // one column block j: ring slot j & 3, accumulator set P (compile time: the loop below alternates)
        auto block = [&](auto par_, int j) {
            constexpr int P = decltype(par_)::value, Q = P ^ 1;
            // block 0 has no predecessor: its epilogue pieces run on stale values and their stores fall outside the buffer (the
            // counted waits see the same number of operations in every block)
            const uint32_t vo0 = j > 0 ? vo[0] : 0xFFFFFFF0u, vo1 = j > 0 ? vo[1] : 0xFFFFFFF0u;
            const uint32_t so = (uint32_t)((j & 3) * SLOT), sn = (uint32_t)(((j + 1) & 3) * SLOT);
            uint32_t vc[8], vn[3];
#pragma unroll
            for (int b = 0; b < 8; ++b) vc[b] = va[b] + so;
#pragma unroll
            for (int b = 0; b < 3; ++b) vn[b] = va[b] + sn;
            static_for<32>([&](auto s_) {
                constexpr int S = decltype(s_)::value;
                if constexpr (S == 29) {
                    // everything of block j + 1 this wave requested has landed; past the barrier every wave's has, and slot (j + 3) & 3
                    // (block j - 1) has been read by everybody
                    if (!(ARS_X & 1)) { if (j == 0) wait_vm<12>(); else wait_vm<16>(); }
                    if (!(ARS_X & 2)) asm volatile("s_barrier" ::: "memory");
                    ARS_SB();
                }
                if constexpr (S == 26 && !(ARS_X & 8)) bias_issue(min(j + 1, nb - 1));     // returned by step 28's wait (in-order LDS queue)
                // W fragment of step S + 3 (the next block's first three from the next slot)
                if constexpr (S + 3 < 32) lds_read16<256 * ((S + 3) >> 3)>(wf[(S + 3) & 3], vc[(S + 3) & 7]);
                else lds_read16<0>(wf[(S + 3) & 3], vn[S + 3 - 32]);
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                if constexpr (S == 0) { mfma_first<0>(acc[P][0], wf[0], bv); mfma_first<128>(acc[P][1], wf[0], bv); }
                else { mfma_acc<4 * S>(acc[P][0], wf[S & 3]); mfma_acc<128 + 4 * S>(acc[P][1], wf[S & 3]); }
                // the stream: rows 3..7 of block j + 2 in steps 0..4, rows 0..2 of block j + 3 in steps 29..31
                if constexpr (S < 5 && !(ARS_X & 1)) dma(j + 2, S + 3);
                if constexpr (S >= 29 && !(ARS_X & 1)) dma(j + 3, S - 29);
                // the previous block's epilogue, a piece per step
                if constexpr (!(ARS_X & 4)) {
                if constexpr (S >= 2 && S < 6) cvt(Q, 0, S - 2);
                if constexpr (S == 6 || S == 7) swp(0, S - 6);
                if constexpr (S >= 8 && S < 12) cvt(Q, 1, S - 8);
                if constexpr (S == 13 || S == 14) swp(1, S - 13);
                if constexpr (S == 12 || S == 16) store(0, (S - 12) >> 2, max(j - 1, 0), vo0);
                if constexpr (S == 20 || S == 24) store(1, (S - 20) >> 2, max(j - 1, 0), vo1);
                }
                if constexpr (S == 28 && !(ARS_X & 8)) bias_collect();
                ARS_SB();
            });
        };
        int j = 0;
        for (; j + 2 <= nb; j += 2) {
            block(std::integral_constant<int, 0>{}, j);
            block(std::integral_constant<int, 1>{}, j + 1);
        }
        if (j < nb) block(std::integral_constant<int, 0>{}, j);
        ARS_T(3 + 5 * sg);
        // ---- the last block's epilogue
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        const int Ql = (nb - 1) & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (Ql) cvt(1, t, i); else cvt(0, t, i); }
            swp(t, 0); swp(t, 1);
            store(t, 0, nb - 1, vo[t]); store(t, 1, nb - 1, vo[t]);
        }
        ARS_T(4 + 5 * sg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ARS_T(5 + 5 * sg);
    }
    ARS_T(13); ARS_RT(14);
}

__global__ __launch_bounds__(256, 1) void gemm_ars16_kernel(ArsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ARS_RESERVE();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, h = lane >> 4;          // 16x16x32: row / column index of the fragment, k-group (8 of the 32 k)
    const uint32_t s0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.X), 0, (int)((size_t)p.M * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (int)((size_t)p.N * p.ldw * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((size_t)p.M * p.ldc * 2), 0x00020000);
    ARS_T(0); ARS_RT(12);

    // ---- the whole bias -> LDS, once, by LDS-DMA (1 KiB = 256 floats per wave-instruction; columns past N read as zeros through the
    // buffer's bounds check): older than every activation slab, so landed when the first activation load is, and published by
    // the barrier that follows it.  No bias: zeros.
    if (p.bias) {
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.N * 4, 0x00020000);
        for (int i = wave; i * 256 < p.N; i += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)(smem + BIAS_OFF + i * 1024), 16, (uint32_t)(lane * 16), (uint32_t)(i * 1024), 0, 0);
    } else {
        for (int i = threadIdx.x; i * 4 < p.N; i += 256) *reinterpret_cast<u32x4*>(smem + BIAS_OFF + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- constants of the weight stream.  Slot image: [32 columns][1,024 B], chunk c of column n at chunk c ^ (n & 15).
    uint32_t vw[8];                                                // per-lane source offsets of this wave's eight rows of a block
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int col = 8 * wave + i;
        vw[i] = (uint32_t)((size_t)col * p.ldw * 2 + ((lane ^ (col & 15)) << 4));
    }
    const uint32_t wstep = 32u * (uint32_t)p.ldw * 2u;             // bytes from one column block to the next
    // fragment addresses in slot 0: column n = lane & 31, k-step s = 8 a + b: byte (32 b + 16 h) ^ ((n & 15) << 4) of 256-byte group a
    uint32_t va[4];                                                // k-step s = 4 a + b, column tile c: + 16384 c + 256 a
#pragma unroll
    for (int b = 0; b < 4; ++b) va[b] = s0 + m * 1024 + (((4 * b + h) ^ m) << 4);
    const uint32_t vbias = s0 + BIAS_OFF + 16 * h;                 // + 128 cb + 64 c: this lane's four floats (columns 16 c + 4 h ..)
    uint32_t ar[2];                                                // activation staging: fragment addresses in this wave's quarter
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) ar[sp] = s0 + wave * SLOT + m * 128 + ((((4 * sp + h) ^ ((m >> 1) & 7))) << 4);

    u32x4 wf[4];
    f32x4 acc[2][2][4];                                            // [set][column tile][row tile]
    f32x4 bv[2];                                                   // the block's bias per column tile (the chains' C operand)
    uint32_t pk[4][4];                                             // [row tile][c0 lo, c0 hi, c1 lo, c1 hi]
    u32x4 bt[2];
#pragma unroll
    for (int q_ = 0; q_ < 2; ++q_)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[q_][c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int sg = 0; sg < MAXSEG; ++sg) {
        const ArsSeg seg = p.segs[blockIdx.x * MAXSEG + sg];
        const int cb0 = __builtin_amdgcn_readfirstlane(seg.cb0), nb = __builtin_amdgcn_readfirstlane(seg.cb1) - cb0;
        if (nb <= 0) break;
        const int R0 = __builtin_amdgcn_readfirstlane(seg.rb) * 256 + wave * 64;
        if (sg > 0) { asm volatile("s_barrier" ::: "memory"); ARS_SB(); }     // every wave has left the ring

        // ---- this wave's 64 rows of X -> a[0:255], through its own quarter of the idle ring: eight 64-k slabs, four in flight.
        // Slab image: [64 rows][128 B], 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 of 32-row
        // fragments); the DMA writes 1 KiB = 8 rows linearly, so the permutation sits on the source address.
        {
            uint32_t vx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * i + (lane >> 3), g = (row >> 1) & 7;
                const int gr = min(R0 + row, p.M - 1);
                vx[i] = (uint32_t)(((size_t)gr * p.ldx + 8 * ((lane & 7) ^ g)) * 2);
            }
            auto slab = [&](int u) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void*)(smem + wave * SLOT + (u & 3) * 8192 + i * 1024), 16, vx[i],
                                                             (uint32_t)(128 * u), 0, 0);
            };
            slab(0); slab(1); slab(2); slab(3);
            static_for<8>([&](auto u_) {
                constexpr int u = decltype(u_)::value;
                ARS_SB();
                wait_vm<8 * (u <= 4 ? 3 : 7 - u)>();               // all but the slabs behind this one have landed
                ARS_SB();
                static_for<4>([&](auto t_) {
                    constexpr int t = decltype(t_)::value;
                    static_for<2>([&](auto sp_) {
                        constexpr int sp = decltype(sp_)::value;
                        lds_read16_acc<64 * t + 4 * (2 * u + sp), (u & 3) * 8192 + t * 2048>(ar[sp]);
                    });
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ARS_SB();
                if constexpr (u + 4 < 8) slab(u + 4);
            });
        }
        ARS_SB();
        ARS_T(1 + 5 * sg);
        asm volatile("s_barrier" ::: "memory");                   // every wave is done with its staging quarter: the ring is free
        ARS_SB();

        // output offsets of this lane's rows (tile t: row R0 + 32 t + m), its 16 bytes start at column 8 h of a 16-column half block
        uint32_t vo[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = R0 + 16 * t + m;
            vo[t] = row < p.M ? (uint32_t)(((size_t)row * p.ldc + 16 * (h & 1) + 8 * (h >> 1)) * 2) : 0xFFFFFFF0u;
        }
        auto dma = [&](int j, int i) {                             // row i of this wave's share of block j (clamped past the end)
            const int jj = min(j, nb - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void*)(smem + (j & 3) * SLOT + (8 * wave + i) * 1024), 16, vw[i],
                                                     (uint32_t)(cb0 + jj) * wstep, 0, 0);
        };
        auto bias_issue = [&](int j) {                             // bias of block j: two ds_read_b128 (in the wave's LDS queue)
            const uint32_t a = vbias + 128u * (uint32_t)(cb0 + j);
            lds_read16<0>(bt[0], a); lds_read16<64>(bt[1], a);
        };
        auto bias_collect = [&]() {                                // once they have returned: -> bv
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const uint32_t u_ = bt[c][e]; bv[c][e] = __uint_as_float(u_); }
        };
        // epilogue pieces of the accumulator set Q (block jq): convert, swap halves between lanes l and l + 32, store
        // row tile t of set Q: lane (m, h) holds columns 16 c + 4 h + 0..3 of row 16 t + m; after the exchange between the 16-lane rows
        // h and h ^ 1 it holds 8 consecutive columns: (h & 1) * 16 + (h >> 1) * 8
        auto cvt = [&](int Q, int t) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                pk[t][2 * c] = pack2(acc[Q][c][t][0], acc[Q][c][t][1]);
                pk[t][2 * c + 1] = pack2(acc[Q][c][t][2], acc[Q][c][t][3]);
            }
        };
        auto swp = [&](int t) { swap16(pk[t][0], pk[t][2]); swap16(pk[t][1], pk[t][3]); };
        auto store = [&](int t, int jq, uint32_t vofs) {
            const u32x4 d = {pk[t][0], pk[t][1], pk[t][2], pk[t][3]};
            __builtin_amdgcn_raw_buffer_store_b128(d, rC, vofs, (uint32_t)(((cb0 + jq) * 32) * 2), 0);
            asm volatile("s_nop 4" :: "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
        };

        // ---- prologue of the stream: blocks 0, 1 and the first three rows of block 2
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(0, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(1, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma(2, i);
        bias_issue(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ARS_SB();
        bias_collect();
        wait_vm<11>();
        asm volatile("s_barrier" ::: "memory");
        ARS_SB();
        ARS_T(2 + 5 * sg);
        lds_read16<0>(wf[0], va[0]); lds_read16<16384>(wf[1], va[0]); lds_read16<0>(wf[2], va[1]);      // steps 0, 1, 2 = (s, c) = (0, 0), (0, 1), (1, 0)
        ARS_SB();

        // one column block j: ring slot j & 3, accumulator set P (compile time: the loop below alternates).  Step S = 2 s + c: the W fragment
        // of column tile c, k-step s (32 of k) feeds four MFMAs (the wave's four 16-row tiles)
        auto block = [&](auto par_, int j) {
            constexpr int P = decltype(par_)::value, Q = P ^ 1;
            uint32_t vob[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) vob[t] = j > 0 ? vo[t] : 0xFFFFFFF0u;
            const uint32_t so = (uint32_t)((j & 3) * SLOT), sn = (uint32_t)(((j + 1) & 3) * SLOT);
            uint32_t vc[4], vn[2];
#pragma unroll
            for (int b = 0; b < 4; ++b) vc[b] = va[b] + so;
#pragma unroll
            for (int b = 0; b < 2; ++b) vn[b] = va[b] + sn;
            static_for<32>([&](auto s_) {
                constexpr int S = decltype(s_)::value, KS = S >> 1, C = S & 1;
                if constexpr (S == 29) {
                    if (!(ARS_X & 1)) { if (j == 0) wait_vm<12>(); else wait_vm<16>(); }
                    if (!(ARS_X & 2)) asm volatile("s_barrier" ::: "memory");
                    ARS_SB();
                }
                if constexpr (S == 26 && !(ARS_X & 8)) bias_issue(min(j + 1, nb - 1));
                {   // W fragment of step S + 3 (the next block's first three from the next slot)
                    constexpr int S3 = (S + 3) & 31, K3 = S3 >> 1, C3 = S3 & 1;
                    if constexpr (S + 3 < 32) lds_read16<16384 * C3 + 256 * (K3 >> 2)>(wf[(S + 3) & 3], vc[K3 & 3]);
                    else lds_read16<16384 * C3 + 256 * (K3 >> 2)>(wf[(S + 3) & 3], vn[K3 & 3]);
                }
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                static_for<4>([&](auto t_) {
                    constexpr int t = decltype(t_)::value;
                    if constexpr (KS == 0) mfma16_first<64 * t>(acc[P][C][t], wf[S & 3], bv[C]);
                    else mfma16_acc<64 * t + 4 * KS>(acc[P][C][t], wf[S & 3]);
                });
                if constexpr (S < 5 && !(ARS_X & 1)) dma(j + 2, S + 3);
                if constexpr (S >= 29 && !(ARS_X & 1)) dma(j + 3, S - 29);
                if constexpr (!(ARS_X & 4)) {
                if constexpr (S == 2 || S == 6 || S == 10 || S == 14) cvt(Q, (S - 2) >> 2);
                if constexpr (S == 4 || S == 8 || S == 12 || S == 16) swp((S - 4) >> 2);
                if constexpr (S == 12 || S == 16 || S == 20 || S == 24) store((S - 12) >> 2, max(j - 1, 0), vob[(S - 12) >> 2]);
                }
                if constexpr (S == 28 && !(ARS_X & 8)) bias_collect();
                ARS_SB();
            });
        };
        int j = 0;
        for (; j + 2 <= nb; j += 2) {
            block(std::integral_constant<int, 0>{}, j);
            block(std::integral_constant<int, 1>{}, j + 1);
        }
        if (j < nb) block(std::integral_constant<int, 0>{}, j);
        ARS_T(3 + 5 * sg);
        // ---- the last block's epilogue
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        const int Ql = (nb - 1) & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (Ql) cvt(1, t); else cvt(0, t);
            swp(t);
            store(t, nb - 1, vo[t]);
        }
        ARS_T(4 + 5 * sg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ARS_T(5 + 5 * sg);
    }
    ARS_T(13); ARS_RT(14);
}

// host: cut the (row block, column block) space into one span per workgroup, charging `a` column blocks for every activation load
static std::vector<ArsSeg> ars_partition(int RB, int CB, int G, double a) {
    const double T = (double)RB * CB;
    auto build = [&](double B, std::vector<ArsSeg>& out) -> int {
        out.assign((size_t)G * MAXSEG, ArsSeg{0, 0, 0, 0});
        int g = 0, ns = 0; double cost = 0;
        for (int rb = 0; rb < RB; ++rb) {
            int c = 0;
            while (c < CB) {
                if (ns == MAXSEG || cost + a + 1.0 > B) { ++g; ns = 0; cost = 0; }
                if (g >= G) return G + 1;
                int take = (int)(B - cost - a);
                if (take < 1) take = 1;
                if (take > CB - c) take = CB - c;
                if (CB - c - take == 1) take = CB - c;              // no one-block leftovers
                out[(size_t)g * MAXSEG + ns] = ArsSeg{rb, c, c + take, 0};
                ++ns; cost += a + take; c += take;
            }
        }
        return g + 1;
    };
    std::vector<ArsSeg> best;
    double lo = T / G, hi = T / G + 3 * a + CB + 2;
    for (int it = 0; it < 50; ++it) {
        const double mid = 0.5 * (lo + hi);
        std::vector<ArsSeg> tmp;
        if (build(mid, tmp) <= G) { hi = mid; best.swap(tmp); } else lo = mid;
    }
    if (best.empty()) build(hi, best);
    return best;
}

__global__ void ref_kernel(const bf16* A, const bf16* B, const float* bias, float* R, int N, int K, int lda, int ldb, const int* rows) {
    const int r = rows[blockIdx.x];
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = bias ? bias[n] : 0.f;
        for (int k = 0; k < K; ++k) s += to_f32(A[(size_t)r * lda + k]) * to_f32(B[(size_t)n * ldb + k]);
        R[(size_t)blockIdx.x * N + n] = s;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 24000, N = argc > 2 ? atoi(argv[2]) : 1536; const double acost = argc > 3 ? atof(argv[3]) : 5.0;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int K = KD;
    if (N % 32) { printf("N multiple of 32\n"); return 1; }
    std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
    std::vector<float> hbias(N);
    uint32_t s = 12345u;
    auto rndf = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    auto rnd = [&]() { const float f = rndf(); uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd();
    for (auto& v : hbias) v = rndf();
    bf16 *A, *B, *C; float *R, *bias; int* rows;
    CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0xff, (size_t)M * N * 2));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ars_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ars16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const bool shape16 = getenv("ARS_SHAPE16") != nullptr;
    auto* const kern = shape16 ? gemm_ars16_kernel : gemm_ars_kernel;
    const int RB = (M + 255) / 256;
    unsigned long long* dbg; CK(hipMalloc(&dbg, 4096 * 16 * 8)); CK(hipMemset(dbg, 0, 4096 * 16 * 8));
    const int G = getenv("ARS_G") ? atoi(getenv("ARS_G")) : 256;
    const std::vector<ArsSeg> hs = ars_partition(RB, N / 32, G, acost);
    {
        int used = 0, maxb = 0, maxs = 0; long tot = 0;
        for (int g = 0; g < G; ++g) { int nb_ = 0, ns_ = 0; for (int i = 0; i < MAXSEG; ++i) { const ArsSeg& q = hs[(size_t)g * MAXSEG + i]; if (q.cb1 > q.cb0) { nb_ += q.cb1 - q.cb0; ++ns_; } }
            used += nb_ > 0; maxb = nb_ > maxb ? nb_ : maxb; maxs = ns_ > maxs ? ns_ : maxs; tot += nb_; }
        printf("partition: %d of %d workgroups used, at most %d blocks in %d pieces each, %ld blocks in all (%d expected)\n", used, G, maxb, maxs, tot, RB * (N / 32));
    }
    ArsSeg* dsegs; CK(hipMalloc(&dsegs, hs.size() * sizeof(ArsSeg))); CK(hipMemcpy(dsegs, hs.data(), hs.size() * sizeof(ArsSeg), hipMemcpyHostToDevice));
    ArsArgs a{A, B, C, bias, M, N, K, K, N, dsegs, dbg};
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    const int nr = 96; std::vector<int> hr(nr);
    for (int i = 0; i < nr; ++i) hr[i] = i < 8 ? i : (i < 16 ? M - 1 - (i - 8) : (int)(((long)i * 7919 * 131) % M));
    CK(hipMalloc(&rows, nr * 4)); CK(hipMalloc(&R, (size_t)nr * N * 4));
    CK(hipMemcpy(rows, hr.data(), nr * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3(nr), dim3(256), 0, 0, A, B, bias, R, N, K, K, K, rows);
    std::vector<float> href((size_t)nr * N); std::vector<uint16_t> hc((size_t)M * N);
    CK(hipMemcpy(href.data(), R, href.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0; long bad = 0;
    for (int i = 0; i < nr; ++i)
        for (int n = 0; n < N; ++n) {
            uint32_t u = (uint32_t)hc[(size_t)hr[i] * N + n] << 16; float c; memcpy(&c, &u, 4);
            const double d = fabs(c - href[(size_t)i * N + n]);
            if (!(d <= 0.02 * fabs(href[(size_t)i * N + n]) + 0.02)) { if (bad < 8) printf("  row %d col %d: got %g want %g\n", hr[i], n, c, href[(size_t)i * N + n]); ++bad; }
            if (d == d) worst = fmax(worst, d);
            scale = fmax(scale, fabs(href[(size_t)i * N + n]));
        }
    printf("check: %ld bad of %ld; worst |C - ref| = %.4g on values up to %.4g (%s)\n", bad, (long)nr * N, worst, scale, bad == 0 ? "ok" : "WRONG");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(256), LDS_BYTES, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(256), LDS_BYTES, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / us * 1e-6;
#ifdef ARS_STAMPS
    {
        std::vector<unsigned long long> hd(4096 * 16);
        CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        const int nwg = G;
        printf("stamps, cycles from the workgroup's start: [A-load, W prologue, loop, tail, drain] per piece\n");
        for (int w = 0; w < nwg; w += nwg / 10 + 1) {
            printf("  wg %4d:", w);
            for (int i = 1; i < 11; ++i) printf(" %7lld", hd[w * 16 + i] ? (long long)(hd[w * 16 + i] - hd[w * 16]) : -1ll);
            printf(" | life %lld cycles = %.2f us (%.2f GHz)\n", (long long)(hd[w * 16 + 13] - hd[w * 16]), (hd[w * 16 + 14] - hd[w * 16 + 12]) * 0.01,
                   (hd[w * 16 + 13] - hd[w * 16]) / ((hd[w * 16 + 14] - hd[w * 16 + 12]) * 10.0));
        }
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int w = 0; w < nwg; ++w) if (hd[w * 16 + 14]) { r0 = hd[w * 16 + 12] < r0 ? hd[w * 16 + 12] : r0; r1 = hd[w * 16 + 14] > r1 ? hd[w * 16 + 14] : r1; }
        printf("first workgroup start -> last workgroup end: %.2f us\n", (r1 - r0) * 0.01);
        (void)tmin; (void)tmax;
    }
#endif
    printf("gemm_ars %s %d x %d x %d, load cost %.1f: %.1f us  %.0f TFLOP/s\n", shape16 ? "16x16x32" : "32x32x16", M, N, K, acost, us, tf);
    return 0;
}
