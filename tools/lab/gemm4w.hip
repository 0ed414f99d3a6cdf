// The 256 x 256 x 64 tile of gemm256.hip partitioned as FOUR waves of 128 x 128 (96 x 128 for the 192-row tile), one per SIMD:
//   NT  Y  = epi(X  . W^T)   X [M][K], W [N][K]      forward projections / FFN      (F.linear, multihead_attention.py:190-208,
//   NN  dX = epi(dY . W)     dY [M][K], W [K][N]      their data gradients            transformer_layer.py:132-134 + autograd)
//
// STATUS (round 5): ARCHIVED lab code, not built and not part of libs2t_hip.so.  Round 4 shipped it behind s2t_set_option("gemm4w", 1)
// (removed with it); to revive it, move it back into fbk_fairseq_st_amd/csrc and restore the hook in s2t_gemm256_try.  Bit-identical to gemm256 on
// every variant, +0..8 % on K = 2,048 products, -5..-20 % on K = 512 ones -- see "What it measured" below.
//
// Why (round 4, profiles/r04_gemm256_experiments.txt): the eight-wave loop is bound by what feeds the matrix cores -- taking its
// LDS-DMA out is worth 17 %, its fragment reads 22 %, its barriers 5 % -- and a second schedule of the same partition changed
// nothing.  With 128 x 128 per wave every fragment read from LDS feeds eight MFMAs instead of four (a third less LDS traffic per
// K-tile: 128 KiB instead of 192), no second wave competes for the SIMD's issue slots, and one barrier per K-tile is enough.
// tools/gemm4w_probe.hip (the bare loop): 85 % of the rate of the same loop with the DMA, the reads and the barrier removed; the
// eight-wave kernel reaches ~50 % of its own.
//
// What it measured: the bare loop (probe) runs at 85 % of its own MFMA-only rate and ~6 % above gemm256's long-K rate, but (1) 256
// accumulators live in AGPRs: hipcc zeroes them with v_accvgpr_write before a tile's first MFMAs (the "first MFMA takes the constant
// 0" form needs VGPR accumulators) and reads them back with v_accvgpr_read in the epilogue, ~2,000 cycles per tile that the
// eight-wave kernel does not pay; (2) one wave per SIMD runs the whole epilogue of its SIMD (32 steps) with nothing else to issue --
// the same total as two waves of 16 steps one after the other; (3) the tile's first K-tile drains the epilogue's stores (vmcnt(0):
// a counted wait would have to assume stores and loads retire in one order).  On K = 512 tiles (eight K-tiles) that outweighs the
// loop; only K >= 1,536 products gain.
//
// Structure:
//   * 4 waves = 2 (M) x 2 (N); 8 x 8 (6 x 8) tiles of v_mfma_f32_16x16x32_bf16 per wave: 256 (192) accumulator registers, which
//     the compiler keeps in the AGPR half of the wave's 512 registers; one workgroup per CU.
//   * the LDS image of gemm256 (gemm_tile.hpp): two K-tile buffers of 64 KiB = A-h0 | A-h1 | B-h0 | B-h1, a half = 128 rows x 128 B
//     (or 64 k-rows x 256 B for the k-strided operand of the NN form), filled by buffer-addressed LDS-DMA with the XOR swizzle on
//     the source side.  Here a half is one wave row's / wave column's operand: wave (wr, wc) reads A-h[wr] and B-h[wc] only.
//   * a K-tile is two phases of 64 (48) MFMAs per wave, the fragments of the NEXT phase in flight under them (two register sets):
//         P0  MFMA k-half 0 | reads k-half 1 of this K-tile
//             wait: own reads returned, own DMA of K-tile t+1 landed ; ONE barrier
//         P1  MFMA k-half 1 | DMA of K-tile t+2 into the buffer just read ; reads k-half 0 of K-tile t+1 from the other buffer
//     Hazards: every wave's reads of buffer X have returned before the barrier, the re-staging DMA issues after it (WAR); every
//     wave's DMA of K-tile t+1 (issued one K-tile earlier) has landed before the barrier, its first reads issue after it (RAW).
//   * persistent: one workgroup per CU walks the tiles; the stream does not stop at a tile boundary (the DMA of the last two
//     K-tiles' P1 stages the next tile's first two K-tiles, so the epilogue runs with the next tile's first K-tiles already in LDS or on their way.
//   * epilogue: gemm256's, per wave, on the new lane -> element map: v_permlane16_swap to 8 consecutive columns per lane, lane turn
//     through a wave-private LDS slot, 16-byte buffer stores held against the store-data hazard (inputs-only asm), 1-bit ReLU record
//     of 32 bytes per lane and tile.
#include "common.hpp"
#include "prof.hpp"
#include "gemm_tile.hpp"
#include <type_traits>

// f(integral_constant<int, G>) for G = G0 .. G1-1: the scheduling builtins want their group sizes as constants
template <int G0, int G1, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (G0 < G1) { f(std::integral_constant<int, G0>{}); static_for<G0 + 1, G1>(f); }
}

template <typename TO, bool TB, int MT, int ACT, int EXT>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HR = 16 * MT;                      // rows of one A half = rows of a wave (128 / 96)
    constexpr int BM = 2 * HR, BN = 256;
    constexpr int APW = HR / 32;                     // 1-KiB pieces (8 rows) of an A half staged by one wave (4 / 3)
    constexpr int NDMA = 2 * APW + 8;                // DMA instructions per wave and K-tile (16 / 14)
    constexpr int NG = 2 * MT;                       // MFMA groups of four per phase (16 / 12)
    constexpr int NR = MT + 8;                       // fragments read per phase (16 / 14)
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, tiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    int tile = xcd_remap(blockIdx.x, G);             // then tile += G: every round is a contiguous run of tiles, an XCD's share contiguous inside it
    if (tile >= tiles) return;
    const int nk = p.K / BK;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1, r16 = lane & 15, q = lane >> 4;

    // ---- staging: wave w fills pieces APW w .. APW w + APW - 1 of each A half and 4 w .. 4 w + 3 of each B half
    struct Offs { uint32_t a[2][APW], b[2][4]; };                      // [half][piece] byte offsets of this lane's 16 bytes at k = 0
    auto offsets = [&](int tl, Offs& o) {
        const int row0 = (tl / tiles_n) * BM, col0 = (tl % tiles_n) * BN;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < APW; ++i) {                            // A: image row r' = 8 piece + lane / 8 of half h <-> tile row h HR + r'
                const int rp = 8 * (APW * wave + i) + (lane >> 3), pos = lane & 7;
                const int gr = min(row0 + h * HR + rp, p.M - 1);
                o.a[h][i] = (uint32_t)(((size_t)gr * p.lda + ((pos ^ (rp & 7)) << 3)) * 2);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (!TB) {                                   // B [N][K]: image row r' <-> tile column 128 h + r'
                    const int rp = 8 * (4 * wave + i) + (lane >> 3), pos = lane & 7;
                    const int gc = min(col0 + 128 * h + rp, p.N - 1);
                    o.b[h][i] = (uint32_t)(((size_t)gc * p.ldb + ((pos ^ (rp & 7)) << 3)) * 2);
                } else {                                               // B [K][N]: image k-row 4 piece + lane / 16, image column c' <-> tile column 128 h + c'
                    const int kr = 4 * (4 * wave + i) + (lane >> 4), pos = lane & 15;
                    const int cp = (pos ^ trswz(kr)) << 3;
                    const int gc = min(col0 + 128 * h + cp, ((p.N + 7) & ~7) - 8);
                    o.b[h][i] = (uint32_t)(((size_t)kr * p.ldb + gc) * 2);
                }
            }
        }
    };
    Offs cur, nxt;
    offsets(tile, cur);
    const uint32_t kstepA = BK * 2, kstepB = TB ? (uint32_t)BK * (uint32_t)p.ldb * 2u : (uint32_t)BK * 2u;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((size_t)p.M * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0,
                                                                         (int)((size_t)(TB ? p.K : p.N) * p.ldb * 2), 0x00020000);
    int sbase = 0;                                                      // K-tiles consumed by earlier tiles: LDS buffer parity of the stream
    bool has_next = false;
    // DMA instruction d (0 .. NDMA-1) of stream position u = t + 2 of the CURRENT tile: past its last K-tile it is K-tile u - nk of the
    // next tile (nk >= 2); with no next tile the source is clamped to the last K-tile (in bounds) and the destination stays the buffer
    // the schedule says is free: the DMA count per K-tile is a constant and nothing reads those bytes afterwards
    auto dma = [&](int u, int d) {
        const bool roll = u >= nk && has_next;
        const int kt = roll ? u - nk : min(u, nk - 1);
        char* base = smem + __builtin_amdgcn_readfirstlane(((sbase + u) & 1) * BUF);
        if (d < 2 * APW) {
            const int h = d / APW, i = d % APW;
            const uint32_t vo = roll ? nxt.a[h][i] : cur.a[h][i];       // (a local: with the conditional as the builtin's argument the HOST pass drops the kernel's stub without a word)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)(base + h * HALF + (APW * wave + i) * 1024), 16, vo, (uint32_t)kt * kstepA, 0, 0);
        } else {
            const int h = (d - 2 * APW) >> 2, i = (d - 2 * APW) & 3;
            const uint32_t vo = roll ? nxt.b[h][i] : cur.b[h][i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)(base + 2 * HALF + h * HALF + (4 * wave + i) * 1024), 16, vo, (uint32_t)kt * kstepB, 0, 0);
        }
    };

    f32x4 acc[MT][8];                                                    // never zeroed: see FIRST below
    u32x4 fa[2][MT], fb[2][8];                                           // [k-half][tile]: the set a phase multiplies, the set the next one will
    const int fr_off = r16 * 128;
    const int swz0 = ((0 + q) ^ (r16 & 7)) << 4, swz1 = ((4 + q) ^ (r16 & 7)) << 4;
    // fragment r (0 .. NR-1) of k-half s from K-tile buffer `buf`: the eight B fragments first, then the A fragments
    auto frag = [&](const char* buf, int s, int r) {
        if (r < 8) {
            if constexpr (!TB) fb[s][r] = *reinterpret_cast<const u32x4*>(buf + 2 * HALF + wc * HALF + r * 2048 + fr_off + (s ? swz1 : swz0));
            else fb[s][r] = tr_frag(buf + 2 * HALF + wc * HALF, 16 * r, s, r16, q);
        } else {
            fa[s][r - 8] = *reinterpret_cast<const u32x4*>(buf + wr * HALF + (r - 8) * 2048 + fr_off + (s ? swz1 : swz0));
        }
    };
    constexpr int BRD = TB ? 2 : 1;                                      // ds_read instructions of one B fragment

    // ---- prologue: both K-tile buffers of the first tile; the first fragments
#pragma unroll
    for (int d = 0; d < NDMA; ++d) dma(0, d);
#pragma unroll
    for (int d = 0; d < NDMA; ++d) dma(1, d);
    if constexpr (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const uint32_t drop_th = (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f);
    const float drop_inv = 1.f / (1.f - p.p_drop);
    const uint32_t drop_ks = drop_seed_key(p.seed), drop_hwm = drop_high_mix(p.seed, 0);
    // buffer descriptors of the output, the extra operand stream and aux_out: M rows each (rows >= M fall outside)
    const void* Eptr = EXT == EXT_RES ? p.residual : EXT == EXT_OLD ? (const void*)p.C : p.aux;
    const int lde = EXT == EXT_RES ? p.ldr : EXT == EXT_OLD ? p.ldc : p.ldaux;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((size_t)p.M * p.ldc * sizeof(TO)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Eptr ? Eptr : (const void*)p.C), 0,
                                                                         (int)((size_t)p.M * lde * sizeof(TO)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(p.aux_out ? p.aux_out : p.C, 0,
                                                                         (int)((size_t)p.M * p.ldaux * sizeof(TO)), 0x00020000);
    // the data registers of a tile's last 16-byte stores (the last C quad, the 1-bit record): held over the loop's back edge until the
    // next tile's offset arithmetic is done (store-data hazard: see the epilogue)
    u32x4 tail_c = {0u, 0u, 0u, 0u}, tail_m = {0u, 0u, 0u, 0u}, tail_m2 = {0u, 0u, 0u, 0u};
    for (;;) {
        has_next = tile + G < tiles;
        if (has_next) offsets(tile + G, nxt);
        asm volatile("" :: "v"(tail_c[0]), "v"(tail_c[1]), "v"(tail_c[2]), "v"(tail_c[3]), "v"(tail_m[0]), "v"(tail_m[1]), "v"(tail_m[2]), "v"(tail_m[3]),
                           "v"(tail_m2[0]), "v"(tail_m2[1]), "v"(tail_m2[2]), "v"(tail_m2[3]) : "memory");
        // FIRST (the first K-tile of an output tile): the first MFMA of every accumulator takes the constant 0 as its C operand, so no
        // accumulator is ever zeroed by moves
        auto ktile = [&](int t, auto first_tag, auto last_tag) {
            constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
            const char* buf = smem + ((sbase + t) & 1) * BUF;
            const char* nbuf = smem + ((sbase + t + 1) & 1) * BUF;
            // ---- P0: k-half 0 | reads of k-half 1.  Source order = the order asked of the scheduler below: reads, then four MFMAs
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): k-half 0's fragments (requested a phase ago) are here
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                frag(buf, 1, g);
                if (g + NG < NR) frag(buf, 1, g + NG);
                const int i = g >> 1, j0 = 4 * (g & 1);
#pragma unroll
                for (int j = j0; j < j0 + 4; ++j)
                    acc[i][j] = mma16<bf16>(fb[0][j], fa[0][i], FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j]);
            }
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                __builtin_amdgcn_sched_group_barrier(0x100, (g < 8 ? BRD : 1) + (g + NG < NR ? 1 : 0), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                         // this wave's reads of `buf` have returned
            // this wave's DMA of K-tile t+1 has landed.  It was issued in P1 of K-tile t-1 and is the youngest vector-memory work of the
            // wave -- except right after an epilogue, whose stores are younger still: a counted wait would have to assume that stores
            // and loads retire in one order (hipcc's own counter model does not), so the first K-tile of a tile drains those too
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- P1: k-half 1 | DMA of K-tile t+2 into `buf`, reads of K-tile t+1's k-half 0 from the other buffer
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                dma(t + 2, g);
                if (g + NG < NDMA) dma(t + 2, g + NG);
                if constexpr (!LAST) {
                    frag(nbuf, 0, g);
                    if (g + NG < NR) frag(nbuf, 0, g + NG);
                }
                const int i = g >> 1, j0 = 4 * (g & 1);
#pragma unroll
                for (int j = j0; j < j0 + 4; ++j) acc[i][j] = mma16<bf16>(fb[1][j], fa[1][i], acc[i][j]);
            }
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                __builtin_amdgcn_sched_group_barrier(0x020, 1 + (g + NG < NDMA ? 1 : 0), 0);
                if constexpr (!LAST) __builtin_amdgcn_sched_group_barrier(0x100, (g < 8 ? BRD : 1) + (g + NG < NR ? 1 : 0), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        // the tile's first fragments (its K-tile 0 is visible: prologue barrier, or the barrier of the previous tile's last K-tile).  They are
        // NOT read under the previous tile's last phase: 64 registers held across the epilogue made every epilogue variant spill
#pragma unroll
        for (int r = 0; r < NR; ++r) frag(smem + (sbase & 1) * BUF, 0, r);
        ktile(0, std::true_type{}, std::false_type{});
        for (int t = 1; t < nk - 1; ++t) ktile(t, std::false_type{}, std::false_type{});
        ktile(nk - 1, std::false_type{}, std::true_type{});

        // ---- this tile's epilogue.  The next tile's first two K-tiles are in LDS or on their way.
        __builtin_amdgcn_sched_barrier(0);
        {
            typedef typename Pack4<TO>::type PK;
            constexpr uint32_t ES = sizeof(TO);
            const int row0 = (tile / tiles_n) * BM, col0 = (tile % tiles_n) * BN;
            const int colw = col0 + wc * 128 + 4 * q;                     // column of this lane's quad in column tile j: + 16 j
            const int roww = row0 + wr * HR + r16;                        // + 16 i
            f32x4 b4[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = colw + 16 * j;
                b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.bias && col + 4 <= p.N) b4[j] = *reinterpret_cast<const f32x4*>(p.bias + col);
                else if (p.bias && col < p.N) {                            // N not a multiple of 4 (a vocabulary of 5,001): the last quad, element by element
#pragma unroll
                    for (int e = 0; e < 4; ++e) b4[j][e] = col + e < p.N ? p.bias[col + e] : 0.f;
                }
            }
            static_assert(sizeof(TO) == 2, "bf16 outputs");
            // The quads of two neighbouring 16-column tiles (pair pp = tiles 2 pp, 2 pp + 1) are exchanged between lane rows q and q ^ 1
            // (v_permlane16_swap) so that every lane holds 8 consecutive columns: 16-byte accesses, 64-byte row segments.  Operand
            // loads come in the same shape and are swapped back.  (gemm256.hip explains the lane turn and the store-data hold.)
            uint32_t vC[4], vE[4], vX[4], vT[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int col = col0 + wc * 128 + pp * 32 + 16 * (q & 1) + 8 * (q >> 1);
                const bool ok = col < p.N;
                vC[pp] = ok ? (uint32_t)(((size_t)roww * p.ldc + col) * ES) : 0xFFFFFFF0u;
                vE[pp] = ok ? (uint32_t)(((size_t)roww * lde + col) * ES) : 0xFFFFFFF0u;
                vX[pp] = ok ? (uint32_t)(((size_t)roww * p.ldaux + col) * ES) : 0xFFFFFFF0u;
                const int colt = col0 + wc * 128 + pp * 32 + 8 * (lane & 3);   // turned order: row lane / 4, chunk lane % 4
                const int rowt = row0 + wr * HR + (lane >> 2);
                vT[pp] = colt < p.N ? (uint32_t)(((size_t)rowt * p.ldc + colt) * ES) : 0xFFFFFFF0u;
            }
            char* const turn = smem + 2 * BUF + wave * 2048;
            const int turn_w = (4 * r16 + 2 * (q & 1) + (q >> 1)) * 16, turn_r = lane * 16;
            u32x4 pend = {0u, 0u, 0u, 0u}, pend2 = {0u, 0u, 0u, 0u};
            u32x4 xhold = {0u, 0u, 0u, 0u};                               // GELU pre-activation stores (aux_out)
            uint32_t pend_v = 0xFFFFFFF0u, pend_s = 0u;
            auto swap2 = [](uint32_t& x, uint32_t& y) {
                const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
                x = r[0]; y = r[1];
            };
            // 1-bit ReLU record (ACT_RELU_MASK writes it, ACT_RELU_BWD_MASK reads it): one bit per output element in the order THIS lane
            // meets them: 32 bytes per lane and tile (two 16-byte accesses, lane-linear in memory: 8 KiB per tile).  Both products have
            // the same M and N, hence the same tiling and the same lane -> element map.  Register k of the record covers the steps
            // 4k .. 4k+3 = 16 packed bf16 pairs; pair i of them owns bit 15 - i (low element) and bit 31 - i (high element).
            constexpr bool MOUT = ACT == ACT_RELU_MASK, MIN = ACT == ACT_RELU_BWD_MASK;
            uint32_t mk[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            const size_t moff = ((size_t)tile * 256 + threadIdx.x) * 32;
            if constexpr (MIN) {
                const u32x4 m0 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.aux) + moff);
                const u32x4 m1 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.aux) + moff + 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) { mk[k] = m0[k]; mk[4 + k] = m1[k]; }
            }
            auto pos_pair = [](uint32_t w) -> uint32_t {   // (hipcc scalarises __builtin_elementwise_min on u16x2 into compares and selects)
                uint32_t r;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w & 0x7FFF7FFFu), "v"(0x00010001u));
                return r;
            };
            auto epi_steps = [&](auto drop_tag) {
                constexpr bool DROP = decltype(drop_tag)::value;
                constexpr int HM = MT / 2;                                // row tiles per operand-load batch
#pragma unroll
                for (int hm = 0; hm < 2; ++hm) {
                    u32x4 e16[HM][4];
#pragma unroll
                    for (int ii = 0; ii < HM; ++ii)
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) {
                            e16[ii][pp] = u32x4{};
                            if constexpr (EXT != EXT_NONE) e16[ii][pp] = buf_load<u32x4>(rE, vE[pp], (uint32_t)((16 * (HM * hm + ii)) * lde) * ES);
                        }
#pragma unroll
                    for (int ii = 0; ii < HM; ++ii)
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) {
                            const int i = HM * hm + ii;
                            const int row = roww + 16 * i;
                            uint32_t e0 = e16[ii][pp][0], e1 = e16[ii][pp][1], e2 = e16[ii][pp][2], e3 = e16[ii][pp][3];
                            if constexpr (EXT != EXT_NONE) { swap2(e0, e2); swap2(e1, e3); }
                            const uint32_t qa = (uint32_t)(((uint64_t)row * p.N + (colw + pp * 32)) >> 2);
                            PK pa, pb;
                            const int ms = i * 4 + pp;                    // step: pairs 4 (ms & 3) .. + 3 of record register ms >> 2
                            const PK oa = epi_quad<TO, ACT, EXT, DROP>(p, acc[i][2 * pp], b4[2 * pp], PK{e0, e1}, qa, drop_ks, drop_hwm, drop_th, drop_inv, pa);
                            const PK ob = epi_quad<TO, ACT, EXT, DROP>(p, acc[i][2 * pp + 1], b4[2 * pp + 1], PK{e2, e3}, qa + 4, drop_ks, drop_hwm, drop_th, drop_inv, pb);
                            uint32_t s0 = oa[0], s1 = oa[1], s2 = ob[0], s3 = ob[1];
                            if constexpr (MOUT) {
                                uint32_t r = mk[ms >> 2];
                                r = (r << 1) | pos_pair(s0); r = (r << 1) | pos_pair(s1); r = (r << 1) | pos_pair(s2); r = (r << 1) | pos_pair(s3);
                                mk[ms >> 2] = r;
                            }
                            if constexpr (MIN) {
                                const uint32_t r = mk[ms >> 2];
                                constexpr uint32_t LOHI = 0x00010001u;
                                const int i0 = 4 * (ms & 3);
                                s0 &= ((r >> (15 - i0)) & LOHI) * 0xFFFFu; s1 &= ((r >> (14 - i0)) & LOHI) * 0xFFFFu;
                                s2 &= ((r >> (13 - i0)) & LOHI) * 0xFFFFu; s3 &= ((r >> (12 - i0)) & LOHI) * 0xFFFFu;
                            }
                            swap2(s0, s2); swap2(s1, s3);
                            {
                                char* slot = turn + (ms & 1) * 1024;
                                *reinterpret_cast<u32x4*>(slot + turn_w) = u32x4{s0, s1, s2, s3};
                                // lanes read what OTHER lanes of the wave wrote: the pair must stay in this order (LDS operations of a
                                // wave execute in issue order; the fence keeps the compiler from moving the read above the write)
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                const u32x4 back = *reinterpret_cast<const u32x4*>(slot + turn_r);
                                if (ms > 0) buf_store(pend, rC, pend_v, pend_s);
                                // store-data hold (gemm256.hip, HAZARD): this step's ds_write data and the quad the previous step stored from
                                // stay untouched for five more states
                                asm volatile("s_nop 4" :: "v"(s0), "v"(s1), "v"(s2), "v"(s3),
                                             "v"(pend2[0]), "v"(pend2[1]), "v"(pend2[2]), "v"(pend2[3]) : "memory");
                                pend2 = pend; pend = back; pend_v = vT[pp]; pend_s = (uint32_t)((16 * i) * p.ldc) * ES;
                            }
                            if constexpr (ACT == ACT_GELU) {
                                if (p.aux_out) {
                                    uint32_t t0 = pa[0], t1 = pa[1], t2 = pb[0], t3 = pb[1];
                                    swap2(t0, t2); swap2(t1, t3);
                                    const u32x4 tq = u32x4{t0, t1, t2, t3};
                                    buf_store(tq, rX, vX[pp], (uint32_t)((16 * i) * p.ldaux) * ES);
                                    asm volatile("s_nop 3" :: "v"(tq[0]), "v"(tq[1]), "v"(tq[2]), "v"(tq[3]),
                                                 "v"(xhold[0]), "v"(xhold[1]), "v"(xhold[2]), "v"(xhold[3]) : "memory");
                                    xhold = tq;
                                }
                            }
                        }
                }
            };
            if (p.p_drop > 0.f) epi_steps(std::true_type{}); else epi_steps(std::false_type{});
            buf_store(pend, rC, pend_v, pend_s);
            tail_c = pend;
            if constexpr (MOUT) {
                const u32x4 m0 = {mk[0], mk[1], mk[2], mk[3]}, m1 = {mk[4], mk[5], mk[6], mk[7]};
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.aux_out) + moff) = m0;
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.aux_out) + moff + 16) = m1;
                tail_m = m0; tail_m2 = m1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        sbase += nk;
        tile += G;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the tail DMAs land in LDS nobody reads; retire them before the wave ends
}

// Same gates as gemm256 (gemm256.hip s2t_gemm256_try, which delegates here when the "gemm4w" option is on); `tiles` / `use192` come from it.
// Returns 0 for the variants this kernel does not take (the caller then launches gemm256).
int s2t_gemm4w_launch(const GemmArgs& a, int trans_b, int ext, bool use192, int tiles, hipStream_t st) {
    const int grid = tiles < 256 ? tiles : 256;
    const size_t lds = 2 * BUF + 8192;             // two K-tile buffers + the epilogue's lane-turn slots (4 waves x 2 KiB)
    bool done = false;
#define S2T_G4W(TB_, MT_, ACT_, EXT_)                                                                                        \
    if (!done && (trans_b != 0) == TB_ && use192 == (MT_ == 6) && a.act == ACT_ && ext == EXT_) {                            \
        static bool attr = false;                                                                                            \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<bf16, TB_, MT_, ACT_, EXT_>),    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }         \
        hipLaunchKernelGGL((gemm4w_kernel<bf16, TB_, MT_, ACT_, EXT_>), dim3(grid), dim3(256), lds, st, a);                  \
        done = true;                                                                                                         \
    }
#define S2T_G4W_MT(TB_, ACT_, EXT_) S2T_G4W(TB_, 8, ACT_, EXT_) S2T_G4W(TB_, 6, ACT_, EXT_)
    // NT forms only: the k-strided operand of the NN forms is read by ds_read_b64_tr_b16, which hipcc orders behind every LDS-DMA in
    // flight (s_waitcnt vmcnt(0) before each read: 2.6x slower here, where DMA and reads share a phase), and the 1-bit ReLU record
    // must have ONE layout for the NT product that writes it and the NN product that reads it: those stay with gemm256
    S2T_G4W_MT(false, ACT_NONE, EXT_NONE) S2T_G4W_MT(false, ACT_RELU, EXT_NONE) S2T_G4W_MT(false, ACT_GELU, EXT_NONE)
    S2T_G4W_MT(false, ACT_NONE, EXT_RES)
#undef S2T_G4W_MT
#undef S2T_G4W
    if (!done) return 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return S2T_EHIP(e);
    return 1;
}
