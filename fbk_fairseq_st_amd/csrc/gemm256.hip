// 256 x 256 x 64 MFMA GEMM for the big bf16 products of the Transformer blocks (M = tokens of the batch):
//   NT  Y  = epi(X  . W^T)   X [M][K], W [N][K]      forward projections / FFN      (F.linear, multihead_attention.py:190-208,
//   NN  dX = epi(dY . W)     dY [M][K], W [K][N]      their data gradients            transformer_layer.py:132-134 + autograd)
//
// Structure (cdna_hip_programming.md section 5, "256^2 8-phase template", rebuilt here from its description):
//   * 8 waves = 2 (M) x 4 (N), each 128 x 64 of the output as 8 x 4 tiles of v_mfma_f32_16x16x32_bf16: 128 accumulator VGPRs,
//     one workgroup per CU (two waves per SIMD);
//   * operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass; the ablation of
//     round 1 put 16 us of a 77 us launch into the register -> LDS writes).  The LDS image is lane-linear per wave-instruction
//     (1 KiB = 8 rows x 128 B, or 4 k-rows x 256 B for a k-strided operand), so the XOR swizzle that keeps the fragment reads
//     conflict-free sits on the per-lane SOURCE address (rule 21);
//   * two K-tile buffers of 64 KiB, each cut into four 16 KiB half-tiles (A-h0, A-h1, B-h0, B-h1).  A K-tile is two super-phases of
//     32 MFMAs per wave (two 64 x 32 quadrants of the wave's tile each; the B fragments of both halves stay in registers):
//         SPa (A-h0 x B-h0, A-h0 x B-h1)      SPb (A-h1 x B-h1, A-h1 x B-h0)            fragment reads per wave: 16 / 8
//     and every super-phase re-stages two half-tiles (four 1-KiB DMA instructions per wave):
//         SPa(t): A-h1(t+1), B-h1(t+1)        SPb(t): A-h0(t+2), B-h0(t+2)
//     ONE counted s_waitcnt vmcnt(4) per K-tile (end of SPb's memory segment: everything but the two half-tiles just issued has
//     landed = all of K-tile t+1), never vmcnt(0) in the loop; raw s_barrier (a __syncthreads() would drain the DMA queue).
//     (Round-2 timeline from in-kernel s_memtime stamps, make dbg + tools/gemm_timeline.py: a four-phase version of this loop --
//     16 MFMAs per barrier pair -- spent 120-250 cycles per barrier hand-off against a 256-cycle MFMA cluster; 32 MFMAs per
//     hand-off and 4 + 4 DMA instructions per K-tile took the big products from 0.95-1.0x to 1.3x the 128 x 128 kernels.)
//   * the two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run one barrier apart: while one group issues its LDS
//     reads and DMA the other runs its MFMA cluster, so the matrix pipe of a SIMD always has one wave's MFMAs to issue.
//   * persistent: one workgroup per CU walks the output tiles, the operand stream never stops at a tile boundary, the epilogue goes
//     straight from the accumulators to memory (buffer addressing, 16-byte stores after a v_permlane16_swap of neighbouring quads).
// Hazards, by barrier count (super-phase p: group 0 runs its memory segment in barrier interval 2p and its MFMAs in 2p+1, group 1
// one interval later).  Write-after-read: A-h0 / B-h0 (read in SPa's memory segment) and A-h1 (SPb's memory segment) are re-staged in
// the NEXT super-phase, which is safe because every wave waits for those reads to return (lgkmcnt(0)) BEFORE the barrier that ends
// its memory segment: group 1's wait precedes barrier 2p+2, group 0's first DMA into the region issues in interval 2p+2.  B-h1 is
// read at the head of SPa's MFMA cluster (NT form) and re-staged two super-phases later (SPa of the next K-tile: interval 2p+4 >
// barrier 2p+3 that both groups' clusters precede).  Read-after-write: the wait that retires K-tile t+1 sits in SPb(t)'s memory
// segment (both groups pass it before barrier 2p+2); its first reader is SPa(t+1)'s memory segment (interval 2p+2 and later).
#include "common.hpp"
#include "prof.hpp"
#include "gemm_epilogue.hpp"
#include <type_traits>

#include "gemm_tile.hpp"

// TB = false: B stored [N][K] (k contiguous);  TB = true: B stored [K][N] (k strided, transposed LDS reads)
// MT = 16-row tiles per wave along M: 8 (256-row tile) or 6 (192-row tile: picked when it fills the 256 CUs better, e.g. N = 512 at
// M = 24,000: 250 tiles of 192 x 256 instead of 188 of 256 x 256)
// Persistent: gridDim.x workgroups (one per CU) walk the output tiles; the operand stream never stops at a tile boundary (the
// half-tiles staged in the last two K-tiles of a tile are the first ones of the next tile), and the stores of a tile drain under
// the next tile's K-loop.
#ifdef S2T_G256_STAGGER
// "CUs out of phase" experiment (profiles/r06_gemm_stagger.txt): workgroups with an odd CU index inside their XCD
// wait this many ticks of the 100 MHz real-time counter before their first tile, so that their store bursts fall into the other half's K-loops
__device__ long long g_g256_stagger = 0;
extern "C" int s2t_g256_set_stagger(long long ticks) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_g256_stagger), &ticks, sizeof(ticks)) == hipSuccess ? 0 : -1;
}
#endif
template <typename TO, bool TB, int MT, int ACT, int EXT, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QM = MT / 2;                       // row tiles of one phase's quadrant (4 / 3)
    constexpr int HR = 16 * QM;                      // rows a wave group owns in one A half-tile (64 / 48)
    constexpr int BM = 4 * HR, BN = 256;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, tiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    int tile = xcd_remap(blockIdx.x, G);             // then tile += G: every round is a contiguous run of tiles, an XCD's share contiguous inside it
    if (tile >= tiles) return;
#ifdef S2T_G256_STAGGER                                 /* experiment twin (tools/gemm_stagger.py): every other CU of an XCD starts late */
    if (g_g256_stagger > 0 && ((blockIdx.x >> 3) & 1)) {
        const long long t0 = wall_clock64();
        while ((long long)wall_clock64() - t0 < g_g256_stagger) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const int nk = p.K / BK;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, q = lane >> 4;
    const int grp = wr;                                                // the stagger group = the M half this wave owns

    // ---- staging: this wave fills pieces 2*wave and 2*wave+1 (1 KiB each) of every half-tile.  An A half-tile = 2 * HR rows =
    // HR / 4 pieces of 8 rows: 16 pieces (two per wave) at MT = 8; 12 at MT = 6 (two for waves 0-3, one for waves 4-7).
    // B half-tile h = the 128 columns [128 h, 128 h + 128) of the tile; wave column wc owns columns 32 wc .. 32 wc + 31 of each.
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Bb = reinterpret_cast<const char*>(p.B);
    constexpr bool A2 = (MT == 8);
    const bool a_two = A2 || wave < 4;                                 // wave-uniform
    struct Offs { uint32_t a[2][2], b[2][2]; };                        // [half][piece] byte offsets of this lane's 16 bytes at k = 0
    auto offsets = [&](int tl, Offs& o) {
        const int row0 = (tl / tiles_n) * BM, col0 = (tl % tiles_n) * BN;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = 2 * wave + i;
                {   // A: image row r' = 8 piece + lane/8 <-> tile row (r'/HR) * 2 HR + h * HR + r' % HR
                    const int pa = A2 ? piece : (wave < 4 ? piece : 8 + (wave - 4));
                    const int rp = 8 * pa + (lane >> 3), pos = lane & 7;
                    const int trow = (rp / HR) * (2 * HR) + h * HR + (rp % HR);
                    const int gr = min(row0 + trow, p.M - 1);
                    o.a[h][i] = (uint32_t)(((size_t)gr * p.lda + ((pos ^ (rp & 7)) << 3)) * 2);
                }
                if constexpr (!TB) {   // B [N][K]: image row r' <-> tile column 128 h + r'
                    const int rp = 8 * piece + (lane >> 3), pos = lane & 7;
                    const int gc = min(col0 + 128 * h + rp, p.N - 1);
                    o.b[h][i] = (uint32_t)(((size_t)gc * p.ldb + ((pos ^ (rp & 7)) << 3)) * 2);
                } else {               // B [K][N]: image k-row 4 piece + lane/16, image column c' <-> tile column 128 h + c'
                    const int kr = 4 * piece + (lane >> 4), pos = lane & 15;
                    const int cp = (pos ^ trswz(kr)) << 3;
                    const int gc = min(col0 + 128 * h + cp, ((p.N + 7) & ~7) - 8);
                    o.b[h][i] = (uint32_t)(((size_t)kr * p.ldb + gc) * 2);
                }
            }
    };
    Offs cur, nxt;
    offsets(tile, cur);
    const uint32_t kstepA = BK * 2, kstepB = TB ? (uint32_t)BK * (uint32_t)p.ldb * 2u : (uint32_t)BK * 2u;
    const int a_dst = A2 ? wave * 2048 : (wave < 4 ? wave * 2048 : 8192 + (wave - 4) * 1024);
    int sbase = 0;                                                      // K-tiles consumed by earlier tiles: LDS buffer parity of the stream
    bool has_next = false;
    // stream position u = t + 1 or t + 2 of the CURRENT tile: past its last K-tile it is K-tile u - nk of the next tile (nk >= 2);
    // with no next tile the source is clamped to the last K-tile (in bounds) and the destination stays the half-tile the schedule
    // says is free: the DMA count per phase is a constant and nothing reads those bytes afterwards
    // The DMA instructions are buffer-addressed (buffer_load_dwordx4 ... offen lds): the lane's byte offset inside the operand is the
    // 32-bit voffset as it stands in `cur` / `nxt`, the K-tile's advance is the scalar soffset -- no 64-bit per-lane address to form
    // and to hand to the memory pipeline per instruction (a global_load_lds costs ~60 cycles of the wave's issue time, and issue slots
    // are what bounds this loop: tools/gemm_x_time.py on the -DS2T_X twins).  -DS2T_DMA_FLAT: the global_load_lds form, for A/B runs.
#ifndef S2T_DMA_FLAT
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((size_t)p.M * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0,
                                                                         (int)((size_t)(TB ? p.K : p.N) * p.ldb * 2), 0x00020000);
#define S2T_DMA(RS_, BASE_, VO_, SO_, DST_) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_, (lds_void*)(DST_), 16, VO_, SO_, 0, 0)
#else
#define S2T_DMA(RS_, BASE_, VO_, SO_, DST_) glds16((BASE_) + (VO_) + (SO_), DST_)
#endif
    auto stageA = [&](int h, int u) {
        char* dst = smem + __builtin_amdgcn_readfirstlane(((sbase + u) & 1) * BUF + h * HALF + a_dst);
        const bool roll = u >= nk && has_next;
        const uint32_t ko = (uint32_t)(roll ? u - nk : min(u, nk - 1)) * kstepA;
        const uint32_t o0 = roll ? nxt.a[h][0] : cur.a[h][0], o1 = roll ? nxt.a[h][1] : cur.a[h][1];
        S2T_DMA(rA, Ab, o0, ko, dst);
        if (a_two) S2T_DMA(rA, Ab, o1, ko, dst + 1024);
    };
    auto stageB = [&](int h, int u) {
        char* dst = smem + __builtin_amdgcn_readfirstlane(((sbase + u) & 1) * BUF + 2 * HALF + h * HALF + wave * 2048);
        const bool roll = u >= nk && has_next;
        const uint32_t ko = (uint32_t)(roll ? u - nk : min(u, nk - 1)) * kstepB;
        const uint32_t o0 = roll ? nxt.b[h][0] : cur.b[h][0], o1 = roll ? nxt.b[h][1] : cur.b[h][1];
        S2T_DMA(rB, Bb, o0, ko, dst);
        S2T_DMA(rB, Bb, o1, ko, dst + 1024);
    };
    // ONE counted wait per K-tile (never 0 in the loop), at the end of SPb's memory segment: only the half-tiles issued in that
    // segment (A-h0, B-h0 of K-tile t+2: 4 DMA instructions, 3 for a wave that stages one A piece) may still be in flight, so all of
    // K-tile t+1 has landed
#define S2T_WAIT_TILE() do { if (a_two) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } while (0)
    // SCHED 1: per K-tile a wave issues G1 = A-h0, B-h0, B-h1 (6 DMA instructions, 5 for a wave with one A piece) and G2 = A-h1 (2 / 1);
    // both of its waits leave exactly one G1 and one G2 in flight (see the schedule below)
#define S2T_WAIT_PIPE() do { if (a_two) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } while (0)
    // reads of a half-tile that is re-staged in the very next super-phase must have RETURNED before this wave passes the barrier
#define S2T_READS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    f32x4 acc[MT][4];                                                    // never zeroed: see FIRST below

    // fragment addresses (bytes inside a half-tile)
    const int swz0 = ((0 + q) ^ (r16 & 7)) << 4, swz1 = ((4 + q) ^ (r16 & 7)) << 4;
    const int a_off = (wr * HR + r16) * 128;                            // + i * 2048
    const int b_off = (wc * 32 + r16) * 128;                            // + j * 2048 (direct image)

    u32x4 fa[QM][2], fb[2][2][2];                                        // fb[half][column tile][k-half]: both B halves stay in registers
    // fragment reads of one phase, k-half 0 first: the MFMAs of k-half 0 start when those have landed (counted lgkmcnt, placed by
    // the compiler) while the k-half 1 fragments are still on their way
    auto readAs = [&](const char* buf, int h, int s_) {
        const char* base = buf + h * HALF + a_off + (s_ ? swz1 : swz0);
#pragma unroll
        for (int i = 0; i < QM; ++i) fa[i][s_] = *reinterpret_cast<const u32x4*>(base + i * 2048);
    };
    u32x4 fa0[QM], fa1[QM];                                             // SCHED 1: the two A fragment sets (this stage's / the next one's)
    auto readA1 = [&](const char* buf, int h, int s_, u32x4 (&dst)[QM]) {
        const char* base = buf + h * HALF + a_off + (s_ ? swz1 : swz0);
#pragma unroll
        for (int i = 0; i < QM; ++i) dst[i] = *reinterpret_cast<const u32x4*>(base + i * 2048);
    };
#ifndef S2T_G256_TR_ASM
#define S2T_G256_TR_ASM 1
#endif
    // TB: per-lane LDS addresses of the k-strided operand's fragments in buffer 0 ([column tile][inner half]); buffer parity toggles
    // bit 16, half-tile and k-half are the instruction's immediate offset (gemm_tile.hpp tr_read_asm)
    uint32_t aB[2][2] = {{0u, 0u}, {0u, 0u}};
    if constexpr (TB && S2T_G256_TR_ASM) {
        const uint32_t s0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) aB[j][hh] = tr_lane_addr(s0, wc * 32 + 16 * j, hh, r16, q);
    }
    auto readBs = [&](const char* buf, int h, int s_) {
        const char* base = buf + 2 * HALF + h * HALF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (!TB) fb[h][j][s_] = *reinterpret_cast<const u32x4*>(base + b_off + j * 2048 + (s_ ? swz1 : swz0));
            else if constexpr (!S2T_G256_TR_ASM) fb[h][j][s_] = tr_frag(base, wc * 32 + 16 * j, s_, r16, q);
            else {
                const uint32_t bb = (uint32_t)(buf - smem);                 // 0 or BUF: wave-uniform
                u32x2 w0, w1;
                if (h == 0 && s_ == 0) { w0 = tr_read_asm<2 * HALF>(aB[j][0] ^ bb); w1 = tr_read_asm<2 * HALF>(aB[j][1] ^ bb); }
                else if (h == 0) { w0 = tr_read_asm<2 * HALF + 8192>(aB[j][0] ^ bb); w1 = tr_read_asm<2 * HALF + 8192>(aB[j][1] ^ bb); }
                else if (s_ == 0) { w0 = tr_read_asm<3 * HALF>(aB[j][0] ^ bb); w1 = tr_read_asm<3 * HALF>(aB[j][1] ^ bb); }
                else { w0 = tr_read_asm<3 * HALF + 8192>(aB[j][0] ^ bb); w1 = tr_read_asm<3 * HALF + 8192>(aB[j][1] ^ bb); }
                fb[h][j][s_] = u32x4{w0[0], w0[1], w1[0], w1[1]};
            }
        }
    };
#define S2T_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifdef S2T_G256_STAMPS
    // diagnostic build only (make dbg; tools/gemm_timeline.py): s_memtime before and after the MFMA cluster of every phase of
    // workgroup 0, waves 0 and 4, into the buffer passed as aux_out.  Perturbs the schedule (each stamp waits for its own return).
    unsigned long long* const DBG = reinterpret_cast<unsigned long long*>(p.aux_out);
    int dbg_n = 0;
    const bool dbg_on = DBG && blockIdx.x == 0 && (wave & 3) == 0;
#define S2T_STAMP(K_)                                                                                        \
    if (dbg_on && dbg_n < 120) {                                                                             \
        unsigned long long t_;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                         \
        if (lane == 0) DBG[(wave >> 2) * 512 + 2 * dbg_n + (K_)] = t_;                                       \
        dbg_n += (K_);                                                                                       \
    }
    unsigned long long tm0_ = 0, tm1_ = 0, tm2_ = 0;
#define S2T_MT(V_) if (dbg_on) asm volatile("s_memtime %0" : "=s"(V_) :: "memory");
#define S2T_MEM_END()                                                                                        \
    if (dbg_on && dbg_n < 120) {                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tm0_), "+s"(tm1_), "+s"(tm2_) :: "memory");               \
        if (lane == 0) { DBG[1024 + (wave >> 2) * 512 + 3 * dbg_n] = tm0_; DBG[1024 + (wave >> 2) * 512 + 3 * dbg_n + 1] = tm1_;  \
                         DBG[1024 + (wave >> 2) * 512 + 3 * dbg_n + 2] = tm2_; }                             \
    }
    // epilogue stamps (round 5): six s_memtime values per tile boundary, collected in SGPRs and written at the next loop top (the
    // counted lgkmcnt waits hipcc places in the epilogue do not know about them: TIMING ONLY, a stamped build's results are not valid)
    unsigned long long es_[7] = {0, 0, 0, 0, 0, 0, 0};
    int ep_n = 0;
#define S2T_ES(I_) if (dbg_on) asm volatile("s_memtime %0" : "=s"(es_[I_]) :: "memory");
#define S2T_ES_FLUSH()                                                                                       \
    if (dbg_on && ep_n < 8) {                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(es_[0]), "+s"(es_[1]), "+s"(es_[2]), "+s"(es_[3]), "+s"(es_[4]), "+s"(es_[5]), "+s"(es_[6]) :: "memory"); \
        if (lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 7; ++i_) DBG[2048 + (wave >> 2) * 256 + 8 * ep_n + i_] = es_[i_]; }  \
        ++ep_n;                                                                                              \
    }
#else
#define S2T_STAMP(K_)
#define S2T_MT(V_)
#define S2T_MEM_END()
#define S2T_ES(I_)
#define S2T_ES_FLUSH()
#endif
    // FIRST (the first K-tile of an output tile): the first MFMA of every accumulator takes the constant 0 as its C operand, so no
    // accumulator is ever zeroed by moves (128 v_mov per lane and tile, which the compiler emitted twice at the loop header)
#define S2T_QUAD(MI, NI)                                                                                     \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                        \
            _Pragma("unroll") for (int i = 0; i < QM; ++i)                                                   \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
                    acc[QM * (MI) + i][2 * (NI) + j] = mma16<bf16>(fb[NI][j][s], fa[i][s],                   \
                        (FIRST && s == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[QM * (MI) + i][2 * (NI) + j]);
#define S2T_MMA2(MI, NA, NB)                                                                                 \
    do {                                                                                                     \
        S2T_STAMP(0)                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        S2T_QUAD(MI, NA) S2T_QUAD(MI, NB)                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        S2T_STAMP(1)                                                                                         \
    } while (0)

    if constexpr (SCHED == 0) {
    // ---- prologue: K-tile 0 and what the steady state stages in SPb of "K-tile -1" (A-h0, B-h0 of K-tile 1)
    stageA(0, 0); stageB(0, 0); stageB(1, 0); stageA(1, 0);
    stageA(0, 1); stageB(0, 1);
    S2T_WAIT_TILE();
    S2T_BAR();
    if (grp == 1) S2T_BAR();                                            // group 1 runs one barrier behind from here on
    } else {
    // ---- SCHED 1 prologue: both K-tile buffers, each as the steady state issues them (G1 = A-h0, B-h0, B-h1; G2 = A-h1)
    stageA(0, 0); stageB(0, 0); stageB(1, 0); stageA(1, 0);
    stageA(0, 1); stageB(0, 1); stageB(1, 1); stageA(1, 1);
    S2T_WAIT_PIPE();                                                    // all but the youngest G1 + G2: K-tile 0 has landed
    S2T_BAR();
    }

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
    // the data registers of a tile's last two 16-byte stores (the last C quad, the 1-bit record): held over the loop's back edge until
    // the next tile's offset arithmetic is done -- they are dead after the epilogue and would otherwise be the first registers that
    // arithmetic writes, within a few cycles of the stores (store-data hazard: see the epilogue's ring)
    u32x4 tail_c = {0u, 0u, 0u, 0u}, tail_m = {0u, 0u, 0u, 0u};
    for (;;) {
        has_next = tile + G < tiles;
        if (has_next) offsets(tile + G, nxt);
        S2T_ES(6) S2T_ES_FLUSH()
#ifndef S2T_NO_HOLD
        asm volatile("" :: "v"(tail_c[0]), "v"(tail_c[1]), "v"(tail_c[2]), "v"(tail_c[3]),
                           "v"(tail_m[0]), "v"(tail_m[1]), "v"(tail_m[2]), "v"(tail_m[3]) : "memory");
#endif
        auto ktile = [&](int t, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const char* buf = smem + ((sbase + t) & 1) * BUF;
            // SPa: quadrants (A-h0 x B-h0), (A-h0 x B-h1).  The B-h1 fragments are read at the head of the MFMA cluster (their latency
            // hides under the first quadrant's MFMAs).  Re-stage, in the OTHER buffer: A-h1 (last read in SPb of K-tile t-1, reads
            // returned before that segment's barrier) and B-h1 (last read in the MFMA cluster of SPa of K-tile t-1) with K-tile t+1.
            S2T_MT(tm0_) readBs(buf, 0, 0); readAs(buf, 0, 0); readBs(buf, 0, 1); readAs(buf, 0, 1);
            if constexpr (TB) { readBs(buf, 1, 0); readBs(buf, 1, 1); }    // transposed reads: two instructions per fragment, kept out of the MFMA cluster
            S2T_MT(tm1_) stageA(1, t + 1); stageB(1, t + 1);
            S2T_MT(tm2_) S2T_MEM_END()
            S2T_READS_DONE();
            S2T_BAR();
            if constexpr (!TB) { readBs(buf, 1, 0); readBs(buf, 1, 1); }
            S2T_MMA2(0, 0, 1);
            S2T_BAR();
            // SPb: quadrants (A-h1 x B-h1), (A-h1 x B-h0) on the B fragments still in registers.  Re-stage, in THIS buffer: A-h0 and B-h0
            // (read in SPa's memory segment just before: returned before its barrier) with K-tile t+2.
            S2T_MT(tm0_) readAs(buf, 1, 0); readAs(buf, 1, 1);
            S2T_MT(tm1_) stageA(0, t + 2); stageB(0, t + 2);
            S2T_MT(tm2_) S2T_MEM_END()
            S2T_READS_DONE();
            S2T_WAIT_TILE();
            S2T_BAR();
            S2T_MMA2(1, 1, 0);
            S2T_BAR();
        };
        // ---- SCHED 1: the same LDS image and the same staging, another schedule.  All eight waves run ONE program, software-pipelined
        // at 16-MFMA stages, with the fragments of the next stage in flight under the MFMAs of the current one (two A fragment sets,
        // both k-halves of the B fragments kept: the same 64 fragment registers as the phases above) -- no wave group waits for the
        // other, the two waves of a SIMD fall into step by themselves, and a barrier only separates "every wave has read half-tile
        // X" from "X is re-staged" (twice per K-tile):
        //     st0  MFMA A-h0.s0 x B.s0   | reads A-h0.s1, B.s1
        //     st1  MFMA A-h0.s1 x B.s1   | B1; reads A-h1.s0; DMA G1(t+2) = A-h0, B-h0 (B-h1 in st2)      A-h0(t), B(t): all read
        //     st2  MFMA A-h1.s0 x B.s0   | reads A-h1.s1; DMA B-h1(t+2)
        //     st3  MFMA A-h1.s1 x B.s1   | B2; reads A-h0.s0, B.s0 of K-tile t+1; DMA G2(t+2) = A-h1  A-h1(t): all read
        // Read-after-write: a wave's wait before B2(t) leaves one G1 and one G2 of its own in flight, i.e. G1 of K-tile t+1 has landed
        // (its first reader is st3(t), behind B2(t)); the wait before B1(t+1) does the same for G2 of K-tile t+1 (A-h1(t+1), first read
        // in st1(t+1) behind B1(t+1)).  Every DMA has six to eight stages (1.5 - 2 K-tiles) to land.
        // diagnostic twins only (make x X=<mask>; tools/gemm_x_time.py): -DS2T_X bit 0 drops the loop's barriers, bit 1 its DMA, bit 2 its
        // fragment reads -- WRONG results, timing only: what each of them costs the schedule
#ifndef S2T_X
#define S2T_X 0
#endif
#define S2T_XBAR() do { if (!(S2T_X & 1)) { S2T_BAR(); } } while (0)
#define S2T_XDMA(...) if (!(S2T_X & 2)) { __VA_ARGS__ }
#define S2T_XRD(...) if (!(S2T_X & 4)) { __VA_ARGS__ }
        auto ktile1 = [&](int t, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const char* buf = smem + ((sbase + t) & 1) * BUF;
            const char* nbuf = smem + ((sbase + t + 1) & 1) * BUF;
#define S2T_STAGE(MI, S_, FA)                                                                                 \
            _Pragma("unroll") for (int NI = 0; NI < 2; ++NI)                                                 \
                _Pragma("unroll") for (int i = 0; i < QM; ++i)                                               \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                            \
                        acc[QM * (MI) + i][2 * NI + j] = mma16<bf16>(fb[NI][j][S_], FA[i],                   \
                            (FIRST && (S_) == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[QM * (MI) + i][2 * NI + j]);
            // sched_barrier(0) pins the order the stages are written in: hipcc otherwise sinks a stage's MFMAs below the hand-placed
            // waits of the next one (register-only instructions are not ordered by an asm "memory" clobber: rule 18) and issues the
            // fragment reads late, right in front of the wait for them.  Inside a stage the DMA instructions go between MFMA groups.
#define S2T_SB() __builtin_amdgcn_sched_barrier(0)
#define S2T_MIX(NMFMA, NDMA)                                                                                  \
            _Pragma("unroll") for (int g_ = 0; g_ < (NDMA); ++g_) {                                          \
                __builtin_amdgcn_sched_group_barrier(0x008, (NMFMA), 0);                                     \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                           \
            }
            // st0 (its fragments -- fa0 = A-h0.s0, fb[.][.][0] -- were requested in st3 of the K-tile before, or at the tile's start: a
            // stage ago.  The explicit wait is a builtin, not asm: hipcc's own counter model then knows they have landed and does not
            // wait for the reads issued right below before the first MFMA)
            S2T_SB();
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0)
            S2T_XRD(readA1(buf, 0, 1, fa1); readBs(buf, 0, 1); readBs(buf, 1, 1);)
            S2T_SB();
            S2T_STAGE(0, 0, fa0)
            S2T_SB();
            // st1
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // A-h0(t) and B(t) are in this wave's registers
            S2T_WAIT_PIPE();
            S2T_XBAR();                                         // B1
            S2T_XRD(readA1(buf, 1, 0, fa0);)
            S2T_SB();
            S2T_XDMA(stageA(0, t + 2); stageB(0, t + 2);)
            S2T_STAGE(0, 1, fa1)
            S2T_MIX(QM, 4)
            S2T_SB();
            // st2
            __builtin_amdgcn_s_waitcnt(0xC07F);
            S2T_XRD(readA1(buf, 1, 1, fa1);)
            S2T_SB();
            S2T_XDMA(stageB(1, t + 2);)
            S2T_STAGE(1, 0, fa0)
            S2T_MIX(2 * QM, 2)
            S2T_SB();
            // st3
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // A-h1(t) too
            S2T_WAIT_PIPE();
            S2T_XBAR();                                         // B2
            S2T_XRD(if (t + 1 < nk) { readA1(nbuf, 0, 0, fa0); readBs(nbuf, 0, 0); readBs(nbuf, 1, 0); })
            S2T_SB();
            S2T_XDMA(stageA(1, t + 2);)
            S2T_STAGE(1, 1, fa1)
            S2T_MIX(2 * QM, 2)
            S2T_SB();
#undef S2T_MIX
#undef S2T_SB
#undef S2T_STAGE
        };
        if constexpr (SCHED == 0) {
        ktile(0, std::true_type{});
        for (int t = 1; t < nk; ++t) ktile(t, std::false_type{});
        } else {
        {   // the tile's first fragments: K-tile 0 is visible (prologue barrier, or B2 of the tile before)
            const char* buf = smem + (sbase & 1) * BUF;
            readA1(buf, 0, 0, fa0); readBs(buf, 0, 0); readBs(buf, 1, 0);
        }
        ktile1(0, std::true_type{});
        for (int t = 1; t < nk; ++t) ktile1(t, std::false_type{});
        }
        // ---- this tile's epilogue (no barrier inside: the other group is one interval away in its own stream).  Running the two groups'
        // epilogues in the SAME interval (group 0 idling through group 1's last cluster, group 1 closing an extra interval after its own)
        // was tried: 7,800 -> 7,000 cycles per boundary, but the masked (per-quad) epilogue then stored a few wrong values per launch --
        // with AND without the lane turn below, non-deterministically, in the same lanes (rows 4 a + 3 of a 16-row block, first dword of
        // a 16-byte chunk); not understood.  With one epilogue per SIMD at a time it does not happen: tools/gemm_turn_check.py holds
        // this arrangement to a twin built with -DS2T_NOTURN bit for bit
        __builtin_amdgcn_sched_barrier(0);
        S2T_ES(0) S2T_ES(1)
        {
            typedef typename Pack4<TO>::type PK;
            constexpr uint32_t ES = sizeof(TO);
            const int row0 = (tile / tiles_n) * BM, col0 = (tile % tiles_n) * BN;
            const int colw = col0 + wc * 32 + 4 * q;                      // column of this lane's quad in tile j: + 128 (j >> 1) + 16 (j & 1)
            const int roww = row0 + wr * (2 * HR) + r16;                  // + HR hm + 16 ii
            f32x4 b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = colw + (j >> 1) * 128 + 16 * (j & 1);
                b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.bias && col + 4 <= p.N) b4[j] = *reinterpret_cast<const f32x4*>(p.bias + col);
                else if (p.bias && col < p.N) {                            // N not a multiple of 4 (a vocabulary of 5,001): the last quad, element by element
#pragma unroll
                    for (int e = 0; e < 4; ++e) b4[j][e] = col + e < p.N ? p.bias[col + e] : 0.f;
                }
            }
            static_assert(sizeof(TO) == 2, "bf16 outputs");
            {
                // bf16: a store of one quad is 8 bytes per lane = sixteen 32-byte row segments per wave-instruction, and the store tail
                // is bound by the NUMBER of such instructions (cdna_hip_programming.md T21).  The quads of two neighbouring 16-column
                // tiles are exchanged between lane rows q and q ^ 1 (v_permlane16_swap) so that every lane holds 8 consecutive columns:
                // 16-byte accesses, half the instructions, 64-byte row segments.  Operand loads come in the same shape and are swapped back.
                uint32_t vC[2], vE[2], vX[2];                             // per-lane byte offsets of (roww, this lane's 8 columns of pair pp)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int col = col0 + wc * 32 + pp * 128 + 16 * (q & 1) + 8 * (q >> 1);
                    const bool ok = col < p.N;
                    vC[pp] = ok ? (uint32_t)(((size_t)roww * p.ldc + col) * ES) : 0xFFFFFFF0u;
                    vE[pp] = ok ? (uint32_t)(((size_t)roww * lde + col) * ES) : 0xFFFFFFF0u;
                    vX[pp] = ok ? (uint32_t)(((size_t)roww * p.ldaux + col) * ES) : 0xFFFFFFF0u;
                }
                // The C stores leave in a TURNED lane order.  In the accumulator order the four lanes that share a 64-byte row segment
                // are 16 lanes apart and the memory pipeline takes ~37 cycles per wave-store (tools/store_probe.hip); with four
                // CONSECUTIVE lanes per segment it takes ~10.  The turn is a lane permutation (lane r16 + 16 q -> lane 4 r16 + chunk(q)) through
                // a wave-private 1 KiB LDS slot (ds_write_b128 at the turned position, ds_read_b128 at the own one: LDS operations of a
                // wave execute in order, no barrier), on the LDS pipe, which is idle in the epilogue; the store of a step is issued one step
                // later so that its read-back has returned.
                char* const turn = smem + 2 * BUF + wave * 2048;
                const int turn_w = (4 * r16 + 2 * (q & 1) + (q >> 1)) * 16, turn_r = lane * 16;
                uint32_t vT[2];                                           // byte offsets in the turned order: row lane / 4, chunk lane % 4
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int col = col0 + wc * 32 + pp * 128 + 8 * (lane & 3);
                    const int row = row0 + wr * (2 * HR) + (lane >> 2);
                    vT[pp] = col < p.N ? (uint32_t)(((size_t)row * p.ldc + col) * ES) : 0xFFFFFFF0u;
                }
                // HAZARD (round 4; tools/gemm_sched_diff.py, tests/test_kernels_gpu.py::test_gemm256_store_data_hazard_twins): the data
                // registers of a 16-byte LDS write or buffer store are read out over several cycles AFTER the instruction has issued,
                // and a VALU write to them in that window lands in the stored data (raw f32 intermediates in the output: the last
                // lanes of every 16-lane row, which are read last).  hipcc pads two wait states after a wide buffer store and none
                // after a wide ds_write -- it emitted `ds_write_b128 v156, v[148:151]` / `v_mov_b32 v148, v251` back to back -- and
                // with the SIMD partner storing at the same time the window gets longer (round 3's "wrong values when the two
                // groups' epilogues overlap").  So every step ends in an asm that READS this step's ds_write data and the quad the
                // PREVIOUS step stored from (inputs only: nothing is redefined, hipcc merely cannot touch those registers earlier)
                // and waits five states: a ds_write's data survives the read-back and store issue that follow it, a store's data one
                // whole step.  -DS2T_NO_HOLD drops it (the reproducer's failing arm).
                u32x4 pend = {0u, 0u, 0u, 0u}, pend2 = {0u, 0u, 0u, 0u};
                u32x4 xhold[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};          // GELU pre-activation stores (aux_out), see there
                uint32_t pend_v = 0xFFFFFFF0u, pend_s = 0u;
                auto swap2 = [](uint32_t& x, uint32_t& y) {
                    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
                    x = r[0]; y = r[1];
                };
                // 1-bit ReLU record (ACT_RELU_MASK writes it, ACT_RELU_BWD_MASK reads it): one bit per output element in the order
                // THIS lane meets them: 16 bytes per lane and tile, lane-linear in memory.  Both products have the same M and N, hence
                // the same tiling and the same lane -> element map: no exchange, one 16-byte access.  Register k of the record covers
                // the steps 4k .. 4k+3 = 16 packed bf16 pairs; pair i of them owns bit 15 - i (low element) and bit 31 - i (high
                // element): written as rec = rec << 1 | min(pair & 0x7FFF7FFF, 0x00010001) (3 VALU per pair), applied as
                // pair &= ((rec >> (15 - i)) & 0x00010001) * 0xFFFF (4 VALU per pair; the stored activation is never negative).
                constexpr bool MOUT = ACT == ACT_RELU_MASK, MIN = ACT == ACT_RELU_BWD_MASK;
                u32x4 mk = {0u, 0u, 0u, 0u};
                const size_t moff = ((size_t)tile * 512 + threadIdx.x) * 16;
                if constexpr (MIN) mk = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.aux) + moff);
                auto pos_pair = [](uint32_t w) -> uint32_t {   // (hipcc scalarises __builtin_elementwise_min on u16x2 into compares and selects)
                    uint32_t r;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w & 0x7FFF7FFFu), "v"(0x00010001u));
                    return r;
                };
                // one straight-line body per dropout setting: with the test inside epi_quad every quad was its own basic block and the
                // steps of a tile could not be interleaved by the scheduler (a wave is alone on its SIMD here: the other group is in its MFMAs)
                auto epi_steps = [&](auto drop_tag) {
                constexpr bool DROP = decltype(drop_tag)::value;
#pragma unroll
                for (int hm = 0; hm < 2; ++hm) {
                    u32x4 e16[QM][2];
#pragma unroll
                    for (int ii = 0; ii < QM; ++ii)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) {
                            e16[ii][pp] = u32x4{};
                            if constexpr (EXT != EXT_NONE) e16[ii][pp] = buf_load<u32x4>(rE, vE[pp], (uint32_t)((hm * HR + 16 * ii) * lde) * ES);
                        }
#pragma unroll
                    for (int ii = 0; ii < QM; ++ii)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) {
                            const int row = roww + hm * HR + 16 * ii;
                            uint32_t e0 = e16[ii][pp][0], e1 = e16[ii][pp][1], e2 = e16[ii][pp][2], e3 = e16[ii][pp][3];
                            if constexpr (EXT != EXT_NONE) { swap2(e0, e2); swap2(e1, e3); }
                            const uint32_t qa = (uint32_t)(((uint64_t)row * p.N + (colw + pp * 128)) >> 2);
                            PK pa, pb;
                            const int ms = (QM * hm + ii) * 2 + pp;              // step: pairs 4 (ms & 3) .. + 3 of record register ms >> 2
                            const PK oa = epi_quad<TO, ACT, EXT, DROP>(p, acc[QM * hm + ii][2 * pp], b4[2 * pp], PK{e0, e1}, qa, drop_ks, drop_hwm, drop_th, drop_inv, pa);
                            const PK ob = epi_quad<TO, ACT, EXT, DROP>(p, acc[QM * hm + ii][2 * pp + 1], b4[2 * pp + 1], PK{e2, e3}, qa + 4, drop_ks, drop_hwm, drop_th, drop_inv, pb);
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
#ifdef S2T_NOTURN                                                    /* diagnostic twin: the stores in accumulator order (tools/gemm_turn_check.py) */
                            buf_store(u32x4{s0, s1, s2, s3}, rC, vC[pp], (uint32_t)((hm * HR + 16 * ii) * p.ldc) * ES);
#else
                            {
                                char* slot = turn + (ms & 1) * 1024;
                                *reinterpret_cast<u32x4*>(slot + turn_w) = u32x4{s0, s1, s2, s3};
                                // lanes read what OTHER lanes of the wave wrote: the pair must stay in this order (LDS operations of a
                                // wave execute in issue order; the fence keeps the compiler from moving the read above the write)
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                const u32x4 back = *reinterpret_cast<const u32x4*>(slot + turn_r);
                                if (ms > 0) buf_store(pend, rC, pend_v, pend_s);
#ifndef S2T_NO_HOLD
                                asm volatile("s_nop 4" :: "v"(s0), "v"(s1), "v"(s2), "v"(s3),
                                             "v"(pend2[0]), "v"(pend2[1]), "v"(pend2[2]), "v"(pend2[3]) : "memory");
#endif
                                pend2 = pend; pend = back; pend_v = vT[pp]; pend_s = (uint32_t)((hm * HR + 16 * ii) * p.ldc) * ES;
                            }
#endif
                            if constexpr (ACT == ACT_GELU) {
                                if (p.aux_out) {
                                    uint32_t t0 = pa[0], t1 = pa[1], t2 = pb[0], t3 = pb[1];
                                    swap2(t0, t2); swap2(t1, t3);
                                    // the pre-activation goes out straight from VALU results: held like the C stores' data
                                    const u32x4 tq = u32x4{t0, t1, t2, t3};
                                    buf_store(tq, rX, vX[pp], (uint32_t)((hm * HR + 16 * ii) * p.ldaux) * ES);
#ifndef S2T_NO_HOLD
                                    asm volatile("s_nop 3" :: "v"(tq[0]), "v"(tq[1]), "v"(tq[2]), "v"(tq[3]),
                                                 "v"(xhold[0][0]), "v"(xhold[0][1]), "v"(xhold[0][2]), "v"(xhold[0][3]) : "memory");
#endif
                                    xhold[0] = tq;
                                }
                            }
                        }
                }
                };
                S2T_ES(2)
                if (p.p_drop > 0.f) epi_steps(std::true_type{}); else epi_steps(std::false_type{});
                S2T_ES(4)
#ifndef S2T_NOTURN
                buf_store(pend, rC, pend_v, pend_s);
                tail_c = pend;
#endif
                if constexpr (MOUT) { *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.aux_out) + moff) = mk; tail_m = mk; }
                S2T_ES(5)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        sbase += nk;
        tile += G;
        cur = nxt;
    }
    if (SCHED == 0 && grp == 0) S2T_BAR();                // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the tail DMAs land in LDS nobody reads; retire them before the wave ends
#undef S2T_MMA2
#undef S2T_QUAD
#undef S2T_BAR
#undef S2T_WAIT_TILE
#undef S2T_WAIT_PIPE
#undef S2T_DMA
#undef S2T_READS_DONE
}

// Shapes this kernel takes: bf16 operands, K a multiple of 64, at least two K-tiles, 16-byte aligned rows, operands below 4 GiB
// (32-bit byte offsets), no gather maps, no split-K.  Returns 0 when the product is not for this kernel (the caller falls
// through to gemm.hip), 1 when launched (dry_run: when it would be), < 0 on error.
// row-tile height: the one that needs less time on 256 CUs = rounds x rows per tile (ties go to 256: fewer B re-reads).  Returns the
// number of tiles, 0 when the product has too few of them for this kernel.
static long g256_tiles(int M, int N, bool& use192) {
    const int tn = (N + 255) / 256;
    const long t256 = (long)((M + 255) / 256) * tn, t192 = (long)((M + 191) / 192) * tn;
    // one workgroup per CU: with too few tiles the 128 x 128 kernels (2-3 tiles per CU) win.  tools/gemm_gate_probe.py: at 188 tiles
    // (the l preset's 9,000 tokens x N = 1,024) this kernel is 1.3-1.45x faster (K = 1,024 .. 4,096), at 126-128 tiles the two tie, at
    // 96 and below the small tiles win -> 160.  "gemm256_min_tiles" (s2t_set_option) overrides the threshold for such measurements.
    if (t192 < (g_s2t_opt_gemm256_min_tiles > 0 ? g_s2t_opt_gemm256_min_tiles : 160)) return 0;
    const int ncu = s2t_persistent_cus();
    use192 = ((t192 + ncu - 1) / ncu) * 192 < ((t256 + ncu - 1) / ncu) * 256;
    return use192 ? t192 : t256;
}

// include/s2t_hip.h: 8 KiB per tile (512 lanes x 16 bytes), 0 when an [M][N] product over K does not come here (the shape part of the
// gates of s2t_gemm256_try and of gemm_run's "small" rule for NT products)
extern "C" size_t s2t_gemm_relu_mask_bytes(int M, int N, int K) {
    if (!g_s2t_opt_gemm256 || K % BK || K < 2 * BK || M < 256 || N < 256 || (N & 7)) return 0;
    if ((unsigned long long)M * (unsigned long long)N >= (1ull << 30)) return 0;        // bf16 output below 2 GiB
    if ((long)((M + 127) / 128) * ((N + 127) / 128) < 192) return 0;
    bool use192 = false;
    return (size_t)g256_tiles(M, N, use192) * 8192;
}

int s2t_gemm256_try(const GemmArgs& a, int out_dtype, int trans_b, hipStream_t st, bool dry_run) {
    if (a.mapA || a.mapB || a.mapC || a.splitk != 1 || a.rowsum) return 0;
    if (a.K % BK || a.K < 2 * BK || a.M < 256 || a.N < 256) return 0;
    if (a.N & 7) {
        // N not a multiple of 8 (the CTC head's 5,001 logits per row): the last 16-byte store of a row covers columns up to the next
        // multiple of 8, i.e. the row padding of the caller's buffer (K.alloc_rows), which must exist; plain bias epilogue only (the
        // dropout mask and the operand streams are addressed by aligned element quads)
        if (trans_b || a.ldc < ((a.N + 7) & ~7) || a.residual || a.accumulate || a.aux || a.aux_out || a.p_drop > 0.f || a.act != ACT_NONE) return 0;
    }
    if (((uintptr_t)a.C & 15) || (a.ldc & 7) || (a.bias && ((uintptr_t)a.bias & 15))) return 0;
    if (a.residual && (((uintptr_t)a.residual & 15) || (a.ldr & 7))) return 0;
    if ((a.aux && ((uintptr_t)a.aux & 15)) || (a.aux_out && ((uintptr_t)a.aux_out & 15)) || ((a.aux || a.aux_out) && (a.ldaux & 7))) return 0;
    if ((a.lda % 8) || (a.ldb % 8) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return 0;
    const size_t rowsB = trans_b ? (size_t)a.K : (size_t)a.N;
    if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || rowsB * a.ldb * 2 >= (1ull << 32)) return 0;
    const size_t osz = out_dtype == S2T_BF16 ? 2 : 4;
    if ((size_t)a.M * a.ldc * osz >= (1ull << 31) || (size_t)a.M * a.ldr * osz >= (1ull << 31) || (size_t)a.M * a.ldaux * osz >= (1ull << 31)) return 0;
    if (trans_b && ((a.N + 7) / 8 * 8 > a.ldb)) return 0;
    if (a.p_drop > 0.f && (unsigned long long)a.M * (unsigned long long)a.N >= (1ull << 34)) return 0;   // the epilogue keeps the mask's quad index in 32 bits
    // epilogue variant: the activation and the ONE extra operand stream are compile-time (gemm256_kernel<.., ACT, EXT>)
    int ext = EXT_NONE;
    if (a.act == ACT_RELU_MASK) { if (trans_b || !a.aux_out || a.residual || a.accumulate) return 0; }
    else if (a.act == ACT_RELU_BWD_MASK) { if (!trans_b || !a.aux || a.residual || a.accumulate || a.bias || a.p_drop > 0.f) return 0; }
    else if (a.act == ACT_RELU_BWD || a.act == ACT_GELU_BWD) { if (a.residual || a.accumulate) return 0; ext = EXT_AUX; }
    else if (a.residual) { if (a.accumulate) return 0; ext = EXT_RES; }
    else if (a.accumulate) ext = EXT_OLD;
    if (out_dtype != S2T_BF16) return 0;            // bf16 outputs only (f32 rows -- logits in fp32 mode, split-K partials -- stay in gemm.hip)
    if (trans_b && (a.bias || a.act == ACT_RELU || a.act == ACT_GELU || a.act == ACT_RELU_MASK || ext == EXT_RES)) return 0;   // data gradients: none / act-bwd / accumulate
    if (!trans_b && (ext == EXT_AUX || ext == EXT_OLD || (a.act != ACT_NONE && ext != EXT_NONE))) return 0;
    bool use192 = false;
    const int tiles = (int)g256_tiles(a.M, a.N, use192);
    if (!tiles) return 0;
    if (dry_run) return 1;                         // every gate passed: the caller names the launch (profiling family) before it happens
    const int grid = tiles < s2t_persistent_cus() ? tiles : s2t_persistent_cus();
    const size_t lds = 2 * BUF + 16384;            // two K-tile buffers + the epilogue's lane-turn slots (8 waves x 2 KiB)
    bool done = false;
    const int sched = g_s2t_opt_gemm256_sched;
#define S2T_G256(TO_, TB_, MT_, ACT_, EXT_, SC_)                                                                             \
    if (!done && (out_dtype == S2T_BF16) == (sizeof(TO_) == 2) && (trans_b != 0) == TB_ && use192 == (MT_ == 6) &&          \
        a.act == ACT_ && ext == EXT_ && sched == SC_) {                                                                      \
        static bool attr = false;                                                                                            \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<TO_, TB_, MT_, ACT_, EXT_, SC_>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }         \
        hipLaunchKernelGGL((gemm256_kernel<TO_, TB_, MT_, ACT_, EXT_, SC_>), dim3(grid), dim3(512), lds, st, a);             \
        done = true;                                                                                                         \
    }
    // SCHED 1 (all eight waves software-pipelined at 16-MFMA stages, every wave in the epilogue at once) measured 0.92 - 1.04 x the
    // shipped schedule and is not in the product library: -DS2T_G256_SCHED1 builds it into the twins of `make twins`, where it is the
    // vehicle of the store-data hazard test (two waves of a SIMD issuing 16-byte stores together)
#ifdef S2T_G256_SCHED1
#define S2T_G256_MT(TO_, TB_, ACT_, EXT_) S2T_G256(TO_, TB_, 8, ACT_, EXT_, 0) S2T_G256(TO_, TB_, 6, ACT_, EXT_, 0) \
                                          S2T_G256(TO_, TB_, 8, ACT_, EXT_, 1) S2T_G256(TO_, TB_, 6, ACT_, EXT_, 1)
#else
    if (sched != 0) return S2T_ENOTSUP;
#define S2T_G256_MT(TO_, TB_, ACT_, EXT_) S2T_G256(TO_, TB_, 8, ACT_, EXT_, 0) S2T_G256(TO_, TB_, 6, ACT_, EXT_, 0)
#endif
    S2T_G256_MT(bf16, false, ACT_NONE, EXT_NONE) S2T_G256_MT(bf16, false, ACT_RELU, EXT_NONE) S2T_G256_MT(bf16, false, ACT_GELU, EXT_NONE)
    S2T_G256_MT(bf16, false, ACT_NONE, EXT_RES) S2T_G256_MT(bf16, false, ACT_RELU_MASK, EXT_NONE)
    S2T_G256_MT(bf16, true, ACT_NONE, EXT_NONE) S2T_G256_MT(bf16, true, ACT_RELU_BWD, EXT_AUX) S2T_G256_MT(bf16, true, ACT_GELU_BWD, EXT_AUX)
    S2T_G256_MT(bf16, true, ACT_RELU_BWD_MASK, EXT_NONE)
    S2T_G256_MT(bf16, true, ACT_NONE, EXT_OLD)
#undef S2T_G256_MT
#undef S2T_G256
    if (!done) return 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return S2T_EHIP(e);
    return 1;
}
