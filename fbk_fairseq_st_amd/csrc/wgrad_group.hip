// Grouped weight-gradient products of the Transformer blocks' Linears, one launch for MANY layers:
//     dW_p[n_out][n_in] += dY_p[tokens][n_out]^T . X_p[tokens][n_in]        db_p[n_out] += column sums of dY_p
// (autograd of F.linear at fairseq/modules/multihead_attention.py:190-208, fairseq/modules/transformer_layer.py:132-134; the
// reference runs one GEMM + one reduction per Linear as backward reaches it).
//
// Why grouped and deferred.  A single dW has a small output (512 x 512 ... 2048 x 512) and a huge reduction (every token of the
// batch), so a launch per Linear must split K over the CUs and meet in f32 atomics: at 64 KB per workgroup the atomic pass alone
// is ~12 us of a ~50 us launch (the chip adds 1.3 TB/s, MI355X_MICROARCH.md "Global float atomics") and the tiles are too small
// for an LDS-DMA pipeline to pay (round 1: 470 TFLOP/s, the slowest GEMM family and the largest share of the update).  Nothing
// consumes a weight gradient before the optimizer (or the gradient all-reduce), so the engine queues (dY, X) pairs during backward
// and hands a few layers' worth to THIS kernel: a work list of 256 x 256 output tiles, each owned by exactly one workgroup for the
// WHOLE reduction -- no split-K, no atomics, f32 read-modify-write of the gradient arena by plain 16-byte accesses, operands by
// LDS-DMA through the pipeline of gemm256.hip (two K-tile buffers in half-tile units, two super-phases of 32 MFMAs per K-tile, one
// counted vmcnt per K-tile, staggered wave groups).  Both operands have the reduction index as their memory ROW, so both are staged
// as they lie in memory ([64 tokens][128 columns] images, 256-byte rows, XOR-swizzled chunks) and gathered by ds_read_b64_tr_b16.
// Token counts that are not a multiple of 64 read zeros for the missing rows (a 1 KiB zero page in device memory).
// The bias gradient rides on the A operand: one extra MFMA per A fragment against a ones operand, spread over the four wave columns.
#include "common.hpp"
#include "prof.hpp"
#include "../../include/s2t_hip.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef short s16x4_w __attribute__((ext_vector_type(4)));

namespace {
constexpr int HALF = 16384, BUF = 65536, BK = 64;

struct Prob {
    const bf16* dY; const bf16* X; float* dW; float* db;
    int n_out, n_in, tokens, ldy, ldx, ldw, pad0, pad1;
};
// one work item = one 256 x 256 output tile over the K-tiles [kt0, kt1) of its problem.  `atomic`: the tile's token range is
// shared with other items (a tile that straddles two workgroups' shares), so the result is ADDED with f32 atomics.
struct Item { int prob, tm, tn, kt0, kt1, atomic; };

__device__ uint4 g_zero_page[64];                                  // 1 KiB of zeros (device globals are zero-initialised)

__device__ __forceinline__ int trswz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ void glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ u32x4 tr_frag(const char* img, int col, int s, int r16, int q) {
    u32x4 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 32 * s + 8 * q + 4 * h + (r16 >> 2);
        const int ch = (col >> 3) + ((r16 & 3) >> 1);
        const char* a = img + row * 256 + ((ch ^ trswz(row)) << 4) + ((r16 & 1) << 3);
        const s16x4_w v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_w*)a);
        const u32x2 w = __builtin_bit_cast(u32x2, v);
        f[2 * h] = w[0]; f[2 * h + 1] = w[1];
    }
    return f;
}
}  // namespace

// 8 waves = 2 (M) x 4 (N).  Tile rows [128 h, 128 h + 128) form A half-tile h, wave row wr owns rows 64 wr .. 64 wr + 63 of each;
// tile columns [128 h, 128 h + 128) form B half-tile h, wave column wc owns columns 32 wc .. 32 wc + 31 of each.
__global__ __launch_bounds__(512, 2) void wgrad_group_kernel(const Prob* __restrict__ probs, const Item* __restrict__ items, int n_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, q = lane >> 4;
    const int grp = wr;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // the table is [round][workgroup slot] (the host balances the slots' token counts); an XCD's workgroups (b, b + 8, ...) take a
    // contiguous run of slots: neighbouring tiles of one dW, which share operand columns, meet in one L2
    for (int it = xcd_remap(blockIdx.x, gridDim.x); it < n_items; it += gridDim.x) {
        const Item I = items[it];
        if (I.kt0 >= I.kt1) break;                                       // this slot's list is shorter than the longest one
        const Prob P = probs[I.prob];
        const int tm = I.tm, tn = I.tn;
        const int row0 = tm * 256, col0 = tn * 256;                      // row = n_out index, col = n_in index
        const int nk = I.kt1 - I.kt0;
        const char* Ab = reinterpret_cast<const char*>(P.dY) + (size_t)I.kt0 * BK * P.ldy * 2;
        const char* Bb = reinterpret_cast<const char*>(P.X) + (size_t)I.kt0 * BK * P.ldx * 2;
        const int tokens_left = P.tokens - I.kt0 * BK;                   // token rows from this item's first K-tile on

        // ---- staging: this wave fills pieces 2 wave and 2 wave + 1 (4 token rows x 256 B each) of every half-tile
        // [half][piece]: byte offset (operands are below 4 GiB: checked on the host) of this lane's 16 bytes at token row
        // krow = 4 piece + lane / 16 of K-tile 0
        uint32_t ga[2][2], gb[2][2];
        const int krow0 = 8 * wave + (lane >> 4);                       // piece 2 wave; piece 2 wave + 1 is 4 rows further
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kr = krow0 + 4 * i, pos = lane & 15, cp = (pos ^ trswz(kr)) << 3;
                const int am = min(row0 + 128 * h + cp, ((P.n_out + 7) & ~7) - 8);
                const int bn = min(col0 + 128 * h + cp, ((P.n_in + 7) & ~7) - 8);
                ga[h][i] = (uint32_t)(((size_t)kr * P.ldy + am) * 2);
                gb[h][i] = (uint32_t)(((size_t)kr * P.ldx + bn) * 2);
            }
        const uint32_t kstepA = (uint32_t)BK * (uint32_t)P.ldy * 2u, kstepB = (uint32_t)BK * (uint32_t)P.ldx * 2u;
        auto stage = [&](const char* base, const uint32_t (&g)[2][2], uint32_t kstep, int region, int h, int t) {
            char* dst = smem + __builtin_amdgcn_readfirstlane((t & 1) * BUF + region * 2 * HALF + h * HALF + wave * 2048);
            const int tt = min(t, nk - 1);                               // past the last K-tile: re-read it into a free half-tile (constant DMA count)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool in = tt * BK + krow0 + 4 * i < tokens_left;   // token rows past the end read zeros (select, no branch)
                const char* src = in ? base + (g[h][i] + (uint32_t)tt * kstep) : zero + 16 * lane;
                glds16(src, dst + 1024 * i);
            }
        };
        auto stageA = [&](int h, int t) { stage(Ab, ga, kstepA, 0, h, t); };
        auto stageB = [&](int h, int t) { stage(Bb, gb, kstepB, 1, h, t); };
#define WG_WAIT_TILE() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
#define WG_READS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WG_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool do_rs = P.db != nullptr && tn == 0;                   // bias gradient: the first column tile sums its A operand
        f32x4 rs[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

        u32x4 fa[4][2], fb[2][2][2];
        auto readAs = [&](const char* buf, int h, int s_) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][s_] = tr_frag(buf + h * HALF, wr * 64 + 16 * i, s_, r16, q);
        };
        auto readBs = [&](const char* buf, int h, int s_) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[h][j][s_] = tr_frag(buf + 2 * HALF + h * HALF, wc * 32 + 16 * j, s_, r16, q);
        };
#define WG_QUAD(MI, NI)                                                                                      \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
                    acc[4 * (MI) + i][2 * (NI) + j] = mma16<bf16>(fb[NI][j][s], fa[i][s], acc[4 * (MI) + i][2 * (NI) + j]);
        // row sums of the A operand: wave column wc takes row tile i == wc of the current A half (both k-halves)
#define WG_RS(MI)                                                                                            \
        if (do_rs) {                                                                                         \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                  \
                const u32x4 f = wc == 0 ? fa[0][s] : wc == 1 ? fa[1][s] : wc == 2 ? fa[2][s] : fa[3][s];     \
                rs[MI] = mma16<bf16>(ones, f, rs[MI]);                                                       \
            }                                                                                                \
        }
#define WG_MMA2(MI, NA, NB)                                                                                  \
    do {                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        WG_QUAD(MI, NA) WG_QUAD(MI, NB) WG_RS(MI)                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)

        // ---- prologue: K-tile 0 and what SPb of "K-tile -1" would have staged (A-h0, B-h0 of K-tile 1)
        stageA(0, 0); stageB(0, 0); stageB(1, 0); stageA(1, 0);
        stageA(0, 1); stageB(0, 1);
        WG_WAIT_TILE();
        WG_BAR();
        if (grp == 1) WG_BAR();                                          // group 1 runs one barrier behind from here on
        for (int t = 0; t < nk; ++t) {
            const char* buf = smem + (t & 1) * BUF;
            // SPa: (A-h0 x B-h0), (A-h0 x B-h1); re-stage A-h1 and B-h1 of the other buffer with K-tile t+1
            readBs(buf, 0, 0); readAs(buf, 0, 0); readBs(buf, 0, 1); readAs(buf, 0, 1); readBs(buf, 1, 0); readBs(buf, 1, 1);
            stageA(1, t + 1); stageB(1, t + 1);
            WG_READS_DONE();
            WG_BAR();
            WG_MMA2(0, 0, 1);
            WG_BAR();
            // SPb: (A-h1 x B-h1), (A-h1 x B-h0); re-stage A-h0 and B-h0 of this buffer with K-tile t+2
            readAs(buf, 1, 0); readAs(buf, 1, 1);
            stageA(0, t + 2); stageB(0, t + 2);
            WG_READS_DONE();
            WG_WAIT_TILE();
            WG_BAR();
            WG_MMA2(1, 1, 0);
            WG_BAR();
        }
        if (grp == 0) WG_BAR();                                          // group 0 waits for group 1's last cluster
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the tail DMAs must land before the epilogue reuses the LDS
        WG_BAR();
#undef WG_MMA2
#undef WG_RS
#undef WG_QUAD

        // ---- epilogue: dW tile += acc, four slabs of 64 rows through LDS (f32, row stride 1040 B), 16-byte read-modify-write
        if (do_rs && q == 0) {
#pragma unroll
            for (int hm = 0; hm < 2; ++hm) {
                const int row = row0 + 128 * hm + wr * 64 + 16 * wc + r16;
                if (row < P.n_out) {
                    if (I.atomic) atomicAdd(P.db + row, rs[hm][0]);
                    else P.db[row] += rs[hm][0];                         // exactly one lane of one workgroup owns this element
                }
            }
        }
        constexpr int RS = 256 * 4 + 16;
        // a tile inside the matrix with 16-byte aligned rows (every tile of the Transformer's Linears): the eight 16-byte chunks a lane
        // adds to per slab are LOADED TOGETHER before the slab goes through LDS -- one memory round trip per slab under the LDS write and
        // its barrier.  (The element-wise form below waits for every chunk before it touches the next: 32 dependent round trips per tile,
        // ~20 us per work item with all CUs in their epilogues at once.)
        const bool interior = !I.atomic && row0 + 256 <= P.n_out && col0 + 256 <= P.n_in && (P.ldw & 3) == 0 &&
                              (reinterpret_cast<uintptr_t>(P.dW) & 15) == 0;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {                                 // slab sl = rows 64 sl .. 64 sl + 63 = A half sl / 2, wave row sl % 2
            f32x4 oldv[8];
            if (interior) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int c = threadIdx.x + 512 * it, lr = c >> 6, cc = c & 63;
                    oldv[it] = *reinterpret_cast<const f32x4*>(P.dW + (size_t)(row0 + 64 * sl + lr) * P.ldw + col0 + 4 * cc);
                }
            }
            if (wr == (sl & 1)) {
                const int hm = sl >> 1;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int lr = 16 * i + r16, lc = (j >> 1) * 128 + wc * 32 + 16 * (j & 1) + 4 * q;
                        *reinterpret_cast<f32x4*>(smem + lr * RS + lc * 4) = acc[4 * hm + i][j];
                    }
            }
            __syncthreads();
            if (I.atomic) {                                              // lanes of a wave on 64 consecutive columns: 256 contiguous bytes per wave-instruction
                for (int idx = threadIdx.x; idx < 64 * 256; idx += 512) {
                    const int lr = idx >> 8, lc = idx & 255, row = row0 + 64 * sl + lr, col = col0 + lc;
                    if (row < P.n_out && col < P.n_in) atomicAdd(P.dW + (size_t)row * P.ldw + col, *reinterpret_cast<const float*>(smem + lr * RS + lc * 4));
                }
            } else if (interior) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int c = threadIdx.x + 512 * it, lr = c >> 6, cc = c & 63;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + lr * RS + cc * 16);
                    *reinterpret_cast<f32x4*>(P.dW + (size_t)(row0 + 64 * sl + lr) * P.ldw + col0 + 4 * cc) = oldv[it] + v;
                }
            } else
            for (int c = threadIdx.x; c < 64 * 64; c += 512) {           // 64 rows x 64 chunks of 4 floats
                const int lr = c >> 6, cc = c & 63;
                const int row = row0 + 64 * sl + lr, col = col0 + 4 * cc;
                if (row >= P.n_out || col >= P.n_in) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + lr * RS + cc * 16);
                float* dst = P.dW + (size_t)row * P.ldw + col;
                if (col + 4 <= P.n_in && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                    f32x4 o = *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = o + v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < P.n_in) dst[e] += v[e];
                }
            }
            __syncthreads();
        }
#undef WG_BAR
#undef WG_WAIT_TILE
#undef WG_READS_DONE
    }
}

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
// ---- work lists.  A list is a table [round][slot]: workgroup slot s runs items s, s + G, s + 2G, ... until an empty one.
// Cost model for comparing lists: an item costs its K-tiles plus C0 (pipeline fill from cold operands + the epilogue's 256 KB
// read-modify-write, measured ~8 K-tile times at 40-K-tile items); a launch takes as long as its most loaded slot.
constexpr int C0 = 8, MINP = 8;
int G = 256;                      // workgroups of the launch = s2t_persistent_cus() when the list is built (under the entry point's mutex)
struct Layout { std::vector<Item> table; int used = 0; long makespan = 0; };

long load_of(const Layout& L) {
    long worst = 0;
    for (int sl = 0; sl < L.used; ++sl) {
        long u = 0;
        for (size_t it = sl; it < L.table.size(); it += L.used) {
            const Item& t = L.table[it];
            if (t.kt0 >= t.kt1) break;
            u += t.kt1 - t.kt0 + C0;
        }
        worst = std::max(worst, u);
    }
    return worst;
}

// Layout 1 -- rounds.  Longest reductions first (stable: the tiles of one dW stay neighbours), dealt in rounds of one item per CU; the
// workgroups of a round sweep the token range of neighbouring tiles in step, which is what lets the L2s / the MALL serve the
// operand columns those tiles share (a schedule that balanced the CUs perfectly by handing each an arbitrary stretch of a line
// of tiles ran 1.6x SLOWER: every tile then streams its 24 MB of operands from HBM alone).  Two cuts along the token range, whose
// pieces meet in f32 atomics:
//  * a tile whose reduction is much longer than a CU's fair share of the launch is cut into equal pieces of about that share, the
//    same token ranges for all tiles of its dW;
//  * a partly filled last round would leave CUs idle for a whole item's time: its items are cut into as many equal pieces as
//    fill the round.
// The right list for uniform groups (the encoder's 616 tiles of 375 K-tiles).
void layout_rounds(std::vector<Item> iv, Layout& out) {
    long units = 0;
    for (const Item& t : iv) units += t.kt1 + 6;
    const int share = (int)((units + G - 1) / G);
    {
        std::vector<Item> cutv;
        cutv.reserve(iv.size());
        for (size_t i = 0; i < iv.size();) {
            size_t j = i;
            while (j < iv.size() && iv[j].prob == iv[i].prob) ++j;           // the tiles of one dW: same reduction length
            const int nk = iv[i].kt1;
            const int f = nk > share + share / 4 ? std::min((nk + share - 1) / share, nk / MINP) : 1;
            if (f <= 1) cutv.insert(cutv.end(), iv.begin() + i, iv.begin() + j);
            else {
                const int per = (nk + f - 1) / f;
                for (int piece = 0; piece < f; ++piece)                      // piece-major: equal token ranges sit next to each other
                    for (size_t k = i; k < j; ++k) {
                        const int k0 = piece * per, k1 = std::min(nk, k0 + per);
                        if (k0 < k1) cutv.push_back(Item{iv[k].prob, iv[k].tm, iv[k].tn, k0, k1, 1});
                    }
            }
            i = j;
        }
        iv.swap(cutv);
    }
    std::stable_sort(iv.begin(), iv.end(), [](const Item& x, const Item& y) { return x.kt1 - x.kt0 > y.kt1 - y.kt0; });
    const int rem = (int)(iv.size() % G);
    if (rem) {
        const int f = G / rem;
        if (f >= 2) {
            std::vector<Item> tail(iv.end() - rem, iv.end());
            iv.resize(iv.size() - rem);
            for (int piece = 0; piece < f; ++piece)
                for (const Item& t : tail) {
                    const int nk = t.kt1 - t.kt0, ff = std::max(1, std::min(f, nk / MINP)), per = (nk + ff - 1) / ff;
                    const int k0 = t.kt0 + piece * per, k1 = std::min(t.kt1, k0 + per);
                    if (piece < ff && k0 < k1) iv.push_back(Item{t.prob, t.tm, t.tn, k0, k1, ff > 1 ? 1 : t.atomic});
                }
        }
    }
    out.used = (int)std::min<size_t>(iv.size(), G);
    out.table.swap(iv);
    out.makespan = load_of(out);
}

// Layout 2 -- fill to a level.  For groups that mix a few very long reductions with many short ones (the decoder's: six K/V
// projections over the ~24,000 source tokens = 48 tiles of 374 K-tiles next to 400 tiles of 40 K-tiles over its own 2,560 tokens):
// the short tiles are dealt over the slots whole (1 or 2 each), then the long dWs are poured into what is left of every slot up to
// a common level T: the tiles of one dW always as a gang on neighbouring slots with the SAME token range (they sweep it in step, first
// thing in their slots), the range cut wherever a gang's slots are full.  Every slot ends within a few K-tiles of T.
bool layout_fill(const std::vector<Item>& tiles, Layout& out) {
    struct Line { size_t first, count; int nk; };
    std::vector<Line> lines;
    long units = 0;
    for (size_t i = 0; i < tiles.size();) {
        size_t j = i;
        while (j < tiles.size() && tiles[j].prob == tiles[i].prob) ++j;
        lines.push_back(Line{i, j - i, tiles[i].kt1});
        units += (long)(j - i) * (tiles[i].kt1 + C0);
        i = j;
    }
    const long fair = units / G;
    std::vector<Line> longs;
    std::vector<Item> shorts;
    long long_k = 0;
    for (const Line& l : lines) {
        if (l.nk + C0 > fair && l.count <= (size_t)G / 2 && l.nk >= 4 * MINP) { longs.push_back(l); long_k += (long)l.count * l.nk; }
        else shorts.insert(shorts.end(), tiles.begin() + l.first, tiles.begin() + l.first + l.count);
    }
    if (longs.empty()) return false;
    std::stable_sort(shorts.begin(), shorts.end(), [](const Item& x, const Item& y) { return x.kt1 > y.kt1; });
    std::vector<std::vector<Item>> tail(G), head(G);
    std::vector<long> base(G, 0);
    for (size_t i = 0; i < shorts.size(); ++i) { tail[i % G].push_back(shorts[i]); base[i % G] += shorts[i].kt1 + C0; }
    long sum_base = 0;
    for (long b : base) sum_base += b;
    std::vector<long> ld;
    long T = (sum_base + long_k + (long)G * C0 + G - 1) / G;
    for (int attempt = 0; attempt < 64; ++attempt, T += std::max(1L, T / 64)) {
        for (auto& h : head) h.clear();
        ld = base;
        size_t sl = 0;
        bool ok = true;
        for (const Line& l : longs) {
            int k0 = 0;
            while (k0 < l.nk && ok) {
                if (sl + l.count > (size_t)G) { ok = false; break; }
                long cap = T;
                for (size_t j = 0; j < l.count; ++j) cap = std::min(cap, T - ld[sl + j] - C0);
                int len = (int)std::min<long>(cap, l.nk - k0);
                if (l.nk - k0 - len > 0 && l.nk - k0 - len < MINP) len = l.nk - k0 - MINP;     // never leave a remainder shorter than MINP
                if (len < MINP) { sl += l.count; continue; }                                    // this gang is full
                for (size_t j = 0; j < l.count; ++j) {
                    const Item& t = tiles[l.first + j];
                    head[sl + j].push_back(Item{t.prob, t.tm, t.tn, k0, k0 + len, len == l.nk ? 0 : 1});
                    ld[sl + j] += len + C0;
                }
                k0 += len;
            }
            if (!ok) break;
        }
        if (!ok) continue;
        size_t rounds = 0;
        for (int i = 0; i < G; ++i) rounds = std::max(rounds, head[i].size() + tail[i].size());
        out.table.assign(rounds * G, Item{0, 0, 0, 0, 0, 0});
        for (int i = 0; i < G; ++i) {
            size_t r = 0;
            for (const Item& t : head[i]) out.table[(r++) * G + i] = t;
            for (const Item& t : tail[i]) out.table[(r++) * G + i] = t;
        }
        out.used = G;
        out.makespan = load_of(out);
        return true;
    }
    return false;
}
}  // namespace

extern "C" int s2t_wgrad_group(int n, const S2TWgradProblem* probs, void* stream) {
    if (n <= 0) return S2T_OK;
    if (!probs) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // A training loop hands over the same list update after update (same shapes, and the caching allocator returns the same activation
    // addresses), so the last few lists are kept: their work tables stay on the device (the two pageable-memory copies of an upload
    // cost ~35 us of GPU timeline per launch, as much as the 40 K-tiles of a decoder-sized product) and an identical list is launched
    // without building anything.
    struct Cached { std::vector<char> key; hipStream_t st = nullptr; int device = -1; void* dev = nullptr; size_t cap = 0; unsigned long long used = 0;
                    size_t pb = 0; int n_items = 0, grid = 0; double flops = 0.0, bytes = 0.0; };
    static Cached cache[4];
    static unsigned long long tick = 0;
    static std::mutex mu;
    const size_t key_bytes = (size_t)n * sizeof(S2TWgradProblem);
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lock(mu);
    ++tick;
    if (G != s2t_persistent_cus()) {             // the option changed: every cached list was laid out for another number of workgroups
        for (Cached& c : cache) c.key.clear();
        G = s2t_persistent_cus();
    }
    Cached* hit = nullptr;
    Cached* lru = &cache[0];
    for (Cached& c : cache) {
        // a cached table serves only the (device, stream) it was uploaded on: its upload and its readers are ordered by that stream
        if (c.dev && c.st == st && c.device == device && c.key.size() == key_bytes && memcmp(c.key.data(), probs, key_bytes) == 0) { hit = &c; break; }
        if (c.used < lru->used) lru = &c;
    }
    if (!hit) {
        std::vector<Prob> pv(n);
        std::vector<Item> iv;
        double flops = 0.0, bytes = 0.0;
        for (int i = 0; i < n; ++i) {
            const S2TWgradProblem& s = probs[i];
            if (!s.dY || !s.X || !s.dW || s.n_out <= 0 || s.n_in <= 0 || s.tokens <= 0) return S2T_EINVAL;
            if ((size_t)s.tokens * s.ldy * 2 >= (1ull << 32) || (size_t)s.tokens * s.ldx * 2 >= (1ull << 32)) return S2T_EINVAL;   // 32-bit byte offsets
            // 16-byte column chunks of both operands (the last one may cover row padding, never the next row), 4-byte aligned f32 rows
            if ((s.ldy & 7) || (s.ldx & 7) || ((uintptr_t)s.dY & 15) || ((uintptr_t)s.X & 15)) return S2T_EINVAL;
            if (((s.n_out + 7) & ~7) > s.ldy || ((s.n_in + 7) & ~7) > s.ldx || s.n_out < 8 || s.n_in < 8) return S2T_EINVAL;
            Prob& p = pv[i];
            p.dY = (const bf16*)s.dY; p.X = (const bf16*)s.X; p.dW = s.dW; p.db = s.db;
            p.n_out = s.n_out; p.n_in = s.n_in; p.tokens = s.tokens; p.ldy = s.ldy; p.ldx = s.ldx; p.ldw = s.ldw; p.pad0 = p.pad1 = 0;
            const int nk = (s.tokens + BK - 1) / BK, tn = (s.n_in + 255) / 256, tmn = (s.n_out + 255) / 256;
            for (int a = 0; a < tmn; ++a)
                for (int b = 0; b < tn; ++b) iv.push_back(Item{i, a, b, 0, nk, 0});
            flops += 2.0 * s.n_out * (double)s.n_in * s.tokens;
            bytes += 2.0 * s.tokens * ((double)s.n_out + s.n_in) + 8.0 * s.n_out * (double)s.n_in;
        }
        Layout lay, alt;
        const bool have_alt = layout_fill(iv, alt);
        layout_rounds(std::move(iv), lay);
        if (have_alt && alt.makespan < lay.makespan) std::swap(lay, alt);
        // device-side tables (problems, then items) in one buffer
        const size_t pb = pv.size() * sizeof(Prob), ib = lay.table.size() * sizeof(Item), need = pb + ib;
        std::vector<char> host(need);
        memcpy(host.data(), pv.data(), pb);
        memcpy(host.data() + pb, lay.table.data(), ib);
        if (need > lru->cap || lru->device != device) {
            // grows rarely (the first updates); a hipFree of the old table waits for the kernels that may still read it
            if (lru->dev) (void)hipFree(lru->dev);
            lru->dev = nullptr; lru->key.clear();
            lru->cap = std::max(need * 2, (size_t)1 << 16);
            hipError_t e = hipMalloc(&lru->dev, lru->cap);
            if (e != hipSuccess) { lru->dev = nullptr; lru->cap = 0; return S2T_EHIP(e); }
        }
        // stream-ordered upload from pageable memory (staged by the runtime before the call returns).  Earlier launches that read
        // this slot were enqueued on the stream recorded in it: if that is another stream, wait for it before overwriting the table
        if (lru->dev && lru->st != st && lru->device == device) (void)hipStreamSynchronize(lru->st);
        hipError_t e = hipMemcpyAsync(lru->dev, host.data(), need, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { lru->key.clear(); return S2T_EHIP(e); }
        lru->key.assign(reinterpret_cast<const char*>(probs), reinterpret_cast<const char*>(probs) + key_bytes);
        lru->st = st; lru->device = device;
        lru->pb = pb; lru->n_items = (int)lay.table.size(); lru->grid = lay.used; lru->flops = flops; lru->bytes = bytes;
        hit = lru;
    }
    hit->used = tick;
    ProfScope prof("wgrad_group", st, hit->flops, hit->bytes);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF); attr = true; }
    hipLaunchKernelGGL(wgrad_group_kernel, dim3(hit->grid), dim3(512), 2 * BUF, st,
                       (const Prob*)hit->dev, (const Item*)((char*)hit->dev + hit->pb), hit->n_items);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
