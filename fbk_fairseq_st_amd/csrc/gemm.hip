// MFMA GEMM with fused epilogues for the Transformer projections / FFN of the S2T path
// (reference call sites: F.linear in fairseq/modules/multihead_attention.py:190-208,
//  fairseq/modules/transformer_layer.py:132-134, examples/speech_recognition/models/conv_transformer.py:227,279,
//  fairseq/models/transformer.py:784-788, and their autograd backward).
//
// C[M,N] = epi( op(A)[M,K] . op(B)[K,N] )
//   TA=0: A stored [M][K] (k contiguous)     TA=1: A stored [K][M]  (dW = dY^T X)
//   TB=0: B stored [N][K] (weight layout)    TB=1: B stored [K][N]  (dX = dY W, dW)
// Tile BMxBN, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) as 16x16 MFMA tiles.
// LDS image of either operand is always [row][128 bytes of k] with the 16-byte chunk index
// XOR-swizzled by (row & 7): conflict-free ds_read_b128 fragment reads (cdna_hip_programming.md T2)
// and conflict-free ds_write_b128/b64 staging in both the direct and the transposing path.
// The kernel is byte-generic: 128 B of k per row = 64 bf16 or 32 f32; one "k-group" = 16 B per lane.
// Double-buffered LDS, global loads of tile t+1 issued before the MFMAs of tile t and written to
// LDS after them (register-staged prefetch, one barrier per k-tile).
#include "common.hpp"
#include "prof.hpp"
#include "gemm_epilogue.hpp"
int s2t_gemm256_try(const GemmArgs& a, int out_dtype, int trans_b, hipStream_t st, bool dry_run);   // gemm256.hip
#include <cstdlib>
#include <type_traits>

template <typename T> __device__ __forceinline__ u32x4 load16_guard(const T* p, int valid) {
    // element-wise guarded load of up to 16 bytes (valid = number of in-bounds elements)
    constexpr int E = Elem<T>::PER16;
    T tmp[E];
#pragma unroll
    for (int i = 0; i < E; ++i) tmp[i] = (i < valid) ? p[i] : from_f32<T>(0.f);
    return *reinterpret_cast<u32x4*>(tmp);
}

// ---- stage a tile whose k is contiguous in global memory: rows x 128 B
template <typename T, int ROWS>
struct StageDirect {
    static constexpr int E = Elem<T>::PER16;
    static constexpr int N = ROWS * 8 / 256;     // 16-B chunks per thread
    u32x4 r[N];
    __device__ __forceinline__ void load(const T* g, int ld, int row0, int nrows, int k0, int K, bool vec,
                                         const int* map = nullptr, int period = 0) {
        if (map && vec && k0 + 8 * E <= K) {
            // gathered rows, branch-free: a guarded load compiles to a branch around it and the N chunks of a thread then run as N
            // dependent (map -> data) round trips one after the other.  Here every map entry is fetched first (clamped index), then
            // every row (clamped source), and out-of-range / -1 chunks are zeroed by a select afterwards.
            int src[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int cid = threadIdx.x + 256 * i, row = cid >> 3, c = cid & 7;
                const int gr = min(row0 + row, nrows - 1), gk = k0 + c * E;
                src[i] = map[(size_t)(gk / period) * nrows + gr];
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int cid = threadIdx.x + 256 * i, c = cid & 7;
                const int gk = k0 + c * E, kk = gk % period;
                r[i] = *reinterpret_cast<const u32x4*>(g + (size_t)max(src[i], 0) * ld + kk);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int cid = threadIdx.x + 256 * i, row = cid >> 3;
                asm volatile("" : "+v"(r[i]));
                if (row0 + row >= nrows || src[i] < 0) r[i] = (u32x4){0, 0, 0, 0};
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int cid = threadIdx.x + 256 * i, row = cid >> 3, c = cid & 7;
            const int gr = row0 + row, gk = k0 + c * E;
            u32x4 v = {0, 0, 0, 0};
            if (gr < nrows && gk < K) {
                long src = gr; int kk = gk;
                if (map) { const int tap = gk / period; kk = gk - tap * period; src = map[(size_t)tap * nrows + gr]; }
                if (src >= 0) {
                    const T* p = g + (size_t)src * ld + kk;
                    if (vec && gk + E <= K) v = *reinterpret_cast<const u32x4*>(p);
                    else v = load16_guard<T>(p, K - gk);
                }
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int cid = threadIdx.x + 256 * i, row = cid >> 3, c = cid & 7;
            *reinterpret_cast<u32x4*>(lds + row * 128 + ((c ^ (row & 7)) << 4)) = r[i];
        }
    }
};

// ---- stage a tile whose k is the strided dimension in global memory ([K][cols], cols contiguous):
//      each work item = 4 consecutive k x E consecutive cols, transposed in registers.
template <typename T, int COLS>
struct StageTrans {
    static constexpr int E = Elem<T>::PER16;
    static constexpr int BK = 128 / (int)sizeof(T);
    static constexpr int NKQ = BK / 4;                  // k-quads per tile (16 bf16 / 8 f32)
    static constexpr int ITEMS = (COLS / E) * NKQ;      // 256 at COLS=128, 128 at COLS=64
    static constexpr int N = (ITEMS + 255) / 256;
    u32x4 r[N][4];
    __device__ __forceinline__ void load(const T* g, int ld, int col0, int ncols, int k0, int K, bool vec,
                                         const int* map = nullptr, int period = 0) {
        if (map && vec && k0 + BK <= K && ncols % E == 0) {
            // gathered k-rows, branch-free (see StageDirect::load): all map entries, then all rows, then the zero select
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int it = min((int)threadIdx.x + 256 * i, ITEMS - 1);
                const int kq = it % NKQ, cg = it / NKQ;
                const int gc = col0 + cg * E;
                int src[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) src[rr] = map[k0 + 4 * kq + rr];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) r[i][rr] = *reinterpret_cast<const u32x4*>(g + (size_t)max(src[rr], 0) * ld + min(gc, ncols - E));
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    asm volatile("" : "+v"(r[i][rr]));
                    if (src[rr] < 0 || gc >= ncols) r[i][rr] = (u32x4){0, 0, 0, 0};
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int it = threadIdx.x + 256 * i;
            const int kq = it % NKQ, cg = it / NKQ;
            const int gc = col0 + cg * E;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int gk = k0 + 4 * kq + rr;
                u32x4 v = {0, 0, 0, 0};
                if (it < ITEMS && gk < K && gc < ncols) {
                    const long src = map ? (long)map[gk] : (long)gk;
                    if (src >= 0) {
                        const T* p = g + (size_t)src * ld + gc;
                        if (vec && gc + E <= ncols) v = *reinterpret_cast<const u32x4*>(p);
                        else v = load16_guard<T>(p, ncols - gc);
                    }
                }
                r[i][rr] = v;
            }
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int it = threadIdx.x + 256 * i;
            if (it >= ITEMS) continue;
            const int kq = it % NKQ, cg = it / NKQ;
            if constexpr (sizeof(T) == 2) {
                // 4(k) x 8(col) bf16 -> for each col 4 k-contiguous bf16 = 8 bytes
                const uint16_t* e0 = reinterpret_cast<const uint16_t*>(&r[i][0]);
                const uint16_t* e1 = reinterpret_cast<const uint16_t*>(&r[i][1]);
                const uint16_t* e2 = reinterpret_cast<const uint16_t*>(&r[i][2]);
                const uint16_t* e3 = reinterpret_cast<const uint16_t*>(&r[i][3]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = cg * 8 + j;
                    u32x2 v;
                    v[0] = (uint32_t)e0[j] | ((uint32_t)e1[j] << 16);
                    v[1] = (uint32_t)e2[j] | ((uint32_t)e3[j] << 16);
                    *reinterpret_cast<u32x2*>(lds + col * 128 + (((kq >> 1) ^ (col & 7)) << 4) + ((kq & 1) << 3)) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = cg * 4 + j;
                    u32x4 v = {r[i][0][j], r[i][1][j], r[i][2][j], r[i][3][j]};
                    *reinterpret_cast<u32x4*>(lds + col * 128 + ((kq ^ (col & 7)) << 4)) = v;
                }
            }
        }
    }
};

template <typename T, int ROWS, bool TRANS> struct StageSel { typedef StageDirect<T, ROWS> type; };
template <typename T, int ROWS> struct StageSel<T, ROWS, true> { typedef StageTrans<T, ROWS> type; };

template <typename TI, typename TO, bool TA, bool TB, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int BK = 128 / (int)sizeof(TI);
    constexpr int MT = BM / 32, NT = BN / 32;            // 16x16 tiles per wave in m / n
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (BM + BN) * 128;                // bytes of one (A,B) buffer pair

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = wg / tiles_n, tn = wg % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;

    const int nk_total = (p.K + BK - 1) / BK;
    const int per = (nk_total + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per, kt1 = min(nk_total, kt0 + per);
    if (kt0 >= kt1) return;

    const TI* A = reinterpret_cast<const TI*>(p.A);
    const TI* B = reinterpret_cast<const TI*>(p.B);
    constexpr int E = Elem<TI>::PER16;
    const bool vecA = (p.lda % E == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool vecB = (p.ldb % E == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

    typename StageSel<TI, BM, TA>::type sa;
    typename StageSel<TI, BN, TB>::type sb;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    sa.load(A, p.lda, row0, p.M, kt0 * BK, p.K, vecA, p.mapA, p.periodA);
    sb.load(B, p.ldb, col0, p.N, kt0 * BK, p.K, vecB, p.mapB, 0);
    sa.store(smem);
    sb.store(smem + BM * 128);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = (kt + 1 < kt1);
        if (more) {
            sa.load(A, p.lda, row0, p.M, (kt + 1) * BK, p.K, vecA, p.mapA, p.periodA);
            sb.load(B, p.ldb, col0, p.N, (kt + 1) * BK, p.K, vecB, p.mapB, 0);
        }
        const char* la = smem + cur * STAGE + (wr * (BM / 2)) * 128;
        const char* lb = smem + cur * STAGE + BM * 128 + (wc * (BN / 2)) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fa[MT], fb[NT];
            const int ch = 4 * s + q;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = 16 * i + r16;           // (wr*(BM/2)) is a multiple of 8 -> same swizzle
                fa[i] = *reinterpret_cast<const u32x4*>(la + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = 16 * j + r16;
                fb[j] = *reinterpret_cast<const u32x4*>(lb + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mma16<TI>(fb[j], fa[i], acc[i][j]);   // swapped roles: a lane ends up with 4 consecutive n of one m
        }
        if (more) {
            sa.store(smem + (cur ^ 1) * STAGE);
            sb.store(smem + (cur ^ 1) * STAGE + BM * 128);
        }
        __syncthreads();
    }

    gemm_epilogue<TO, BM, BN, MT, NT, 256>(p, acc, smem, row0, col0, wr * (BM / 2), wc * (BN / 2), q, r16);
}

// ------------------------------------------------------------------------------------------------------------
// Fast path (aligned operands, K % BK == 0, no gather): per-thread source pointers and LDS offsets are computed
// once (rows/cols beyond the edge are clamped to valid memory and discarded by the epilogue: no guards in the loop),
// and global loads run TWO k-tiles ahead of the MFMAs in two named register sets, so a load has two compute
// phases to land before it is written to LDS (PMC showed the guarded 1-deep variant 54 % parked in s_waitcnt
// and issuing 9 VALU per MFMA).
template <typename T, int ROWS, int NTH = 256> struct FastDirect {
    static constexpr int E = Elem<T>::PER16;
    static constexpr int BK = 128 / (int)sizeof(T);
    static constexpr int N = ROWS * 8 / NTH;
    struct Regs { u32x4 r[N]; };
    const char* base;          // wave-uniform (kernel argument): lets the loads use the saddr + 32-bit voffset form
    uint32_t goff[N];          // byte offsets from base (operands are < 4 GiB)
    int off[N];
    __device__ __forceinline__ void init(const T* g, int ld, int row0, int nrows, int k0) {
        base = reinterpret_cast<const char*>(g);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int cid = threadIdx.x + NTH * i, row = cid >> 3, c = cid & 7;
            const int gr = min(row0 + row, nrows - 1);
            goff[i] = (uint32_t)(((size_t)gr * ld + k0 + c * E) * sizeof(T));
            off[i] = row * 128 + ((c ^ (row & 7)) << 4);
        }
    }
    // step = 128 while tiles remain, 0 afterwards (the tail re-reads the last tile: branch-free and in bounds)
    __device__ __forceinline__ void load(Regs& R, int step) {
#pragma unroll
        for (int i = 0; i < N; ++i) { R.r[i] = *reinterpret_cast<const u32x4*>(base + goff[i]); goff[i] += step; }
    }
    __device__ __forceinline__ void store(char* lds, const Regs& R) const {
#pragma unroll
        for (int i = 0; i < N; ++i) *reinterpret_cast<u32x4*>(lds + off[i]) = R.r[i];
    }
};

template <typename T, int COLS, int NTH = 256> struct FastTrans {
    static constexpr int E = Elem<T>::PER16;
    static constexpr int BK = 128 / (int)sizeof(T);
    static constexpr int NKQ = BK / 4;
    static constexpr int ITEMS = (COLS / E) * NKQ;
    static constexpr int N = (ITEMS + NTH - 1) / NTH;
    struct Regs { u32x4 r[N][4]; };
    const T* ptr[N];
    size_t ld;
    __device__ __forceinline__ void init(const T* g, int ld_, int col0, int ncols, int k0) {
        ld = (size_t)ld_;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int it = min((int)threadIdx.x + NTH * i, ITEMS - 1);
            const int kq = it % NKQ, cg = it / NKQ;
            const int gc = min(col0 + cg * E, (ncols + E - 1) / E * E - E);   // the last chunk may hang over into the row padding (ld >= roundup(ncols))
            ptr[i] = g + (size_t)(k0 + 4 * kq) * ld + gc;
        }
    }
    __device__ __forceinline__ void load(Regs& R, int step) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) R.r[i][rr] = *reinterpret_cast<const u32x4*>(ptr[i] + rr * ld);
            if (step) ptr[i] += (size_t)BK * ld;
        }
    }
    __device__ __forceinline__ void store(char* lds, const Regs& R) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int it = threadIdx.x + NTH * i;
            if (it >= ITEMS) continue;
            const int kq = it % NKQ, cg = it / NKQ;
            if constexpr (sizeof(T) == 2) {
                const uint16_t* e0 = reinterpret_cast<const uint16_t*>(&R.r[i][0]);
                const uint16_t* e1 = reinterpret_cast<const uint16_t*>(&R.r[i][1]);
                const uint16_t* e2 = reinterpret_cast<const uint16_t*>(&R.r[i][2]);
                const uint16_t* e3 = reinterpret_cast<const uint16_t*>(&R.r[i][3]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = cg * 8 + j;
                    u32x2 v;
                    v[0] = (uint32_t)e0[j] | ((uint32_t)e1[j] << 16);
                    v[1] = (uint32_t)e2[j] | ((uint32_t)e3[j] << 16);
                    *reinterpret_cast<u32x2*>(lds + col * 128 + (((kq >> 1) ^ (col & 7)) << 4) + ((kq & 1) << 3)) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = cg * 4 + j;
                    u32x4 v = {R.r[i][0][j], R.r[i][1][j], R.r[i][2][j], R.r[i][3][j]};
                    *reinterpret_cast<u32x4*>(lds + col * 128 + ((kq ^ (col & 7)) << 4)) = v;
                }
            }
        }
    }
};

// bf16 operand whose reduction index is the ROW index in memory (dW = dY^T X, dX = dY W), 128 tile columns: the
// [64 k][128 col] tile is copied to LDS as it lies in memory (256-byte rows, 16-byte chunks XOR-swizzled) and the MFMA
// fragments are gathered by the hardware transpose read ds_read_b64_tr_b16 (cdna_hip_programming.md T10, image (b)):
// no register transposition, no sub-dword packing, half the LDS write instructions of FastTrans.
__device__ __forceinline__ int tr_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <int COLS, int NTH = 256> struct FastTr {
    static_assert(COLS == 128, "256-byte LDS rows");
    static constexpr int N = 1024 / NTH;                       // 64 rows x 16 chunks over the workgroup
    static constexpr bool kTrRead = true;
    struct Regs { u32x4 r[N]; };
    const char* base;
    uint32_t goff[N];
    int off[N];
    uint32_t tile_bytes;                                       // 64 rows further down
    __device__ __forceinline__ void init(const bf16* g, int ld, int col0, int ncols, int k0, int tid = threadIdx.x) {
        base = reinterpret_cast<const char*>(g);
        tile_bytes = (uint32_t)(64u * (uint32_t)ld * 2u);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int cid = tid + NTH * i, row = cid >> 4, c = cid & 15;
            const int gc = min(col0 + c * 8, (ncols + 7) / 8 * 8 - 8);
            goff[i] = (uint32_t)(((size_t)(k0 + row) * ld + gc) * 2);
            off[i] = row * 256 + ((c ^ tr_swz(row)) << 4);
        }
    }
    __device__ __forceinline__ void load(Regs& R, int step) {
        const uint32_t st = step ? tile_bytes : 0u;
#pragma unroll
        for (int i = 0; i < N; ++i) { R.r[i] = *reinterpret_cast<const u32x4*>(base + goff[i]); goff[i] += st; }
    }
    __device__ __forceinline__ void store(char* lds, const Regs& R) const {
#pragma unroll
        for (int i = 0; i < N; ++i) *reinterpret_cast<u32x4*>(lds + off[i]) = R.r[i];
    }
};

template <typename T, int ROWS, bool TRANS, int NTH = 256> struct FastSel { typedef FastDirect<T, ROWS, NTH> type; };
template <typename T, int ROWS, int NTH> struct FastSel<T, ROWS, true, NTH> { typedef FastTrans<T, ROWS, NTH> type; };
template <int NTH> struct FastSel<bf16, 128, true, NTH> { typedef FastTr<128, NTH> type; };
template <typename S> struct UsesTrRead { static constexpr bool value = false; };
template <int C, int NTH> struct UsesTrRead<FastTr<C, NTH>> { static constexpr bool value = true; };

typedef short s16x4 __attribute__((ext_vector_type(4)));
// fragment t (16 tile columns) of k-half s from a FastTr image; cw = first tile column of this wave
__device__ __forceinline__ u32x4 tr_fragment(const char* img, int cw, int t, int s, int r16, int q) {
    u32x4 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 32 * s + 8 * q + 4 * h + (r16 >> 2);
        const int ch = ((cw + 16 * t) >> 3) + ((r16 & 3) >> 1);
        const char* a = img + row * 256 + ((ch ^ tr_swz(row)) << 4) + ((r16 & 1) << 3);
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
        const u32x2 w = __builtin_bit_cast(u32x2, v);
        f[2 * h] = w[0]; f[2 * h + 1] = w[1];
    }
    return f;
}

// la / lb: LDS image of the A / B tile; arow / brow: first tile row (= output row / column) of this wave
// RSUM: additionally rs[i] += A-fragment x ones, i.e. the row sums of op(A) over this k-tile (every column of the
// 16x16 result holds the same sum): the bias gradient of a Linear rides on the weight-gradient product's A operand.
template <typename TI, int MT, int NT, bool TRA, bool TRB, bool RSUM = false>
__device__ __forceinline__ void mma_tile(const char* la, const char* lb, int arow, int brow, int r16, int q, f32x4 (&acc)[MT][NT],
                                         f32x4 (*rs)[MT] = nullptr) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 fa[MT], fb[NT];
        const int sw = ((4 * s + q) ^ (r16 & 7)) << 4;          // (16*i + r16) & 7 == r16 & 7
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (TRA) fa[i] = tr_fragment(la, arow, i, s, r16, q);
            else fa[i] = *reinterpret_cast<const u32x4*>(la + (arow + 16 * i + r16) * 128 + sw);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if constexpr (TRB) fb[j] = tr_fragment(lb, brow, j, s, r16, q);
            else fb[j] = *reinterpret_cast<const u32x4*>(lb + (brow + 16 * j + r16) * 128 + sw);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mma16<TI>(fb[j], fa[i], acc[i][j]);   // swapped roles: a lane ends up with 4 consecutive n of one m
        if constexpr (RSUM) {
            const u32x4 ones = sizeof(TI) == 2 ? (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                               : (u32x4){0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u};
#pragma unroll
            for (int i = 0; i < MT; ++i) (*rs)[i] = mma16<TI>(ones, fa[i], (*rs)[i]);
        }
    }
}

// NW = 4: waves 2 x 2, each (BM/2) x (BN/2).  NW = 8: waves 2 x 4, each (BM/2) x (BN/4): twice the wavefronts per CU to
// cover the global-load latency of the k-loop, at 1.5x the LDS fragment traffic per MFMA.
template <typename TI, typename TO, bool TA, bool TB, int BM, int BN, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) void gemm_fast_kernel(GemmArgs p, const bool g_deep_ok) {   // 2nd argument = min waves per SIMD
    constexpr int BK = 128 / (int)sizeof(TI);
    constexpr int NTH = NW * 64, WN = NW / 2;
    constexpr int MT = BM / 32, NT = BN / (16 * WN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (BM + BN) * 128;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int tiles = tiles_m * tiles_n;
    int wg, kz;
    {
        // Workgroups are dealt to the 8 XCDs round-robin on their linear id.  Without split-K an XCD gets a contiguous run
        // of tiles (neighbours share A rows / B columns in its L2).  With split-K the k-slices partition the ROWS of both
        // operands, so all tiles of one slice belong on one XCD: it then streams its slice of the operands from HBM once
        // (PMC: the dW products read 2.2x their operand bytes with the tile-major deal).
        const int L = blockIdx.z * tiles + blockIdx.x, xcd = L & 7, j = L >> 3, sk = p.splitk;
        if (sk > 1 && (sk & 7) == 0) { kz = xcd + 8 * (j / tiles); wg = j % tiles; }
        else if ((sk == 2 || sk == 4) && tiles % (8 / sk) == 0) { kz = xcd % sk; wg = (xcd / sk) * (tiles / (8 / sk)) + j; }
        else { kz = blockIdx.z; wg = xcd_remap(blockIdx.x, tiles); }
    }
    const int tm = wg / tiles_n, tn = wg % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int nk_total = p.K / BK;
    const int per = (nk_total + p.splitk - 1) / p.splitk;
    const int kt0 = kz * per, kt1 = min(nk_total, kt0 + per);
    const int nk = kt1 - kt0;
    if (nk <= 0) return;

    typedef typename FastSel<TI, BM, TA, NTH>::type SA;
    typedef typename FastSel<TI, BN, TB, NTH>::type SB;
    SA sa; SB sb;
    sa.init(reinterpret_cast<const TI*>(p.A), p.lda, row0, p.M, kt0 * BK);
    sb.init(reinterpret_cast<const TI*>(p.B), p.ldb, col0, p.N, kt0 * BK);
    typename SA::Regs a0, a1;
    typename SB::Regs b0, b1;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WN, wc = wave % WN, r16 = lane & 15, q = lane >> 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    char* l0 = smem;
    char* l1 = smem + STAGE;
    constexpr bool TRA = UsesTrRead<SA>::value, TRB = UsesTrRead<SB>::value;
    const int arow = wr * (BM / 2), brow = wc * (BN / WN);
    constexpr int BOFF = BM * 128;

    int left = nk;                                  // tiles not yet requested from global memory
    auto step = [&]() { left -= 1; return left > 0 ? 128 : 0; };
    // DEEP: the 64 x 64 four-wave form (the small products: decoder-side rows, M of a few thousand) is a chain of memory round trips --
    // two k-tiles in flight make ~1.3 us per k-tile whatever the tile costs to multiply.  A thread stages only 2 + 2 (transposed
    // operand: 2 + 4) 16-byte chunks per k-tile there, so DEPTH register sets fit (8 x 16 = 128 VGPRs, or 4 x 24): all of K = 512 is
    // requested before the first MFMA.  Same k order, same MFMAs: bit-identical results.  Taken when the k-tile count is a multiple of
    // DEPTH (every K of the model is), so that the unrolled rotation needs no guard around its loads.
    constexpr int DEPTH = (NW == 4 && BM == 64 && BN == 64 && sizeof(TI) == 2) ? ((TA || TB) ? 4 : 8) : 0;
    bool deep = false;
    if constexpr (DEPTH > 0) deep = g_deep_ok && nk % DEPTH == 0 && !(TA && p.rowsum != nullptr);
    if (!deep) {
        { const int st = step(); sa.load(a0, st); sb.load(b0, st); }  // tile 0
        { const int st = step(); sa.load(a1, st); sb.load(b1, st); }  // tile 1 (or tile 0 again)
    }
    // bias gradient: the first column tile's wc == 0 waves also sum their A rows (wave-uniform choice)
    const bool do_rs = TA && p.rowsum != nullptr && tn == 0 && wc == 0;
    f32x4 rs[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const char* l) {
        if constexpr (TA) {
            if (do_rs) { mma_tile<TI, MT, NT, TRA, TRB, true>(l, l + BOFF, arow, brow, r16, q, acc, &rs); return; }
        }
        mma_tile<TI, MT, NT, TRA, TRB>(l, l + BOFF, arow, brow, r16, q, acc);
    };
    if constexpr (NW == 8) {
        // 8-wave form: a thread stages only 2 + 2 chunks per k-tile, so a THIRD register set fits under 128 VGPRs (4 waves per
        // SIMD kept): loads run three k-tiles ahead of the MFMAs (PMC: 42 % of the wave cycles were parked in s_waitcnt / barriers
        // with two).  Tile k lives in set k % 3 and is written to stage k % 2 one step before it is consumed; unrolled by 6.
        typename SA::Regs a2;
        typename SB::Regs b2;
        { const int st = step(); sa.load(a2, st); sb.load(b2, st); }  // tile 2
        sa.store(l0, a0); sb.store(l0 + BM * 128, b0);
        __syncthreads();
#define S2T_STEP(RA, RB, SA_, SB_, LCUR, LNXT)                                                  \
        { const int st = step(); sa.load(RA, st); sb.load(RB, st); }   /* tile t+3 */            \
        mma(LCUR);                                                      /* tile t   */            \
        sa.store(LNXT, SA_); sb.store(LNXT + BM * 128, SB_);            /* tile t+1 */            \
        __syncthreads();
        const int nkk = nk;
        int t = 0;
        for (; t + 6 <= nkk; t += 6) {
            S2T_STEP(a0, b0, a1, b1, l0, l1)
            S2T_STEP(a1, b1, a2, b2, l1, l0)
            S2T_STEP(a2, b2, a0, b0, l0, l1)
            S2T_STEP(a0, b0, a1, b1, l1, l0)
            S2T_STEP(a1, b1, a2, b2, l0, l1)
            S2T_STEP(a2, b2, a0, b0, l1, l0)
        }
        const int rem = nkk - t;                       // 0..5 steps left, the rotation is back at its start
        if (rem > 0) { S2T_STEP(a0, b0, a1, b1, l0, l1)
        if (rem > 1) { S2T_STEP(a1, b1, a2, b2, l1, l0)
        if (rem > 2) { S2T_STEP(a2, b2, a0, b0, l0, l1)
        if (rem > 3) { S2T_STEP(a0, b0, a1, b1, l1, l0)
        if (rem > 4) { S2T_STEP(a1, b1, a2, b2, l0, l1) } } } } }
#undef S2T_STEP
    } else if (deep) {
        if constexpr (DEPTH > 0) {
            typename SA::Regs ra[DEPTH];
            typename SB::Regs rb[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { const int st = step(); sa.load(ra[d], st); sb.load(rb[d], st); }
            for (int t0 = 0; t0 < nk; t0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {                     // tile t0 + d: set d, stage d & 1 (DEPTH is even)
                    char* l = (d & 1) ? l1 : l0;
                    sa.store(l, ra[d]); sb.store(l + BM * 128, rb[d]);
                    __syncthreads();
                    { const int st = step(); sa.load(ra[d], st); sb.load(rb[d], st); }   // tile t0 + d + DEPTH (past the end: the last tile again)
                    mma(l);
                }
            }
            // the epilogue stores the accumulators into the same LDS at once: every wave's fragment reads of the last k-tile must be done
            // (the other loops end their last iteration on a barrier; without this one an eight-wave variant of this loop produced NaNs)
            __syncthreads();
        }
    } else {
    sa.store(l0, a0); sb.store(l0 + BM * 128, b0);
    __syncthreads();
    for (int t = 0; t < nk; t += 2) {
        { const int st = step(); sa.load(a0, st); sb.load(b0, st); }  // tile t+2 -> set 0
        mma(l0);                                                       // tile t
        sa.store(l1, a1); sb.store(l1 + BM * 128, b1);                 // tile t+1
        __syncthreads();
        { const int st = step(); sa.load(a1, st); sb.load(b1, st); }  // tile t+3 -> set 1
        if (t + 1 < nk) mma(l1);
        sa.store(l0, a0); sb.store(l0 + BM * 128, b0);                 // tile t+2
        __syncthreads();
    }
    }
    if constexpr (TA) {
        if (do_rs && q == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = row0 + arow + 16 * i + r16;
                if (row < p.M) atomicAdd(p.rowsum + row, rs[i][0]);
            }
        }
    }
    gemm_epilogue<TO, BM, BN, MT, NT, NTH>(p, acc, smem, row0, col0, arow, brow, q, r16);
}

// ------------------------------------------------------------------------------------------------------------
// dW = dY^T X with two k-slices per workgroup.  The f32 atomic pass of split-K runs at ~1.4 TB/s (one dword per clock and
// L2 channel) and cost ~22 us of an 80 us dW launch at 8 slices.  Here a workgroup is two independent 4-wave groups (each the
// 2x2 / 64x64-per-wave structure of gemm_fast_kernel with its own LDS stages) that walk neighbouring k-slices of the SAME
// output tile; their accumulators meet in LDS and leave as one atomic pass: half the atomic traffic at the same number of
// wavefronts per CU (8 waves, one workgroup per CU).  bf16 operands (both by transposed LDS reads), f32 output.
__global__ __launch_bounds__(512, 1) void gemm_tn2_kernel(GemmArgs p) {
    constexpr int BM = 128, BN = 128, BK = 64, MT = 4, NT = 4, STAGE = (BM + BN) * 128, BOFF = BM * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, tiles = tiles_m * tiles_n;
    int wg, kz;
    {
        const int L = blockIdx.z * tiles + blockIdx.x, xcd = L & 7, j = L >> 3, sk = p.splitk;     // slices -> XCDs as in gemm_fast_kernel
        if (sk > 1 && (sk & 7) == 0) { kz = xcd + 8 * (j / tiles); wg = j % tiles; }
        else if ((sk == 2 || sk == 4) && tiles % (8 / sk) == 0) { kz = xcd % sk; wg = (xcd / sk) * (tiles / (8 / sk)) + j; }
        else { kz = blockIdx.z; wg = xcd_remap(blockIdx.x, tiles); }
    }
    const int tm = wg / tiles_n, tn = wg % tiles_n, row0 = tm * BM, col0 = tn * BN;
    const int nk_total = p.K / BK;
    const int per = (nk_total + 2 * p.splitk - 1) / (2 * p.splitk);          // k-tiles per half slice
    const int kt0 = (2 * kz + grp) * per, nk = max(0, min(nk_total, kt0 + per) - kt0);
    const int nk_max = min(per, max(0, nk_total - 2 * kz * per));              // group 0's count: the barrier count of both groups

    typedef FastTr<128, 256> S;
    S sa, sb;
    const int k0 = min(kt0, nk_total - 1) * BK;                                // an empty half still points at valid memory
    sa.init(reinterpret_cast<const bf16*>(p.A), p.lda, row0, p.M, k0, tid);
    sb.init(reinterpret_cast<const bf16*>(p.B), p.ldb, col0, p.N, k0, tid);
    S::Regs a0, a1, b0, b1;

    const int lane = threadIdx.x & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, r16 = lane & 15, q = lane >> 4;
    const int arow = wr * 64, brow = wc * 64;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_rs = p.rowsum != nullptr && tn == 0 && wc == 0;
    f32x4 rs[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    char* l0 = smem + grp * 2 * STAGE;
    char* l1 = l0 + STAGE;
    int left = nk;
    auto step = [&]() { left -= 1; return left > 0 ? 128 : 0; };
    // B fragments one at a time (4 live fragment registers fewer than mma_tile: this kernel sits at the 256-VGPR limit and a
    // spill reload in the loop would drain the global prefetch through its vmcnt wait)
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    auto mma = [&](const char* l) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4 fa[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = tr_fragment(l, arow, i, s2, r16, q);
            if (do_rs) {
#pragma unroll
                for (int i = 0; i < MT; ++i) rs[i] = mma16<bf16>(ones, fa[i], rs[i]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const u32x4 fb = tr_fragment(l + BOFF, brow, j, s2, r16, q);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = mma16<bf16>(fb, fa[i], acc[i][j]);
            }
        }
    };
    { const int st = step(); sa.load(a0, st); sb.load(b0, st); }
    { const int st = step(); sa.load(a1, st); sb.load(b1, st); }
    sa.store(l0, a0); sb.store(l0 + BOFF, b0);
    __syncthreads();
    for (int t = 0; t < nk_max; t += 2) {
        { const int st = step(); sa.load(a0, st); sb.load(b0, st); }
        if (t < nk) mma(l0);
        sa.store(l1, a1); sb.store(l1 + BOFF, b1);
        __syncthreads();
        { const int st = step(); sa.load(a1, st); sb.load(b1, st); }
        if (t + 1 < nk) mma(l1);
        sa.store(l0, a0); sb.store(l0 + BOFF, b0);
        __syncthreads();
    }
    if (do_rs && q == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = row0 + arow + 16 * i + r16;
            if (row < p.M) atomicAdd(p.rowsum + row, rs[i][0]);
        }
    }
    // ---- the two halves meet in LDS (f32 tile, row stride 528 B as in gemm_epilogue), then one pass over the output
    constexpr int RS = BN * 4 + 16;
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4*>(smem + (arow + 16 * i + r16) * RS + (brow + 16 * j + 4 * q) * 4) = acc[i][j];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4* d = reinterpret_cast<f32x4*>(smem + (arow + 16 * i + r16) * RS + (brow + 16 * j + 4 * q) * 4);
                *d = (*d + acc[i][j]) * p.alpha;
            }
    }
    __syncthreads();
    float* C = reinterpret_cast<float*>(p.C);
    for (int idx = threadIdx.x; idx < BM * BN; idx += 512) {
        const int lr = idx / BN, lc = idx % BN, row = row0 + lr, col = col0 + lc;
        if (row >= p.M || col >= p.N) continue;
        const float v = *reinterpret_cast<const float*>(smem + lr * RS + lc * 4);
        float* dst = C + (size_t)row * p.ldc + col;
        if (p.splitk > 1) atomicAdd(dst, v);
        else *dst = p.accumulate ? *dst + v : v;
    }
}

template <typename TI, bool TA, bool TB>
static bool fast_ok(const GemmArgs& a) {
    constexpr int E = Elem<TI>::PER16;
    constexpr int BK = 128 / (int)sizeof(TI);
    if (a.mapA || a.mapB || a.K % BK) return false;
    if ((a.lda % E) || (a.ldb % E) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return false;
    const size_t rowsA = TA ? (size_t)a.K : (size_t)a.M, rowsB = TB ? (size_t)a.K : (size_t)a.N;
    if (rowsA * a.lda * sizeof(TI) >= (1ull << 32) || rowsB * a.ldb * sizeof(TI) >= (1ull << 32)) return false;   // 32-bit offsets
    // transposed operands are read in whole 16-byte column chunks: the last one may cover row padding, never the next row
    if (TA && (a.M < E || (a.M + E - 1) / E * E > a.lda)) return false;
    if (TB && (a.N < E || (a.N + E - 1) / E * E > a.ldb)) return false;
    return true;
}

extern "C" int s2t_colsum(int dtype, const void* X, int ld, int M, int N, float* out, void* stream);

template <typename TI, typename TO, bool TA, bool TB, int BM, int BN>
static int launch(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    dim3 grid(tiles, 1, a.splitk);
    const bool fast = fast_ok<TI, TA, TB>(a);
    if (a.rowsum && !(TA && fast)) {
        // the row sums ride on the fast kernels' transposed A operand only: otherwise a separate column-sum pass over A [K][M]
        if (!TA) return S2T_EINVAL;
        const int rc = s2t_colsum(sizeof(TI) == 2 ? S2T_BF16 : S2T_F32, a.A, a.lda, a.K, a.M, a.rowsum, st);
        if (rc != S2T_OK) return rc;
        a.rowsum = nullptr;
    }
    size_t lds = 2 * (BM + BN) * 128;
    if ((size_t)BM * (BN * 4 + 16) > lds) lds = (size_t)BM * (BN * 4 + 16);   // epilogue staging of the f32 tile
    if (fast) {
        if constexpr (BM == 128 && BN == 128 && sizeof(TI) == 2 && sizeof(TO) == 4 && TA && TB) {
            // weight-gradient products: two k-slices per workgroup, half the atomic traffic
            const int nk_total2 = a.K / 64;
            if (!a.mapC && nk_total2 >= 4 * a.splitk && (a.splitk == 1 || a.splitk % 2 == 0)) {
                GemmArgs b = a;
                b.splitk = a.splitk > 1 ? a.splitk / 2 : 1;                    // same wavefronts per CU: s slices of 4 waves -> s/2 of 8
                static bool attr4 = false;
                if (!attr4) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); attr4 = true; }
                hipLaunchKernelGGL(gemm_tn2_kernel, dim3(tiles, 1, b.splitk), dim3(512), 131072, st, b);
                S2T_LAUNCH_CHECK();
                return S2T_OK;
            }
        }
        if constexpr (BM == 128 && BN == 128 && sizeof(TI) == 2 && !TA) {
            // measured (tools/microbench.py, M = 24000): 8 waves gain 10-18 % on the NT / NN products (the k-loop is bound by
            // global-load latency: twice the wavefronts per CU cover more of it), and lose ~10 % on TN where both operands
            // are gathered by transposed LDS reads (fragment traffic dominates)
            static bool attr3 = false;
            if (!attr3) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fast_kernel<TI, TO, TA, TB, BM, BN, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr3 = true; }
            hipLaunchKernelGGL((gemm_fast_kernel<TI, TO, TA, TB, BM, BN, 8>), grid, dim3(512), lds, st, a, g_s2t_opt_gemm_deep != 0);
            S2T_LAUNCH_CHECK();
            return S2T_OK;
        }
        if (lds > 65536) {
            static bool attr = false;
            if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fast_kernel<TI, TO, TA, TB, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        }
        hipLaunchKernelGGL((gemm_fast_kernel<TI, TO, TA, TB, BM, BN>), grid, dim3(256), lds, st, a, g_s2t_opt_gemm_deep != 0);
        S2T_LAUNCH_CHECK();
        return S2T_OK;
    }
    if (lds > 65536) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<TI, TO, TA, TB, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    }
    hipLaunchKernelGGL((gemm_kernel<TI, TO, TA, TB, BM, BN>), grid, dim3(256), lds, st, a);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

template <typename TI, typename TO, int BM, int BN>
static int launch_t(const GemmArgs& a, int ta, int tb, hipStream_t st) {
    if (!ta && !tb) return launch<TI, TO, false, false, BM, BN>(a, st);
    if (!ta && tb) return launch<TI, TO, false, true, BM, BN>(a, st);
    if (ta && tb) return launch<TI, TO, true, true, BM, BN>(a, st);
    return S2T_ENOTSUP;   // A^T . B^T never occurs on this path
}

static int gemm_run(int in_dtype, int out_dtype, int trans_a, int trans_b, int M, int N, int K,
                    const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                    const float* bias, const void* residual, int ldr, const void* aux, void* aux_out,
                    int ldaux, int act, int accumulate, int splitk, float alpha,
                    const int* mapA, int periodA, const int* mapB, const int* mapC,
                    float p_drop, unsigned long long seed, float* rowsum, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return (M < 0 || N < 0 || K < 0) ? S2T_EINVAL : S2T_OK;
    if (!A || !B || !C) return S2T_EINVAL;
    if (splitk < 1) splitk = 1;
    if (splitk > 1 && (out_dtype != S2T_F32 || bias || residual || act != ACT_NONE)) return S2T_EINVAL;
    if ((act == ACT_RELU_BWD || act == ACT_GELU_BWD || act == ACT_RELU_BWD_MASK) && !aux) return S2T_EINVAL;
    if (act == ACT_RELU_MASK && !aux_out) return S2T_EINVAL;
    if (act < ACT_NONE || act > ACT_RELU_BWD_MASK) return S2T_EINVAL;
    if (mapA && (trans_a || periodA <= 0 || periodA % 8)) return S2T_EINVAL;
    if (mapB && !trans_b) return S2T_EINVAL;
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && splitk > 1)) return S2T_EINVAL;
    {   // K tail (e.g. K = V_src = 5001 in the CTC head's data gradient): whole k-tiles go through the fast kernel, the
        // remainder is accumulated by a second, tiny launch instead of sending the whole product down the guarded path
        const int BKe = in_dtype == S2T_BF16 ? 64 : 32, tail = K % BKe;
        if (tail && K >= 8 * BKe && !mapA && !mapB && !mapC && splitk == 1 && !bias && !residual && !aux && !aux_out &&
            act == ACT_NONE && p_drop == 0.f) {
            const size_t es = in_dtype == S2T_BF16 ? 2 : 4;
            const int Km = K - tail;
            int rc = gemm_run(in_dtype, out_dtype, trans_a, trans_b, M, N, Km, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0,
                              nullptr, nullptr, 0, ACT_NONE, accumulate, 1, alpha, nullptr, 0, nullptr, nullptr, 0.f, 0ull, rowsum, stream);
            if (rc != S2T_OK) return rc;
            const char* At = (const char*)A + (trans_a ? (size_t)Km * lda : (size_t)Km) * es;
            const char* Bt = (const char*)B + (trans_b ? (size_t)Km * ldb : (size_t)Km) * es;
            return gemm_run(in_dtype, out_dtype, trans_a, trans_b, M, N, tail, At, lda, Bt, ldb, C, ldc, nullptr, nullptr, 0,
                            nullptr, nullptr, 0, ACT_NONE, 1, 1, alpha, nullptr, 0, nullptr, nullptr, 0.f, 0ull, rowsum, stream);
        }
    }
    GemmArgs a{A, B, C, bias, residual, aux, aux_out, M, N, K, lda, ldb, ldc, ldr, ldaux, act, accumulate, splitk, alpha,
               mapA, periodA, mapB, mapC, p_drop, seed, rowsum};
    if (rowsum && (!trans_a || mapA || mapB)) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const double esz = in_dtype == S2T_BF16 ? 2.0 : 4.0, osz = out_dtype == S2T_BF16 ? 2.0 : 4.0;
    // small problems: 64x64 tiles so that more workgroups exist than CUs
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * splitk;
    // products with a k-strided operand (NN, TN) stage it through transposed LDS reads only in the 128-wide kernels (the 64-wide
    // form transposes in registers): they switch to 64x64 tiles much later (tools/small_gemm.py: NN 2560 x 512 x 8000 93 -> 67 us,
    // TN 512 x 512 18 -> 14.6 us with the 128-wide kernels; NT is better off with 64x64 below ~192 tiles)
    // f32 operands: the exact-f32 MFMA runs at 1/16 of the bf16 rate, so a product is bound by its tiles' MFMA time, not by staging, and
    // what counts is that every CU has tiles: thresholds of their own (s2t_set_option "gemm_f32_small_nt" / "_kt": below that many 128 x 128
    // tiles the 64 x 64 form; "gemm_f32_narrow": below that many the 128 x 64 form)
    const bool f32in = in_dtype == S2T_F32 && !mapA && !mapB && !mapC;
    const bool small = f32in ? t128 < ((trans_a || trans_b) ? g_s2t_opt_f32_small_kt : g_s2t_opt_f32_small_nt)
                             : t128 < ((trans_a || trans_b) ? g_s2t_opt_small_kt : g_s2t_opt_small_nt);
    // (128 x 64 tiles for products with 40-160 tiles of 128 x 128 -- twice the workgroups on the idle CUs -- measured 2-3 % SLOWER on
    // the decoder's M = 2,560 / 3,000 products, tools/dec_gemm_time.py: not used)
    const bool narrow = !small && (N <= 64 || (f32in && t128 < g_s2t_opt_f32_narrow));    // N <= 64: conv2 implicit GEMM, 64 output channels
    // families for the roofline report (one kernel template each): dW-shaped (TN), forward (NT), dX-shaped (NN) products on
    // 128x128 tiles, their small-problem 64x64 forms, and the implicit-GEMM convolution
    // ONE KERNEL TEMPLATE per family: the 256 x 256 x 64 LDS-DMA kernel in its forward (NT) and data-gradient (NN) forms, and the
    // register-staged kernels of this file
    static const char* const kFam[3][2] = {{"gemm_nt", "gemm_nt_small"}, {"gemm_nn", "gemm_nn_small"}, {"gemm_tn", "gemm_tn_small"}};
    // big forward / data-gradient products: 256 x 256 x 64 LDS-DMA kernel
    const bool big = in_dtype == S2T_BF16 && !trans_a && !small && g_s2t_opt_gemm256 && s2t_gemm256_try(a, out_dtype, trans_b, st, true) == 1;
    ProfScope prof(mapA || mapB ? "gemm_gather" : (big ? (trans_b ? "gemm256_nn" : "gemm256_nt") : kFam[trans_a ? 2 : (trans_b ? 1 : 0)][small ? 1 : 0]),
                   st, 2.0 * M * (double)N * K, esz * ((double)M * K + (double)N * K) + osz * (double)M * N);
    if (big) {
        const int r = s2t_gemm256_try(a, out_dtype, trans_b, st, false);
        if (r != 0) return r < 0 ? r : S2T_OK;
    }
    if (act == ACT_RELU_MASK || act == ACT_RELU_BWD_MASK) return S2T_ENOTSUP;   // the 1-bit record exists only in the 256-wide kernel's tile order
#define S2T_PICK(TI_, TO_)                                                                   \
    return small ? launch_t<TI_, TO_, 64, 64>(a, trans_a, trans_b, st)                       \
                 : (narrow ? launch_t<TI_, TO_, 128, 64>(a, trans_a, trans_b, st)            \
                           : launch_t<TI_, TO_, 128, 128>(a, trans_a, trans_b, st))
    if (in_dtype == S2T_BF16 && out_dtype == S2T_BF16) { S2T_PICK(bf16, bf16); }
    if (in_dtype == S2T_BF16 && out_dtype == S2T_F32) { S2T_PICK(bf16, float); }
    if (in_dtype == S2T_F32 && out_dtype == S2T_F32) { S2T_PICK(float, float); }
#undef S2T_PICK
    return S2T_ENOTSUP;
}

extern "C" int s2t_gemm_gather(int in_dtype, int out_dtype, int trans_a, int trans_b, int M, int N, int K,
                               const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                               const float* bias, const void* residual, int ldr, const void* aux, void* aux_out,
                               int ldaux, int act, int accumulate, int splitk, float alpha,
                               const int* mapA, int periodA, const int* mapB, const int* mapC,
                               float p_drop, unsigned long long seed, void* stream) {
    return gemm_run(in_dtype, out_dtype, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, aux, aux_out,
                    ldaux, act, accumulate, splitk, alpha, mapA, periodA, mapB, mapC, p_drop, seed, nullptr, stream);
}

// Parameter gradients of y = x W^T + b in one pass over dY (autograd of F.linear, fairseq/modules/multihead_attention.py:190-208,
// transformer_layer.py:132-134):  dW[n_out][n_in] += dY^T X  (f32, split-K atomics)  and, when db is given,
// db[n_out] += column sums of dY, taken from the same LDS tiles the weight-gradient product reads.
extern "C" int s2t_linear_wgrad(int in_dtype, int n_out, int n_in, int tokens, const void* dY, int ldy, const void* X, int ldx,
                                float* dW, int ldw, float* db, int splitk, void* stream) {
    if (n_out <= 0 || n_in <= 0 || tokens <= 0) return (n_out < 0 || n_in < 0 || tokens < 0) ? S2T_EINVAL : S2T_OK;
    if (!dY || !X || !dW) return S2T_EINVAL;
    return gemm_run(in_dtype, S2T_F32, 1, 1, n_out, n_in, tokens, dY, ldy, X, ldx, dW, ldw, nullptr, nullptr, 0, nullptr, nullptr, 0,
                    ACT_NONE, 1, splitk < 1 ? 1 : splitk, 1.f, nullptr, 0, nullptr, nullptr, 0.f, 0ull, db, stream);
}

extern "C" int s2t_gemm(int in_dtype, int out_dtype, int trans_a, int trans_b, int M, int N, int K,
                        const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        const float* bias, const void* residual, int ldr, const void* aux, void* aux_out,
                        int ldaux, int act, int accumulate, int splitk, float alpha, void* stream) {
    return s2t_gemm_gather(in_dtype, out_dtype, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr,
                           aux, aux_out, ldaux, act, accumulate, splitk, alpha, nullptr, 0, nullptr, nullptr, 0.f, 0ull, stream);
}

// Column sums of a [M][N] activation-gradient matrix into an f32 vector (bias gradients):
// out[n] (+)= sum_m X[m][n].  Each lane owns 16 bytes of columns (8 bf16 / 4 f32), a workgroup = 4 row slots x
// 64 lanes; rows are grid-strided in chunks, partial sums meet in LDS, one f32 atomic per column per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, int ld, int M, int N, float* out, int rows_per_block, int vec, int lpr) {
    // lpr = lanes per row (power of two <= 64): a narrow matrix (N <= 64*E/2 columns) puts 64/lpr rows on one wave-instruction
    // instead of leaving most lanes idle
    constexpr int E = Elem<T>::PER16;
    __shared__ float sh[256][E + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rpw = 64 / lpr;                                   // rows per wave-instruction
    const int c0 = (blockIdx.x * lpr + (lane % lpr)) * E;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    if (c0 < N) {
        if (vec && c0 + E <= N) {
            for (int m = m0 + w * rpw + lane / lpr; m < m1; m += 4 * rpw) {
                T tmp[E];
                *reinterpret_cast<u32x4*>(tmp) = *reinterpret_cast<const u32x4*>(X + (size_t)m * ld + c0);
#pragma unroll
                for (int e = 0; e < E; ++e) s[e] += to_f32(tmp[e]);
            }
        } else {
            for (int m = m0 + w * rpw + lane / lpr; m < m1; m += 4 * rpw)
#pragma unroll
                for (int e = 0; e < E; ++e) if (c0 + e < N) s[e] += to_f32(X[(size_t)m * ld + c0 + e]);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) sh[threadIdx.x][e] = s[e];
    __syncthreads();
    for (int i = threadIdx.x; i < lpr * E; i += 256) {
        const int cl = i / E, e = i % E;
        const int c = (blockIdx.x * lpr + cl) * E + e;
        if (c >= N) continue;
        float t = 0.f;
        for (int k = cl; k < 256; k += lpr) t += sh[k][e];         // every thread whose lane % lpr == cl (lpr divides 64)
        atomicAdd(out + c, t);
    }
}

extern "C" int s2t_colsum(int dtype, const void* X, int ld, int M, int N, float* out, void* stream) {
    if (M <= 0 || N <= 0) return S2T_OK;
    if (!X || !out) return S2T_EINVAL;
    const int E = dtype == S2T_BF16 ? 8 : 4;
    const int vec = (ld % E == 0) && (((uintptr_t)X & 15) == 0);
    int lpr = 64;
    while (lpr > 1 && (lpr / 2) * E >= N) lpr /= 2;             // fewest lanes (power of two) that cover one row
    const int col_blocks = (N + lpr * E - 1) / (lpr * E);
    int rpb = (int)(((long)M * col_blocks + 1023) / 1024);       // aim at ~1024 workgroups (atomic contention per column)
    rpb = rpb < 256 ? 256 : rpb;
    dim3 grid(col_blocks, (M + rpb - 1) / rpb);
    if (dtype == S2T_BF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)X, ld, M, N, out, rpb, vec, lpr);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)X, ld, M, N, out, rpb, vec, lpr);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
