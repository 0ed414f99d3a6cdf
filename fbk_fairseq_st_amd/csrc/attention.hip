// Fused multi-head attention (forward + backward) for the S2T Transformer encoder/decoder.
// Reference semantics: fairseq/modules/multihead_attention.py:190-366 and
// torch.nn.functional.multi_head_attention_forward: S = (q*d^-1/2) k^T,
// -inf on padded keys (suffix padding given as per-batch key lengths) and above the diagonal
// (causal, decoder self-attention), softmax in fp32, dropout on P, O = P V.
//
// Flash-style: scores are never written to HBM.  K/V tiles of 64 keys are staged in LDS, QK^T and
// PV run on MFMA (16x16 tiles, f32 accumulate), softmax statistics are kept per query row in
// registers with wavefront (16-lane) shuffles.  Workgroup = 4 waves, each wave owns 16 rows.
// Tensors are addressed as  X[t][b][h][j] = base + t*st + b*sb + h*DH + j  so the kernels read
// q/k/v straight out of the fused QKV projection output (time-major) and write the
// context in the (T,B,D) layout the out-projection GEMM consumes.
//
// Backward = 3 kernels: delta (rowsum(dO*O)), dK/dV (one workgroup per 64 keys, loops over query
// tiles) and dQ (one workgroup per 64 queries, loops over key tiles): no atomics, deterministic.
#include "common.hpp"
#include "prof.hpp"
#include <cstdlib>
#include <type_traits>

template <typename T, int DH> struct ACfg {
    static constexpr int E = Elem<T>::PER16;
    static constexpr int ROWB = DH * (int)sizeof(T);      // bytes per row of a [64][DH] tile
    static constexpr int NCH = ROWB / 16;
    static constexpr int KG = NCH / 4;                      // MFMA k-groups across DH
    static constexpr int PROWB = 64 * (int)sizeof(T);     // bytes per row of a [*][64] tile
    static constexpr int PCH = PROWB / 16;
    static constexpr int PKG = PCH / 4;                     // k-groups across 64 keys / queries
    static constexpr int ND = DH / 16;                      // 16-wide output tiles across DH
    static constexpr int TILE = 64 * ROWB;                  // bytes of a [64][DH] tile
    static constexpr int TTILE = DH * PROWB;                // bytes of a [DH][64] tile (same)
    static constexpr int PTILE = 16 * PROWB;                // per-wave [16][64] round-trip tile
};

__device__ __forceinline__ int swz(int row, int c, int nch) { return c ^ (row & ((nch < 8 ? nch : 8) - 1)); }

// [64 rows][DH] tile, rows = consecutive time steps of one (b,h)
template <typename T, int DH>
__device__ __forceinline__ void stage_rows(char* lds, const T* g, long stride, int row0, int nvalid) {
    typedef ACfg<T, DH> C;
    for (int cid = threadIdx.x; cid < 64 * C::NCH; cid += 256) {
        const int row = cid / C::NCH, c = cid % C::NCH;
        u32x4 v = {0, 0, 0, 0};
        if (row0 + row < nvalid) v = *reinterpret_cast<const u32x4*>(g + (long)(row0 + row) * stride + c * C::E);
        *reinterpret_cast<u32x4*>(lds + row * C::ROWB + (swz(row, c, C::NCH) << 4)) = v;
    }
}

// transposed tile [DH rows = feature][64 cols = time step]
template <typename T, int DH>
__device__ __forceinline__ void stage_cols(char* lds, const T* g, long stride, int row0, int nvalid) {
    typedef ACfg<T, DH> C;
    if constexpr (sizeof(T) == 2) {
        for (int it = threadIdx.x; it < 32 * C::NCH; it += 256) {
            const int rp = it & 31, dc = it >> 5;
            u32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
            const int r0 = row0 + 2 * rp;
            if (r0 < nvalid) v0 = *reinterpret_cast<const u32x4*>(g + (long)r0 * stride + dc * 8);
            if (r0 + 1 < nvalid) v1 = *reinterpret_cast<const u32x4*>(g + (long)(r0 + 1) * stride + dc * 8);
            const uint16_t* a = reinterpret_cast<const uint16_t*>(&v0);
            const uint16_t* b = reinterpret_cast<const uint16_t*>(&v1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = dc * 8 + e;
                const uint32_t w = (uint32_t)a[e] | ((uint32_t)b[e] << 16);
                *reinterpret_cast<uint32_t*>(lds + d * C::PROWB + (swz(d, rp >> 2, C::PCH) << 4) + ((rp & 3) << 2)) = w;
            }
        }
    } else {
        for (int it = threadIdx.x; it < 64 * C::NCH; it += 256) {
            const int r = it & 63, dc = it >> 6;
            u32x4 v = {0, 0, 0, 0};
            if (row0 + r < nvalid) v = *reinterpret_cast<const u32x4*>(g + (long)(row0 + r) * stride + dc * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = dc * 4 + e;
                *reinterpret_cast<uint32_t*>(lds + d * C::PROWB + (swz(d, r >> 2, C::PCH) << 4) + ((r & 3) << 2)) = v[e];
            }
        }
    }
}

__device__ __forceinline__ u32x4 frag(const char* lds, int row, int chunk, int rowb, int nch) {
    return *reinterpret_cast<const u32x4*>(lds + row * rowb + (swz(row, chunk, nch) << 4));
}

// accumulator-layout [16][64] values (acc[j][r]: row 4q+r, col 16j+r16) -> per-wave LDS tile, as T
template <typename T>
__device__ __forceinline__ void put_tile(char* lds, const f32x4 (&acc)[4], int q, int r16) {
    constexpr int PROWB = 64 * (int)sizeof(T), PCH = PROWB / 16;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * q + r, byte = (16 * j + r16) * (int)sizeof(T);
            *reinterpret_cast<T*>(lds + row * PROWB + (swz(row, byte >> 4, PCH) << 4) + (byte & 15)) = from_f32<T>(acc[j][r]);
        }
}
// transposed variant: acc rows become LDS columns ([16 cols... used when the wave owns 16 "rows"
// of the TRANSPOSED product: acc[j][r] = X^T[own 4q+r][other 16j+r16] is stored as is (same as put_tile).

// short query blocks over long key ranges (the decoder's encoder-attention: Tq = 40, Tk = 368) also take the second-generation
// kernels: a workgroup is mostly padding rows there, but each key tile costs a fraction of what the first-generation kernels spend
#define S2T_ATTN_V2_MIN_TQ g_s2t_opt_attn_v2_min_tq      /* s2t_set_option("attn_v2_min_tq") */

struct AttnArgs {
    const void *Q, *K, *V; void* O; float* LSE;          // LSE [B][H][Tq]
    const void *dO; void *dQ, *dK, *dV; const float* Delta;
    long q_st, q_sb, k_st, k_sb, v_st, v_sb, o_st, o_sb;  // element strides (time, batch)
    long dq_st, dq_sb, dk_st, dk_sb, dv_st, dv_sb, do_st, do_sb;
    const int* klen;                                       // [B] valid keys (null = Tk)
    int B, H, Tq, Tk, causal;
    int dist_pen;                                          // 1: scores -= max(0, ln|query - key|)  (LocalAttention + LogPenalty)
    float scale, p_drop; unsigned long long seed;
};

// dropout element index of (b, h, query, key): key rows are padded to a multiple of 4 so that 4 consecutive keys from a
// multiple of 4 form one hash quad (common.hpp drop_hash4) in every kernel's register layout
__device__ __forceinline__ uint64_t drop_index(const AttnArgs& p, int b, int h, int qrow, int key) {
    return (((uint64_t)b * p.H + h) * p.Tq + qrow) * (uint64_t)((p.Tk + 3) & ~3) + key;
}
__device__ __forceinline__ float drop_scale(const AttnArgs& p, int b, int h, int qrow, int key) {
    if (p.p_drop <= 0.f) return 1.f;
    const uint32_t th = (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f);
    return dropout_keep(p.seed, drop_index(p, b, h, qrow, key), th) ? 1.f / (1.f - p.p_drop) : 0.f;
}

// pair of bf16 bit patterns (non-negative values) times the keep decisions of a pair of 16-bit uniforms: field >= th  <=>  the
// saturating field - (th - 1) is not zero.  thm1x2 = (th - 1) in both halves.
__device__ __forceinline__ uint32_t drop_pair(uint32_t pair, uint32_t fields, uint32_t thm1x2) {
    uint32_t d, k, r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(fields), "v"(thm1x2));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(k) : "v"(d), "v"(0x00010001u));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(pair), "v"(k));
    return r;
}

// distance penalty of examples/speech_recognition/modules/local_attention.py:131-133 with LogPenalty
// (modules/conv_transformer_layer.py:22-27): max(0, ln|i - j|), subtracted from the scaled scores before the softmax.
// A constant additive bias: the backward pass only sees it through P.
__device__ __forceinline__ float dist_pen_ln(int qrow, int key) {
    const int d = qrow > key ? qrow - key : key - qrow;
    return d > 1 ? __logf((float)d) : 0.f;
}
__device__ __forceinline__ float dist_pen_log2(int qrow, int key) {
    const int d = qrow > key ? qrow - key : key - qrow;
    return d > 1 ? __builtin_amdgcn_logf((float)d) : 0.f;          // v_log_f32 = log2
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
    typedef ACfg<T, DH> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;
    char* ldsVt = smem + C::TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    char* ldsP = smem + 2 * C::TILE + wave * C::PTILE;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const T* Qg = reinterpret_cast<const T*>(p.Q) + (long)b * p.q_sb + (long)h * DH;
    const T* Kg = reinterpret_cast<const T*>(p.K) + (long)b * p.k_sb + (long)h * DH;
    const T* Vg = reinterpret_cast<const T*>(p.V) + (long)b * p.v_sb + (long)h * DH;

    u32x4 qf[C::KG];
#pragma unroll
    for (int g = 0; g < C::KG; ++g) {
        qf[g] = (u32x4){0, 0, 0, 0};
        if (q0 + r16 < p.Tq) qf[g] = *reinterpret_cast<const u32x4*>(Qg + (long)(q0 + r16) * p.q_st + (4 * g + q) * C::E);
    }
    f32x4 o[C::ND];
#pragma unroll
    for (int n = 0; n < C::ND; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l[4] = {0.f, 0.f, 0.f, 0.f};

    int kv_end = klen;
    if (p.causal) kv_end = min(kv_end, blockIdx.x * 64 + 64);     // block-uniform bound
    for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
        __syncthreads();
        stage_rows<T, DH>(ldsK, Kg, p.k_st, kv0, klen);
        stage_cols<T, DH>(ldsVt, Vg, p.v_st, kv0, klen);
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < C::KG; ++g)
                s[j] = mma16<T>(qf[g], frag(ldsK, 16 * j + r16, 4 * g + q, C::ROWB, C::NCH), s[j]);
        }
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = kv0 + 16 * j + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qrow = q0 + 4 * q + r;
                const bool ok = key < klen && (!p.causal || key <= qrow);
                s[j][r] = ok ? s[j][r] * p.scale - (p.dist_pen ? dist_pen_ln(qrow, key) : 0.f) : -INFINITY;
                mx[r] = fmaxf(mx[r], s[j][r]);
            }
        }
        float alpha[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = mx[r];
            v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2));
            v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
            const float mn = fmaxf(m[r], v);
            const float mu = (mn == -INFINITY) ? 0.f : mn;
            alpha[r] = __expf(m[r] - mu);          // m = -inf -> 0
            m[r] = mn;
            mx[r] = mu;
            rs[r] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(s[j][r] - mx[r]);
                rs[r] += pv;
                s[j][r] = pv * drop_scale(p, b, h, q0 + 4 * q + r, kv0 + 16 * j + r16);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = rs[r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            l[r] = l[r] * alpha[r] + v;
        }
#pragma unroll
        for (int n = 0; n < C::ND; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[n][r] *= alpha[r];
        put_tile<T>(ldsP, s, q, r16);                  // per-wave region: only this wave reads it back
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): LDS writes of this wave done
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < C::PKG; ++g) {
            const u32x4 pa = frag(ldsP, r16, 4 * g + q, C::PROWB, C::PCH);
#pragma unroll
            for (int n = 0; n < C::ND; ++n)
                o[n] = mma16<T>(pa, frag(ldsVt, 16 * n + r16, 4 * g + q, C::PROWB, C::PCH), o[n]);
        }
    }
    T* Og = reinterpret_cast<T*>(p.O) + (long)b * p.o_sb + (long)h * DH;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * q + r;
        if (qrow >= p.Tq) continue;
        const float inv = l[r] > 0.f ? 1.f / l[r] : 0.f;
#pragma unroll
        for (int n = 0; n < C::ND; ++n) Og[(long)qrow * p.o_st + 16 * n + r16] = from_f32<T>(o[n][r] * inv);
        if (r16 == 0 && p.LSE) p.LSE[((long)b * p.H + h) * p.Tq + qrow] = m[r] + logf(l[r]);
    }
}

// ------------------------------------------------------------------------------------ forward, bf16 d = 64, long queries
// Second-generation forward for the encoder shapes (Tq >= 128): a wave owns 32 queries (two 16-query column blocks), a
// workgroup 128, so every K / V fragment read from LDS feeds two MFMAs.  The products are taken transposed,
//   S^T = K Q^T   (A = K rows from LDS, B = Q fragments kept in registers)      acc[r] = S[query r16][key 16j + 4q + r]
//   O^T = V^T P^T (A = V gathered by ds_read_b64_tr_b16, B = P straight from the S^T accumulators)
// so a query is a lane column in both: the softmax statistics are per lane (two xor-shuffles per reduction instead of
// a 16-lane butterfly), the rescale of O needs no exchange, and P never travels through LDS: the 8 keys a lane holds
// for a 32-key block (4q..4q+3 of its two 16-key tiles) are exactly the k-slice of its P^T operand once V's transposed
// reads fetch the same keys.  V is staged as it lies in memory (no transposing scatter); K/V tiles are double-buffered
// with register prefetch, one barrier per 64 keys.  Dropout mask = the same per-(query, key) hash as every other kernel.
// The workgroups of one (batch, head) read the same K/V (or Q/dO) tiles: put them on one XCD so its L2 serves the re-reads
// (PMC: the forward fetched 2.1x its operand bytes with the tiles of a head dealt round-robin over the 8 XCDs).
// Linear id L -> XCD L % 8 (dispatch order x fastest); returns the tile index within the head, sets the head index.
__device__ __forceinline__ int head_xcd_remap(int& head, int nheads, int ntile) {
    const int L = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (nheads & 7) { head = (int)blockIdx.z * gridDim.y + blockIdx.y; return blockIdx.x; }
    const int slot = L >> 3;
    head = (L & 7) + 8 * (slot / ntile);
    return slot % ntile;
}
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {          // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
// max over the four 16-lane rows of a wave (lanes r, r + 16, r + 32, r + 48): two VALU lane swaps instead of two LDS round trips.
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second, v_permlane32_swap the upper
// half of the first with the lower half of the second: with both operands = x, (a, b) = (x0 x0 x2 x2, x1 x1 x3 x3), then halves.
__device__ __forceinline__ float max_over_rows(float v) {
    uint32_t x = __builtin_bit_cast(uint32_t, v);
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    v = fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
    x = __builtin_bit_cast(uint32_t, v);
    r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
// Diagnostic build only (make dbg; tools/attn_timeline.py): s_memtime at the phase boundaries of every tile, first and a middle
// workgroup, lane 0 of each wave, into a buffer of their own (s2t_dbg_attn_stamps).  Each stamp waits for its own return
// (lgkmcnt(0)), which perturbs the schedule.  [kernel 0 fwd / 1 dq / 2 dkv][workgroup slot 4][wave 4][tile 8][stamp 8]; tile 7 = entry, loop start, loop end, exit
#ifdef S2T_ATTN_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;
extern "C" int s2t_dbg_attn_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define ASTAMP_INIT(KID)                                                                                         \
    const int lin_ = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                        \
    const int slot_ = lin_ == 0 ? 0 : (lin_ == 700 ? 1 : (lin_ == 1100 ? 2 : (lin_ == 1500 ? 3 : -1)));                                                    \
    unsigned long long* const stp_ = (g_attn_stamps && slot_ >= 0) ? g_attn_stamps + (((KID) * 4 + slot_) * 4 + (threadIdx.x >> 6)) * 64 : nullptr;
#define ASTAMP(T, K)                                                                                             \
    if (stp_ && (T) < 8) {                                                                                       \
        unsigned long long t_;                                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                             \
        if ((threadIdx.x & 63) == 0) stp_[(T) * 8 + (K)] = t_;                                                   \
    }
#define ASTAMP_EARLY(KID)  /* kernel entry: the init computes its own slot, the stamp goes to [tile 7][0] */                \
    ASTAMP_INIT(KID) ASTAMP(7, 0)
#else
#define ASTAMP_INIT(KID)
#define ASTAMP(T, K)
#define ASTAMP_EARLY(KID)
#endif
__global__ __launch_bounds__(256, 3) void attn_fwd2_kernel(AttnArgs p) {
    constexpr int DH = 64;
    ASTAMP_EARLY(0)
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x (K 8 KiB | V 8 KiB), 512 B of score offsets, Q 16 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    int head;
    const int bx = head_xcd_remap(head, p.B * p.H, gridDim.x);
    const int b = head / p.H, h = head % p.H, qw = bx * 128 + wave * 32;
    const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const bf16* Qg = reinterpret_cast<const bf16*>(p.Q) + (long)b * p.q_sb + (long)h * DH;
    const bf16* Kg = reinterpret_cast<const bf16*>(p.K) + (long)b * p.k_sb + (long)h * DH;
    const bf16* Vg = reinterpret_cast<const bf16*>(p.V) + (long)b * p.v_sb + (long)h * DH;

    f32x4 o[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int n = 0; n < 4; ++n) o[qb][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};

    int kv_end = klen;
    if (p.causal) kv_end = min(kv_end, bx * 128 + 128);
    const int ntile = (kv_end + 63) / 64;

    // staging: thread -> (row, chunk) of the K and of the V tile, two 16-byte pieces each
    // Staging by LDS-DMA (global_load_lds, 16 bytes per lane, 1 KiB = 8 tile rows per wave-instruction, lane-linear in LDS): no
    // staging registers (the kernel then fits three workgroups per CU) and no ds_write pass.  The XOR swizzle of the 16-byte
    // chunks is applied on the SOURCE side: the lane that fills position (row, pos) fetches chunk pos ^ swz(row), so chunk c sits
    // at position c ^ swz(row): K rows are read back as b128 rows (swz = row & 7), V by transposing reads (swz = row & 6).
    // Rows past klen are NOT zeroed: the address is clamped to the last row of the tensor (finite data), their scores are masked
    // to -inf and their probabilities are exact zeros.
    const uint32_t kst2 = (uint32_t)p.k_st * 2u, vst2 = (uint32_t)p.v_st * 2u;
    auto stage = [&](int kv0, char* st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r8 = 8 * (wave + 4 * i), row = r8 + (lane >> 3), pos = lane & 7;
            // 32-bit byte offsets from the head's (uniform) base, one v_mad_u32_u24 each: row strides below 16 MiB and tensors below
            // 4 GiB (checked by the launcher).  The 64-bit form was two quarter-rate multiplies and a 64-bit mad per address, every tile.
            const uint32_t r = (uint32_t)min(kv0 + row, p.Tk - 1);
            char* dst = st + __builtin_amdgcn_readfirstlane(r8 * 128);
            const char* ka = reinterpret_cast<const char*>(Kg) + (__umul24(r, kst2) + (uint32_t)((pos ^ (row & 7)) << 4));
            const char* va = reinterpret_cast<const char*>(Vg) + (__umul24(r, vst2) + (uint32_t)((pos ^ (row & 6)) << 4));
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)ka, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)va, (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
        }
    };
    const uint32_t drop_th16 = (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) >> 16;
    const uint32_t drop_thm1x2 = (drop_th16 - 1u) * 0x00010001u;
    const bool has_drop = drop_th16 > 0;               // a rate below 2^-16 keeps every element (field >= 0), as in every other kernel
    const float drop_inv = 1.f / (1.f - p.p_drop);
    const uint32_t drop_ks = drop_seed_key(p.seed);
    const float sc2 = p.scale * 1.44269504088896f;
    // the whole dropout index space of this call in one 32-bit quad word (always, short of 17 G attention probabilities)
    const bool plain = !p.causal && !p.dist_pen && (uint64_t)p.B * p.H * p.Tq * (uint64_t)((p.Tk + 3) & ~3) < (1ull << 34);

    // Score offsets of the last key tile: 0 for its keys below klen, -inf past them; the row of zeros next to it serves every other
    // tile.  The QK^T accumulators START from these (read from LDS, no VALU), so the tail of a plain softmax needs no mask per element
    // and no second code path (a second path between QK^T and the softmax costs this kernel spilled registers).
    float* sInit = reinterpret_cast<float*>(smem + 32768);              // [2][64]
    if (threadIdx.x < 128) {
        const int i = threadIdx.x & 63;
        sInit[threadIdx.x] = (threadIdx.x < 64 || (ntile - 1) * 64 + i < klen) ? 0.f : -INFINITY;
    }
    // Prologue: the first K/V tile's DMA goes out BEFORE the Q fragment loads, so the two round trips to memory overlap (one after the
    // other they were ~6,000 cycles of a workgroup's ~50,000).  The Q fragments must have landed before the loop: left pending, the
    // compiler's wait for them sits behind the loop's own prefetch in the in-order counter and becomes a vmcnt(0) -- the full
    // latency of the K/V prefetch, exposed, every iteration.
    if (ntile > 0) stage(0, smem);
    // The workgroup's 128 query rows are parked in LDS (LDS-DMA, the K tile's swizzle) and their fragments re-read every tile: 16
    // VGPRs less than holding them (the kernel sits at its 168-register budget for three waves per SIMD).  Rows past Tq repeat the
    // last query (finite); their columns are never stored.
    char* sQ = smem + 32768 + 512;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r8 = 8 * (wave + 4 * i), row = r8 + (lane >> 3), pos = lane & 7;
        const long r = min(bx * 128 + row, p.Tq - 1);
        char* dst = sQ + __builtin_amdgcn_readfirstlane(r8 * 128);
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(Qg + r * p.q_st + ((pos ^ (row & 7)) << 3)),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): the DMA is not a register write, the compiler inserts no wait for it
    __syncthreads();
    ASTAMP(7, 1)
    for (int t = 0; t < ntile; ++t) {
        const int kv0 = t * 64;
        const char* sK = smem + (t & 1) * 16384;
        const char* sV = sK + 8192;
        ASTAMP(t, 0)
        if (t + 1 < ntile) stage(kv0 + 64, smem + ((t + 1) & 1) * 16384);   // its readers of two tiles ago passed the last barrier
        ASTAMP(t, 1)
        // ---- S^T = K Q^T
        f32x4 s[2][4];
        const float* sInit0 = sInit + ((plain && t == ntile - 1) ? 64 : 0) + 4 * q;
        u32x4 qf[2][2];                                // [query block][k-group of d]
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int row = wave * 32 + 16 * qb + r16;
                qf[qb][g] = *reinterpret_cast<const u32x4*>(sQ + row * 128 + (((4 * g + q) ^ (row & 7)) << 4));
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 kf[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int row = 16 * j + r16;
                kf[g] = *reinterpret_cast<const u32x4*>(sK + row * 128 + (((4 * g + q) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                s[qb][j] = *reinterpret_cast<const f32x4*>(sInit0 + 16 * j);
#pragma unroll
                for (int g = 0; g < 2; ++g) s[qb][j] = mma16<bf16>(kf[g], qf[qb][g], s[qb][j]);
            }
        }
        // ---- online softmax, per lane = per query (base-2 domain: p = exp2(s * scale * log2 e - max)); interior tiles carry no
        // masks.  VALU is what bounds this kernel (32 MFMAs against ~700 VALU slots per 64 keys), so the per-element work is
        // kept to: a share of a max3 on the RAW score (the scale is positive), one fma, the exp2, a share of a packed add for the
        // row sum, a share of the bf16 pair convert, and for dropout a 16-bit compare + select on the value.  The 1/(1-p) of the
        // dropout is a constant factor of O: it is applied once, with the 1/l normalisation, at the end.
        u32x4 pf[2][2];                                // [query block][32-key block]
        // Two instantiations: FAST = interior tile of a plain softmax with an index space below 2^34 elements (no masks, no
        // distance penalty, the dropout quad index stays in its low word); the other handles everything dynamically.
        auto softmax_tile = [&](auto fast_tag, auto drop_tag) {
            constexpr bool FAST = decltype(fast_tag)::value, DROP = decltype(drop_tag)::value;      // DROP only narrows FAST
            const bool pen = !FAST && p.dist_pen;
            const float k2 = pen ? 1.f : sc2;          // pen: the scores are moved to the log2 domain first (scale, then - log2 distance)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int qrow = qw + 16 * qb + r16;
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (!FAST) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = s[qb][j][r];
                            const int key = kv0 + 16 * j + 4 * q + r;
                            if (pen) v = v * sc2 - dist_pen_log2(qrow, key);
                            v = (key < klen && (!p.causal || key <= qrow)) ? v : -INFINITY;
                            s[qb][j][r] = v;
                        }
                    }
                    mx = fmaxf(fmaxf(mx, s[qb][j][0]), s[qb][j][1]);
                    mx = fmaxf(fmaxf(mx, s[qb][j][2]), s[qb][j][3]);
                }
                mx = max_over_rows(mx);
                const float mn = fmaxf(m[qb], mx * k2);                    // -inf * k2 = -inf (k2 > 0)
                const float mu = (mn == -INFINITY) ? 0.f : mn;
                const float alpha = __builtin_amdgcn_exp2f(m[qb] - mu);    // m = -inf -> 0
                m[qb] = mn;
                f32x2_t rs2 = {0.f, 0.f};
                // dropout: the quad of keys 4q .. 4q+3 of 16-key tile j has index quad0 + 4 j (rows are a multiple of 4 keys long)
                const bool drop = FAST ? DROP : has_drop;
                const uint64_t quad0 = drop ? drop_index(p, b, h, qrow, kv0 + 4 * q) >> 2 : 0;
                const uint32_t qlo = (uint32_t)quad0, hwm0 = drop_high_mix(p.seed, quad0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][j][r], k2, -mu));
                    rs2 += (f32x2_t){pv[0], pv[1]};
                    rs2 += (f32x2_t){pv[2], pv[3]};
                    uint32_t p01 = pack_bf16(pv[0], pv[1]), p23 = pack_bf16(pv[2], pv[3]);
                    if (drop) {
                        // the mask on the PACKED pairs: the hash words hold the 16-bit fields in the order of the pairs (fields 0, 1 =
                        // y.lo, y.hi; 2, 3 = x.lo, x.hi), so "field >= threshold" is a saturating packed subtract of threshold - 1,
                        // a packed min with 1, and the packed product of the bf16 bit patterns with that 0 / 1: 3 VALU per pair
                        // where extract + compare + select per element took 6
                        const u32x2 hq = FAST ? drop_hash4_lo(drop_ks, hwm0, qlo + 4u * j) : drop_hash4(p.seed, quad0 + 4 * j);
                        p01 = drop_pair(p01, hq[1], drop_thm1x2); p23 = drop_pair(p23, hq[0], drop_thm1x2);
                    }
                    pf[qb][j >> 1][2 * (j & 1)] = p01;
                    pf[qb][j >> 1][2 * (j & 1) + 1] = p23;
                }
                l[qb] = l[qb] * alpha + (rs2[0] + rs2[1]);
#pragma unroll
                for (int n = 0; n < 4; ++n) o[qb][n] *= alpha;
            }
        };
        if (!plain) softmax_tile(std::false_type{}, std::false_type{});
        else if (has_drop) softmax_tile(std::true_type{}, std::true_type{});
        else softmax_tile(std::true_type{}, std::false_type{});
        ASTAMP(t, 2)
        // ---- O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                u32x4 vf;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row = 32 * kb + 16 * hh + 4 * q + (r16 >> 2);
                    const int ch = 2 * n + ((r16 & 3) >> 1);
                    const char* a = sV + row * 128 + ((ch ^ (row & 6)) << 4) + ((r16 & 1) << 3);
                    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a);
                    const u32x2 w = __builtin_bit_cast(u32x2, v);
                    vf[2 * hh] = w[0]; vf[2 * hh + 1] = w[1];
                }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) o[qb][n] = mma16<bf16>(vf, pf[qb][kb], o[qb][n]);
            }
        ASTAMP(t, 3)
        __builtin_amdgcn_s_waitcnt(0x0F70);                              // tile t + 1 has landed (this wave's share; the barrier covers the rest)
        ASTAMP(t, 4)
        __syncthreads();
        ASTAMP(t, 5)
    }
    ASTAMP(7, 2)
    bf16* Og = reinterpret_cast<bf16*>(p.O) + (long)b * p.o_sb + (long)h * DH;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = qw + 16 * qb + r16;
        float lt = l[qb];
        lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
        if (qrow >= p.Tq) continue;
        const float inv = lt > 0.f ? drop_inv / lt : 0.f;                 // the dropout's 1/(1-p) rides on the normalisation
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            u32x2 w;
            w[0] = pack_bf16(o[qb][n][0] * inv, o[qb][n][1] * inv);
            w[1] = pack_bf16(o[qb][n][2] * inv, o[qb][n][3] * inv);
            *reinterpret_cast<u32x2*>(Og + (long)qrow * p.o_st + 16 * n + 4 * q) = w;
        }
        if (q == 0 && p.LSE) p.LSE[((long)b * p.H + h) * p.Tq + qrow] = m[qb] * 0.693147180559945f + logf(lt);
    }
    ASTAMP(7, 3)
}

// ------------------------------------------------------------------------------------ delta
// Delta[b][h][t] = sum_j dO[t,b,h,j] * O[t,b,h,j]   (one 16-lane group per row)
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs p) {
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 4, li = threadIdx.x & 15;
    const long total = (long)p.B * p.H * p.Tq;
    float s = 0.f;
    if (g < total) {
        const int t = g % p.Tq, h = (g / p.Tq) % p.H, b = g / (p.Tq * p.H);
        const T* o = reinterpret_cast<const T*>(p.O) + (long)t * p.o_st + (long)b * p.o_sb + (long)h * DH;
        const T* d = reinterpret_cast<const T*>(p.dO) + (long)t * p.do_st + (long)b * p.do_sb + (long)h * DH;
        for (int j = li; j < DH; j += 16) s += to_f32(o[j]) * to_f32(d[j]);
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
    if (g < total && li == 0) {
        const int t = g % p.Tq, h = (g / p.Tq) % p.H, b = g / (p.Tq * p.H);
        const_cast<float*>(p.Delta)[((long)b * p.H + h) * p.Tq + t] = s;
    }
}

// ------------------------------------------------------------------------------------ dK, dV
// Workgroup: 64 keys of one (b,h); wave w owns keys [16w,16w+16).  Loop over 64-query tiles.
//   St = K Q^T (rows = keys)    Pt = exp(St*scale - LSE[q])     dV += (D*Pt) dO
//   dPt = V dO^T                dSt = Pt * (D*dPt - Delta[q]) * scale       dK += dSt Q
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs p) {
    typedef ACfg<T, DH> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsQ = smem;                       // [64 q][DH]
    char* ldsDO = smem + C::TILE;            // [64 q][DH]
    char* ldsQt = smem + 2 * C::TILE;        // [DH][64 q]
    char* ldsDOt = smem + 3 * C::TILE;       // [DH][64 q]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    char* ldsP = smem + 4 * C::TILE + wave * 2 * C::PTILE;
    char* ldsS = ldsP + C::PTILE;
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64 + wave * 16;
    const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const T* Qg = reinterpret_cast<const T*>(p.Q) + (long)b * p.q_sb + (long)h * DH;
    const T* Kg = reinterpret_cast<const T*>(p.K) + (long)b * p.k_sb + (long)h * DH;
    const T* Vg = reinterpret_cast<const T*>(p.V) + (long)b * p.v_sb + (long)h * DH;
    const T* dOg = reinterpret_cast<const T*>(p.dO) + (long)b * p.do_sb + (long)h * DH;
    const float* lse = p.LSE + ((long)b * p.H + h) * p.Tq;
    const float* dlt = p.Delta + ((long)b * p.H + h) * p.Tq;

    u32x4 kf[C::KG], vf[C::KG];
#pragma unroll
    for (int g = 0; g < C::KG; ++g) {
        kf[g] = vf[g] = (u32x4){0, 0, 0, 0};
        if (k0 + r16 < klen) {
            kf[g] = *reinterpret_cast<const u32x4*>(Kg + (long)(k0 + r16) * p.k_st + (4 * g + q) * C::E);
            vf[g] = *reinterpret_cast<const u32x4*>(Vg + (long)(k0 + r16) * p.v_st + (4 * g + q) * C::E);
        }
    }
    f32x4 dk[C::ND], dv[C::ND];
#pragma unroll
    for (int n = 0; n < C::ND; ++n) { dk[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[n] = dk[n]; }

    const int qstart = p.causal ? (blockIdx.x * 64) : 0;      // queries before the first key see nothing
    for (int qt = qstart; qt < p.Tq; qt += 64) {
        __syncthreads();
        stage_rows<T, DH>(ldsQ, Qg, p.q_st, qt, p.Tq);
        stage_rows<T, DH>(ldsDO, dOg, p.do_st, qt, p.Tq);
        stage_cols<T, DH>(ldsQt, Qg, p.q_st, qt, p.Tq);
        stage_cols<T, DH>(ldsDOt, dOg, p.do_st, qt, p.Tq);
        __syncthreads();
        f32x4 st[4], dp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = st[j];
#pragma unroll
            for (int g = 0; g < C::KG; ++g) {
                st[j] = mma16<T>(kf[g], frag(ldsQ, 16 * j + r16, 4 * g + q, C::ROWB, C::NCH), st[j]);
                dp[j] = mma16<T>(vf[g], frag(ldsDO, 16 * j + r16, 4 * g + q, C::ROWB, C::NCH), dp[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qrow = qt + 16 * j + r16;
            const bool qok = qrow < p.Tq;
            const float L = qok ? lse[qrow] : 0.f, Dl = qok ? dlt[qrow] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + 4 * q + r;
                const bool ok = qok && key < klen && (!p.causal || key <= qrow);
                const float pv = ok ? __expf(st[j][r] * p.scale - (p.dist_pen ? dist_pen_ln(qrow, key) : 0.f) - L) : 0.f;
                const float ds = ok ? drop_scale(p, b, h, qrow, key) : 0.f;
                st[j][r] = pv * ds;                                  // D*P   (for dV)
                dp[j][r] = pv * (ds * dp[j][r] - Dl) * p.scale;      // dS    (for dK)
            }
        }
        put_tile<T>(ldsP, st, q, r16);
        put_tile<T>(ldsS, dp, q, r16);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < C::PKG; ++g) {
            const u32x4 pa = frag(ldsP, r16, 4 * g + q, C::PROWB, C::PCH);
            const u32x4 sa = frag(ldsS, r16, 4 * g + q, C::PROWB, C::PCH);
#pragma unroll
            for (int n = 0; n < C::ND; ++n) {
                dv[n] = mma16<T>(pa, frag(ldsDOt, 16 * n + r16, 4 * g + q, C::PROWB, C::PCH), dv[n]);
                dk[n] = mma16<T>(sa, frag(ldsQt, 16 * n + r16, 4 * g + q, C::PROWB, C::PCH), dk[n]);
            }
        }
    }
    T* dKg = reinterpret_cast<T*>(p.dK) + (long)b * p.dk_sb + (long)h * DH;
    T* dVg = reinterpret_cast<T*>(p.dV) + (long)b * p.dv_sb + (long)h * DH;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = k0 + 4 * q + r;
        if (key >= p.Tk) continue;
#pragma unroll
        for (int n = 0; n < C::ND; ++n) {
            dKg[(long)key * p.dk_st + 16 * n + r16] = from_f32<T>(dk[n][r]);
            dVg[(long)key * p.dv_st + 16 * n + r16] = from_f32<T>(dv[n][r]);
        }
    }
}

// ------------------------------------------------------------------------------------ dQ
// Workgroup: 64 queries of one (b,h); wave w owns rows [16w,16w+16).  Loop over 64-key tiles.
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
    typedef ACfg<T, DH> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;                       // [64 keys][DH]
    char* ldsV = smem + C::TILE;             // [64 keys][DH]
    char* ldsKt = smem + 2 * C::TILE;        // [DH][64 keys]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    char* ldsS = smem + 3 * C::TILE + wave * C::PTILE;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const T* Qg = reinterpret_cast<const T*>(p.Q) + (long)b * p.q_sb + (long)h * DH;
    const T* Kg = reinterpret_cast<const T*>(p.K) + (long)b * p.k_sb + (long)h * DH;
    const T* Vg = reinterpret_cast<const T*>(p.V) + (long)b * p.v_sb + (long)h * DH;
    const T* dOg = reinterpret_cast<const T*>(p.dO) + (long)b * p.do_sb + (long)h * DH;
    const float* lse = p.LSE + ((long)b * p.H + h) * p.Tq;
    const float* dlt = p.Delta + ((long)b * p.H + h) * p.Tq;

    u32x4 qf[C::KG], dof[C::KG];
#pragma unroll
    for (int g = 0; g < C::KG; ++g) {
        qf[g] = dof[g] = (u32x4){0, 0, 0, 0};
        if (q0 + r16 < p.Tq) {
            qf[g] = *reinterpret_cast<const u32x4*>(Qg + (long)(q0 + r16) * p.q_st + (4 * g + q) * C::E);
            dof[g] = *reinterpret_cast<const u32x4*>(dOg + (long)(q0 + r16) * p.do_st + (4 * g + q) * C::E);
        }
    }
    float L[4], Dl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * q + r;
        L[r] = qrow < p.Tq ? lse[qrow] : 0.f;
        Dl[r] = qrow < p.Tq ? dlt[qrow] : 0.f;
    }
    f32x4 dq[C::ND];
#pragma unroll
    for (int n = 0; n < C::ND; ++n) dq[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kv_end = klen;
    if (p.causal) kv_end = min(kv_end, blockIdx.x * 64 + 64);
    for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
        __syncthreads();
        stage_rows<T, DH>(ldsK, Kg, p.k_st, kv0, klen);
        stage_rows<T, DH>(ldsV, Vg, p.v_st, kv0, klen);
        stage_cols<T, DH>(ldsKt, Kg, p.k_st, kv0, klen);
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j];
#pragma unroll
            for (int g = 0; g < C::KG; ++g) {
                s[j] = mma16<T>(qf[g], frag(ldsK, 16 * j + r16, 4 * g + q, C::ROWB, C::NCH), s[j]);
                dp[j] = mma16<T>(dof[g], frag(ldsV, 16 * j + r16, 4 * g + q, C::ROWB, C::NCH), dp[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = kv0 + 16 * j + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qrow = q0 + 4 * q + r;
                const bool ok = qrow < p.Tq && key < klen && (!p.causal || key <= qrow);
                const float pv = ok ? __expf(s[j][r] * p.scale - (p.dist_pen ? dist_pen_ln(qrow, key) : 0.f) - L[r]) : 0.f;
                const float ds = ok ? drop_scale(p, b, h, qrow, key) : 0.f;
                dp[j][r] = pv * (ds * dp[j][r] - Dl[r]) * p.scale;
            }
        }
        put_tile<T>(ldsS, dp, q, r16);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < C::PKG; ++g) {
            const u32x4 sa = frag(ldsS, r16, 4 * g + q, C::PROWB, C::PCH);
#pragma unroll
            for (int n = 0; n < C::ND; ++n)
                dq[n] = mma16<T>(sa, frag(ldsKt, 16 * n + r16, 4 * g + q, C::PROWB, C::PCH), dq[n]);
        }
    }
    T* dQg = reinterpret_cast<T*>(p.dQ) + (long)b * p.dq_sb + (long)h * DH;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * q + r;
        if (qrow >= p.Tq) continue;
#pragma unroll
        for (int n = 0; n < C::ND; ++n) dQg[(long)qrow * p.dq_st + 16 * n + r16] = from_f32<T>(dq[n][r]);
    }
}

// ------------------------------------------------------------------------------------ backward, second generation
// Same construction as attn_fwd2_kernel (bf16, d = 64): every product is oriented so that the index a wave OWNS is the
// lane column of its accumulators, P / dS are consumed as MFMA operands straight from the accumulators of the product
// that made them (the 8 values a lane holds for a 32-long contraction block are its k-slice once the other operand's
// transposed LDS read fetches the same rows), nothing is round-tripped through LDS, a wave owns 32 rows (two blocks of
// 16) so each staged fragment feeds two MFMAs, tiles are double-buffered with register prefetch.
//   dK/dV kernel (owns keys, loops over 64-query tiles of Q and dO):
//     S  = Q K^T,  dP = dO V^T        acc[r] = X[query 16i+4q+r][key r16]     (A = LDS row reads, B = K / V registers)
//     dV^T += dO^T (D*P),  dK^T += Q^T dS   acc[r] = Y^T[d 16n+4q+r][key r16]  (A = transposed reads of the same tiles)
//   dQ kernel (owns queries, loops over 64-key tiles of K and V):
//     S^T = K Q^T, dP^T = V dO^T      acc[r] = X^T[key 16j+4q+r][query r16]
//     dQ^T += K^T dS^T                acc[r] = dQ^T[d 16n+4q+r][query r16]
__device__ __forceinline__ u32x4 row_frag128(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const u32x4*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}
// A operand X^T[d = 16n + r16][rows 32*blk + {4q..4q+3, 16+4q..16+4q+3}] of a [64 rows][64 d] tile (chunk swizzle row & 7)
__device__ __forceinline__ u32x4 tr_frag128(const char* tile, int blk, int n, int r16, int q) {
    u32x4 f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = 32 * blk + 16 * hh + 4 * q + (r16 >> 2);
        const int ch = 2 * n + ((r16 & 3) >> 1);
        const char* a = tile + row * 128 + ((ch ^ (row & 7)) << 4) + ((r16 & 1) << 3);
        const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a);
        const u32x2 w = __builtin_bit_cast(u32x2, v);
        f[2 * hh] = w[0]; f[2 * hh + 1] = w[1];
    }
    return f;
}
// value of lane (lane & ~3) | r of every 4-lane group (DPP quad_perm broadcast: one VALU move, no LDS traffic)
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int r) {
    switch (r) {
        case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true);
        case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true);
        case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xf, 0xf, true);
        default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xf, 0xf, true);
    }
}
// Two [64 rows][64 d] bf16 tiles (A at st, B at st + 8 KiB) staged by LDS-DMA: 16 bytes per lane, 8 tile rows per wave-instruction,
// lane-linear in LDS, so the XOR swizzle (chunk c of row r at position c ^ (r & 7)) is applied on the source side.  Rows past the
// tensor are clamped to its last row (finite data): whoever consumes the tile masks their contribution, nothing is zeroed here.
// No staging registers, no ds_write pass; the caller waits (vmcnt) before the barrier that publishes the tile.
__device__ __forceinline__ void stage2_dma(const bf16* A, long a_st, const bf16* B, long b_st, int r0, int nrows, char* st, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r8 = 8 * (wave + 4 * i), row = r8 + (lane >> 3), pos = lane & 7;
        // 32-bit byte offsets from the (uniform) bases, one v_mad_u32_u24 each: row strides below 16 MiB, tensors below 4 GiB (launcher)
        const uint32_t r = (uint32_t)min(r0 + row, nrows - 1);
        const uint32_t ch = (uint32_t)((pos ^ (row & 7)) << 4);
        char* dst = st + __builtin_amdgcn_readfirstlane(r8 * 128);
        const char* aa = reinterpret_cast<const char*>(A) + (__umul24(r, (uint32_t)a_st * 2u) + ch);
        const char* ba = reinterpret_cast<const char*>(B) + (__umul24(r, (uint32_t)b_st * 2u) + ch);
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)aa, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)ba, (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
    }
}
#define S2T_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)               /* vmcnt(0); lgkmcnt / expcnt untouched */

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv2_kernel(AttnArgs p) {
    constexpr int DH = 64;
    ASTAMP_EARLY(2)
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x (Q 8 KiB | dO 8 KiB)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
    int head;
    const int bx = head_xcd_remap(head, p.B * p.H, gridDim.x);
    const int b = head / p.H, h = head % p.H, kw = bx * 128 + wave * 32;
    const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const bf16* Qg = reinterpret_cast<const bf16*>(p.Q) + (long)b * p.q_sb + (long)h * DH;
    const bf16* Kg = reinterpret_cast<const bf16*>(p.K) + (long)b * p.k_sb + (long)h * DH;
    const bf16* Vg = reinterpret_cast<const bf16*>(p.V) + (long)b * p.v_sb + (long)h * DH;
    const bf16* dOg = reinterpret_cast<const bf16*>(p.dO) + (long)b * p.do_sb + (long)h * DH;
    const float* lse = p.LSE + ((long)b * p.H + h) * p.Tq;
    const float* dlt = p.Delta + ((long)b * p.H + h) * p.Tq;

    // the workgroup's own 128 keys: K and V rows parked in LDS for the whole kernel (B operands of S and dP); holding the
    // 8 fragments per lane in registers instead pushed the kernel past 256 VGPRs
    char* sKown = smem + 32768 + 1024;                              // [128 keys][64 d] K, then V
    char* sVown = sKown + 16384;
    f32x4 dkT[2][4], dvT[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int n = 0; n < 4; ++n) { dkT[kb][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvT[kb][n] = dkT[kb][n]; }
    const uint32_t drop_th16 = (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float drop_inv = 1.f / (1.f - p.p_drop);
    const uint32_t drop_ks = drop_seed_key(p.seed), drop_hwm = drop_high_mix(p.seed, 0);   // < 2^34 elements (launcher): quad index in one word
    const uint32_t tkq = (uint32_t)((p.Tk + 3) >> 2);                                        // quads per (padded) row of the index space
    const uint32_t drop_thm1x2 = (drop_th16 - 1u) * 0x00010001u;
    const uint32_t own_bit = ((r16 & 1) ? 0x00010000u : 1u) << ((r16 >> 1) & 1);            // key r16 & 3 of a quad: bits 0, 16, 1, 17

    const int qstart = p.causal ? bx * 128 : 0;                 // queries before the first key of the workgroup see nothing
    const int ntile = qstart < p.Tq ? (p.Tq - qstart + 63) / 64 : 0;
    // per-query statistics of the tile (LSE, Delta) travel with it: 4-byte LDS-DMA by waves 0 / 1
    float* sStat = reinterpret_cast<float*>(smem + 32768);          // [2 stages][2][64]
    auto stage = [&](int qt, int stg) {
        stage2_dma(Qg, p.q_st, dOg, p.do_st, qt, p.Tq, smem + stg * 16384, wave, lane);
        if (wave < 2) {
            // rows past Tq (the last tile): LSE = +inf, so their probabilities -- and with them dS -- are exact zeros without a mask
            // per element (their Q / dO rows repeat the last query: finite); the DMA writes nothing for the lanes that are off
            float* dst = sStat + stg * 128 + wave * 64;
            if (wave == 1 || qt + lane < p.Tq) {
                // a wave-uniform base + a 32-bit lane offset: a 64-bit per-lane pointer here was spilled, and its reload -- issued right
                // behind the tile's DMAs -- made the compiler's wait a vmcnt(0): the whole prefetch latency, exposed, every tile
                const char* sb = reinterpret_cast<const char*>(wave == 0 ? lse : dlt);
                const char* src = sb + (uint32_t)(min(qt + lane, p.Tq - 1) * 4);
                __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
            } else dst[lane] = INFINITY;
        }
    };
    if (ntile > 0) stage(qstart, 0);                  // first: the DMA overlaps the loads of the own keys below (one round trip, not two)
    {   // all eight loads of a thread go out together (keys past klen read the last row of the tensor and are zeroed by a select)
        u32x4 kk[4], vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cid = threadIdx.x + 256 * i, row = cid >> 3, c = cid & 7;
            const long key = min(bx * 128 + row, p.Tk - 1);
            kk[i] = *reinterpret_cast<const u32x4*>(Kg + key * p.k_st + c * 8);
            vv[i] = *reinterpret_cast<const u32x4*>(Vg + key * p.v_st + c * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cid = threadIdx.x + 256 * i, row = cid >> 3, c = cid & 7;
            const bool in = bx * 128 + row < klen;
#pragma unroll
            for (int w = 0; w < 4; ++w) { kk[i][w] = in ? kk[i][w] : 0u; vv[i][w] = in ? vv[i][w] : 0u; }
            *reinterpret_cast<u32x4*>(sKown + row * 128 + ((c ^ (row & 7)) << 4)) = kk[i];
            *reinterpret_cast<u32x4*>(sVown + row * 128 + ((c ^ (row & 7)) << 4)) = vv[i];
        }
    }
    S2T_WAIT_VM0();
    __syncthreads();
    const float sc2 = p.scale * 1.44269504088896f;
    const bool generic = p.causal || p.dist_pen;
    ASTAMP(7, 1)
    for (int t = 0; t < ntile; ++t) {
        const int qt = qstart + t * 64;
        const char* sQ = smem + (t & 1) * 16384;
        const char* sDO = sQ + 8192;
        const float* sL = sStat + (t & 1) * 128;
        ASTAMP(t, 0)
        if (t + 1 < ntile) stage(qt + 64, (t + 1) & 1);
        ASTAMP(t, 1)
        // EDGE: masks evaluated per element (query rows past Tq -- their tile rows repeat the last query --, the causal triangle,
        // the distance penalty); DROP: dropout on.  Common factors leave the inner loop: dS carries neither the softmax scale (dK is
        // scaled once at the end) nor, like P, the dropout's 1/(1-p) (it multiplies dP inside one fma, and dV at the end).
        auto tile = [&](auto edge_tag, auto drop_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value, DROP = decltype(drop_tag)::value;
#pragma unroll 1
        for (int ib = 0; ib < 2; ++ib) {                  // 32 queries at a time: their P / dS operands are consumed at once
        u32x4 pf[2], sf[2];                             // [key block]: D*P and dS as B operands
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * ib + ii;
            u32x4 qa[2], da[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) { qa[g] = row_frag128(sQ, 16 * i + r16, 4 * g + q); da[g] = row_frag128(sDO, 16 * i + r16, 4 * g + q); }
            const f32x4 L = *reinterpret_cast<const f32x4*>(sL + 16 * i + 4 * q) * 1.44269504088896f;
            const f32x4 Dl = *reinterpret_cast<const f32x4*>(sL + 64 + 16 * i + 4 * q);
            // dropout: lane r16 hashes (query 4q + (r16 & 3), key quad r16 >> 2) of both key blocks
            const uint32_t qrow_quads = DROP ? (uint32_t)(((b * p.H + h) * p.Tq + qt + 16 * i + 4 * q + (r16 & 3))) * tkq : 0u;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int krow = wave * 32 + 16 * kb + r16;
                    st = mma16<bf16>(qa[g], row_frag128(sKown, krow, 4 * g + q), st);
                    dp = mma16<bf16>(da[g], row_frag128(sVown, krow, 4 * g + q), dp);
                }
                const int key = kw + 16 * kb + r16;
                // dropout: this lane hashes the quad (query 4q + (r16 & 3), keys 4 (r16 >> 2) ..+3) and turns its four fields into keep
                // BITS at once (field >= threshold as a saturating packed subtract of threshold - 1 and a packed min with 1: keys 0 / 1
                // -> bits 0 / 16 of one word, keys 2 / 3 of the other, merged as bits 0, 16, 1, 17).  The element (query 4q + r, own
                // key) then costs one DPP quad broadcast fused into an AND with the own key's bit, one compare and the two selects --
                // broadcasting both hash words and extracting a lane-dependent 16-bit field per element took 2.4x as many instructions.
                uint32_t nib = 0u;
                if constexpr (DROP) {
                    const u32x2 hq = drop_hash4_lo(drop_ks, drop_hwm, qrow_quads + ((uint32_t)key >> 2));
                    uint32_t dy, dx, ky, kx;
                    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dy) : "v"(hq[1]), "v"(drop_thm1x2));
                    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dx) : "v"(hq[0]), "v"(drop_thm1x2));
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(ky) : "v"(dy), "v"(0x00010001u));
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(kx) : "v"(dx), "v"(0x00010001u));
                    nib = drop_th16 > 0 ? (ky | (kx << 1)) : 0x00030003u;    // a rate below 2^-16 keeps every element, as in every other kernel
                }
                float pv[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qrow = qt + 16 * i + 4 * q + r;
                    float arg = __builtin_fmaf(st[r], sc2, -L[r]);
                    if constexpr (EDGE) { if (p.dist_pen) arg -= dist_pen_log2(qrow, key); }
                    float e = __builtin_amdgcn_exp2f(arg);
                    if constexpr (EDGE) e = (qrow < p.Tq && (!p.causal || key <= qrow)) ? e : 0.f;     // keys past klen: discarded at the store
                    if constexpr (DROP) {
                        // the bits of (query 4q + r, keys of the own quad) sit in lane (r16 & ~3) | r of the same 4-lane group
                        const bool keep = (quad_bcast(nib, r) & own_bit) != 0u;
                        pv[r] = keep ? e : 0.f;
                        ds[r] = e * __builtin_fmaf(keep ? dp[r] : 0.f, drop_inv, -Dl[r]);
                    } else {
                        pv[r] = e;
                        ds[r] = e * (dp[r] - Dl[r]);
                    }
                }
                pf[kb][2 * ii] = pack_bf16(pv[0], pv[1]); pf[kb][2 * ii + 1] = pack_bf16(pv[2], pv[3]);
                sf[kb][2 * ii] = pack_bf16(ds[0], ds[1]); sf[kb][2 * ii + 1] = pack_bf16(ds[2], ds[3]);
            }
        }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const u32x4 dot = tr_frag128(sDO, ib, n, r16, q), qtf = tr_frag128(sQ, ib, n, r16, q);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    dvT[kb][n] = mma16<bf16>(dot, pf[kb], dvT[kb][n]);
                    dkT[kb][n] = mma16<bf16>(qtf, sf[kb], dkT[kb][n]);
                }
            }
        }
        };
        const bool edge = generic;                       // a plain softmax has no edge: the query tail is handled by LSE = +inf
        if (p.p_drop > 0.f) { if (edge) tile(std::true_type{}, std::true_type{}); else tile(std::false_type{}, std::true_type{}); }
        else if (edge) tile(std::true_type{}, std::false_type{});
        else tile(std::false_type{}, std::false_type{});
        ASTAMP(t, 3)
        S2T_WAIT_VM0();                                   // tile t + 1 has landed (this wave's share; the barrier covers the rest)
        ASTAMP(t, 4)
        __syncthreads();
        ASTAMP(t, 5)
    }
    ASTAMP(7, 2)
    const float dv_scale = p.p_drop > 0.f ? drop_inv : 1.f;
    bf16* dKg = reinterpret_cast<bf16*>(p.dK) + (long)b * p.dk_sb + (long)h * DH;
    bf16* dVg = reinterpret_cast<bf16*>(p.dV) + (long)b * p.dv_sb + (long)h * DH;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kw + 16 * kb + r16;
        if (key >= p.Tk) continue;
        const float kz = key < klen ? 1.f : 0.f;          // padded keys get exact zeros
#pragma unroll
        for (int n = 0; n < 4; ++n) { dkT[kb][n] *= kz * p.scale; dvT[kb][n] *= kz * dv_scale; }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            u32x2 w;
            w[0] = pack_bf16(dkT[kb][n][0], dkT[kb][n][1]); w[1] = pack_bf16(dkT[kb][n][2], dkT[kb][n][3]);
            *reinterpret_cast<u32x2*>(dKg + (long)key * p.dk_st + 16 * n + 4 * q) = w;
            w[0] = pack_bf16(dvT[kb][n][0], dvT[kb][n][1]); w[1] = pack_bf16(dvT[kb][n][2], dvT[kb][n][3]);
            *reinterpret_cast<u32x2*>(dVg + (long)key * p.dv_st + 16 * n + 4 * q) = w;
        }
    }
    ASTAMP(7, 3)
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(AttnArgs p) {
    constexpr int DH = 64;
    ASTAMP_EARLY(1)
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x (K 8 KiB | V 8 KiB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    int head;
    const int bx = head_xcd_remap(head, p.B * p.H, gridDim.x);
    const int b = head / p.H, h = head % p.H, qw = bx * 128 + wave * 32;
    const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const bf16* Qg = reinterpret_cast<const bf16*>(p.Q) + (long)b * p.q_sb + (long)h * DH;
    const bf16* Kg = reinterpret_cast<const bf16*>(p.K) + (long)b * p.k_sb + (long)h * DH;
    const bf16* Vg = reinterpret_cast<const bf16*>(p.V) + (long)b * p.v_sb + (long)h * DH;
    const bf16* dOg = reinterpret_cast<const bf16*>(p.dO) + (long)b * p.do_sb + (long)h * DH;
    const float* lse = p.LSE + ((long)b * p.H + h) * p.Tq;
    const float* dlt = p.Delta + ((long)b * p.H + h) * p.Tq;

    int kv_end = klen;
    if (p.causal) kv_end = min(kv_end, bx * 128 + 128);
    const int ntile = (kv_end + 63) / 64;
    if (ntile > 0) stage2_dma(Kg, p.k_st, Vg, p.v_st, 0, p.Tk, smem, wave, lane);      // first: overlaps the operand loads below
    u32x4 qf[2][2], dof[2][2];
    float L[2], Dl[2];
    const bf16* Og = reinterpret_cast<const bf16*>(p.O) + (long)b * p.o_sb + (long)h * DH;
    // All thirteen loads of a lane go out back to back (rows past Tq read the last row and are zeroed by a select): inside
    // `if (row < Tq)` blocks that also consumed them, each block waited out its own round trip to memory -- four in a row, ~10,000
    // cycles of a workgroup's ~42,000.
    u32x4 of[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const long row = min(qw + 16 * qb + r16, p.Tq - 1);
        L[qb] = lse[row] * 1.44269504088896f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            qf[qb][g] = *reinterpret_cast<const u32x4*>(Qg + row * p.q_st + (4 * g + q) * 8);
            dof[qb][g] = *reinterpret_cast<const u32x4*>(dOg + row * p.do_st + (4 * g + q) * 8);
            of[qb][g] = *reinterpret_cast<const u32x4*>(Og + row * p.o_st + (4 * g + q) * 8);
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int row = qw + 16 * qb + r16;
        const bool in = row < p.Tq;
        float part = 0.f;                                   // Delta = rowsum(dO * O): this kernel owns the query, so it makes it
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int w = 0; w < 4; ++w) { qf[qb][g][w] = in ? qf[qb][g][w] : 0u; dof[qb][g][w] = in ? dof[qb][g][w] : 0u; }
            const bf16* oe = reinterpret_cast<const bf16*>(&of[qb][g]);
            const bf16* de = reinterpret_cast<const bf16*>(&dof[qb][g]);
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)oe[e] * (float)de[e];
        }
        L[qb] = in ? L[qb] : 0.f;
        part += __shfl_xor(part, 16); part += __shfl_xor(part, 32);
        Dl[qb] = part;
        if (q == 0 && in) const_cast<float*>(dlt)[row] = part;                 // the dK/dV kernel, launched after this one, reads it
    }
    f32x4 dqT[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int n = 0; n < 4; ++n) dqT[qb][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t drop_th16 = (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float drop_inv = 1.f / (1.f - p.p_drop);
    const uint32_t drop_ks = drop_seed_key(p.seed), drop_hwm = drop_high_mix(p.seed, 0);   // < 2^34 elements (launcher): quad index in one word
    const uint32_t tkq = (uint32_t)((p.Tk + 3) >> 2);

    const float sc2 = p.scale * 1.44269504088896f;
    const bool generic = p.causal || p.dist_pen;
    // Q / dO fragments must have landed before the loop: left pending, the compiler's wait for them sits behind the loop's own
    // prefetch in the in-order counter and becomes a vmcnt(0), i.e. the whole K/V prefetch latency, exposed, every iteration
    S2T_WAIT_VM0();                                   // (the first K/V tile went out ahead of those loads: one round trip, not two)
    __syncthreads();
    ASTAMP(7, 1)
    for (int t = 0; t < ntile; ++t) {
        const int kv0 = t * 64;
        const char* sK = smem + (t & 1) * 16384;
        const char* sV = sK + 8192;
        ASTAMP(t, 0)
        if (t + 1 < ntile) stage2_dma(Kg, p.k_st, Vg, p.v_st, kv0 + 64, p.Tk, smem + ((t + 1) & 1) * 16384, wave, lane);
        ASTAMP(t, 1)
        u32x4 sf[2][2];                                 // [query block][32-key block]: dS^T as B operand
        // EDGE: masks per element (keys past klen -- their tile rows hold whatever lies there --, the causal triangle, the distance
        // penalty).  dS carries neither the softmax scale (dQ is scaled once at the end) nor a separate 1/(1-p) multiply.
        // TAIL: the last key tile of a plain softmax (klen not a multiple of 64) -- the interior arithmetic plus one compare + select
        // per element for the keys past klen (the generic EDGE path takes twice an interior tile's time)
        auto ds_tile = [&](auto edge_tag, auto drop_tag, auto tail_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value, DROP = decltype(drop_tag)::value, TAIL = decltype(tail_tag)::value;
        const int tail_thr = klen - kv0 - 4 * q;                           // key 16 j + 4 q + r is valid <=> 16 j + r < tail_thr
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 ka[2], va[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) { ka[g] = row_frag128(sK, 16 * j + r16, 4 * g + q); va[g] = row_frag128(sV, 16 * j + r16, 4 * g + q); }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2; ++g) { st = mma16<bf16>(ka[g], qf[qb][g], st); dp = mma16<bf16>(va[g], dof[qb][g], dp); }
                const int qrow = qw + 16 * qb + r16;
                const int key0 = kv0 + 16 * j + 4 * q;
                u32x2 hq = {0u, 0u};
                if constexpr (DROP) hq = drop_hash4_lo(drop_ks, drop_hwm, (uint32_t)((b * p.H + h) * p.Tq + qrow) * tkq + ((uint32_t)key0 >> 2));   // keys 4q .. 4q+3 = one quad
                float ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // query rows past Tq are never stored; padded / future keys must not reach dQ
                    float arg = __builtin_fmaf(st[r], sc2, -L[qb]);
                    if constexpr (EDGE) { if (p.dist_pen) arg -= dist_pen_log2(qrow, key0 + r); }
                    float e = __builtin_amdgcn_exp2f(arg);
                    if constexpr (EDGE) {
                        const int key = key0 + r;
                        e = (key < klen && (!p.causal || key <= qrow)) ? e : 0.f;
                    }
                    if constexpr (TAIL) e = (16 * j + r < tail_thr) ? e : 0.f;
                    if constexpr (DROP) ds[r] = e * __builtin_fmaf(drop_field(hq, r) >= drop_th16 ? dp[r] : 0.f, drop_inv, -Dl[qb]);
                    else ds[r] = e * (dp[r] - Dl[qb]);
                }
                sf[qb][j >> 1][2 * (j & 1)] = pack_bf16(ds[0], ds[1]); sf[qb][j >> 1][2 * (j & 1) + 1] = pack_bf16(ds[2], ds[3]);
            }
        }
        };
        const bool tail = kv0 + 64 > klen;
        if (generic) { if (p.p_drop > 0.f) ds_tile(std::true_type{}, std::true_type{}, std::false_type{}); else ds_tile(std::true_type{}, std::false_type{}, std::false_type{}); }
        else if (p.p_drop > 0.f) { if (tail) ds_tile(std::false_type{}, std::true_type{}, std::true_type{}); else ds_tile(std::false_type{}, std::true_type{}, std::false_type{}); }
        else if (tail) ds_tile(std::false_type{}, std::false_type{}, std::true_type{});
        else ds_tile(std::false_type{}, std::false_type{}, std::false_type{});
        ASTAMP(t, 2)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const u32x4 kt = tr_frag128(sK, jb, n, r16, q);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) dqT[qb][n] = mma16<bf16>(kt, sf[qb][jb], dqT[qb][n]);
            }
        ASTAMP(t, 3)
        S2T_WAIT_VM0();                                   // tile t + 1 has landed (this wave's share; the barrier covers the rest)
        ASTAMP(t, 4)
        __syncthreads();
        ASTAMP(t, 5)
    }
    ASTAMP(7, 2)
    bf16* dQg = reinterpret_cast<bf16*>(p.dQ) + (long)b * p.dq_sb + (long)h * DH;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = qw + 16 * qb + r16;
        if (qrow >= p.Tq) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            u32x2 w;
            w[0] = pack_bf16(dqT[qb][n][0] * p.scale, dqT[qb][n][1] * p.scale); w[1] = pack_bf16(dqT[qb][n][2] * p.scale, dqT[qb][n][3] * p.scale);
            *reinterpret_cast<u32x2*>(dQg + (long)qrow * p.dq_st + 16 * n + 4 * q) = w;
        }
    }
    ASTAMP(7, 3)
}

// ------------------------------------------------------------------------------------ backward, one kernel (short key sequences)
// The two second-generation kernels each rebuild S, dP, the softmax and the dropout mask of every (query, key) pair -- and that
// element-wise work, not the MFMAs, is what they are bound by (PMC, round 3: VALU : MFMA = 498 : 32 per wave-tile).  When ALL keys of a
// head fit in one workgroup's LDS (Tk <= 384: the encoder's 1,500 frames / 4 = 375) one workgroup can own the head: eight waves x 48
// keys, K and V parked in LDS for the whole launch (96 KiB), 32-query tiles of Q and dO streamed by LDS-DMA.  Per tile every wave
// makes P and dS for its 48 keys ONCE (the dK/dV kernel's arithmetic, instruction for instruction, so the dropout mask is the
// forward's), feeds dV^T += dO^T P and dK^T += Q^T dS from the accumulators as before, and also writes dS (bf16) to a
// [32 queries][384 keys] LDS tile; after a barrier the eight waves split that tile's dQ^T = K^T dS^T by (query block, d block) and run
// the contraction over all 384 keys from LDS (K^T through the transposed reads that also serve dK/dV): complete per tile, so no
// atomics and no cross-workgroup sums.  Delta comes from attn_delta_kernel.  Plain softmax only (no causal mask, no distance penalty).
constexpr int FB_KEYS = 384, FB_DS_ROWB = 800;          // dS tile row: 768 B of keys + 32 B (a query's 8-byte reads fall on distinct banks)
constexpr int FB_STAGE = 8192, FB_STAT = 2 * FB_STAGE, FB_K = FB_STAT + 1024, FB_V = FB_K + FB_KEYS * 128, FB_DS = FB_V + FB_KEYS * 128,
              FB_LDS = FB_DS + 32 * FB_DS_ROWB;
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(AttnArgs p) {
    constexpr int DH = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
    const int head = blockIdx.x, b = head / p.H, h = head % p.H, kw = wave * 48;
    const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    const bf16* Qg = reinterpret_cast<const bf16*>(p.Q) + (long)b * p.q_sb + (long)h * DH;
    const bf16* Kg = reinterpret_cast<const bf16*>(p.K) + (long)b * p.k_sb + (long)h * DH;
    const bf16* Vg = reinterpret_cast<const bf16*>(p.V) + (long)b * p.v_sb + (long)h * DH;
    const bf16* dOg = reinterpret_cast<const bf16*>(p.dO) + (long)b * p.do_sb + (long)h * DH;
    const float* lse = p.LSE + ((long)b * p.H + h) * p.Tq;
    const float* dlt = p.Delta + ((long)b * p.H + h) * p.Tq;
    char* sKown = smem + FB_K;                                        // [384 keys][64 d], chunk swizzle row & 7
    char* sVown = smem + FB_V;
    char* sDS = smem + FB_DS;                                         // [32 queries][FB_DS_ROWB]: dS of the tile, keys contiguous
    float* sStat = reinterpret_cast<float*>(smem + FB_STAT);          // [2 stages][2][32]: LSE, Delta of the tile's queries

    f32x4 dkT[3][4], dvT[3][4];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb)
#pragma unroll
        for (int n = 0; n < 4; ++n) { dkT[kb][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvT[kb][n] = dkT[kb][n]; }
    const uint32_t drop_th16 = (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float drop_inv = 1.f / (1.f - p.p_drop);
    const uint32_t drop_ks = drop_seed_key(p.seed), drop_hwm = drop_high_mix(p.seed, 0);
    const uint32_t tkq = (uint32_t)((p.Tk + 3) >> 2);
    const uint32_t drop_thm1x2 = (drop_th16 - 1u) * 0x00010001u;
    const uint32_t own_bit = ((r16 & 1) ? 0x00010000u : 1u) << ((r16 >> 1) & 1);

    const int ntile = (p.Tq + 31) / 32;
    // one tile = 32 rows of Q (waves 0-3, 8 rows each) and of dO (waves 4-7) by LDS-DMA, LSE / Delta by waves 0 / 1 (lanes < 32)
    auto stage = [&](int qt, int stg) {
        const int r8 = 8 * (wave & 3), row = r8 + (lane >> 3), pos = lane & 7;
        const uint32_t r = (uint32_t)min(qt + row, p.Tq - 1);
        const uint32_t ch = (uint32_t)((pos ^ (row & 7)) << 4);
        char* dst = smem + stg * FB_STAGE + __builtin_amdgcn_readfirstlane((wave < 4 ? 0 : 4096) + r8 * 128);
        const char* src = wave < 4 ? reinterpret_cast<const char*>(Qg) + (__umul24(r, (uint32_t)p.q_st * 2u) + ch)
                                   : reinterpret_cast<const char*>(dOg) + (__umul24(r, (uint32_t)p.do_st * 2u) + ch);
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        if (wave < 2 && lane < 32) {
            // rows past Tq: LSE = +inf, so their probabilities -- and with them dS -- are exact zeros without a mask per element
            float* sd = sStat + stg * 64 + wave * 32;
            if (wave == 1 || qt + lane < p.Tq) {
                const char* sb = reinterpret_cast<const char*>(wave == 0 ? lse : dlt);
                const char* ss = sb + (uint32_t)(min(qt + lane, p.Tq - 1) * 4);
                __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)ss, (__attribute__((address_space(3))) void*)sd, 4, 0, 0);
            } else sd[lane] = INFINITY;
        }
    };
    if (ntile > 0) stage(0, 0);
    {   // the head's keys and values: twelve 16-byte loads per thread, all out together; rows past klen are zeros (their dK / dV are
        // never stored and they add nothing to dQ)
        u32x4 kk[6], vv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int cid = threadIdx.x + 512 * i, row = cid >> 3, c = cid & 7;
            const long key = min(row, p.Tk - 1);
            kk[i] = *reinterpret_cast<const u32x4*>(Kg + key * p.k_st + c * 8);
            vv[i] = *reinterpret_cast<const u32x4*>(Vg + key * p.v_st + c * 8);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int cid = threadIdx.x + 512 * i, row = cid >> 3, c = cid & 7;
            const bool in = row < klen;
#pragma unroll
            for (int w = 0; w < 4; ++w) { kk[i][w] = in ? kk[i][w] : 0u; vv[i][w] = in ? vv[i][w] : 0u; }
            *reinterpret_cast<u32x4*>(sKown + row * 128 + ((c ^ (row & 7)) << 4)) = kk[i];
            *reinterpret_cast<u32x4*>(sVown + row * 128 + ((c ^ (row & 7)) << 4)) = vv[i];
        }
    }
    S2T_WAIT_VM0();
    __syncthreads();
    const float sc2 = p.scale * 1.44269504088896f;
    const int nblk = (min(klen, FB_KEYS) + 31) / 32;                  // 32-key blocks that hold a valid key
    const int dq_j = wave & 1, dq_n = wave >> 1;                      // this wave's share of a tile's dQ^T: queries 16 j .., d 16 n ..
    bf16* dQg = reinterpret_cast<bf16*>(p.dQ) + (long)b * p.dq_sb + (long)h * DH;
    for (int t = 0; t < ntile; ++t) {
        const int qt = t * 32;
        const char* sQ = smem + (t & 1) * FB_STAGE;
        const char* sDO = sQ + 4096;
        const float* sL = sStat + (t & 1) * 64;
        if (t + 1 < ntile) stage(qt + 32, (t + 1) & 1);
        auto tile = [&](auto drop_tag) {
            constexpr bool DROP = decltype(drop_tag)::value;
            u32x4 pf[3], sf[3];                             // [key block]: D*P and dS as B operands (k = the tile's 32 queries)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                u32x4 qa[2], da[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) { qa[g] = row_frag128(sQ, 16 * ii + r16, 4 * g + q); da[g] = row_frag128(sDO, 16 * ii + r16, 4 * g + q); }
                const f32x4 L = *reinterpret_cast<const f32x4*>(sL + 16 * ii + 4 * q) * 1.44269504088896f;
                const f32x4 Dl = *reinterpret_cast<const f32x4*>(sL + 32 + 16 * ii + 4 * q);
                const uint32_t qrow_quads = DROP ? (uint32_t)(((b * p.H + h) * p.Tq + qt + 16 * ii + 4 * q + (r16 & 3))) * tkq : 0u;
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    const int krow = kw + 16 * kb + r16;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        st = mma16<bf16>(qa[g], row_frag128(sKown, krow, 4 * g + q), st);
                        dp = mma16<bf16>(da[g], row_frag128(sVown, krow, 4 * g + q), dp);
                    }
                    uint32_t nib = 0u;
                    if constexpr (DROP) {                     // as in attn_bwd_dkv2_kernel: keep bits of the own quad, one hash per lane
                        const u32x2 hq = drop_hash4_lo(drop_ks, drop_hwm, qrow_quads + ((uint32_t)krow >> 2));
                        uint32_t dy, dx, ky, kx;
                        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dy) : "v"(hq[1]), "v"(drop_thm1x2));
                        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dx) : "v"(hq[0]), "v"(drop_thm1x2));
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(ky) : "v"(dy), "v"(0x00010001u));
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(kx) : "v"(dx), "v"(0x00010001u));
                        nib = drop_th16 > 0 ? (ky | (kx << 1)) : 0x00030003u;
                    }
                    float pv[4], ds[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], sc2, -L[r]));
                        if constexpr (DROP) {
                            const bool keep = (quad_bcast(nib, r) & own_bit) != 0u;
                            pv[r] = keep ? e : 0.f;
                            ds[r] = e * __builtin_fmaf(keep ? dp[r] : 0.f, drop_inv, -Dl[r]);
                        } else {
                            pv[r] = e;
                            ds[r] = e * (dp[r] - Dl[r]);
                        }
                    }
                    pf[kb][2 * ii] = pack_bf16(pv[0], pv[1]); pf[kb][2 * ii + 1] = pack_bf16(pv[2], pv[3]);
                    sf[kb][2 * ii] = pack_bf16(ds[0], ds[1]); sf[kb][2 * ii + 1] = pack_bf16(ds[2], ds[3]);
                    // the same dS, keys contiguous, for the dQ product: element (query 16 ii + 4 q + r, key krow)
                    const uint32_t w01 = sf[kb][2 * ii], w23 = sf[kb][2 * ii + 1];
                    char* drow = sDS + (16 * ii + 4 * q) * FB_DS_ROWB + krow * 2;
                    *reinterpret_cast<uint16_t*>(drow) = (uint16_t)w01;
                    *reinterpret_cast<uint16_t*>(drow + FB_DS_ROWB) = (uint16_t)(w01 >> 16);
                    *reinterpret_cast<uint16_t*>(drow + 2 * FB_DS_ROWB) = (uint16_t)w23;
                    *reinterpret_cast<uint16_t*>(drow + 3 * FB_DS_ROWB) = (uint16_t)(w23 >> 16);
                }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const u32x4 dot = tr_frag128(sDO, 0, n, r16, q), qtf = tr_frag128(sQ, 0, n, r16, q);
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    dvT[kb][n] = mma16<bf16>(dot, pf[kb], dvT[kb][n]);
                    dkT[kb][n] = mma16<bf16>(qtf, sf[kb], dkT[kb][n]);
                }
            }
        };
        if (p.p_drop > 0.f) tile(std::true_type{}); else tile(std::false_type{});
        __syncthreads();                                  // every wave's dS of this tile is in LDS
        {   // dQ^T[d 16 n + 4 q + r][query 16 j + r16] = sum over all keys K^T dS^T: A = K^T by transposed reads, B = the query's keys
            // (two accumulators over alternating key blocks, or the fragments of four blocks requested together, spill: the
            // element-wise phase above sits at the 256-register edge of two waves per SIMD)
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
            const char* brow = sDS + (16 * dq_j + r16) * FB_DS_ROWB + 8 * q;
            for (int blk = 0; blk < nblk; ++blk) {
                const u32x4 kt = tr_frag128(sKown, blk, dq_n, r16, q);
                const u32x2 lo = *reinterpret_cast<const u32x2*>(brow + 64 * blk), hi = *reinterpret_cast<const u32x2*>(brow + 64 * blk + 32);
                a4 = mma16<bf16>(kt, (u32x4){lo[0], lo[1], hi[0], hi[1]}, a4);
            }
            const int qrow = qt + 16 * dq_j + r16;
            if (qrow < p.Tq) {
                u32x2 w;
                w[0] = pack_bf16(a4[0] * p.scale, a4[1] * p.scale); w[1] = pack_bf16(a4[2] * p.scale, a4[3] * p.scale);
                *reinterpret_cast<u32x2*>(dQg + (long)qrow * p.dq_st + 16 * dq_n + 4 * q) = w;
            }
        }
        S2T_WAIT_VM0();                                   // tile t + 1 has landed (this wave's share; the barrier covers the rest)
        __syncthreads();                                  // and nobody still reads this tile's dS
    }
    const float dv_scale = p.p_drop > 0.f ? drop_inv : 1.f;
    bf16* dKg = reinterpret_cast<bf16*>(p.dK) + (long)b * p.dk_sb + (long)h * DH;
    bf16* dVg = reinterpret_cast<bf16*>(p.dV) + (long)b * p.dv_sb + (long)h * DH;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
        const int key = kw + 16 * kb + r16;
        if (key >= p.Tk) continue;
        const float kz = key < klen ? 1.f : 0.f;          // padded keys get exact zeros
#pragma unroll
        for (int n = 0; n < 4; ++n) { dkT[kb][n] *= kz * p.scale; dvT[kb][n] *= kz * dv_scale; }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            u32x2 w;
            w[0] = pack_bf16(dkT[kb][n][0], dkT[kb][n][1]); w[1] = pack_bf16(dkT[kb][n][2], dkT[kb][n][3]);
            *reinterpret_cast<u32x2*>(dKg + (long)key * p.dk_st + 16 * n + 4 * q) = w;
            w[0] = pack_bf16(dvT[kb][n][0], dvT[kb][n][1]); w[1] = pack_bf16(dvT[kb][n][2], dvT[kb][n][3]);
            *reinterpret_cast<u32x2*>(dVg + (long)key * p.dv_st + 16 * n + 4 * q) = w;
        }
    }
}

// ------------------------------------------------------------------------------------ C ABI
// the second-generation kernels address the rows of a staged tensor by 32-bit byte offsets built with a 24-bit multiply
static bool span32(long stride, int rows) {
    return stride >= 0 && stride * 2 < (1l << 24) && (long)rows * stride * 2 + 256 < (1l << 32);
}
static bool strides_ok(int dtype, const long* s, int n) {
    const int e = dtype == S2T_BF16 ? 8 : 4;
    for (int i = 0; i < n; ++i) if (s[i] % e) return false;
    return true;
}

template <typename T, int DH> static int fwd_launch(const AttnArgs& a, hipStream_t st) {
    typedef ACfg<T, DH> C;
    dim3 grid((a.Tq + 63) / 64, a.H, a.B);
    const size_t lds = 2 * C::TILE + 4 * C::PTILE;
    hipLaunchKernelGGL((attn_fwd_kernel<T, DH>), grid, dim3(256), lds, st, a);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
template <typename T, int DH> static int bwd_launch(const AttnArgs& a, hipStream_t st) {
    typedef ACfg<T, DH> C;
    const long rows = (long)a.B * a.H * a.Tq;
    bool dkv2 = false, dq2 = false;
    if constexpr (sizeof(T) == 2 && DH == 64) {
        const bool v1 = g_s2t_opt_attn_v1 != 0;                            // s2t_set_option("attn_v1")
        const bool al = !(a.dk_st % 4) && !(a.dk_sb % 4) && !(a.dv_st % 4) && !(a.dv_sb % 4) && !(a.dq_st % 4) && !(a.dq_sb % 4) &&
                        !((uintptr_t)a.dK & 7) && !((uintptr_t)a.dV & 7) && !((uintptr_t)a.dQ & 7);
        // the second-generation kernels keep the dropout quad index in one 32-bit word
        const bool idx32 = (unsigned long long)a.B * a.H * a.Tq * (unsigned long long)((a.Tk + 3) & ~3) < (1ull << 34);
        // one kernel for the whole backward when a head's keys fit in one workgroup (plain softmax, Tk <= 384: the encoder's self-attention)
        if (!v1 && g_s2t_opt_attn_bwd_fused && al && idx32 && !a.causal && !a.dist_pen && a.Tk >= 128 && a.Tk <= FB_KEYS && a.Tq >= 128 &&
            span32(a.q_st, a.Tq) && span32(a.do_st, a.Tq)) {
            hipLaunchKernelGGL((attn_delta_kernel<T, DH>), dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, a);
            S2T_LAUNCH_CHECK();
            static bool attr = false;
            if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS); attr = true; }
            hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(a.B * a.H), dim3(512), FB_LDS, st, a);
            S2T_LAUNCH_CHECK();
            return S2T_OK;
        }
        dkv2 = !v1 && al && idx32 && a.Tk >= 128 && span32(a.q_st, a.Tq) && span32(a.do_st, a.Tq);
        dq2 = !v1 && al && idx32 && span32(a.k_st, a.Tk) && span32(a.v_st, a.Tk) && (a.Tq >= 128 || (a.Tq >= S2T_ATTN_V2_MIN_TQ && a.Tk >= 128)) && !(a.o_st % 8) && !(a.o_sb % 8) && !((uintptr_t)a.O & 15);
        if (dq2) {                                       // first: it also writes Delta for the dK/dV kernel
            hipLaunchKernelGGL(attn_bwd_dq2_kernel, dim3((a.Tq + 127) / 128, a.H, a.B), dim3(256), 32768, st, a);
            S2T_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL((attn_delta_kernel<T, DH>), dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, a);
            S2T_LAUNCH_CHECK();
        }
        if (dkv2) {
            {
                static bool attr = false;
                if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 + 1024 + 32768); attr = true; }
            }
            hipLaunchKernelGGL(attn_bwd_dkv2_kernel, dim3((a.Tk + 127) / 128, a.H, a.B), dim3(256), 32768 + 1024 + 32768, st, a);
            S2T_LAUNCH_CHECK();
        }
    } else {
        hipLaunchKernelGGL((attn_delta_kernel<T, DH>), dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, a);
        S2T_LAUNCH_CHECK();
    }
    if (!dkv2) {
        static bool attr = false;
        const size_t lds = 4 * C::TILE + 8 * C::PTILE;
        if (!attr && lds > 65536) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<T, DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH>), dim3((a.Tk + 63) / 64, a.H, a.B), dim3(256), lds, st, a);
        S2T_LAUNCH_CHECK();
    }
    if (!dq2) {
        static bool attr = false;
        const size_t lds = 3 * C::TILE + 4 * C::PTILE;
        if (!attr && lds > 65536) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<T, DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH>), dim3((a.Tq + 63) / 64, a.H, a.B), dim3(256), lds, st, a);
        S2T_LAUNCH_CHECK();
    }
    return S2T_OK;
}

static int check_common(int dtype, int head_dim, const AttnArgs& a) {
    if (dtype != S2T_F32 && dtype != S2T_BF16) return S2T_ENOTSUP;
    if (head_dim != 64 && head_dim != 32) return S2T_ENOTSUP;
    if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tk <= 0) return S2T_EINVAL;
    if (a.p_drop < 0.f || a.p_drop >= 1.f || (a.dist_pen != 0 && a.dist_pen != 1)) return S2T_EINVAL;
    return S2T_OK;
}

extern "C" int s2t_attn_fwd(int dtype, int head_dim, int B, int H, int Tq, int Tk,
                            const void* Q, long q_st, long q_sb, const void* K, long k_st, long k_sb,
                            const void* V, long v_st, long v_sb, void* O, long o_st, long o_sb, float* LSE,
                            const int* klen, int causal, int dist_penalty, float scale, float p_drop, unsigned long long seed,
                            void* stream) {
    AttnArgs a{};
    a.dist_pen = dist_penalty;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.LSE = LSE;
    a.q_st = q_st; a.q_sb = q_sb; a.k_st = k_st; a.k_sb = k_sb; a.v_st = v_st; a.v_sb = v_sb; a.o_st = o_st; a.o_sb = o_sb;
    a.klen = klen; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.causal = causal; a.scale = scale; a.p_drop = p_drop; a.seed = seed;
    if (B == 0 || Tq == 0) return S2T_OK;
    int rc = check_common(dtype, head_dim, a);
    if (rc) return rc;
    if (!Q || !K || !V || !O) return S2T_EINVAL;
    const long s[] = {q_st, q_sb, k_st, k_sb, v_st, v_sb};
    if (!strides_ok(dtype, s, 6)) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("attn_fwd", st, 4.0 * B * H * (double)Tq * Tk * head_dim * (causal ? 0.5 : 1.0), 0.0);
    if (dtype == S2T_BF16 && head_dim == 64 && (Tq >= 128 || (Tq >= S2T_ATTN_V2_MIN_TQ && Tk >= 128)) && (o_st % 4) == 0 && (o_sb % 4) == 0 && ((uintptr_t)O & 7) == 0) {
        const bool v1 = g_s2t_opt_attn_v1 != 0 || !span32(k_st, Tk) || !span32(v_st, Tk);     // s2t_set_option("attn_v1")
        if (!v1) {
            hipLaunchKernelGGL(attn_fwd2_kernel, dim3((Tq + 127) / 128, H, B), dim3(256), 32768 + 512 + 16384, st, a);
            S2T_LAUNCH_CHECK();
            return S2T_OK;
        }
    }
    if (dtype == S2T_BF16) return head_dim == 64 ? fwd_launch<bf16, 64>(a, st) : fwd_launch<bf16, 32>(a, st);
    return head_dim == 64 ? fwd_launch<float, 64>(a, st) : fwd_launch<float, 32>(a, st);
}

extern "C" int s2t_attn_bwd(int dtype, int head_dim, int B, int H, int Tq, int Tk,
                            const void* Q, long q_st, long q_sb, const void* K, long k_st, long k_sb,
                            const void* V, long v_st, long v_sb, const void* O, long o_st, long o_sb,
                            const void* dO, long do_st, long do_sb, const float* LSE, float* Delta,
                            void* dQ, long dq_st, long dq_sb, void* dK, long dk_st, long dk_sb,
                            void* dV, long dv_st, long dv_sb,
                            const int* klen, int causal, int dist_penalty, float scale, float p_drop, unsigned long long seed,
                            void* stream) {
    AttnArgs a{};
    a.dist_pen = dist_penalty;
    a.Q = Q; a.K = K; a.V = V; a.O = const_cast<void*>(O); a.LSE = const_cast<float*>(LSE);
    a.dO = dO; a.dQ = dQ; a.dK = dK; a.dV = dV; a.Delta = Delta;
    a.q_st = q_st; a.q_sb = q_sb; a.k_st = k_st; a.k_sb = k_sb; a.v_st = v_st; a.v_sb = v_sb; a.o_st = o_st; a.o_sb = o_sb;
    a.dq_st = dq_st; a.dq_sb = dq_sb; a.dk_st = dk_st; a.dk_sb = dk_sb; a.dv_st = dv_st; a.dv_sb = dv_sb; a.do_st = do_st; a.do_sb = do_sb;
    a.klen = klen; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.causal = causal; a.scale = scale; a.p_drop = p_drop; a.seed = seed;
    if (B == 0 || Tq == 0) return S2T_OK;
    int rc = check_common(dtype, head_dim, a);
    if (rc) return rc;
    if (!Q || !K || !V || !O || !dO || !LSE || !Delta || !dQ || !dK || !dV) return S2T_EINVAL;
    const long s[] = {q_st, q_sb, k_st, k_sb, v_st, v_sb, do_st, do_sb};
    if (!strides_ok(dtype, s, 8)) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("attn_bwd", st, 14.0 * B * H * (double)Tq * Tk * head_dim * (causal ? 0.5 : 1.0), 0.0);
    if (dtype == S2T_BF16) return head_dim == 64 ? bwd_launch<bf16, 64>(a, st) : bwd_launch<bf16, 32>(a, st);
    return head_dim == 64 ? bwd_launch<float, 64>(a, st) : bwd_launch<float, 32>(a, st);
}

// ------------------------------------------------------------------------------------ head-averaged attention probabilities
// fairseq/models/transformer.py:756-782 (TransformerDecoder.extract_features): the decoder returns, for `alignment_layer` (the last
// layer by default), the encoder-attention probabilities averaged over the heads -- what generate.py --print-alignment and the
// ensemble's attention averaging consume (fairseq_cli/generate.py:82,219; sequence_generator.py:757-768).  The fused attention kernels
// never materialise P, so this recomputes it for the ONE layer that is asked for: out[b][tq][tk] = mean_h softmax_tk(scale q_h . k_h),
// keys past klen[b] get probability 0 (key padding: -inf scores).  Eval-mode semantics (no dropout on P), as the reference's
// need_weights path returns the softmax output after dropout only in training -- alignment is an inference feature.
// One workgroup per (query, batch); plain f32 VALU dot products: a decoding step has Tq = 1 and a few hundred keys.
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_avg_kernel(const T* __restrict__ Q, long q_st, long q_sb, const T* __restrict__ Kp, long k_st,
                                                             long k_sb, const int* __restrict__ klen, float* __restrict__ out, int H, int dh,
                                                             int Tq, int Tk, int heads_used, float scale) {
    __shared__ float sh[16];
    __shared__ float qs[128];
    const int tq = blockIdx.x, b = blockIdx.y;
    const int kl = klen ? min(klen[b], Tk) : Tk;
    constexpr int NK = 8;                                     // keys per thread: Tk <= 2048 (host check)
    float acc[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) acc[i] = 0.f;
    for (int h = 0; h < heads_used; ++h) {
        __syncthreads();
        if ((int)threadIdx.x < dh) qs[threadIdx.x] = to_f32(Q[tq * q_st + b * q_sb + h * dh + threadIdx.x]) * scale;
        __syncthreads();
        float s[NK];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int tk = threadIdx.x + 256 * i;
            s[i] = -INFINITY;
            if (tk < kl) {
                const T* kr = Kp + (long)tk * k_st + b * k_sb + h * dh;
                float d = 0.f;
                for (int j = 0; j < dh; ++j) d += qs[j] * to_f32(kr[j]);
                s[i] = d;
            }
            m = fmaxf(m, s[i]);
        }
        m = block_max(m, sh);
        float l = 0.f;
#pragma unroll
        for (int i = 0; i < NK; ++i) { s[i] = (s[i] == -INFINITY) ? 0.f : expf(s[i] - m); l += s[i]; }
        l = block_sum(l, sh);
        const float inv = l > 0.f ? 1.f / (l * heads_used) : 0.f;
#pragma unroll
        for (int i = 0; i < NK; ++i) acc[i] += s[i] * inv;
    }
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int tk = threadIdx.x + 256 * i;
        if (tk < Tk) out[((long)b * Tq + tq) * Tk + tk] = acc[i];
    }
}

extern "C" int s2t_attn_probs_avg(int dtype, int head_dim, int B, int H, int Tq, int Tk, const void* Q, long q_st, long q_sb,
                                  const void* K, long k_st, long k_sb, const int* klen, int heads_used, float scale, float* out,
                                  void* stream) {
    if (B == 0 || Tq == 0 || Tk == 0) return S2T_OK;
    if (!Q || !K || !out || head_dim <= 0 || head_dim > 128 || H <= 0 || heads_used <= 0 || heads_used > H) return S2T_EINVAL;
    if (Tk > 2048) return S2T_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)Tq, (unsigned)B);
    if (dtype == S2T_BF16) hipLaunchKernelGGL(attn_probs_avg_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)Q, q_st, q_sb, (const bf16*)K, k_st, k_sb,
                                               klen, out, H, head_dim, Tq, Tk, heads_used, scale);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(attn_probs_avg_kernel<float>, grid, dim3(256), 0, st, (const float*)Q, q_st, q_sb, (const float*)K, k_st,
                                                  k_sb, klen, out, H, head_dim, Tq, Tk, heads_used, scale);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
