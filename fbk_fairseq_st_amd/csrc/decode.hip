// Incremental decoding + beam search: every launch of one decoding step (include/s2t_hip.h: s2t_decode_step).
//
// Reference behaviour: fairseq/sequence_generator.py:243-447 (loop body), fairseq/search.py:55-83 (BeamSearch.step),
// fairseq/models/transformer.py:674-782 and fairseq/modules/transformer_layer.py:243-377 with incremental_state,
// fairseq/modules/multihead_attention.py:246-283 (cached keys/values), :407-420 (reorder_incremental_state).
//
// Shape of the problem on MI355X: N = B * beam <= 128 rows against ~52 MB (bf16, m preset) of decoder weights, a K/V cache that grows by
// one row per step, and B * Ts encoder rows per layer.  Nothing is compute-bound; the step is a chain of dependent phases, each a few
// microseconds, so what counts is (1) how many phases there are (a kernel boundary costs ~1.3 us inside a captured graph, a grid barrier
// ~4 us: MI355X_MICROARCH.md price list -- so the seams stay kernel boundaries and the whole step is captured once), (2) that no phase
// re-reads or copies state (index indirection `anc` instead of re-ordering the cache; encoder K/V once per sentence), (3) that every
// phase fills the chip from work that is independent per (sentence, head) or (sentence, ffn slice): the hypotheses of one sentence share a
// workgroup, which multiplies the weight slice it streams from L2 by a 16-row MFMA tile (rows >= beam are duplicates of the last row
// and never stored).  Per-head / per-slice shares of the two output projections and of fc2 are written as f32 slabs and summed, in a
// fixed order, by the NEXT phase's prologue together with residual and bias: deterministic, no atomics, no extra launch.
#include "common.hpp"
#include "s2t_hip.h"

// Diagnostic build only (make dec_stamps: -DS2T_DEC_STAMPS): shader-clock stamps of workgroup (0, 0), wave 0 at the phase boundaries of
// the step's kernels, read back with s2t_decode_read_stamps (tools/decode_stamps.py).  The product build compiles none of it.
#ifdef S2T_DEC_STAMPS
__device__ unsigned long long s2t_dec_stamps[8 * 16];
#define DSTAMP(kid, i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { s2t_dec_stamps[(kid) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
                                 if ((i) == 0) s2t_dec_stamps[(kid) * 16 + 14] = __builtin_amdgcn_s_memrealtime(); \
                                 else s2t_dec_stamps[(kid) * 16 + 15] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define DSTAMP(kid, i) do { } while (0)
#endif

#ifndef S2T_DEC_NB_FFN
#define S2T_DEC_NB_FFN 8
#endif
#ifndef S2T_DEC_NB_CROSS
#define S2T_DEC_NB_CROSS 8
#endif
namespace {
constexpr int DH = 64;                 // head size (256/4, 512/8, 1024/16: every preset of the reference and of SURVEY 8-P)
constexpr int NTHREADS = 256;
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct FR { static constexpr int PER = 16 / (int)sizeof(T), KS = 4 * PER; };   // elements per 16 B, k per mma16

__host__ __device__ inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
// the PER elements of a 16-byte fragment as f32
template <typename T> struct Unp;
template <> struct Unp<float> {
    float f[4];
    __device__ __forceinline__ explicit Unp(u32x4 v) {
        const f32x4 x = __builtin_bit_cast(f32x4, v);
        f[0] = x[0]; f[1] = x[1]; f[2] = x[2]; f[3] = x[3];
    }
};
template <> struct Unp<bf16> {
    float f[8];
    __device__ __forceinline__ explicit Unp(u32x4 v) {
        const uint32_t w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
        f[0] = __builtin_bit_cast(float, w0 << 16); f[1] = __builtin_bit_cast(float, w0 & 0xffff0000u);
        f[2] = __builtin_bit_cast(float, w1 << 16); f[3] = __builtin_bit_cast(float, w1 & 0xffff0000u);
        f[4] = __builtin_bit_cast(float, w2 << 16); f[5] = __builtin_bit_cast(float, w2 & 0xffff0000u);
        f[6] = __builtin_bit_cast(float, w3 << 16); f[7] = __builtin_bit_cast(float, w3 & 0xffff0000u);
    }
};

// bf16 dot products of the self-attention launch: the lane's query slice is rounded to bf16 once (as the training attention kernels hold
// q) and a 16-byte key fragment costs four v_dot2_f32_bf16 -- unpacking eight bf16 and eight FMAs were 16 of the ~25 VALU instructions
// per (position, hypothesis) of the score loop, on a wave that issues one every ~8 cycles.  f32 keeps the plain multiply-adds.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
// ------------------------------------------------------------------------------------------------ cross-lane reductions on DPP
// The kernels below run one wave per SIMD on dependent chains, so a reduction's LATENCY is what counts: a ds_bpermute butterfly
// (__shfl_xor) costs ~100+ cycles per step, a DPP operand ~8.  quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror leave
// every lane of a 4 / 8 / 16-lane group with the group's result; the four 16-lane rows meet through v_readlane.  All 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float rl_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
template <int G> __device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v);
    if (G >= 8) v += dpp_f<0x141>(v);
    if (G >= 16) v += dpp_f<0x140>(v);
    if (G == 64) v = (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
    return v;
}
__device__ __forceinline__ float wave_max64(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(rl_f(v, 0), rl_f(v, 16)), fmaxf(rl_f(v, 32), rl_f(v, 48)));
}
__device__ __forceinline__ int wave_min64(int v) {
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// arg-max over the wave with ties to the smaller index: (value, index) of the winner in every lane
__device__ __forceinline__ void wave_argmax(float v, int i, float& mv, int& mi) {
    mv = wave_max64(v);
    mi = wave_min64(v == mv ? i : 0x7fffffff);
}

// ------------------------------------------------------------------------------------------------ LDS plans (host and device agree)
struct SelfLds { size_t a_ln, q, kt, vt, sc, anc, o, red, total; };
__host__ __device__ inline SelfLds self_lds(int RT, int D, int maxpos, int es) {
    SelfLds l; size_t o = 128; const int per = 16 / es;           // the first 128 bytes: the prologue's reduction scratch
    l.a_ln = o; o += up16((size_t)RT * (D + per) * es);
    l.q = o;    o += up16((size_t)RT * DH * 4);
    l.kt = o;   o += up16((size_t)RT * DH * es);                  // this step's key / value rows (position t), as stored in the cache
    l.vt = o;   o += up16((size_t)RT * DH * es);
    l.sc = o;   o += up16((size_t)RT * maxpos * 4);
    l.anc = o;  o += up16((size_t)RT * maxpos * 4);
    l.o = o;    o += up16((size_t)RT * (DH + per) * es);
    l.red = o;  o += up16((size_t)16 * RT * DH * 4);
    l.total = o; return l;
}
struct CrossLds { size_t a_ln, q, sc, p, o, total; };
__host__ __device__ inline CrossLds cross_lds(int R, int D, int Tsp, int es) {
    CrossLds l; size_t o = 128; const int per = 16 / es;
    l.a_ln = o; o += up16((size_t)R * (D + per) * es);
    l.q = o;    o += up16((size_t)R * (DH + per) * es);
    l.sc = o;   o += up16((size_t)R * Tsp * 4);
    l.p = o;    o += up16((size_t)R * (Tsp + per) * es);
    l.o = o;    o += up16((size_t)R * (DH + per) * es);
    l.total = o; return l;
}
struct FfnLds { size_t a_ln, h, total; };
__host__ __device__ inline FfnLds ffn_lds(int R, int D, int hs, int es) {
    FfnLds l; size_t o = 128; const int per = 16 / es;
    l.a_ln = o; o += up16((size_t)R * (D + per) * es);
    l.h = o;    o += up16((size_t)R * (hs + per) * es);
    l.total = o; return l;
}

// ------------------------------------------------------------------------------------------------ shared prologue
// Every launch of a step is a chain of dependent phases on one wave per SIMD, so its length is (bytes streamed) / (what one CU keeps in
// flight) + (number of DEPENDENT memory round trips) x (1-2 us each for lines another CU wrote).  The rules the kernels below follow:
// everything whose address is known at the top of the launch is requested there, in the order in which it is needed (loads return in
// order); workgroup barriers between the phases order LDS only (lds_barrier), so that requests stay in flight across them -- a
// __syncthreads() waits for every outstanding load of the wave.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
// rows n0 .. n0+R-1:  v = x_in + bias + sum_p part_in[p]   (fixed order: the result does not depend on scheduling)
// the writer workgroup stores v (the residual stream after the previous block) to x_out; every workgroup normalises its copy:
// dst[r * ld + d] = T(LayerNorm(v)[d])  (fairseq/modules/layer_norm.py: eps inside the square root, biased variance, f32 statistics).
// All 256 threads work on all rows at once: a thread owns up to four float4 of the [R][D] block (item = row * D/4 + column/4) and requests
// x, bias, gamma, beta and the first NB shares of each of them together (the shares were written by other CUs: every one is an L2 miss);
// the row statistics go wave (DPP) -> LDS -> thread.  `red`: 32 floats of LDS.
// mid(): called once, after those requests and before the first barrier: the caller requests what else it streams (weights) behind them
// and returns false when the workgroup has nothing to do (the sentence is past its last step): the prologue then returns false at once.
struct Pro {
    const float *x_in; const void* part_in; const float* bias; float* x_out; const float *g, *b; int np; float eps;
};
// four consecutive elements of a share (stored in the compute type: f32 in f32 mode, bf16 in bf16 mode, where they are half the bytes of
// the step's second largest stream and rounded no more coarsely than that mode's activations), as loaded and as f32
template <typename T> struct Raw4;
template <> struct Raw4<float> {
    typedef f32x4 type;
    static __device__ __forceinline__ type ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ f32x4 cvt(type r) { return r; }
};
template <> struct Raw4<bf16> {
    typedef u32x2 type;
    static __device__ __forceinline__ type ld(const bf16* p) { return *reinterpret_cast<const u32x2*>(p); }
    static __device__ __forceinline__ f32x4 cvt(type w) {
        return f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                     __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
    }
};
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float (&v)[4]) {
    typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}
// one pass: the 1024 items from `base` (whole rows: 1024 is a multiple of the items per row)
template <typename T, int NB, int NK, bool FULL, typename Mid>
__device__ __forceinline__ bool pro_pass(const Pro& p, bool writer, int N, int D, int n0, int R, T* dst, int ld, float* red, int base, Mid&& mid) {
    typedef typename Raw4<T>::type raw_t;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int sh = D == 256 ? 6 : (D == 512 ? 7 : 8);           // log2(float4 items per row)
    const int IPR = 1 << sh, WPR = IPR >> 6, total = R << sh;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {
        int row[NK]; unsigned off[NK]; bool ok[NK];
        f32x4 v[NK];
        raw_t t[NB][NK];
        const int col = (tid & (IPR - 1)) * 4;                   // IPR divides 256: a thread's four items are in one column
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int it = base + k * 256 + tid;
            ok[k] = it < total;
            row[k] = min(it, total - 1) >> sh;                   // past the end: the last row again (same column), never stored
            off[k] = (unsigned)((n0 + row[k]) * D + col);        // element offset inside an [N][D] slab
        }
        auto request = [&](int q0) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int q = min(q0 + u, p.np - 1);
#pragma unroll
                for (int k = 0; k < NK; ++k) t[u][k] = Raw4<T>::ld(reinterpret_cast<const T*>(p.part_in) + (size_t)q * N * D + off[k]);
            }
        };
#pragma unroll
        for (int k = 0; k < NK; ++k) v[k] = *reinterpret_cast<const f32x4*>(p.x_in + off[k]);
        const f32x4 bs = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : zero;
        if (p.np > 0) request(0);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(p.g + col), bb = *reinterpret_cast<const f32x4*>(p.b + col);
        if (!mid()) return false;
        if (p.bias) {
#pragma unroll
            for (int k = 0; k < NK; ++k) v[k] += bs;
        }
        for (int q0 = 0; q0 < p.np; q0 += NB) {
            if (q0 > 0) request(q0);
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const bool use = FULL || q0 + u < p.np;        // FULL: np is a multiple of NB -- no select per addend
#pragma unroll
                for (int k = 0; k < NK; ++k) v[k] += use ? Raw4<T>::cvt(t[u][k]) : zero;
            }
        }
        if (writer) {
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (ok[k]) *reinterpret_cast<f32x4*>(p.x_out + off[k]) = v[k];
        }
        float st[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) st[k] = group_sum<64>(ok[k] ? (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]) : 0.f);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NK; ++k) red[k * 4 + w] = st[k];
        }
        lds_barrier();
        float mean[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int first = ((row[k] << sh) - base) >> 6;
            float m = 0.f;
            for (int j = 0; j < WPR; ++j) m += red[first + j];
            mean[k] = m / (float)D;
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const f32x4 c = v[k] - mean[k];
            st[k] = group_sum<64>(ok[k] ? (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]) : 0.f);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NK; ++k) red[16 + k * 4 + w] = st[k];
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int first = ((row[k] << sh) - base) >> 6;
            float q2 = 0.f;
            for (int j = 0; j < WPR; ++j) q2 += red[16 + first + j];
            const float rstd = rsqrtf(q2 / (float)D + p.eps);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[k][j] - mean[k]) * rstd * gg[j] + bb[j];
            if (ok[k]) store4<T>(dst + (size_t)row[k] * ld + col, o);
        }
        lds_barrier();
    }
    return true;
}
// the first pass is straight-line code around mid() (values it defines -- weight fragments -- are not carried around a loop)
template <typename T, int NB, typename Mid>
__device__ __forceinline__ bool dec_prologue(const Pro& p, bool writer, int N, int D, int n0, int R, T* dst, int ld, float* red, Mid&& mid) {
    // NK = 256-item slots of a pass (a thread's items): three when the rows fit (beam 5 x D 512 = 640 items), so that the fourth slot's
    // clamped duplicates cost neither requests nor arithmetic; chosen once, outside the straight-line pass
    const int total = R * (D / 4);
    if (total <= 768) {
        if (p.np == NB) return pro_pass<T, NB, 3, true>(p, writer, N, D, n0, R, dst, ld, red, 0, mid);
        return pro_pass<T, NB, 3, false>(p, writer, N, D, n0, R, dst, ld, red, 0, mid);
    }
    if (!pro_pass<T, NB, 4, false>(p, writer, N, D, n0, R, dst, ld, red, 0, mid)) return false;
    for (int base = 1024; base < total; base += 1024) pro_pass<T, NB, 4, false>(p, writer, N, D, n0, R, dst, ld, red, base, []() { return true; });
    return true;
}

// ---- softmax over the positions of R rows of scores in LDS.  GRP lanes per row: a wave for up to four rows, half a wave for up to
// eight, a 16-lane row above -- so that the rows of the beam run side by side instead of one after the other on the same wave.
// fin(row, position, probability) stores the result.  f32 throughout (multihead_attention.py:338-339).
template <int GRP> __device__ __forceinline__ float grp_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    if (GRP == 16) return v;
    const float lo = fmaxf(rl_f(v, 0), rl_f(v, 16)), hi = fmaxf(rl_f(v, 32), rl_f(v, 48));
    if (GRP == 32) return (threadIdx.x & 32) ? hi : lo;
    return fmaxf(lo, hi);
}
template <int GRP> __device__ __forceinline__ float grp_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    if (GRP == 16) return v;
    const float lo = rl_f(v, 0) + rl_f(v, 16), hi = rl_f(v, 32) + rl_f(v, 48);
    if (GRP == 32) return (threadIdx.x & 32) ? hi : lo;
    return lo + hi;
}
// FAST (bf16 mode: the probabilities are rounded to bf16 or multiplied into bf16 values next): v_exp_f32 on (x - m) log2 e and one reciprocal
// per row instead of expf and a division per element -- a third of the loop's instructions on a wave that issues one every ~8 cycles
template <int GRP, bool FAST, typename Fin>
__device__ __forceinline__ void softmax_rows_g(float* sc, int ldsc, int R, int n, Fin&& fin) {
    constexpr int NG = NTHREADS / GRP;
    const int g = threadIdx.x / GRP, gl = threadIdx.x % GRP;
    for (int r0 = 0; r0 < R; r0 += NG) {                       // the same trip count in every lane: the DPP reductions need all 64 active
        const bool act = r0 + g < R;
        float* row = sc + (size_t)min(r0 + g, R - 1) * ldsc;      // an idle group reads the last row and stores nothing
        float m = -INFINITY;
        for (int p = gl; p < n; p += GRP) m = fmaxf(m, row[p]);
        m = grp_max<GRP>(m);
        float z = 0.f;
        for (int p = gl; p < n; p += GRP) {
            const float e = FAST ? __builtin_amdgcn_exp2f((row[p] - m) * 1.44269504088896f) : expf(row[p] - m);
            if (act) row[p] = e;
            z += e;
        }
        z = grp_sum<GRP>(z);
        const float iz = FAST ? __builtin_amdgcn_rcpf(z) : 0.f;
        if (act)
            for (int p = gl; p < n; p += GRP) fin(r0 + g, p, FAST ? row[p] * iz : row[p] / z);
    }
}
template <bool FAST, typename Fin>
__device__ __forceinline__ void softmax_rows(float* sc, int ldsc, int R, int n, Fin&& fin) {
    if (R <= 4) softmax_rows_g<64, FAST>(sc, ldsc, R, n, fin);
    else if (R <= 8) softmax_rows_g<32, FAST>(sc, ldsc, R, n, fin);
    else softmax_rows_g<16, FAST>(sc, ldsc, R, n, fin);
}

// ---- weights in FRAGMENT-MAJOR order (s2t_decode_pack_weight): W [N][K] -> [N / 16 column tiles][K / KS k-steps][64 lanes][16 bytes], the
// 16 bytes of lane l of (tile, step) being W[16 tile + (l & 15)][KS step + PER (l >> 4) ..]: exactly the B operand of one MFMA, so a wave's
// load of a fragment is ONE contiguous KiB (eight whole cache lines) instead of 16 rows x 64 bytes (sixteen half lines).  A CU keeps a
// limited number of line requests in flight: measured on the feed-forward launch, the wait for 256 KB of weights went from 24,500 to
// 13,400 cycles (profiles/r06_decode_experiments.txt).
// (a wave-uniform base -- the callers' tile indices come from readfirstlane -- plus a 32-bit lane offset: one scalar base and immediate
// offsets for a run of fragments instead of a 64-bit address pair in VGPRs per load)
template <typename T> __device__ __forceinline__ const T* wfrag(const T* Wp, int ksteps, int tile, int step) {
    const T* ub = Wp + ((size_t)tile * ksteps + step) * (64 * FR<T>::PER);
    return ub + (unsigned)((threadIdx.x & 63) * FR<T>::PER);
}

// acc[i] += A (LDS fragment stream at ap, 16 rows) x W_i^T for NT column tiles of a fragment-major weight, K deep
// (K a multiple of CH * KS; the tiles' fragments at wfrag(Wp, ksteps, tile0 + i tile_step, step)).  Every global fragment of a chunk of CH k-steps is requested before the first MFMA of the chunk (NT * CH
// 16-byte loads in flight per lane); nothing in the chunk is conditional, so the compiler keeps one straight-line block per chunk.
template <typename T, int NT, int CH>
__device__ __forceinline__ void mma_rows(const T* ap, const T* Wp, int ksteps, int tile0, int tile_step, int K, f32x4 (&acc)[NT]) {
    constexpr int KS = FR<T>::KS;
    for (int k0 = 0; k0 < K; k0 += CH * KS) {
        u32x4 b[NT][CH];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) b[i][c] = ld16(wfrag<T>(Wp, ksteps, tile0 + i * tile_step, k0 / KS + c));
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const u32x4 af = ld16(ap + k0 + c * KS);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = mma16<T>(af, b[i][c], acc[i]);
        }
    }
}

// one head's (or slice's) share of an output projection: out[row][col] = sum_k A[row][k] * W[col][KS step_off + k], k < KST * KS
// A: LDS, T [R][lda] (rows >= R read row R-1: their results are never stored); Wp: the fragment-major weight (`ksteps` k-steps per row
// of tiles); all D output columns.
// A wave owns the column tiles w, w+4, ... and requests the weight fragments of TG of them at a time (D/64 is a multiple of TG).
template <typename T, int KST>
__device__ __forceinline__ void share_out_t(const T* a_s, int lda, const T* __restrict__ Wp, int ksteps, int step_off, T* __restrict__ out, int D, int R) {
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS, TG = KST >= 16 ? 1 : (KST == 8 ? 2 : 4);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const T* ap = a_s + (size_t)min(lane & 15, R - 1) * lda + PER * (lane >> 4);
    u32x4 af[KST];
#pragma unroll
    for (int c = 0; c < KST; ++c) af[c] = ld16(ap + c * KS);
    const int per_wave = D / 64;
    for (int i0 = 0; i0 < per_wave; i0 += TG) {
        u32x4 b[TG][KST];
#pragma unroll
        for (int g = 0; g < TG; ++g)
#pragma unroll
            for (int c = 0; c < KST; ++c) b[g][c] = ld16(wfrag<T>(Wp, ksteps, w + 4 * (i0 + g), step_off + c));
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KST; ++c) acc = mma16<T>(af[c], b[g][c], acc);
            const int col = (w + 4 * (i0 + g)) * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * (lane >> 4) + i;
                if (r < R) out[(size_t)r * D + col] = from_f32<T>(acc[i]);
            }
        }
    }
}
// the same product with the weight fragments already in registers (requested at the top of the kernel): b[i][c] = fragment c of the
// wave's i-th column tile (tile w + 4 i)
template <typename T, int NTILE, int KST>
__device__ __forceinline__ void load_share_w(u32x4 (&b)[NTILE][KST], const T* __restrict__ Wp, int ksteps, int step_off) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < NTILE; ++i)
#pragma unroll
        for (int c = 0; c < KST; ++c) b[i][c] = ld16(wfrag<T>(Wp, ksteps, w + 4 * i, step_off + c));
}
template <typename T, int NTILE, int KST>
__device__ __forceinline__ void share_regs(const T* a_s, int lda, const u32x4 (&b)[NTILE][KST], T* __restrict__ out, int D, int R) {
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const T* ap = a_s + (size_t)min(lane & 15, R - 1) * lda + PER * (lane >> 4);
    u32x4 af[KST];
#pragma unroll
    for (int c = 0; c < KST; ++c) af[c] = ld16(ap + c * KS);
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KST; ++c) acc = mma16<T>(af[c], b[i][c], acc);
        const int col = (w + 4 * i) * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * (lane >> 4) + q;
            if (r < R) out[(size_t)r * D + col] = from_f32<T>(acc[q]);
        }
    }
}
// the fragments of NT column tiles (tile0 + i tile_step) over a K of KST k-steps, and the product from them
template <typename T, int NT, int KST>
__device__ __forceinline__ void load_rows_w(u32x4 (&b)[NT][KST], const T* Wp, int tile0, int tile_step) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int c = 0; c < KST; ++c) b[i][c] = ld16(wfrag<T>(Wp, KST, tile0 + i * tile_step, c));
}
template <typename T, int NT, int KST>
__device__ __forceinline__ void mma_regs(const T* ap, const u32x4 (&b)[NT][KST], f32x4 (&acc)[NT]) {
    constexpr int KS = FR<T>::KS;
#pragma unroll
    for (int c = 0; c < KST; ++c) {
        const u32x4 af = ld16(ap + c * KS);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = mma16<T>(af, b[i][c], acc[i]);
    }
}
template <typename T>
__device__ __forceinline__ void share_out(const T* a_s, int lda, int klen, const T* __restrict__ Wp, int ksteps, int step_off, T* __restrict__ out,
                                          int D, int R) {
    switch (klen / FR<T>::KS) {
        case 2: share_out_t<T, 2>(a_s, lda, Wp, ksteps, step_off, out, D, R); break;
        case 4: share_out_t<T, 4>(a_s, lda, Wp, ksteps, step_off, out, D, R); break;
        case 8: share_out_t<T, 8>(a_s, lda, Wp, ksteps, step_off, out, D, R); break;
        default: share_out_t<T, 16>(a_s, lda, Wp, ksteps, step_off, out, D, R); break;       // 16: f32 with 256 hidden units per slice
    }
}

// ------------------------------------------------------------------------------------------------ S: self-attention block of one head
// grid (heads, B): workgroups that stream the same head's weights have equal blockIdx.x, i.e. (round-robin dispatch) share an XCD's L2.
struct SelfArgs {
    Pro pro; int B, beam, N, D, heads, maxpos, max_len; float scale;
    const void *w_qkv; const float* b_qkv; const void* w_o; void* cache; const int* anc; const int* steps; void* part_out;
};
// RT = rows held in registers by the attention loops: the beam itself for beam <= 8 (exact: no wasted lanes), 16 above (rows >= beam
// repeat row beam-1 and are never stored)
// DD = D when the weight fragments of the whole launch fit in registers (bf16, D <= 512: q|k|v 3 * D/32 fragments per lane, the output
// projection's D/32 after them in the same registers).  DD = 0: D at run time, fragments requested chunk by chunk where they are used
// (f32, D = 1024).
// Order of requests (they return in order): x / bias / shares / gamma / beta of the prologue; the ancestor rows; q|k|v weights; then,
// once the ancestors are in LDS, the cached keys AND values of the first PU * NSLOT positions (all of them while t < 128 at beam 5) --
// they depend on nothing this launch computes, so they travel under the LayerNorm and the q|k|v product; the output projection's
// fragments go out when the q|k|v fragments are dead.  Position t's own key / value rows are used from LDS, not read back.
template <typename T, int RT, int DD>
__global__ __launch_bounds__(NTHREADS) void dec_self_kernel(SelfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS, DC = DH / PER, NSLOT = NTHREADS / DC, PU = RT <= 5 ? 4 : (RT <= 8 ? 2 : 1);
    constexpr bool PRE = DD > 0, PRE_O = PRE && RT <= 8;
    constexpr int KSTD = PRE ? DD / KS : 1, NTO = PRE_O ? DD / 64 : 1, CH0 = PU * NSLOT;
    const int h = blockIdx.x, s = blockIdx.y, R = a.beam, n0 = s * R, D = PRE ? DD : a.D, N = a.N, maxpos = a.maxpos;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int t_raw = a.steps[s];
    DSTAMP(0, 0);
    u32x4 wqkv[3][KSTD], wo[NTO][DH / KS];
    const SelfLds L = self_lds(RT, D, maxpos, (int)sizeof(T));
    float* red0 = reinterpret_cast<float*>(smem);
    T* a_ln = reinterpret_cast<T*>(smem + L.a_ln);
    float* q_s = reinterpret_cast<float*>(smem + L.q);
    T* k_s = reinterpret_cast<T*>(smem + L.kt);
    T* v_s = reinterpret_cast<T*>(smem + L.vt);
    float* sc = reinterpret_cast<float*>(smem + L.sc);
    int* anc_s = reinterpret_cast<int*>(smem + L.anc);
    T* o_s = reinterpret_cast<T*>(smem + L.o);
    float* red = reinterpret_cast<float*>(smem + L.red);
    const int lda = D + PER;
    T* cache = reinterpret_cast<T*>(a.cache);
    int t = 0;
    const bool go = dec_prologue<T, 16>(a.pro, h == 0, N, D, n0, R, a_ln, lda, red0, [&]() __attribute__((always_inline)) {
        t = t_raw;
        if (t > a.max_len) return false;
        // ancestors of the sentence's rows at positions < t, RT rows x 256 positions requested together; the weights go out behind the
        // first batch and before its LDS stores (which wait for those RT loads only)
        for (int p0 = 0; p0 < t || p0 == 0; p0 += NTHREADS) {
            int av[RT];
            const int pc = min(p0 + tid, max(t - 1, 0));
#pragma unroll
            for (int r = 0; r < RT; ++r) av[r] = a.anc[(size_t)(n0 + (r < R ? r : R - 1)) * maxpos + pc];
            if constexpr (PRE) { if (p0 == 0) load_rows_w<T, 3, KSTD>(wqkv, reinterpret_cast<const T*>(a.w_qkv), h * 4 + w, D / 16); }
            if (p0 + tid < t) {
#pragma unroll
                for (int r = 0; r < RT; ++r) if (r < R) anc_s[r * maxpos + p0 + tid] = av[r];
            }
        }
        if (tid < R) anc_s[tid * maxpos + t] = n0 + tid;          // position t: the row itself
        return true;
    });
    if (!go) return;
    DSTAMP(0, 1);

    const int slot = tid / DC, dc = tid % DC;
    // cached keys and values of positions [0, CH0): DC lanes share one 64-wide row (16 bytes each), NSLOT rows per pass, PU passes, all RT
    // hypotheses.  Positions >= t are served from LDS below; their requests here repeat position t-1 (never used).
    u32x4 kv[PU][RT], vv[PU][RT];
    {
        const int tm1 = max(t - 1, 0);
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int pc = min(u * NSLOT + slot, tm1);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int rr = r < R ? r : R - 1;
                const T* src = cache + ((size_t)pc * N + anc_s[rr * maxpos + pc]) * 2 * D + h * DH + dc * PER;
                kv[u][r] = ld16(src);
                vv[u][r] = ld16(src + D);
            }
        }
    }
    {   // q | k | v columns of this head: wave w owns column tile w of each of the three (3 x 16 columns), K = D
        const T* ap = a_ln + (size_t)min(lane & 15, R - 1) * lda + PER * (lane >> 4);
        f32x4 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) mma_regs<T, 3, KSTD>(ap, wqkv, acc);
        else mma_rows<T, 3, 8>(ap, reinterpret_cast<const T*>(a.w_qkv), D / KS, h * 4 + w, D / 16, D, acc);
        const int col = 16 * w + (lane & 15);
        const float bq = a.b_qkv[h * DH + col], bk = a.b_qkv[D + h * DH + col], bv = a.b_qkv[2 * D + h * DH + col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * (lane >> 4) + i;
            if (r < R) {
                q_s[r * DH + col] = (acc[0][i] + bq) * a.scale;               // multihead_attention.py:155-160: q = (Wq x + b) * d^-1/2
                const T kt = from_f32<T>(acc[1][i] + bk), vt = from_f32<T>(acc[2][i] + bv);
                T* row = cache + ((size_t)t * N + n0 + r) * 2 * D + h * DH + col;
                row[0] = kt; row[D] = vt;                                       // the cache row of position t: later steps read it
                k_s[r * DH + col] = kt; v_s[r * DH + col] = vt;                 // this step reads it from here
            }
        }
    }
    if constexpr (PRE_O) load_share_w<T, NTO, DH / KS>(wo, reinterpret_cast<const T*>(a.w_o), D / KS, h * (DH / KS));
    lds_barrier();
    DSTAMP(0, 2);

    // scores[r][pos] = q_r . k_{anc(r, pos)}.  Positions past t repeat position t (same value written twice), rows past the beam the last row.
    constexpr bool DOT2 = sizeof(T) == 2;                // bf16: packed query slices and v_dot2_f32_bf16
    float qr[RT][DOT2 ? 1 : PER];                        // this lane's PER columns of every row's query, and of the step's own keys
    uint32_t qp[RT][DOT2 ? 4 : 1];
    u32x4 kown[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int rr = r < R ? r : R - 1;
        if constexpr (DOT2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) qp[r][j] = pack2_bf16(q_s[rr * DH + dc * PER + 2 * j], q_s[rr * DH + dc * PER + 2 * j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) qr[r][j] = q_s[rr * DH + dc * PER + j];
        }
        kown[r] = ld16(k_s + rr * DH + dc * PER);
    }
    auto score_chunk = [&](int pos0, const u32x4 (&kk)[PU][RT]) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int pos = pos0 + u * NSLOT + slot;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int rr = r < R ? r : R - 1;
                float d = 0.f;
                if constexpr (DOT2) {
                    const u32x4 kf = pos >= t ? kown[r] : kk[u][r];
#pragma unroll
                    for (int j = 0; j < 4; ++j) d = dot2_bf16(qp[r][j], kf[j], d);
                } else {
                    const Unp<T> kf(pos >= t ? kown[r] : kk[u][r]);
#pragma unroll
                    for (int j = 0; j < PER; ++j) d += qr[r][j] * kf.f[j];
                }
                d = group_sum<DC>(d);
                if (dc == 0) sc[rr * maxpos + min(pos, t)] = d;
            }
        }
    };
    score_chunk(0, kv);
    for (int pos0 = CH0; pos0 <= t; pos0 += CH0) {
        u32x4 kk[PU][RT];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int pc = min(pos0 + u * NSLOT + slot, t - 1);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int rr = r < R ? r : R - 1;
                kk[u][r] = ld16(cache + ((size_t)pc * N + anc_s[rr * maxpos + pc]) * 2 * D + h * DH + dc * PER);
            }
        }
        score_chunk(pos0, kk);
    }
    lds_barrier();
    DSTAMP(0, 3);
    softmax_rows<sizeof(T) == 2>(sc, maxpos, R, t + 1, [&](int r, int p, float v) { sc[r * maxpos + p] = v; });
    lds_barrier();
    DSTAMP(0, 4);
    {   // o[r][:] = sum_pos p[r][pos] * v_{anc(r, pos)}
        float acc[RT][PER];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int j = 0; j < PER; ++j) acc[r][j] = 0.f;
        u32x4 vown[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) vown[r] = ld16(v_s + (r < R ? r : R - 1) * DH + dc * PER);
        auto pv_chunk = [&](int pos0, const u32x4 (&vk)[PU][RT]) {
            if constexpr (sizeof(T) == 2 && PU % 2 == 0) {
                // bf16: two positions at a time.  The two probabilities are packed as bf16 (as the training attention kernels hold P), the
                // two value fragments are interleaved element by element (v_perm_b32) and every output column takes one
                // v_dot2_f32_bf16: 17 instructions per 16 multiply-adds where unpack + FMA took 32.
#pragma unroll
                for (int u = 0; u < PU; u += 2) {
                    const int pa = pos0 + u * NSLOT + slot, pb = pa + NSLOT;
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        const int rr = r < R ? r : R - 1;
                        const float wa = pa <= t ? sc[rr * maxpos + min(pa, t)] : 0.f, wb = pb <= t ? sc[rr * maxpos + min(pb, t)] : 0.f;
                        const uint32_t pp = pack2_bf16(wa, wb);
                        const u32x4 va = pa >= t ? vown[r] : vk[u][r], vb = pb >= t ? vown[r] : vk[u + 1][r];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const uint32_t lo = __builtin_amdgcn_perm(vb[w], va[w], 0x05040100u);      // (va element 2w, vb element 2w)
                            const uint32_t hi = __builtin_amdgcn_perm(vb[w], va[w], 0x07060302u);      // (va element 2w+1, vb element 2w+1)
                            acc[r][2 * w] = dot2_bf16(pp, lo, acc[r][2 * w]);
                            acc[r][2 * w + 1] = dot2_bf16(pp, hi, acc[r][2 * w + 1]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int pos = pos0 + u * NSLOT + slot;
                    const int pc = min(pos, t);
                    const bool live = pos <= t;
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        const int rr = r < R ? r : R - 1;
                        const float p = live ? sc[rr * maxpos + pc] : 0.f;
                        const Unp<T> vf(pos >= t ? vown[r] : vk[u][r]);
#pragma unroll
                        for (int j = 0; j < PER; ++j) acc[r][j] += p * vf.f[j];
                    }
                }
            }
        };
        pv_chunk(0, vv);
        for (int pos0 = CH0; pos0 <= t; pos0 += CH0) {
            u32x4 vk[PU][RT];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int pc = min(pos0 + u * NSLOT + slot, t - 1);
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const int rr = r < R ? r : R - 1;
                    vk[u][r] = ld16(cache + ((size_t)pc * N + anc_s[rr * maxpos + pc]) * 2 * D + D + h * DH + dc * PER);
                }
            }
            pv_chunk(pos0, vk);
        }
        // sum over the NSLOT position slots: inside a 16-lane row by DPP (two slots per row when DC = 8), the 16 rows of the workgroup
        // through LDS
#pragma unroll
        for (int r = 0; r < RT; ++r) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float v = acc[r][j];
                if (DC == 8) v += dpp_f<0x128>(v);                             // row_ror:8 = the other slot of this row
                if ((lane & 15) < DC) red[((size_t)(tid >> 4) * RT + r) * DH + dc * PER + j] = v;
            }
        }
    }
    lds_barrier();
    DSTAMP(0, 5);
    for (int i = tid; i < R * DH; i += NTHREADS) {
        const int r = i / DH, d = i % DH;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += red[((size_t)g * RT + r) * DH + d];
        o_s[(size_t)r * (DH + PER) + d] = from_f32<T>(v);
    }
    lds_barrier();
    DSTAMP(0, 6);
    if constexpr (PRE_O) share_regs<T, NTO, DH / KS>(o_s, DH + PER, wo, reinterpret_cast<T*>(a.part_out) + ((size_t)h * N + n0) * D, D, R);
    else share_out<T>(o_s, DH + PER, DH, reinterpret_cast<const T*>(a.w_o), D / KS, h * (DH / KS), reinterpret_cast<T*>(a.part_out) + ((size_t)h * N + n0) * D, D, R);
    DSTAMP(0, 7);
}

// ------------------------------------------------------------------------------------------------ C: encoder-attention block of one head
struct CrossArgs {
    Pro pro; int B, beam, N, D, heads, Ts, Tsp, max_len; float scale;
    const void* w_q; const float* b_q; const void* w_o; const void* kv_enc; const void* vt_enc; const int* klen; const int* steps; void* part_out;
};
// DD as in dec_self_kernel.  TP = Tsp / 128 when the sentence's encoder keys and values of this head fit in registers beside the weights
// (Tsp <= 256: 2 TP position tiles of keys and Tsp / KS k-steps of one value column tile per wave): they are requested at the top of the
// launch with the weights, so nothing is waited for after the LayerNorm.  TP = 0: requested where they are used.
template <typename T, int DD, int TP>
__global__ __launch_bounds__(NTHREADS) void dec_cross_kernel(CrossArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS, KST = DH / KS;
    constexpr bool PRE = DD > 0, PKV = TP > 0;
    constexpr int KSTD = PRE ? DD / KS : 1, NTO = PRE ? DD / 64 : 1, TPT = PKV ? 2 * TP : 1, NV = PKV ? 128 * TP / KS : 1;
    const int h = blockIdx.x, s = blockIdx.y, R = a.beam, n0 = s * R, D = PRE ? DD : a.D, N = a.N, Ts = a.Ts, Tsp = PKV ? 128 * TP : a.Tsp;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int t_raw = a.steps[s];
    DSTAMP(1, 0);
    u32x4 wq[1][KSTD], wo[NTO][KST], kf[TPT][KST], vf[1][NV];
    const CrossLds L = cross_lds(R, D, Tsp, (int)sizeof(T));
    float* red0 = reinterpret_cast<float*>(smem);
    T* a_ln = reinterpret_cast<T*>(smem + L.a_ln);
    T* q_s = reinterpret_cast<T*>(smem + L.q);
    float* sc = reinterpret_cast<float*>(smem + L.sc);
    T* p_s = reinterpret_cast<T*>(smem + L.p);
    T* o_s = reinterpret_cast<T*>(smem + L.o);
    const int lda = D + PER, ldq = DH + PER, ldp = Tsp + PER;
    const int arow = min(lane & 15, R - 1);
    // this sentence's keys and values of this head, fragment-major (s2t_decode_prepare_enc):
    //   keys [Tsp / 16 position tiles][KST steps][64 lanes][16 B], values (transposed) [4 column tiles][Tsp / KS steps][64][16 B]
    const T* Kp = reinterpret_cast<const T*>(a.kv_enc) + ((size_t)s * a.heads + h) * (size_t)Tsp * DH;
    const T* Vp = reinterpret_cast<const T*>(a.vt_enc) + ((size_t)s * a.heads + h) * (size_t)Tsp * DH;

    const bool go = dec_prologue<T, S2T_DEC_NB_CROSS>(a.pro, h == 0, N, D, n0, R, a_ln, lda, red0, [&]() __attribute__((always_inline)) {
        if (t_raw > a.max_len) return false;
        if constexpr (PRE) load_rows_w<T, 1, KSTD>(wq, reinterpret_cast<const T*>(a.w_q), h * 4 + w, 0);
        if constexpr (PKV) {
            load_rows_w<T, TPT, KST>(kf, Kp, w, 4);                 // position tiles w, w + 4, ...
            load_rows_w<T, 1, NV>(vf, Vp, w, 0);                    // value column tile w
        }
        if constexpr (PRE) load_share_w<T, NTO, KST>(wo, reinterpret_cast<const T*>(a.w_o), D / KS, h * KST);
        return true;
    });
    if (!go) return;
    DSTAMP(1, 1);
    {   // q of this head: wave w owns 16 of its 64 columns
        const T* ap = a_ln + (size_t)arow * lda + PER * (lane >> 4);
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (PRE) mma_regs<T, 1, KSTD>(ap, wq, acc);
        else mma_rows<T, 1, 8>(ap, reinterpret_cast<const T*>(a.w_q), D / KS, h * 4 + w, 0, D, acc);
        const int col = 16 * w + (lane & 15);
        const float bq = a.b_q[h * DH + col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * (lane >> 4) + i;
            if (r < R) q_s[(size_t)r * ldq + col] = from_f32<T>((acc[0][i] + bq) * a.scale);
        }
    }
    lds_barrier();
    DSTAMP(1, 2);
    {   // scores over the sentence's encoder rows: 16 positions per MFMA tile, K = 64; padding rows -> -inf (multihead_attention.py:318-327)
        const int klen = a.klen ? min(a.klen[s], Ts) : Ts;
        const T* ap = q_s + (size_t)arow * ldq + PER * (lane >> 4);
        u32x4 af[KST];
#pragma unroll
        for (int c = 0; c < KST; ++c) af[c] = ld16(ap + c * KS);
        auto put = [&](int pt, const f32x4& acc) {
            const int pos = pt * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * (lane >> 4) + i;
                if (r < R) sc[(size_t)r * Tsp + pos] = pos < klen ? acc[i] : -INFINITY;
            }
        };
        if constexpr (PKV) {
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < KST; ++c) acc = mma16<T>(af[c], kf[i][c], acc);
                put(w + 4 * i, acc);
            }
        } else {
            constexpr int TG = 8 / KST;                                // position tiles requested together (8 fragments in flight)
            const int last = Tsp / 16 - 4 + w;                         // this wave's last tile (Tsp is a multiple of 128)
            for (int pt0 = w; pt0 <= last; pt0 += 4 * TG) {
                u32x4 b[TG][KST];
                int ptg[TG];
#pragma unroll
                for (int g = 0; g < TG; ++g) {
                    ptg[g] = min(pt0 + 4 * g, last);                   // past the end: the last tile again (same values stored twice)
#pragma unroll
                    for (int c = 0; c < KST; ++c) b[g][c] = ld16(wfrag<T>(Kp, KST, ptg[g], c));
                }
#pragma unroll
                for (int g = 0; g < TG; ++g) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < KST; ++c) acc = mma16<T>(af[c], b[g][c], acc);
                    put(ptg[g], acc);
                }
            }
        }
    }
    lds_barrier();
    DSTAMP(1, 3);
    softmax_rows<sizeof(T) == 2>(sc, Tsp, R, Tsp, [&](int r, int p, float v) { p_s[(size_t)r * ldp + p] = from_f32<T>(v); });
    lds_barrier();
    DSTAMP(1, 4);
    {   // o = P V: wave w owns 16 of the 64 value columns; V is read through its transposed, fragment-major copy
        const T* ap = p_s + (size_t)arow * ldp + PER * (lane >> 4);
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (PKV) mma_regs<T, 1, NV>(ap, vf, acc);
        else mma_rows<T, 1, 128 / KS>(ap, Vp, Tsp / KS, w, 0, Tsp, acc);
        const int col = 16 * w + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * (lane >> 4) + i;
            if (r < R) o_s[(size_t)r * ldq + col] = from_f32<T>(acc[0][i]);
        }
    }
    lds_barrier();
    DSTAMP(1, 5);
    if constexpr (PRE) share_regs<T, NTO, KST>(o_s, ldq, wo, reinterpret_cast<T*>(a.part_out) + ((size_t)h * N + n0) * D, D, R);
    else share_out<T>(o_s, ldq, DH, reinterpret_cast<const T*>(a.w_o), D / KS, h * KST, reinterpret_cast<T*>(a.part_out) + ((size_t)h * N + n0) * D, D, R);
    DSTAMP(1, 6);
}

// ------------------------------------------------------------------------------------------------ F: one slice of the feed-forward block
struct FfnArgs {
    Pro pro; int B, beam, N, D, ffn, hs, gelu, max_len;
    const void* w_fc1; const float* b_fc1; const void* w_fc2; const int* steps; void* part_out;
};
template <typename T, int TPW, int DD>          // TPW = hs / 64 column tiles of fc1 per wave; DD as in dec_self_kernel
__global__ __launch_bounds__(NTHREADS) void dec_ffn_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS, HS = 64 * TPW;
    constexpr bool PRE = DD > 0, PRE2 = PRE && TPW <= 2;         // fc2's fragments too while they fit (TPW 4: 512 VGPRs with fc1's)
    constexpr int KSTD = PRE ? DD / KS : 1, NTO = PRE2 ? DD / 64 : 1, KST2 = PRE2 ? HS / KS : 1;
    const int j = blockIdx.x, s = blockIdx.y, R = a.beam, n0 = s * R, D = PRE ? DD : a.D, N = a.N, hs = HS;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int t_raw = a.steps[s];
    DSTAMP(2, 0);
    u32x4 w1[TPW][KSTD], w2[NTO][KST2];
    const FfnLds L = ffn_lds(R, D, hs, (int)sizeof(T));
    float* red0 = reinterpret_cast<float*>(smem);
    T* a_ln = reinterpret_cast<T*>(smem + L.a_ln);
    T* h_s = reinterpret_cast<T*>(smem + L.h);
    const int lda = D + PER, ldh = hs + PER;
    const int arow = min(lane & 15, R - 1);

    const bool go = dec_prologue<T, S2T_DEC_NB_FFN>(a.pro, j == 0, N, D, n0, R, a_ln, lda, red0, [&]() __attribute__((always_inline)) {
        if (t_raw > a.max_len) return false;
        if constexpr (PRE) load_rows_w<T, TPW, KSTD>(w1, reinterpret_cast<const T*>(a.w_fc1), j * (HS / 16) + w * TPW, 1);
        if constexpr (PRE2) load_share_w<T, NTO, KST2>(w2, reinterpret_cast<const T*>(a.w_fc2), a.ffn / KS, j * (HS / KS));
        return true;
    });
    if (!go) return;
    DSTAMP(2, 1);
    {   // hidden units j*hs .. +hs: wave w owns TPW column tiles (transformer_layer.py:367-368: fc1, activation)
        const T* ap = a_ln + (size_t)arow * lda + PER * (lane >> 4);
        f32x4 acc[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) mma_regs<T, TPW, KSTD>(ap, w1, acc);
        else mma_rows<T, TPW, (TPW == 4 ? 4 : 8)>(ap, reinterpret_cast<const T*>(a.w_fc1), D / KS, j * (HS / 16) + w * TPW, 1, D, acc);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int col = (w * TPW + i) * 16 + (lane & 15);
            const float bb = a.b_fc1[j * hs + col];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * (lane >> 4) + q;
                float v = acc[i][q] + bb;
                v = a.gelu ? gelu_f(v) : fmaxf(v, 0.f);
                if (r < R) h_s[(size_t)r * ldh + col] = from_f32<T>(v);
            }
        }
    }
    lds_barrier();
    DSTAMP(2, 2);
    if constexpr (PRE2) share_regs<T, NTO, KST2>(h_s, ldh, w2, reinterpret_cast<T*>(a.part_out) + ((size_t)j * N + n0) * D, D, R);
    else share_out<T>(h_s, ldh, hs, reinterpret_cast<const T*>(a.w_fc2), a.ffn / KS, j * (HS / KS), reinterpret_cast<T*>(a.part_out) + ((size_t)j * N + n0) * D, D, R);
    DSTAMP(2, 3);
}

// ------------------------------------------------------------------------------------------------ final LayerNorm (rows to global)
struct FinalArgs { Pro pro; int beam, N, D, max_len; const int* steps; void* xn; };
template <typename T>
__global__ __launch_bounds__(NTHREADS) void dec_final_kernel(FinalArgs a) {
    __shared__ float red0[32];
    const int s = blockIdx.x, n0 = s * a.beam;
    const int t_raw = a.steps[s];
    DSTAMP(6, 0);
    dec_prologue<T, 16>(a.pro, true, a.N, a.D, n0, a.beam, reinterpret_cast<T*>(a.xn) + (size_t)n0 * a.D, a.D, red0,
                        [&]() { return t_raw <= a.max_len; });
    DSTAMP(6, 1);
}

// ------------------------------------------------------------------------------------------------ output projection, all N rows
// logits[n][v] = xn[n] . W[v]: a workgroup owns 64 vocabulary columns (one 16-column tile per wave) and every row tile (MT of them), so
// the weight matrix is read exactly once per step.  K goes in chunks of KCH = 16 k-steps (all of D = 512 in bf16): the wave's weight
// fragments of the chunk are requested first, then the chunk's columns of all N rows are staged in LDS (up to 24 sixteen-byte pieces per
// thread in flight: one round trip for 80 rows), then KCH x MT MFMAs.
struct LogitArgs { int N, D, V, ldv; const void* xn; const void* w; float* logits; };
constexpr int LOGIT_KCH = 16;
template <typename T, int MT>
__global__ __launch_bounds__(NTHREADS) void dec_logits_kernel(LogitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS, KCH = LOGIT_KCH, KC = KCH * KS, LDX = KC + PER, PPR = KC / PER, UB = 24;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, D = a.D, N = a.N;
    const int ct = blockIdx.x * 4 + w;
    T* x_s = reinterpret_cast<T*>(smem);                      // [N][LDX]
    const T* X = reinterpret_cast<const T*>(a.xn);
    const T* Wp = reinterpret_cast<const T*>(a.w);               // fragment-major, ceil(V / 16) tiles (rows past V are zero)
    const int ctc = min(ct, (a.V + 15) / 16 - 1);
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = min(mt * 16 + (lane & 15), N - 1) * LDX + PER * (lane >> 4);
    DSTAMP(5, 0);
    const int kc_w = min(KC, D), ppr = kc_w / PER;              // D = 256 in bf16: one chunk of 8 k-steps
    const int psh = 31 - __builtin_clz(ppr);                    // pieces per row: a power of two (32 or 64)
    for (int kc0 = 0; kc0 < D; kc0 += KC) {
        u32x4 b[KCH];
#pragma unroll
        for (int c = 0; c < KCH; ++c) b[c] = ld16(wfrag<T>(Wp, D / KS, ctc, min(kc0 / KS + c, D / KS - 1)));
        if (kc0 > 0) lds_barrier();                              // the previous chunk's fragment reads are done
        for (int i0 = 0; i0 < N * ppr; i0 += UB * NTHREADS) {
            u32x4 piece[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = min(i0 + u * NTHREADS + tid, N * ppr - 1);
                piece[u] = ld16(X + (size_t)(i >> psh) * D + kc0 + (i & (ppr - 1)) * PER);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * NTHREADS + tid;
                if (i < N * ppr) *reinterpret_cast<u32x4*>(x_s + (size_t)(i >> psh) * LDX + (i & (ppr - 1)) * PER) = piece[u];
            }
        }
        lds_barrier();
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            if (c * KS < kc_w) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mma16<T>(ld16(x_s + arow[mt] + c * KS), b[c], acc[mt]);
            }
        }
    }
    DSTAMP(5, 1);
    const int col = ct * 16 + (lane & 15);
    if (col < a.V) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = mt * 16 + 4 * (lane >> 4) + i;
                if (r < N) a.logits[(size_t)r * a.ldv + col] = acc[mt][i];
            }
        }
    }
    DSTAMP(5, 2);
}

// ------------------------------------------------------------------------------------------------ search, part 1: one row
// log-softmax of row n (same arithmetic and summation order as log_softmax_kernel of loss_embed.hip), the score rules of
// sequence_generator.py:263-282 (NaN -> -inf; pad -inf; unk penalty; only EOS at max_len; no EOS before min_len), + the cumulative score
// of the hypothesis (search.py:62-69), then the K2 = 2*beam best (value descending, column ascending on ties).
struct RowArgs {
    int beam, N, V, ldv, K2, pad, unk, eos, max_len, min_len, step0_all; float it, unk_penalty;
    const float* logits; const int* steps; const float* cum_hist; const float* init_scores; float* cand_val; int* cand_idx;
};
__device__ __forceinline__ bool cand_after(float v, int i, float lv, int li) { return v < lv || (v == lv && i > li); }
__device__ __forceinline__ bool cand_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }
template <int VPT>
__global__ __launch_bounds__(NTHREADS) void dec_row_kernel(RowArgs a) {
    __shared__ float sh[16];
    __shared__ float wv[4 * 32];
    __shared__ int wi[4 * 32];
    const int n = blockIdx.x, s = n / a.beam, tid = threadIdx.x, V = a.V;
    const int t = a.steps[s];
    if (t > a.max_len) return;
    DSTAMP(3, 0);
    const float* x = a.logits + (size_t)n * a.ldv;
    float val[VPT];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = tid + i * NTHREADS;
        val[i] = x[min(v, V - 1)] * a.it;
        if (v >= V) val[i] = -INFINITY;
        m = fmaxf(m, val[i]);
    }
    m = block_max(m, sh);
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = tid + i * NTHREADS;
        if (v < V) z += expf(val[i] - m);
    }
    z = block_sum(z, sh);
    const float lse = m + logf(z);
    DSTAMP(3, 1);
    const bool live = t > 0 || a.step0_all || (n % a.beam) == 0;          // step 0: every slot holds the same <bos> (search.py:64-66)
    const float base = t > 0 ? a.cum_hist[(size_t)t * a.N + n] : (a.init_scores ? a.init_scores[n] : 0.f);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = tid + i * NTHREADS;
        float lp = val[i] - lse;
        if (lp != lp) lp = -INFINITY;
        if (v == a.pad) lp = -INFINITY;
        if (v == a.unk) lp -= a.unk_penalty;
        if (t >= a.max_len) { if (v != a.eos) lp = -INFINITY; }
        else if (t < a.min_len && v == a.eos) lp = -INFINITY;
        val[i] = (live && v < V) ? lp + base : -INFINITY;
    }
    // K2 best of the row.  Threshold first: every lane's largest value; the K2-th largest lane maximum of a wave is a value that at
    // least K2 columns reach, so the largest such value over the four waves (`thr`) is a lower bound of the row's K2-th best: only
    // columns >= thr can be among the K2 best.  They are few (about K2) unless the row is full of ties or of -inf: those go to a list
    // in LDS and wave 0 orders it (value descending, column ascending).  A row whose list would overflow takes the exact slow way.
    const int w = tid >> 6, lane = tid & 63, K2 = a.K2;
    DSTAMP(3, 2);
    {
        float lm = -INFINITY;
#pragma unroll
        for (int i = 0; i < VPT; ++i) lm = fmaxf(lm, val[i]);
        // the K2-th largest of the wave's 64 lane maxima: every lane counts the lanes that come before it in the order (value descending,
        // lane ascending) -- 64 broadcasts from scalar registers, no cross-lane dependency chain (K2 rounds of arg-max took 11,000 cycles
        // here on a wave that runs alone on its SIMD; this takes ~2,500) -- and the lane with K2 - 1 before it holds the value
        int before = 0;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const float o = rl_f(lm, j);
            before += (o > lm || (o == lm && j < lane)) ? 1 : 0;
        }
        const unsigned long long hit = __ballot(before == K2 - 1);
        const float kth = rl_f(lm, (int)__builtin_ctzll(hit | (1ull << 63)));
        if (lane == 0) wv[w] = kth;
        if (tid == 0) wi[127] = 0;                                 // list length
    }
    DSTAMP(3, 3);
    __syncthreads();
    const float thr = fmaxf(fmaxf(wv[0], wv[1]), fmaxf(wv[2], wv[3]));
    __syncthreads();
    DSTAMP(3, 4);
    constexpr int CAP = 120;                                       // list entries (wv / wi [4 .. 123])
    if (thr > -INFINITY) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = tid + i * NTHREADS;
            if (v < V && val[i] >= thr) {
                const int at = atomicAdd(&wi[127], 1);
                if (at < CAP) { wv[4 + at] = val[i]; wi[4 + at] = v; }
            }
        }
    }
    __syncthreads();
    const int cnt = wi[127];
    DSTAMP(3, 5);
    if (thr > -INFINITY && cnt <= CAP) {
        if (w == 0 && cnt <= 64) {
            // one candidate per lane: its place in the order (value descending, column ascending) = how many of the others come before
            // it, counted through `cnt` scalar broadcasts (the list is about K2 + a few long; K2 rounds of arg-max over it took 5,400 cycles)
            float v = -INFINITY; int ix = 0x7fffffff;
            if (lane < cnt) { v = wv[4 + lane]; ix = wi[4 + lane]; }
            int before = 0;
            for (int j = 0; j < cnt; ++j) {
                const float ov = rl_f(v, j); const int oi = __builtin_amdgcn_readlane(ix, j);
                before += (ov > v || (ov == v && oi < ix)) ? 1 : 0;
            }
            if (lane < cnt && before < K2) { a.cand_val[(size_t)n * K2 + before] = v; a.cand_idx[(size_t)n * K2 + before] = ix; }
        } else if (w == 0) {
            float c0v = -INFINITY, c1v = -INFINITY; int c0i = 0x7fffffff, c1i = 0x7fffffff;
            if (lane < cnt) { c0v = wv[4 + lane]; c0i = wi[4 + lane]; }
            if (lane + 64 < cnt) { c1v = wv[4 + lane + 64]; c1i = wi[4 + lane + 64]; }
            float lv = INFINITY; int li = -1;
            for (int k = 0; k < K2; ++k) {
                float cv = -INFINITY; int ci = 0x7fffffff;
                if (c0i != 0x7fffffff && cand_after(c0v, c0i, lv, li)) { cv = c0v; ci = c0i; }
                if (c1i != 0x7fffffff && cand_after(c1v, c1i, lv, li) && cand_better(c1v, c1i, cv, ci)) { cv = c1v; ci = c1i; }
                float mv; int mi;
                wave_argmax(cv, ci, mv, mi);
                if (lane == 0) { a.cand_val[(size_t)n * K2 + k] = mv; a.cand_idx[(size_t)n * K2 + k] = mi; }
                lv = mv; li = mi;
            }
        }
    } else {
        // exact slow way (ties / -inf rows: the forced-EOS step, the idle slots of step 0): every wave takes the K2 best of ITS columns
        // by K2 rounds of arg-max (only the lane that owned the winner looks for its next candidate), wave 0 merges the four lists
        __syncthreads();
        auto scan = [&](float lv, int li, float& bv, int& bi) {
            bv = -INFINITY; bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int v = tid + i * NTHREADS;
                if (v < V && cand_after(val[i], v, lv, li) && cand_better(val[i], v, bv, bi)) { bv = val[i]; bi = v; }
            }
        };
        float bv; int bi;
        scan(INFINITY, -1, bv, bi);
        for (int k = 0; k < K2; ++k) {
            float mv; int mi;
            wave_argmax(bv, bi, mv, mi);
            if (lane == 0) { wv[w * 32 + k] = mv; wi[w * 32 + k] = mi; }
            if (bi == mi && mi != 0x7fffffff) scan(mv, mi, bv, bi);
        }
        __syncthreads();
        if (w == 0) {
            float c0v = -INFINITY, c1v = -INFINITY; int c0i = 0x7fffffff, c1i = 0x7fffffff;
            if (lane < 4 * K2) { c0v = wv[(lane / K2) * 32 + lane % K2]; c0i = wi[(lane / K2) * 32 + lane % K2]; }
            if (lane + 64 < 4 * K2) { c1v = wv[((lane + 64) / K2) * 32 + (lane + 64) % K2]; c1i = wi[((lane + 64) / K2) * 32 + (lane + 64) % K2]; }
            float lv = INFINITY; int li = -1;
            for (int k = 0; k < K2; ++k) {
                float cv = -INFINITY; int ci = 0x7fffffff;
                if (c0i != 0x7fffffff && cand_after(c0v, c0i, lv, li)) { cv = c0v; ci = c0i; }
                if (c1i != 0x7fffffff && cand_after(c1v, c1i, lv, li) && cand_better(c1v, c1i, cv, ci)) { cv = c1v; ci = c1i; }
                float mv; int mi;
                wave_argmax(cv, ci, mv, mi);
                if (mi == 0x7fffffff) { mv = -INFINITY; mi = 0; }       // fewer than K2 columns: never selected (k < number of columns)
                if (lane == 0) { a.cand_val[(size_t)n * K2 + k] = mv; a.cand_idx[(size_t)n * K2 + k] = mi; }
                lv = mv; li = mi;
            }
        }
    }
    DSTAMP(3, 6);
}

// ------------------------------------------------------------------------------------------------ search, part 2: one sentence
// merge of the rows' candidate lists = BeamSearch.step's top-k over beam x V (search.py:70-83); EOS finalisation bookkeeping
// (sequence_generator.py:383-415, finalize_hypos :502-600: which candidates end, in which order, when the sentence is finished);
// the next beam = the first `beam` candidates in rank order that are not EOS, black-listing slots only an EOS could fill (:417-446);
// the record of this selection (token, parent, cumulative score); the ancestor table that replaces reorder_incremental_state; the
// embedding of the chosen tokens = the decoder input of the next step (transformer.py:720-737).
// The bookkeeping is one wave with lane = candidate rank: ballots give every candidate its place among the finalised / the selected.
struct SentArgs {
    int beam, N, D, V, K2, pad, eos, max_len, maxpos, step0_all; float embed_scale;
    const float* cand_val; const int* cand_idx; int* steps; int* anc; int* tok_hist; int* par_hist; float* cum_hist; int* blacklist;
    int* nfin; int* finished; int* fin_step; int* fin_row; float* fin_score; const void* embed; const float* pos_table; float* x0;
};
template <typename T>
__global__ __launch_bounds__(NTHREADS) void dec_sent_kernel(SentArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ __attribute__((aligned(16))) float l_val[16 * 32];
    __shared__ __attribute__((aligned(16))) int l_idx[16 * 32];
    __shared__ int pick_par[16], pick_tok[16];
    int* anc_l = reinterpret_cast<int*>(smem);                     // [beam][maxpos]
    const int s = blockIdx.x, tid = threadIdx.x, beam = a.beam, n0 = s * beam, K2 = a.K2, N = a.N;
    const int t = a.steps[s];
    if (t > a.max_len) return;
    DSTAMP(4, 0);
    const bool first = t == 0 && !a.step0_all;
    const long ncols = first ? a.V : (long)beam * a.V;
    const int k = (int)min((long)K2, ncols - 1);                   // search.py:71-75: pad is never selected
    const int rows = first ? 1 : beam;
    // everything this launch reads is requested at once: the rows' candidate lists, the sentence's flags, its ancestor rows
    const int ne = rows * K2;                                      // <= 512 entries: two per thread
    float ev[2]; int ei[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = min(tid + u * NTHREADS, ne - 1);
        ev[u] = a.cand_val[(size_t)n0 * K2 + i]; ei[u] = a.cand_idx[(size_t)n0 * K2 + i];
    }
    int bl = 0, done = 0, nf = 0;
    if (tid < 64) {                                                // wave 0 keeps the sentence's flags in registers
        bl = tid < beam ? a.blacklist[n0 + tid] : 0;
        done = a.finished[s]; nf = a.nfin[s];
    }
    for (int p0 = 0; p0 < t; p0 += NTHREADS) {                     // old ancestor rows of the sentence (read before anything is rewritten)
        int av[16];
        const int pc = min(p0 + tid, t - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) av[r] = a.anc[(size_t)(n0 + min(r, beam - 1)) * a.maxpos + pc];
        if (p0 + tid < t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (r < beam) anc_l[r * a.maxpos + p0 + tid] = av[r];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * NTHREADS;
        if (i < ne) { l_val[i] = ev[u]; l_idx[i] = (i / K2) * a.V + ei[u]; }      // flat index: row * V + column
    }
    __syncthreads();
    DSTAMP(4, 1);
    if (tid < 64) {
        // k-way merge of the `rows` sorted lists: lane j < rows offers the head of list j; the winner of round r becomes candidate r,
        // kept by lane r.  (Measured alternatives for 50 entries against the 5,900 cycles of these k = 10 rounds: every entry counting
        // the entries before it over the lists in LDS 10,700; the same count through scalar broadcasts of one entry per lane 11,300.)
        int head = 0;
        float my_val = -INFINITY; int my_tok = 0, my_row = n0;
        for (int r = 0; r < k; ++r) {
            float v = -INFINITY; int flat = 0x7fffffff;
            if (tid < rows && head < K2) { v = l_val[tid * K2 + head]; flat = l_idx[tid * K2 + head]; }
            float mv; int mf;
            wave_argmax(v, flat, mv, mf);
            if (flat == mf && tid < rows) ++head;
            if (tid == r) { my_val = mv; my_tok = mf % a.V; my_row = n0 + mf / a.V; }
        }
        DSTAMP(4, 2);
        const int nb = min(beam, k);
        const unsigned long long below = (1ull << tid) - 1ull;
        bool eosm = tid < k && my_tok == a.eos && my_val != -INFINITY;
        if (tid < nb) eosm = eosm && !bl;
        // finalised this step: the EOS candidates among the first `beam`, in rank order, while the sentence has room (:502-600)
        const bool top_eos = tid < nb && eosm && !done;
        const unsigned long long M = __ballot(top_eos);
        const int slot_f = nf + __popcll(M & below);
        if (top_eos && slot_f < beam) {
            a.fin_step[s * beam + slot_f] = t; a.fin_row[s * beam + slot_f] = my_row; a.fin_score[s * beam + slot_f] = my_val;
        }
        const int nf_new = min(beam, nf + __popcll(M));
        if (tid == 0) {
            a.nfin[s] = nf_new;
            if (M != 0ull && (nf_new == beam || t == a.max_len)) a.finished[s] = 1;
        }
        // next beam: candidates that are not EOS (nor black-listed) in rank order, then the others in rank order (:417-446)
        if (tid < nb) eosm = eosm || bl;
        const unsigned long long E = __ballot(tid < k && eosm), NE = __ballot(tid < k && !eosm);
        const int place = eosm ? __popcll(NE) + __popcll(E & below) : __popcll(NE & below);
        if (tid < k && place < beam) {
            pick_par[place] = my_row; pick_tok[place] = my_tok;
            a.tok_hist[(size_t)(t + 1) * N + n0 + place] = my_tok;
            a.par_hist[(size_t)(t + 1) * N + n0 + place] = my_row;
            a.cum_hist[(size_t)(t + 1) * N + n0 + place] = my_val;
            a.blacklist[n0 + place] = eosm ? 1 : 0;
        }
    }
    __syncthreads();
    DSTAMP(4, 3);
    if (t < a.max_len) {
        // the next step's decoder input: embedding of the chosen token * scale + position row (transformer.py:720-737), four columns per
        // item, four items per thread requested together (all of beam 5 x D 512 in one round trip)
        const int dq = a.D / 4, items = beam * dq;
        for (int i0 = 0; i0 < items; i0 += 4 * NTHREADS) {
            f32x4 ev4[4], pv4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * NTHREADS + tid, items - 1), j = i / dq, d = (i - j * dq) * 4;
                const int tok = pick_tok[j];
                ev4[u] = Raw4<T>::cvt(Raw4<T>::ld(reinterpret_cast<const T*>(a.embed) + (size_t)tok * a.D + d));
                pv4[u] = *reinterpret_cast<const f32x4*>(a.pos_table + (size_t)(tok == a.pad ? a.pad : a.pad + 2 + t) * a.D + d);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * NTHREADS + tid, j = i / dq, d = (i - j * dq) * 4;
                if (i < items) {
                    f32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = a.embed_scale * ev4[u][q] + pv4[u][q];
                    *reinterpret_cast<f32x4*>(a.x0 + (size_t)(n0 + j) * a.D + d) = o;
                }
            }
        }
        for (int j = 0; j < beam; ++j) {
            const int par = pick_par[j];
            for (int p = tid; p < t; p += NTHREADS) a.anc[(size_t)(n0 + j) * a.maxpos + p] = anc_l[(par - n0) * a.maxpos + p];
            if (tid == 0) a.anc[(size_t)(n0 + j) * a.maxpos + t] = par;
        }
    }
    DSTAMP(4, 4);
    if (tid == 0) a.steps[s] = t + 1;
}

struct BeginArgs {
    int beam, N, D, pad, bos; float embed_scale;
    int* steps; int* tok_hist; int* blacklist; int* nfin; int* finished; const void* embed; const float* pos_table; float* x0;
};
template <typename T>
__global__ __launch_bounds__(NTHREADS) void dec_begin_kernel(BeginArgs a) {
    const int s = blockIdx.x, tid = threadIdx.x, n0 = s * a.beam;
    if (tid == 0) { a.steps[s] = 0; a.nfin[s] = 0; a.finished[s] = 0; }
    if (tid < a.beam) { a.blacklist[n0 + tid] = 0; a.tok_hist[n0 + tid] = a.bos; }
    const T* e = reinterpret_cast<const T*>(a.embed) + (size_t)a.bos * a.D;
    const float* pe = a.pos_table + (size_t)(a.bos == a.pad ? a.pad : a.pad + 1) * a.D;
    for (int j = 0; j < a.beam; ++j)
        for (int d = tid; d < a.D; d += NTHREADS) a.x0[(size_t)(n0 + j) * a.D + d] = a.embed_scale * to_f32(e[d]) + pe[d];
}

// fragment-major copies of a layer's encoder-side K and V^T (from kv [Ts][B][2D]), zero beyond Ts:
//   kp [B][heads][Tsp / 16 position tiles][64 / KS steps][64 lanes][16 B]: lane l of (pt, step) = K[16 pt + (l & 15)][KS step + PER (l >> 4) ..]
//   vp [B][heads][4 column tiles][Tsp / KS steps][64 lanes][16 B]:        lane l of (ct, step) = V[KS step + PER (l >> 4) ..][16 ct + (l & 15)]
template <typename T>
__global__ __launch_bounds__(NTHREADS) void dec_enc_pack_kernel(const T* __restrict__ kv, T* __restrict__ kp, T* __restrict__ vp, int Ts, int Tsp, int B,
                                                                int D, int heads) {
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS;
    const size_t pieces = (size_t)B * heads * Tsp * DH / PER;        // 16-byte pieces of each output
    for (size_t p = (size_t)blockIdx.x * NTHREADS + threadIdx.x; p < 2 * pieces; p += (size_t)gridDim.x * NTHREADS) {
        const bool isv = p >= pieces;
        const size_t q = isv ? p - pieces : p;
        const int lane = (int)(q & 63);
        size_t r = q >> 6;
        T out[PER];
        if (!isv) {
            const int step = (int)(r % (DH / KS)); r /= (DH / KS);
            const int pt = (int)(r % (Tsp / 16)); r /= (Tsp / 16);
            const int h = (int)(r % heads), b = (int)(r / heads);
            const int pos = pt * 16 + (lane & 15), d0 = step * KS + PER * (lane >> 4);
#pragma unroll
            for (int j = 0; j < PER; ++j) out[j] = pos < Ts ? kv[((size_t)pos * B + b) * 2 * D + h * DH + d0 + j] : from_f32<T>(0.f);
            *reinterpret_cast<u32x4*>(kp + q * PER) = *reinterpret_cast<const u32x4*>(out);
        } else {
            const int step = (int)(r % (Tsp / KS)); r /= (Tsp / KS);
            const int ct = (int)(r % 4); r /= 4;
            const int h = (int)(r % heads), b = (int)(r / heads);
            const int d = ct * 16 + (lane & 15), pos0 = step * KS + PER * (lane >> 4);
#pragma unroll
            for (int j = 0; j < PER; ++j) out[j] = pos0 + j < Ts ? kv[((size_t)(pos0 + j) * B + b) * 2 * D + D + h * DH + d] : from_f32<T>(0.f);
            *reinterpret_cast<u32x4*>(vp + q * PER) = *reinterpret_cast<const u32x4*>(out);
        }
    }
}
// W [N][K] (row stride ldw) -> fragment-major [ceil(N / 16)][K / KS][64][16 B] (wfrag), rows past N zero
template <typename T>
__global__ __launch_bounds__(NTHREADS) void dec_pack_kernel(const T* __restrict__ W, int ldw, int N, int K, T* __restrict__ Wp) {
    constexpr int PER = FR<T>::PER, KS = FR<T>::KS;
    const int ksteps = K / KS;
    const size_t pieces = (size_t)((N + 15) / 16) * ksteps * 64;
    for (size_t p = (size_t)blockIdx.x * NTHREADS + threadIdx.x; p < pieces; p += (size_t)gridDim.x * NTHREADS) {
        const int lane = (int)(p & 63);
        const size_t r = p >> 6;
        const int step = (int)(r % ksteps), tile = (int)(r / ksteps);
        const int n = tile * 16 + (lane & 15), k = step * KS + PER * (lane >> 4);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (n < N) v = ld16(W + (size_t)n * ldw + k);
        *reinterpret_cast<u32x4*>(Wp + p * PER) = v;
    }
}

bool desc_ok(const S2TDecodeDesc* d) {
    if (!d || !d->layer) return false;
    if (d->dtype != S2T_F32 && d->dtype != S2T_BF16) return false;
    if (d->B < 1 || d->beam < 1 || d->beam > 16 || d->B * d->beam > 128) return false;
    if ((d->D != 256 && d->D != 512 && d->D != 1024) || d->D != d->heads * DH) return false;
    if (d->ffn_slices < 1 || d->ffn % d->ffn_slices) return false;
    const int hs = d->ffn / d->ffn_slices;
    if (hs != 64 && hs != 128 && hs != 256) return false;
    if (d->max_len < 1 || d->max_len + 1 > 1024 || d->Ts < 1 || d->Tsp % 128 || d->Tsp < d->Ts) return false;
    if (d->V < 2 * d->beam + 1 || d->V > 32768 || d->ldv < d->V || d->layers < 1) return false;
    return true;
}
inline int self_rt(int beam) { return beam <= 8 ? beam : 16; }
struct LdsNeed { size_t self, cross, ffn, sent, logits; };
LdsNeed lds_need(const S2TDecodeDesc* d) {
    const int es = d->dtype == S2T_BF16 ? 2 : 4;
    LdsNeed n;
    n.self = self_lds(self_rt(d->beam), d->D, d->max_len + 1, es).total;
    n.cross = cross_lds(d->beam, d->D, d->Tsp, es).total;
    n.ffn = ffn_lds(d->beam, d->D, d->ffn / d->ffn_slices, es).total;
    n.sent = up16((size_t)d->beam * (d->max_len + 1) * 4);
    n.logits = (size_t)d->B * d->beam * (LOGIT_KCH * (64 / es) + 16 / es) * es;
    return n;
}
constexpr size_t LDS_CAP = 152 * 1024;                // leaves 8 KiB for the static arrays (dec_sent_kernel: 4.3 KiB)

template <typename K> hipError_t allow_lds(K kern) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP);
}
#define DEC_ALLOW(k) do { const hipError_t e_ = allow_lds(k); if (e_ != hipSuccess) return e_; } while (0)
template <typename T> constexpr int pre_dd(int D) { return (sizeof(T) == 2 && D <= 512) ? D : 0; }
template <typename T, int DD> hipError_t configure_dd() {
    DEC_ALLOW((dec_self_kernel<T, 1, DD>)); DEC_ALLOW((dec_self_kernel<T, 2, DD>)); DEC_ALLOW((dec_self_kernel<T, 3, DD>));
    DEC_ALLOW((dec_self_kernel<T, 4, DD>)); DEC_ALLOW((dec_self_kernel<T, 5, DD>)); DEC_ALLOW((dec_self_kernel<T, 6, DD>));
    DEC_ALLOW((dec_self_kernel<T, 7, DD>)); DEC_ALLOW((dec_self_kernel<T, 8, DD>)); DEC_ALLOW((dec_self_kernel<T, 16, DD>));
    DEC_ALLOW((dec_cross_kernel<T, DD, 0>)); DEC_ALLOW((dec_cross_kernel<T, DD, 1>)); DEC_ALLOW((dec_cross_kernel<T, DD, 2>));
    DEC_ALLOW((dec_ffn_kernel<T, 1, DD>)); DEC_ALLOW((dec_ffn_kernel<T, 2, DD>)); DEC_ALLOW((dec_ffn_kernel<T, 4, DD>));
    return hipSuccess;
}
template <typename T> hipError_t configure() {
    hipError_t e = configure_dd<T, 0>();
    if (e != hipSuccess) return e;
    if constexpr (sizeof(T) == 2) {
        if ((e = configure_dd<T, 256>()) != hipSuccess) return e;
        if ((e = configure_dd<T, 512>()) != hipSuccess) return e;
    }
    DEC_ALLOW(dec_sent_kernel<T>);
    DEC_ALLOW((dec_logits_kernel<T, 1>)); DEC_ALLOW((dec_logits_kernel<T, 2>)); DEC_ALLOW((dec_logits_kernel<T, 3>)); DEC_ALLOW((dec_logits_kernel<T, 4>));
    DEC_ALLOW((dec_logits_kernel<T, 5>)); DEC_ALLOW((dec_logits_kernel<T, 6>)); DEC_ALLOW((dec_logits_kernel<T, 7>)); DEC_ALLOW((dec_logits_kernel<T, 8>));
    return hipSuccess;
}
// launches of the three layer kernels for one compile-time DD
template <typename T, int DD>
void launch_self(int rt, dim3 grid, size_t lds, hipStream_t st, const SelfArgs& a) {
#define DEC_SELF(RT_) hipLaunchKernelGGL((dec_self_kernel<T, RT_, DD>), grid, dim3(NTHREADS), lds, st, a)
    switch (rt) {
        case 1: DEC_SELF(1); break; case 2: DEC_SELF(2); break; case 3: DEC_SELF(3); break; case 4: DEC_SELF(4); break;
        case 5: DEC_SELF(5); break; case 6: DEC_SELF(6); break; case 7: DEC_SELF(7); break; case 8: DEC_SELF(8); break;
        default: DEC_SELF(16); break;
    }
#undef DEC_SELF
}
template <typename T, int DD>
void launch_cross(int tp, dim3 grid, size_t lds, hipStream_t st, const CrossArgs& a) {
    if (tp == 1) hipLaunchKernelGGL((dec_cross_kernel<T, DD, 1>), grid, dim3(NTHREADS), lds, st, a);
    else if (tp == 2) hipLaunchKernelGGL((dec_cross_kernel<T, DD, 2>), grid, dim3(NTHREADS), lds, st, a);
    else hipLaunchKernelGGL((dec_cross_kernel<T, DD, 0>), grid, dim3(NTHREADS), lds, st, a);
}
template <typename T, int DD>
void launch_ffn(int hs, dim3 grid, size_t lds, hipStream_t st, const FfnArgs& a) {
    if (hs == 64) hipLaunchKernelGGL((dec_ffn_kernel<T, 1, DD>), grid, dim3(NTHREADS), lds, st, a);
    else if (hs == 128) hipLaunchKernelGGL((dec_ffn_kernel<T, 2, DD>), grid, dim3(NTHREADS), lds, st, a);
    else hipLaunchKernelGGL((dec_ffn_kernel<T, 4, DD>), grid, dim3(NTHREADS), lds, st, a);
}
}  // namespace
int g_s2t_opt_decode_stop_after = 0;      // diagnostic (s2t_set_option "decode_stop_after"): > 0 ends a step after that many launches
namespace {
#define DEC_STOP_CHECK() do { if (g_s2t_opt_decode_stop_after > 0 && ++launched >= g_s2t_opt_decode_stop_after) { S2T_LAUNCH_CHECK(); return S2T_OK; } } while (0)
template <typename T>
int step_impl(const S2TDecodeDesc* d, hipStream_t st) {
    int launched = 0;
    const int B = d->B, R = d->beam, N = B * R, D = d->D, H = d->heads, FS = d->ffn_slices, hs = d->ffn / FS, maxpos = d->max_len + 1;
    const LdsNeed need = lds_need(d);
    float* X[2] = {d->x0, d->x1};
    void* P[2] = {d->part0, d->part1};
    const float scale = 1.0f / sqrtf((float)DH);
    const int dd = pre_dd<T>(D);                           // weight fragments held in registers for the whole launch (bf16, D <= 512)
    int k = 0;                                             // launch k reads X[k & 1], the shares in P[(k + 1) & 1]; writes X[(k + 1) & 1], P[k & 1]
    int np = 0;
    const float* bias = nullptr;
    auto pro = [&](const void* g, const void* b) {
        Pro p; p.x_in = X[k & 1]; p.part_in = P[(k + 1) & 1]; p.bias = bias; p.x_out = X[(k + 1) & 1];
        p.g = (const float*)g; p.b = (const float*)b; p.np = np; p.eps = d->ln_eps; return p;
    };
    for (int l = 0; l < d->layers; ++l) {
        const S2TDecodeLayer& y = d->layer[l];
        {
            SelfArgs a; a.pro = pro(y.ln1_g, y.ln1_b); a.B = B; a.beam = R; a.N = N; a.D = D; a.heads = H; a.maxpos = maxpos; a.max_len = d->max_len;
            a.scale = scale; a.w_qkv = y.w_qkv; a.b_qkv = (const float*)y.b_qkv; a.w_o = y.w_o; a.cache = y.kv_cache; a.anc = d->anc;
            a.steps = d->steps; a.part_out = P[k & 1];
            if (dd == 256) launch_self<T, (sizeof(T) == 2 ? 256 : 0)>(self_rt(R), dim3(H, B), need.self, st, a);
            else if (dd == 512) launch_self<T, (sizeof(T) == 2 ? 512 : 0)>(self_rt(R), dim3(H, B), need.self, st, a);
            else launch_self<T, 0>(self_rt(R), dim3(H, B), need.self, st, a);
            ++k; np = H; bias = (const float*)y.b_o;
            DEC_STOP_CHECK();
        }
        {
            CrossArgs a; a.pro = pro(y.lnx_g, y.lnx_b); a.B = B; a.beam = R; a.N = N; a.D = D; a.heads = H; a.Ts = d->Ts; a.Tsp = d->Tsp;
            a.max_len = d->max_len; a.scale = scale; a.w_q = y.w_xq; a.b_q = (const float*)y.b_xq; a.w_o = y.w_xo; a.kv_enc = y.kv_enc;
            a.vt_enc = y.vt_enc; a.klen = d->enc_klen; a.steps = d->steps; a.part_out = P[k & 1];
            const int tp = d->Tsp == 128 ? 1 : (d->Tsp == 256 ? 2 : 0);     // encoder keys / values in registers for the launch
            if (dd == 256) launch_cross<T, (sizeof(T) == 2 ? 256 : 0)>(tp, dim3(H, B), need.cross, st, a);
            else if (dd == 512) launch_cross<T, (sizeof(T) == 2 ? 512 : 0)>(tp, dim3(H, B), need.cross, st, a);
            else launch_cross<T, 0>(tp, dim3(H, B), need.cross, st, a);
            ++k; np = H; bias = (const float*)y.b_xo;
            DEC_STOP_CHECK();
        }
        {
            FfnArgs a; a.pro = pro(y.ln2_g, y.ln2_b); a.B = B; a.beam = R; a.N = N; a.D = D; a.ffn = d->ffn; a.hs = hs; a.gelu = d->gelu;
            a.max_len = d->max_len; a.w_fc1 = y.w_fc1; a.b_fc1 = (const float*)y.b_fc1; a.w_fc2 = y.w_fc2; a.steps = d->steps; a.part_out = P[k & 1];
            if (dd == 256) launch_ffn<T, (sizeof(T) == 2 ? 256 : 0)>(hs, dim3(FS, B), need.ffn, st, a);
            else if (dd == 512) launch_ffn<T, (sizeof(T) == 2 ? 512 : 0)>(hs, dim3(FS, B), need.ffn, st, a);
            else launch_ffn<T, 0>(hs, dim3(FS, B), need.ffn, st, a);
            ++k; np = FS; bias = (const float*)y.b_fc2;
            DEC_STOP_CHECK();
        }
    }
    {
        FinalArgs a; a.pro = pro(d->lnf_g, d->lnf_b); a.beam = R; a.N = N; a.D = D; a.max_len = d->max_len; a.steps = d->steps; a.xn = d->xn;
        hipLaunchKernelGGL(dec_final_kernel<T>, dim3(B), dim3(NTHREADS), 0, st, a);
        DEC_STOP_CHECK();
    }
    {
        LogitArgs a; a.N = N; a.D = D; a.V = d->V; a.ldv = d->ldv; a.xn = d->xn; a.w = d->w_out; a.logits = d->logits;
        const dim3 grid((d->V + 63) / 64);
#define DEC_LOGITS(MT_) hipLaunchKernelGGL((dec_logits_kernel<T, MT_>), grid, dim3(NTHREADS), need.logits, st, a)
        switch ((N + 15) / 16) {
            case 1: DEC_LOGITS(1); break; case 2: DEC_LOGITS(2); break; case 3: DEC_LOGITS(3); break; case 4: DEC_LOGITS(4); break;
            case 5: DEC_LOGITS(5); break; case 6: DEC_LOGITS(6); break; case 7: DEC_LOGITS(7); break; default: DEC_LOGITS(8); break;
        }
#undef DEC_LOGITS
        DEC_STOP_CHECK();
    }
    {
        RowArgs a; a.beam = R; a.N = N; a.V = d->V; a.ldv = d->ldv; a.K2 = 2 * R; a.pad = d->pad; a.unk = d->unk; a.eos = d->eos;
        a.max_len = d->max_len; a.min_len = d->min_len; a.step0_all = d->step0_all_slots; a.it = d->inv_temperature; a.unk_penalty = d->unk_penalty;
        a.logits = d->logits; a.steps = d->steps; a.cum_hist = d->cum_hist; a.init_scores = d->init_scores; a.cand_val = d->cand_val;
        a.cand_idx = d->cand_idx;
        // columns per thread: the kernel is bound by its per-column VALU work on a wave that runs alone on its SIMD, so the unrolled loops
        // are sized to the vocabulary (V = 5,000: 20, not 32)
#define DEC_ROW(VPT_) hipLaunchKernelGGL(dec_row_kernel<VPT_>, dim3(N), dim3(NTHREADS), 0, st, a)
        const int vpt = (d->V + NTHREADS - 1) / NTHREADS;
        if (vpt <= 8) DEC_ROW(8); else if (vpt <= 12) DEC_ROW(12); else if (vpt <= 16) DEC_ROW(16); else if (vpt <= 20) DEC_ROW(20);
        else if (vpt <= 24) DEC_ROW(24); else if (vpt <= 32) DEC_ROW(32); else if (vpt <= 40) DEC_ROW(40); else if (vpt <= 48) DEC_ROW(48);
        else if (vpt <= 64) DEC_ROW(64); else if (vpt <= 96) DEC_ROW(96); else DEC_ROW(128);
#undef DEC_ROW
    }
    {
        SentArgs a; a.beam = R; a.N = N; a.D = D; a.V = d->V; a.K2 = 2 * R; a.pad = d->pad; a.eos = d->eos; a.max_len = d->max_len; a.maxpos = maxpos;
        a.step0_all = d->step0_all_slots; a.embed_scale = d->embed_scale; a.cand_val = d->cand_val; a.cand_idx = d->cand_idx; a.steps = d->steps;
        a.anc = d->anc; a.tok_hist = d->tok_hist; a.par_hist = d->par_hist; a.cum_hist = d->cum_hist; a.blacklist = d->blacklist; a.nfin = d->nfin;
        a.finished = d->finished; a.fin_step = d->fin_step; a.fin_row = d->fin_row; a.fin_score = d->fin_score; a.embed = d->embed;
        a.pos_table = d->pos_table; a.x0 = d->x0;
        hipLaunchKernelGGL(dec_sent_kernel<T>, dim3(B), dim3(NTHREADS), need.sent, st, a);
    }
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

template <typename T>
int begin_impl(const S2TDecodeDesc* d, int bos, hipStream_t st) {
    static bool configured = false;                         // per instantiation: raises the dynamic-LDS limit of the step's kernels once
    if (!configured) {
        const hipError_t e = configure<T>();
        if (e != hipSuccess) return S2T_EHIP(e);
        configured = true;
    }
    BeginArgs a; a.beam = d->beam; a.N = d->B * d->beam; a.D = d->D; a.pad = d->pad; a.bos = bos; a.embed_scale = d->embed_scale; a.steps = d->steps;
    a.tok_hist = d->tok_hist; a.blacklist = d->blacklist; a.nfin = d->nfin; a.finished = d->finished; a.embed = d->embed; a.pos_table = d->pos_table;
    a.x0 = d->x0;
    hipLaunchKernelGGL(dec_begin_kernel<T>, dim3(d->B), dim3(NTHREADS), 0, st, a);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
}  // namespace

extern "C" size_t s2t_decode_lds_bytes(const S2TDecodeDesc* d) {
    if (!desc_ok(d)) return 0;
    const LdsNeed n = lds_need(d);
    size_t m = n.self > n.cross ? n.self : n.cross;
    m = m > n.ffn ? m : n.ffn;
    m = m > n.logits ? m : n.logits;
    return m > n.sent ? m : n.sent;
}

static int decode_check(const S2TDecodeDesc* d) {
    if (!d) return S2T_EINVAL;
    if (!desc_ok(d)) return S2T_ENOTSUP;
    if (s2t_decode_lds_bytes(d) > LDS_CAP) return S2T_ENOTSUP;
    if (!d->x0 || !d->x1 || !d->part0 || !d->part1 || !d->xn || !d->logits || !d->steps || !d->anc || !d->cand_val || !d->cand_idx ||
        !d->tok_hist || !d->par_hist || !d->cum_hist || !d->blacklist || !d->nfin || !d->finished || !d->fin_step || !d->fin_row ||
        !d->fin_score || !d->lnf_g || !d->lnf_b || !d->w_out || !d->embed || !d->pos_table)
        return S2T_EINVAL;
    for (int l = 0; l < d->layers; ++l) {
        const S2TDecodeLayer& y = d->layer[l];
        const void* need[] = {y.ln1_g, y.ln1_b, y.w_qkv, y.b_qkv, y.w_o, y.b_o, y.lnx_g, y.lnx_b, y.w_xq, y.b_xq, y.w_xo, y.b_xo, y.ln2_g, y.ln2_b,
                              y.w_fc1, y.b_fc1, y.w_fc2, y.b_fc2, y.kv_enc, y.vt_enc, y.kv_cache};
        for (const void* p : need) if (!p) return S2T_EINVAL;
    }
    return S2T_OK;
}

extern "C" int s2t_decode_begin(const S2TDecodeDesc* d, int bos, void* stream) {
    const int rc = decode_check(d);
    if (rc != S2T_OK) return rc;
    if (bos < 0 || bos >= d->V) return S2T_EINVAL;
    return d->dtype == S2T_BF16 ? begin_impl<bf16>(d, bos, (hipStream_t)stream) : begin_impl<float>(d, bos, (hipStream_t)stream);
}

extern "C" int s2t_decode_step(const S2TDecodeDesc* d, void* stream) {
    const int rc = decode_check(d);
    if (rc != S2T_OK) return rc;
    return d->dtype == S2T_BF16 ? step_impl<bf16>(d, (hipStream_t)stream) : step_impl<float>(d, (hipStream_t)stream);
}

extern "C" int s2t_decode_prepare_enc(int dtype, const void* kv_enc, void* kp_enc, void* vp_enc, int Ts, int Tsp, int B, int D, int heads, void* stream) {
    if (!kv_enc || !kp_enc || !vp_enc || Ts < 1 || Tsp < Ts || Tsp % 128 || B < 1 || heads < 1 || D != heads * DH) return S2T_EINVAL;
    const size_t total = (size_t)2 * B * heads * Tsp * DH / (dtype == S2T_BF16 ? 8 : 4);
    const unsigned grid = (unsigned)((total + NTHREADS - 1) / NTHREADS < 4096 ? (total + NTHREADS - 1) / NTHREADS : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(dec_enc_pack_kernel<bf16>, dim3(grid), dim3(NTHREADS), 0, st, (const bf16*)kv_enc, (bf16*)kp_enc, (bf16*)vp_enc, Ts, Tsp, B, D, heads);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(dec_enc_pack_kernel<float>, dim3(grid), dim3(NTHREADS), 0, st, (const float*)kv_enc, (float*)kp_enc, (float*)vp_enc, Ts, Tsp, B, D, heads);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_decode_pack_weight(int dtype, const void* W, int ldw, int N, int K, void* Wp, void* stream) {
    if (!W || !Wp || N < 1 || K < 1 || ldw < K) return S2T_EINVAL;
    const int ks = dtype == S2T_BF16 ? 32 : 16;
    if ((dtype != S2T_BF16 && dtype != S2T_F32) || K % ks || ((uintptr_t)W & 15) || (ldw * (dtype == S2T_BF16 ? 2 : 4)) % 16) return S2T_ENOTSUP;
    const size_t pieces = (size_t)((N + 15) / 16) * (K / ks) * 64;
    const unsigned grid = (unsigned)((pieces + NTHREADS - 1) / NTHREADS < 8192 ? (pieces + NTHREADS - 1) / NTHREADS : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(dec_pack_kernel<bf16>, dim3(grid), dim3(NTHREADS), 0, st, (const bf16*)W, ldw, N, K, (bf16*)Wp);
    else hipLaunchKernelGGL(dec_pack_kernel<float>, dim3(grid), dim3(NTHREADS), 0, st, (const float*)W, ldw, N, K, (float*)Wp);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_decode_graph_create(const S2TDecodeDesc* d, int n_steps, void** graph_exec) {
    if (!graph_exec || n_steps < 1 || n_steps > 64) return S2T_EINVAL;
    *graph_exec = nullptr;
    const int rc = decode_check(d);
    if (rc != S2T_OK) return rc;
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) return S2T_EHIP(e);
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    int out = S2T_OK;
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
    if (e == hipSuccess) {
        for (int i = 0; i < n_steps && out == S2T_OK; ++i) out = d->dtype == S2T_BF16 ? step_impl<bf16>(d, cs) : step_impl<float>(d, cs);
        e = hipStreamEndCapture(cs, &g);
    }
    if (e == hipSuccess && out == S2T_OK) e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    (void)hipStreamDestroy(cs);
    if (out != S2T_OK) return out;
    if (e != hipSuccess) return S2T_EHIP(e);
    *graph_exec = ex;
    return S2T_OK;
}
extern "C" int s2t_decode_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return S2T_EINVAL;
    const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    return e == hipSuccess ? S2T_OK : S2T_EHIP(e);
}
extern "C" int s2t_decode_graph_destroy(void* graph_exec) {
    if (!graph_exec) return S2T_OK;
    const hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return e == hipSuccess ? S2T_OK : S2T_EHIP(e);
}

// diagnostic: the stamps of the -DS2T_DEC_STAMPS build (8 kernels x 16 u64: slots 0..13 shader clock, 14 / 15 the 100 MHz real-time counter at
// the first / last stamp); S2T_ENOTSUP in the product build.  Not part of include/s2t_hip.h.
extern "C" int s2t_decode_read_stamps(unsigned long long* out_host) {
#ifdef S2T_DEC_STAMPS
    if (!out_host) return S2T_EINVAL;
    const hipError_t e = hipMemcpyFromSymbol(out_host, HIP_SYMBOL(s2t_dec_stamps), sizeof(unsigned long long) * 8 * 16);
    return e == hipSuccess ? S2T_OK : S2T_EHIP(e);
#else
    (void)out_host;
    return S2T_ENOTSUP;
#endif
}
