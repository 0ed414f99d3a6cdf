// Shared device helpers for the S2T HIP kernels (gfx950 / CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define S2T_F32 0
#define S2T_BF16 1

// error codes returned through the C ABI
#define S2T_OK 0
#define S2T_EINVAL (-22)
#define S2T_ENOTSUP (-95)
#define S2T_EHIP(e) (-(1000 + (int)(e)))

#define S2T_LAUNCH_CHECK()                              \
    do {                                                \
        hipError_t _e = hipGetLastError();              \
        if (_e != hipSuccess) return S2T_EHIP(_e);      \
    } while (0)

// kernel-route options (runtime.hip, s2t_set_option)
extern int g_s2t_opt_gemm256, g_s2t_opt_attn_v1, g_s2t_opt_attn_v2_min_tq, g_s2t_opt_gemm256_min_tiles, g_s2t_opt_gemm256_sched;
extern int g_s2t_opt_reserve_cus, g_s2t_opt_f32_small_nt, g_s2t_opt_f32_small_kt, g_s2t_opt_f32_narrow, g_s2t_opt_small_nt, g_s2t_opt_small_kt, g_s2t_opt_attn_bwd_fused, g_s2t_opt_gemm_deep, g_s2t_opt_ln_small;
// workgroups of a persistent one-per-CU launch: the chip's 256 CUs minus the ones left to a collective that runs beside it
// (s2t_set_option "reserve_cus": RCCL's kernels hold CUs while a bucket travels; a 128 KiB-LDS workgroup that finds its CU taken
// waits for a whole round of the others)
inline int s2t_persistent_cus() { return 256 - g_s2t_opt_reserve_cus; }

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int PER16 = 4; };
template <> struct Elem<bf16> { static constexpr int PER16 = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// One 16x16 MFMA "k-group" on a 16-byte operand fragment per lane.
//   bf16: one v_mfma_f32_16x16x32_bf16 (lane holds A[row=l&15][k=8*(l>>4)+j], j<8)
//   f32 : four chained v_mfma_f32_16x16x4_f32 (lane holds k=4*(l>>4)+j; instruction j takes
//         element j, i.e. a fixed permutation of k shared by A and B -> exact f32 fma chain)
// C/D layout for both: col = lane&15, row = 4*(lane>>4) + reg.
template <typename T> __device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16<bf16>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<float>(u32x4 a, u32x4 b, f32x4 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf_ = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf_[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf_[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf_[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf_[3], c, 0, 0, 0);
    return c;
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block reductions (blockDim.x multiple of 64, <= 1024); `sh` holds >= 16 floats
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// S2T_ACT_* of include/s2t_hip.h
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_RELU_BWD = 3, ACT_GELU_BWD = 4, ACT_RELU_MASK = 5, ACT_RELU_BWD_MASK = 6 };
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Dropout masks are a stateless integer hash of (seed, element index): nothing is stored, the backward pass
// re-evaluates it, and every element is independent so GEMM / attention epilogues can evaluate it in whatever
// register layout they hold.  One hash serves the element QUAD (4i .. 4i+3) as four 16-bit uniforms (the rate is
// quantised to 1/65536).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// The hash in two steps, for kernels whose VALU time is the bound (attention: 72 M probabilities per encoder layer, hashed in the
// forward, the dK/dV and the dQ kernels; the GEMM epilogues, which run beside no MFMA): the seed scramble `ks` is wave-uniform
// (scalar ALU, mix32), and the contribution `hwm` of the index's high word (almost always zero) is the same for every quad that
// shares that word, so both are taken once per tile / row.  The per-quad part is built from full-rate operations only -- 32-bit
// multiplies run at quarter rate on CDNA, three of them made 23 issue slots per quad; v_mad_u32_u24 (x + (x mod 2^24) * C with C
// even: a bijection that mixes upwards) and v_bfrev_b32 (turns the well-mixed top bits into the next multiply's low bits) make 12.
// Checked offline (tools/drop_hash_check.py, profiles/r02_drop_hash_check.txt): avalanche |P(flip) - 1/2| < 0.01 on all 64 output
// bits for every input bit, chi-square of the 16-bit fields ~1.0, keep-bit correlations between neighbouring elements / rows < 1e-3,
// per-row and per-column keep counts binomial.
__device__ __forceinline__ uint32_t drop_seed_key(uint64_t seed) { return mix32((uint32_t)seed * 0x9E3779B9u + 0x7F4A7C15u); }
__device__ __forceinline__ uint32_t drop_high_mix(uint64_t seed, uint64_t quad) {
    const uint32_t hw = (uint32_t)(quad >> 32) + (uint32_t)(seed >> 32);
    return hw ^ (hw << 13) ^ (hw >> 7) ^ (hw << 27);
}
__device__ __forceinline__ u32x2 drop_hash4_lo(uint32_t ks, uint32_t hwm, uint32_t quad_lo) {
    uint32_t a = quad_lo ^ ks;
    a = __umul24(a, 0x3C6EF2u) + a; a = __builtin_bitreverse32(a);
    a = __umul24(a, 0x9E3778u) + a; a = __builtin_bitreverse32(a);
    a = __umul24(a, 0x85EBCAu) + a; a ^= a >> 16;
    const uint32_t x = a ^ hwm;
    const uint32_t y = __builtin_bitreverse32(__umul24(x ^ 0x68E31DA4u, 0xC2B2AEu) + x);
    return (u32x2){x, y};
}
__device__ __forceinline__ u32x2 drop_hash4(uint64_t seed, uint64_t quad) {
    // the seed is scrambled first (wave-uniform: scalar ALU) so that call sites whose seeds differ by small offsets do not
    // reuse one mask at xor-neighbouring positions
    return drop_hash4_lo(drop_seed_key(seed), drop_high_mix(seed, quad), (uint32_t)quad);
}
// 16-bit uniform of element f (0..3) of the quad: f0 = y.lo, f1 = y.hi, f2 = x.lo, f3 = x.hi
__device__ __forceinline__ uint32_t drop_field(u32x2 h, int f) {
    const uint32_t w = (f & 2) ? h[0] : h[1];
    return (f & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh /* p * 2^32 */) {
    return drop_field(drop_hash4(seed, idx >> 2), (int)(idx & 3)) >= (thresh >> 16);
}

// bijective XCD-aware remap of a linear workgroup id (cdna_hip_programming.md section 5, T1):
// blocks b and b+8 share an XCD, so give each XCD a contiguous chunk of the tile space.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}
