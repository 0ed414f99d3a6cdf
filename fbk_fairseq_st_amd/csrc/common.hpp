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

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Philox-4x32-10 counter RNG: one call -> 4 uniform u32. Used for dropout masks that are
// regenerated (not stored) in the backward pass from (seed, offset).
__device__ __forceinline__ uint4 philox4x32(uint64_t seed, uint64_t ctr) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// keep-decision for element `idx` of a tensor: a stateless 2-round integer hash of (seed, idx) -- ~14 integer ops
// per element against ~70 for Philox-4x32-10, and every element is independent, so the GEMM / attention
// epilogues can evaluate it in whatever register layout they hold (the backward pass re-evaluates it).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh /* p * 2^32 */) {
    const uint32_t h = mix32((uint32_t)idx + (uint32_t)seed * 0x9E3779B9u);
    return mix32(h ^ (uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32) ^ 0x85ebca6bu) >= thresh;
}

// bijective XCD-aware remap of a linear workgroup id (cdna_hip_programming.md section 5, T1):
// blocks b and b+8 share an XCD, so give each XCD a contiguous chunk of the tile space.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}
