// Pieces of the big-tile bf16 GEMM kernel (gemm256.hip: eight waves of 128 x 64; the archived four-wave experiment tools/lab/gemm4w.hip used them too):
// the LDS image constants, the transposed fragment read, the epilogue arithmetic of one output quad and the buffer-addressed accesses.
#pragma once
#include "common.hpp"
#include "gemm_epilogue.hpp"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef short s16x4_t __attribute__((ext_vector_type(4)));

static constexpr int HALF = 16384;            // bytes of one half-tile
static constexpr int BUF = 65536;             // one K-tile buffer: A-h0 | A-h1 | B-h0 | B-h1
static constexpr int BK = 64;

__device__ __forceinline__ int trswz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

// fragment of a k-strided operand from a [64 k][128 col] image (256-byte rows, chunks XOR-swizzled by trswz): 16 columns from
// `col`, k-half s; two ds_read_b64_tr_b16 (cdna_hip_programming.md T10, image (b))
__device__ __forceinline__ u32x4 tr_frag(const char* img, int col, int s, int r16, int q) {
    u32x4 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 32 * s + 8 * q + 4 * h + (r16 >> 2);
        const int ch = (col >> 3) + ((r16 & 3) >> 1);
        const char* a = img + row * 256 + ((ch ^ trswz(row)) << 4) + ((r16 & 1) << 3);
        const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a);
        const u32x2 w = __builtin_bit_cast(u32x2, v);
        f[2 * h] = w[0]; f[2 * h + 1] = w[1];
    }
    return f;
}

// The same read as inline assembly (round 5; wgrad_group.hip has the twin and the reasoning): through the builtin hipcc puts
// `s_waitcnt vmcnt(0)` in front of the first transposed read of a memory segment -- the intrinsic has no memory operand, so the
// wait-count pass assumes it reads what any LDS-DMA in flight writes -- which drained the NN kernels' prefetch once per K-tile.
// addr = LDS byte address of the lane's 8 bytes in buffer 0 / half 0 / k-half 0, OFF = the compile-time rest (16-bit offset field).
template <int OFF>
__device__ __forceinline__ u32x2 tr_read_asm(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t tr_lane_addr(uint32_t img, int col, int hh, int r16, int q) {
    const int row = 8 * q + 4 * hh + (r16 >> 2);
    const int ch = (col >> 3) + ((r16 & 3) >> 1);
    return img + row * 256 + ((ch ^ trswz(row)) << 4) + ((r16 & 1) << 3);        // trswz(row + 32 s) == trswz(row)
}

// ---- epilogue straight from the accumulators (the LDS ring keeps filling for the next tile meanwhile): the MFMAs ran as
// (W rows x X rows), so a lane holds 4 consecutive columns of one row: C[row][col .. col + 3] -> one 8-byte (bf16) or 16-byte (f32)
// access per operand.  Same arithmetic, in the same order, as gemm_finish (gemm_epilogue.hpp): alpha, bias, activation, dropout
// (one hash per aligned element quad = exactly the mask of the standalone kernel), residual / accumulate.
// EXT: the one extra operand stream an epilogue may read besides the bias: 0 none, 1 residual, 2 the old C (accumulate), 3 aux (the
// activation-backward operand).  Its 16 loads of a half tile are issued together and waited for once (one memory round trip per half
// tile: with one workgroup per CU nothing else hides that latency).
enum { EXT_NONE = 0, EXT_RES = 1, EXT_OLD = 2, EXT_AUX = 3 };
template <typename TO> struct Pack4 { typedef u32x2 type; };
template <> struct Pack4<float> { typedef u32x4 type; };

// arithmetic of one output quad (row, col .. col + 3); `pre` receives the GELU pre-activation (aux_out)
template <typename TO, int ACT, int EXT, bool DROP>
__device__ __forceinline__ typename Pack4<TO>::type epi_quad(const GemmArgs& p, f32x4 v, f32x4 b4, typename Pack4<TO>::type ext,
                                                             uint32_t quad, uint32_t drop_ks, uint32_t drop_hwm, uint32_t drop_th, float drop_inv,
                                                             typename Pack4<TO>::type& pre_out) {
    typedef typename Pack4<TO>::type PK;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = v[e] * p.alpha + b4[e];
    TO xe[4];
    *reinterpret_cast<PK*>(xe) = ext;
    TO pre[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (ACT == ACT_RELU || ACT == ACT_RELU_MASK) x[e] = fmaxf(x[e], 0.f);
        else if constexpr (ACT == ACT_GELU) { pre[e] = from_f32<TO>(x[e]); x[e] = gelu_f(x[e]); }
        else if constexpr (ACT == ACT_RELU_BWD) x[e] = (to_f32(xe[e]) > 0.f) ? x[e] : 0.f;
        else if constexpr (ACT == ACT_GELU_BWD) x[e] *= gelu_grad_f(to_f32(xe[e]));
    }
    if constexpr (ACT == ACT_GELU) pre_out = *reinterpret_cast<const PK*>(pre);
    // DROP = false: the caller knows p_drop == 0.  DROP = true: tested here, per quad, ON PURPOSE -- as one straight-line body the masked
    // epilogue (1,081 VALU instructions) ran 7 % slower on fc1 than as per-quad blocks; the unmasked one ran 6 % faster straight-line
    if (DROP && p.p_drop > 0.f) {
        // the output has fewer than 2^34 elements (host check): the quad index is one 32-bit word, the seed scramble and the
        // high-word term are per-launch constants (common.hpp drop_hash4_lo: the same mask as every other kernel)
        const u32x2 dh = drop_hash4_lo(drop_ks, drop_hwm, quad);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (drop_field(dh, e) >= (drop_th >> 16)) ? x[e] * drop_inv : 0.f;
    }
    if constexpr (EXT == EXT_RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] += to_f32(xe[e]);
    }
    TO o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = from_f32<TO>(x[e]);
    if constexpr (EXT == EXT_OLD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<TO>(to_f32(o[e]) + to_f32(xe[e]));
    }
    return *reinterpret_cast<const PK*>(o);
}

// buffer-addressed 8 / 16-byte accesses: per-lane 32-bit voffset + wave-uniform soffset (no 64-bit address arithmetic per access);
// offsets past the descriptor's size (rows >= M; the voffset of a column >= N is forced there) load zeros / store nothing
template <typename PK> __device__ __forceinline__ PK buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff);
template <> __device__ __forceinline__ u32x2 buf_load<u32x2>(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}
template <> __device__ __forceinline__ u32x4 buf_load<u32x4>(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store(u32x2 v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store(u32x4 v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

