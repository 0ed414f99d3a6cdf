// Second subsampling convolution, forward: nn.Conv2d(C, C, 3, stride 2, padding 1) + bias + activation on the channels-last
// tensors of the subsampler (examples/speech_recognition/models/conv_transformer.py:143-150,202-214; weights :348-354).
//
//   z2[t4][b][f4][co] = act( bias[co] + sum_{kh,kw,ci} y1n[b][2 t4 + kh - 1][2 f4 + kw - 1][ci] * w[co][ci][kh][kw] )
//
// The gathered-GEMM form (s2t_gemm_gather with per-tap row maps) reads every input pixel 2.25 times through index tables and took
// 170 us for the bench shape (208 TFLOP/s; 35 GFLOP, 245 MB in, 61 MB out).  Here a workgroup walks units of two output rows of
// one utterance: the five input rows they touch (25 KB) are staged ONCE by LDS-DMA (double-buffered), the A operand of the MFMA is
// gathered from LDS by per-lane addresses (stride-2 pixels: a tap's 16 pixels are 256 B apart, so the 16-byte channel chunks are
// XOR-swizzled with the pixel PAIR index and neighbouring pixel pairs of the upper half-row are swapped -- both applied on the DMA's
// source side -- to keep the 16 lanes of a read on distinct banks), and the weights never touch LDS: wave w owns output channels
// 16 w .. 16 w + 15 and keeps their 576-long rows as 18 operand fragments in registers for the whole launch.
// bf16, C = 64 only (the reference's default front end; other shapes keep the gathered GEMM).
#include "common.hpp"
#include "prof.hpp"
#include "../../include/s2t_hip.h"

namespace {
constexpr int C2 = 64;                       // channels in = out
constexpr int ROWB = 128;                    // bytes of one pixel (64 bf16)
__device__ uint4 g_conv2_zero[64];           // 1 KiB of zeros: DMA source of the padding rows

// physical slot of input pixel column `col` inside its LDS row (an involution: bit 4 selects, bit 0 flips)
__device__ __forceinline__ int pix_slot(int col) { return col ^ ((col >> 4) & 1); }
// position of 16-byte chunk `ch` (8 channels) inside its pixel
__device__ __forceinline__ int chunk_pos(int ch, int col) { return ch ^ ((col >> 1) & 7); }
}  // namespace

// One unit = (utterance b, output rows t4 = 2 u, 2 u + 1).  LDS stage: [5 input rows][F2P pixel slots][128 B]; F2P = F2 rounded up to 8
// (one DMA wave-instruction fills 8 pixel slots).  A trailing 128-byte zero pixel serves the left / right padding columns.
template <int ACT>
__global__ __launch_bounds__(256, 3) void conv2_fwd_kernel(const bf16* __restrict__ y1n, const bf16* __restrict__ w2p, const float* __restrict__ bias,
                                                           bf16* __restrict__ z2, bf16* __restrict__ pre, int B, int T2, int F2, int T4, int F4,
                                                           int F2P, int units_per_b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
    const int stage_bytes = 5 * F2P * ROWB;
    char* zero_px = smem + 2 * stage_bytes;                           // 128 B of zeros
    if (threadIdx.x < 8) reinterpret_cast<u32x4*>(zero_px)[threadIdx.x] = (u32x4){0, 0, 0, 0};

    // ---- weights of this wave's 16 output channels: 18 B-side... A-side fragments (rows = co, k = tap * 64 + ci), straight from memory
    u32x4 wf[18];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
        wf[ks] = *reinterpret_cast<const u32x4*>(w2p + (size_t)(16 * wave + r16) * (9 * C2) + ks * 32 + 8 * q);
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * wave + 4 * q + r];

    const int n_units = B * units_per_b;
    const int n_instr = 5 * (F2P / 8);                                // DMA wave-instructions per stage
    auto stage = [&](int u, int s) {
        const int b = u / units_per_b, t4a = 2 * (u - b * units_per_b);
        char* st = smem + s * stage_bytes;
        for (int i = wave; i < n_instr; i += 4) {
            const int row = i / (F2P / 8), p8 = i - row * (F2P / 8);  // input row 0..4 of the stage, group of 8 pixel slots
            const int t2 = 2 * t4a - 1 + row;
            const int slot = 8 * p8 + (lane >> 3), col = pix_slot(slot), cp = lane & 7;
            const bool ok = t2 >= 0 && t2 < T2 && col < F2;
            const char* src = ok ? reinterpret_cast<const char*>(y1n + (((size_t)b * T2 + t2) * F2 + col) * C2) + (chunk_pos(cp, col) << 4)
                                 : reinterpret_cast<const char*>(g_conv2_zero) + 16 * lane;
            char* dst = st + __builtin_amdgcn_readfirstlane((row * F2P + 8 * p8) * ROWB);
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    // per-lane geometry of the three 16-pixel m-tiles of a unit (output pixel op = 16 mt + r16 of 2 F4; clamped past the end)
    int orow[3], f4v[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int op = min(16 * mt + r16, 2 * F4 - 1);
        orow[mt] = op >= F4 ? 1 : 0; f4v[mt] = op - orow[mt] * F4;
    }

    int u = blockIdx.x;
    if (u < n_units) stage(u, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int it = 0; u < n_units; u += gridDim.x, ++it) {
        const int s = it & 1;
        const char* st = smem + s * stage_bytes;
        if (u + gridDim.x < n_units) stage(u + gridDim.x, s ^ 1);     // its readers of two units ago passed the last barrier
        f32x4 acc[3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) acc[mt] = bv;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const int col = 2 * f4v[mt] + kw - 1;
                const bool in = col >= 0 && col < F2;
                const char* px = st + ((2 * orow[mt] + kh) * F2P + pix_slot(max(col, 0))) * ROWB;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const char* a = in ? px + (chunk_pos(4 * hf + q, col) << 4) : zero_px + 16 * q;
                    const u32x4 pf = *reinterpret_cast<const u32x4*>(a);
                    acc[mt] = mma16<bf16>(wf[2 * tap + hf], pf, acc[mt]);             // D[co = 4 q + r][pixel = r16]
                }
            }
        }
        // ---- epilogue: activation, 8-byte stores of 4 channels per lane (the four waves complete a pixel's 128 bytes)
        const int b = u / units_per_b, t4a = 2 * (u - b * units_per_b);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int op = 16 * mt + r16;
            const int t4 = t4a + orow[mt];
            if (op < 2 * F4 && t4 < T4) {
                const size_t o = (((size_t)t4 * B + b) * F4 + f4v[mt]) * C2 + 16 * wave + 4 * q;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mt][r];
                if constexpr (ACT == ACT_GELU) {
                    u32x2 pw;
                    pw[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v[0], v[1]}, __attribute__((ext_vector_type(2))) __bf16));
                    pw[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v[2], v[3]}, __attribute__((ext_vector_type(2))) __bf16));
                    *reinterpret_cast<u32x2*>(pre + o) = pw;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_f(to_f32(from_f32<bf16>(v[r])));      // gelu of the stored pre-activation (as the GEMM epilogue)
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                u32x2 w;
                w[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v[0], v[1]}, __attribute__((ext_vector_type(2))) __bf16));
                w[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v[2], v[3]}, __attribute__((ext_vector_type(2))) __bf16));
                *reinterpret_cast<u32x2*>(z2 + o) = w;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // the next unit's rows have landed (this wave's share; the barrier covers the rest)
        __syncthreads();
    }
}

extern "C" int s2t_conv2_fwd(int dtype, const void* y1n, const void* w2p, const float* bias, void* z2, void* pre, int B, int T2, int F2, int C,
                             int act, void* stream) {
    if (B <= 0 || T2 <= 0 || F2 <= 0) return S2T_OK;
    if (!y1n || !w2p || !bias || !z2) return S2T_EINVAL;
    if ((act != ACT_RELU && act != ACT_GELU) || (act == ACT_GELU && !pre)) return S2T_EINVAL;
    if (dtype != S2T_BF16 || C != C2 || F2 > 128) return S2T_ENOTSUP;
    if (((uintptr_t)y1n | (uintptr_t)w2p | (uintptr_t)z2 | (uintptr_t)pre) & 15) return S2T_ENOTSUP;
    const int T4 = (T2 + 1) / 2, F4 = (F2 + 1) / 2;
    if (2 * F4 > 48) return S2T_ENOTSUP;                               // three 16-pixel m-tiles per unit
    const int F2P = (F2 + 7) & ~7, upb = (T4 + 1) / 2;
    const size_t lds = (size_t)2 * 5 * F2P * ROWB + 128;
    if (lds > 64 * 1024) return S2T_ENOTSUP;
    const long units = (long)B * upb;
    const int grid = (int)(units < 768 ? units : 768);                 // three workgroups per CU, persistent
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("conv2_fwd", st, 2.0 * B * T4 * F4 * (double)C2 * 9 * C2, 2.0 * B * ((double)T2 * F2 + (double)T4 * F4) * C2);
    if (act == ACT_GELU)
        hipLaunchKernelGGL(conv2_fwd_kernel<ACT_GELU>, dim3(grid), dim3(256), lds, st, (const bf16*)y1n, (const bf16*)w2p, bias, (bf16*)z2, (bf16*)pre,
                           B, T2, F2, T4, F4, F2P, upb);
    else
        hipLaunchKernelGGL(conv2_fwd_kernel<ACT_RELU>, dim3(grid), dim3(256), lds, st, (const bf16*)y1n, (const bf16*)w2p, bias, (bf16*)z2, (bf16*)nullptr,
                           B, T2, F2, T4, F4, F2P, upb);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------------------------ data gradient
//   dy1n[b][t2][f2][ci] = dropout( sum over the taps (kh, kw) with 2 t4 + kh - 1 = t2, 2 f4 + kw - 1 = f2 of
//                                  sum_co dpre[t4][b][f4][co] * w[co][ci][kh][kw] )
// An input pixel is reached by 1, 2, 2 or 4 taps depending on the parities (pt, pf) of (t2, f2): four classes with their own
// reduction length.  The gathered form ran one product per class with scattered output rows (260 us for the four).  Here a unit is
// four input rows of one utterance (two per row parity, 40 pixels per class = three 16-pixel m-tiles); the three dpre rows they
// reach (7.7 KB) are staged by LDS-DMA; wave w owns input channels 16 w .. 16 w + 15 with the weights' 18 fragments in registers
// (w2q = s2t_permute_conv_w(mode 1): [ci][slot * C + co], taps in class-major slot order).  The dropout mask of y1n is applied on
// the way out, indexed by the element's position in y1n like every other mask.
namespace {
// class c = 2 pt + pf: first slot 0 / 1 / 3 / 5 and 1 / 2 / 2 / 4 taps; slot -> (kh, kw) = (1,1) (1,0) (1,2) (0,1) (2,1) (0,0) (0,2) (2,0) (2,2)
__device__ __forceinline__ int dpx_slot(int col) { return col ^ ((col >> 3) & 1); }      // neighbouring pairs of the odd octets swapped
__device__ __forceinline__ int dchunk_pos(int ch, int col) { return ch ^ (col & 7); }
}  // namespace

__global__ __launch_bounds__(256, 3) void conv2_dgrad_kernel(const bf16* __restrict__ dpre, const bf16* __restrict__ w2q, bf16* __restrict__ dy1n,
                                                             int B, int T2, int F2, int T4, int F4, int F4P, int units_per_b, float p_drop,
                                                             unsigned long long seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
    const int stage_bytes = 3 * F4P * ROWB;
    char* zero_px = smem + 2 * stage_bytes;
    if (threadIdx.x < 8) reinterpret_cast<u32x4*>(zero_px)[threadIdx.x] = (u32x4){0, 0, 0, 0};
    u32x4 wf[18];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
        wf[ks] = *reinterpret_cast<const u32x4*>(w2q + (size_t)(16 * wave + r16) * (9 * C2) + ks * 32 + 8 * q);
    const uint32_t drop_th16 = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float drop_inv = 1.f / (1.f - p_drop);

    const int n_units = B * units_per_b;
    const int n_instr = 3 * (F4P / 8);
    auto stage = [&](int u, int s) {
        const int b = u / units_per_b, a = 2 * (u - b * units_per_b);                 // unit = input rows 2a .. 2a + 3 <- dpre rows a .. a + 2
        char* st = smem + s * stage_bytes;
        for (int i = wave; i < n_instr; i += 4) {
            const int row = i / (F4P / 8), p8 = i - row * (F4P / 8);
            const int t4 = a + row;
            const int slot = 8 * p8 + (lane >> 3), col = dpx_slot(slot), cp = lane & 7;
            const bool ok = t4 < T4 && col < F4;
            const char* src = ok ? reinterpret_cast<const char*>(dpre + (((size_t)t4 * B + b) * F4 + col) * C2) + (dchunk_pos(cp, col) << 4)
                                 : reinterpret_cast<const char*>(g_conv2_zero) + 16 * lane;
            char* dst = st + __builtin_amdgcn_readfirstlane((row * F4P + 8 * p8) * ROWB);
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    const int npc = (F2 + 1) / 2;                                      // pixels of a row in the even-column class (odd class: F2 / 2)

    int u = blockIdx.x;
    if (u < n_units) stage(u, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int it = 0; u < n_units; u += gridDim.x, ++it) {
        const char* st = smem + (it & 1) * stage_bytes;
        if (u + gridDim.x < n_units) stage(u + gridDim.x, (it & 1) ^ 1);
        const int b = u / units_per_b, a = 2 * (u - b * units_per_b);
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int pt = cls >> 1, pf = cls & 1;
            const int ncol = pf ? F2 / 2 : npc;                        // pixels per row in this class
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const int pi = 16 * mt + r16;                          // pixel of the class inside the unit: row pi / ncol (0 / 1), column j
                const bool valid = ncol > 0 && pi < 2 * ncol;
                const int pic = valid ? pi : 0;
                const int crow = pic >= ncol ? 1 : 0, j = pic - crow * ncol;
                const int t2 = 2 * a + 2 * crow + pt, f2 = 2 * j + pf;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    if (ti < (cls == 0 ? 1 : (cls == 3 ? 4 : 2))) {
                        const int slot = (cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5))) + ti;
                        const int kh = slot == 0 ? 1 : slot == 1 ? 1 : slot == 2 ? 1 : slot == 3 ? 0 : slot == 4 ? 2 : slot == 5 ? 0 : slot == 6 ? 0 : 2;
                        const int kw = slot == 0 ? 1 : slot == 1 ? 0 : slot == 2 ? 2 : slot == 3 ? 1 : slot == 4 ? 1 : slot == 5 ? 0 : slot == 6 ? 2 : slot == 7 ? 0 : 2;
                        const int t4r = (t2 + 1 - kh) / 2 - a, f4 = (f2 + 1 - kw) / 2;        // exact divisions for this class's taps
                        const bool in = f4 >= 0 && f4 < F4 && t4r >= 0;                       // rows past T4 were staged as zeros
                        const char* px = st + (min(max(t4r, 0), 2) * F4P + dpx_slot(min(max(f4, 0), F4P - 1))) * ROWB;
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const char* ad = in ? px + (dchunk_pos(4 * hf + q, f4) << 4) : zero_px + 16 * q;
                            acc = mma16<bf16>(wf[2 * slot + hf], *reinterpret_cast<const u32x4*>(ad), acc);    // D[ci = 4 q + r][pixel = r16]
                        }
                    }
                }
                if (valid && t2 < T2) {
                    const size_t o = (((size_t)b * T2 + t2) * F2 + f2) * C2 + 16 * wave + 4 * q;      // element index in dy1n (the mask's index)
                    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
                    if (p_drop > 0.f) {
                        const u32x2 hq = drop_hash4(seed, (uint64_t)o >> 2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = drop_field(hq, r) >= drop_th16 ? v[r] * drop_inv : 0.f;
                    }
                    u32x2 w;
                    w[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v[0], v[1]}, __attribute__((ext_vector_type(2))) __bf16));
                    w[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v[2], v[3]}, __attribute__((ext_vector_type(2))) __bf16));
                    // 8-byte pieces, four waves completing a pixel's 128 bytes (parking the unit's rows in LDS and storing whole
                    // pixels measured slower: 163 vs 149 us -- the launch is bound by its index / mask arithmetic, not by the stores)
                    *reinterpret_cast<u32x2*>(dy1n + o) = w;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
}

extern "C" int s2t_conv2_dgrad(int dtype, const void* dpre, const void* w2q, void* dy1n, int B, int T2, int F2, int C, float p_drop,
                               unsigned long long seed, void* stream) {
    if (B <= 0 || T2 <= 0 || F2 <= 0) return S2T_OK;
    if (!dpre || !w2q || !dy1n || p_drop < 0.f || p_drop >= 1.f) return S2T_EINVAL;
    if (dtype != S2T_BF16 || C != C2) return S2T_ENOTSUP;
    if (((uintptr_t)dpre | (uintptr_t)w2q | (uintptr_t)dy1n) & 15) return S2T_ENOTSUP;
    const int T4 = (T2 + 1) / 2, F4 = (F2 + 1) / 2;
    if (F2 + 1 > 48 || F4 > 64) return S2T_ENOTSUP;                     // a class's two rows fit three 16-pixel m-tiles
    const int F4P = (F4 + 8) & ~7;                                       // one slot beyond F4 so that the pair swap stays inside the row
    const int upb = (T2 + 3) / 4;
    const size_t lds = (size_t)2 * 3 * F4P * ROWB + 128;
    const long units = (long)B * upb;
    const int grid = (int)(units < 768 ? units : 768);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("conv2_dgrad", st, 2.0 * B * T4 * F4 * (double)C2 * 9 * C2, 2.0 * B * ((double)T2 * F2 + (double)T4 * F4) * C2);
    hipLaunchKernelGGL(conv2_dgrad_kernel, dim3(grid), dim3(256), lds, st, (const bf16*)dpre, (const bf16*)w2q, (bf16*)dy1n, B, T2, F2, T4, F4, F4P, upb,
                       p_drop, seed);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
