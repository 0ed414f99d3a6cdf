// ConvAttention2D blocks of the conv_transformer front end (reference:
// examples/speech_recognition/modules/conv_attention_2d.py:46-135, called from conv_transformer.py:216-222 with
// query = key = value = x and no padding mask).
//
// Layout: every tensor is channels-last over the subsampler's pixel rows r = (t*B + b)*F + f  (the layout conv2 / fc3
// already use): x [M][64], qkv [M][16] (channels 0-3 q, 4-7 k, 8-11 v, 12-15 zero padding so that rows are 16-byte
// multiples), cat [M][8] (0-3 time attention, 4-7 frequency attention).  The two 3x3 convolutions run as gathered GEMMs
// (gemm.hip, row maps built by the host); this file holds what is specific to the block:
//   * BatchNorm statistics / apply(+ReLU, + residual) / backward for a small generic channel count,
//   * time attention  softmax_t'(q k^T) v   per (batch, head) plane [T][F]   (head_dim = F = 20 for 80-mel input),
//   * frequency attention softmax_f'(q^T k) v^T (scores summed over every frame),
//   * packing of the convolution weights for the three gathered-GEMM forms.
// All arithmetic is f32; the planes are tiny (4 heads x F <= 32 columns), so these are plain VALU kernels staged
// through LDS -- HBM-bound element work, not MFMA shapes.
#include "common.hpp"
#include "prof.hpp"
#include <cstdlib>
#include "../../include/s2t_hip.h"

namespace {

constexpr int QKV_C = 16, CAT_C = 8, HEADS = 4;

#define A2D_DISPATCH_T(dtype, EXPR_BF16, EXPR_F32) \
    do { if ((dtype) == S2T_BF16) { EXPR_BF16; } else if ((dtype) == S2T_F32) { EXPR_F32; } else return S2T_ENOTSUP; } while (0)

// ------------------------------------------------------------------ BatchNorm pieces (channels-last, C <= 64, row stride ld)
// sums layout: channel c of group g = c / Cg  ->  sums[g*2*Cg + (c % Cg)] (first moment) and [.. + Cg] (second):
// each group is the [C | C] block s2t_bn_finalize expects, so q / k / v finalise separately from one statistics pass.
// mode 0: moments of z' = prescale[c] * z.   mode 1 (backward): with dyn = dy * [z'*scale + shift > 0] (ReLU after BN):
// sum dyn and sum dyn * xhat, xhat = (z' - mean) * rstd.
template <typename T>
__global__ __launch_bounds__(256) void a2d_stats_kernel(const T* __restrict__ z, const T* __restrict__ dy, const float* __restrict__ prescale,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        double* __restrict__ sums, long M, int C, int ld_z, int ld_dy, int Cg, int mode, int rpb) {
    __shared__ double sh[2][256];
    const int CP = C <= 16 ? 16 : (C <= 32 ? 32 : 64);          // threads per row
    const int c = threadIdx.x % CP, rsub = threadIdx.x / CP, rstep = 256 / CP;
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < M ? r0 + rpb : M;
    float a0 = 0.f, a1 = 0.f;
    double d0 = 0.0, d1 = 0.0;
    if (c < C) {
        const float ps = prescale ? prescale[c] : 1.f;
        const float mu = mode ? mean[c] : 0.f, rs = mode ? rstd[c] : 0.f, sc = mode ? scale[c] : 0.f, sf = mode ? shift[c] : 0.f;
        int n = 0;
        for (long r = r0 + rsub; r < r1; r += rstep) {
            const float v = to_f32(z[r * ld_z + c]) * ps;
            if (mode == 0) { a0 += v; a1 += v * v; }
            else {
                const float g = (v * sc + sf > 0.f) ? to_f32(dy[r * ld_dy + c]) : 0.f;
                a0 += g; a1 += g * (v - mu) * rs;
            }
            if (++n == 64) { d0 += a0; d1 += a1; a0 = a1 = 0.f; n = 0; }     // short f32 runs, double totals
        }
        d0 += a0; d1 += a1;
    }
    sh[0][threadIdx.x] = d0; sh[1][threadIdx.x] = d1;
    __syncthreads();
    if (threadIdx.x < CP && c < C) {
        double s0 = 0.0, s1 = 0.0;
        for (int k = 0; k < rstep; ++k) { s0 += sh[0][k * CP + c]; s1 += sh[1][k * CP + c]; }
        const int g = c / Cg, cc = c % Cg;
        atomicAdd(sums + g * 2 * Cg + cc, s0);
        atomicAdd(sums + g * 2 * Cg + Cg + cc, s1);
    }
}

// y = relu(prescale*z*scale + shift) [+ res]
template <typename T>
__global__ __launch_bounds__(256) void a2d_bn_act_kernel(const T* __restrict__ z, const float* __restrict__ prescale, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res, T* __restrict__ y,
                                                         long M, int C, int ld_z, int ld_y) {
    const long n = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / C; const int c = (int)(i % C);
        float v = to_f32(z[r * ld_z + c]) * (prescale ? prescale[c] : 1.f) * scale[c] + shift[c];
        v = fmaxf(v, 0.f);
        if (res) v += to_f32(res[r * ld_y + c]);
        y[r * ld_y + c] = from_f32<T>(v);
    }
}

// dz = prescale * scale * (dyn - mean(dyn) - xhat * mean(dyn*xhat))   (training; scale = gamma*rstd)
//    = prescale * scale * dyn                                           (eval: running statistics are constants)
template <typename T>
__global__ __launch_bounds__(256) void a2d_bn_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z, const float* __restrict__ prescale,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const double* __restrict__ sums, T* __restrict__ dz, long M, int C, int ld_dy, int ld_z,
                                                         int Cg, double count, int training) {
    const long n = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / C; const int c = (int)(i % C);
        const float ps = prescale ? prescale[c] : 1.f;
        const float v = to_f32(z[r * ld_z + c]) * ps;
        const float g = (v * scale[c] + shift[c] > 0.f) ? to_f32(dy[r * ld_dy + c]) : 0.f;
        float o = g;
        if (training) {
            const int gi = c / Cg, cc = c % Cg;
            const float m0 = (float)(sums[gi * 2 * Cg + cc] / count), m1 = (float)(sums[gi * 2 * Cg + Cg + cc] / count);
            o = g - m0 - (v - mean[c]) * rstd[c] * m1;
        }
        dz[r * ld_z + c] = from_f32<T>(o * scale[c] * ps);
    }
}

__global__ void a2d_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int Cg) {
    const int c = threadIdx.x;
    if (c < Cg) { dbeta[c] += (float)sums[c]; dgamma[c] += (float)sums[Cg + c]; }
}

// 8 consecutive channels per access (one 16-byte load for bf16, two for f32)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<bf16>(const bf16* p, float (&v)[8]) {
    bf16 t[8];
    *reinterpret_cast<u32x4*>(t) = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<bf16>(bf16* p, const float (&v)[8]) {
    bf16 t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16)v[e];
    *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(t);
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}
// per-thread copies of the parameters of its 8 channels (channels >= C: neutral values)
__device__ __forceinline__ void par8(const float* p, int c0, int C, float fill, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (p && c0 + e < C) ? p[c0 + e] : fill;
}

// ---- vectorised forms of the three BatchNorm kernels below: rows of `ld` = 8*CH channels (CH a power of two <= 8), 16-byte aligned;
// a thread keeps one 8-channel chunk (its parameters stay in registers) and walks rows.  (The scalar forms moved 2 bytes per lane
// and divided by C per element: 67 us for a 15 MB tensor.)
template <typename T>
__global__ __launch_bounds__(256) void a2d_stats8_kernel(const T* __restrict__ z, const T* __restrict__ dy, const float* __restrict__ prescale,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         double* __restrict__ sums, long M, int C, int ld_z, int ld_dy, int Cg, int mode, int rpb, int CH) {
    __shared__ double sh[2][256][8];
    const int chunk = threadIdx.x % CH, rsub = threadIdx.x / CH, rstep = 256 / CH, c0 = chunk * 8;
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < M ? r0 + rpb : M;
    float ps[8], mu[8], rs[8], sc[8], sf[8];
    par8(prescale, c0, C, 1.f, ps); par8(mode ? mean : nullptr, c0, C, 0.f, mu); par8(mode ? rstd : nullptr, c0, C, 0.f, rs);
    par8(mode ? scale : nullptr, c0, C, 0.f, sc); par8(mode ? shift : nullptr, c0, C, 0.f, sf);
    float a0[8], a1[8];
    double d0[8], d1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a0[e] = a1[e] = 0.f; d0[e] = d1[e] = 0.0; }
    int n = 0;
    for (long r = r0 + rsub; r < r1; r += rstep) {
        float v[8], g[8];
        ld8<T>(z + r * ld_z + c0, v);
        if (mode) ld8<T>(dy + r * ld_dy + c0, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = v[e] * ps[e];
            if (mode == 0) { a0[e] += x; a1[e] += x * x; }
            else { const float gg = (x * sc[e] + sf[e] > 0.f) ? g[e] : 0.f; a0[e] += gg; a1[e] += gg * (x - mu[e]) * rs[e]; }
        }
        if (++n == 64) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { d0[e] += a0[e]; d1[e] += a1[e]; a0[e] = a1[e] = 0.f; }
            n = 0;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[0][threadIdx.x][e] = d0[e] + a0[e]; sh[1][threadIdx.x][e] = d1[e] + a1[e]; }
    __syncthreads();
    if (threadIdx.x < CH * 8) {
        const int c = threadIdx.x, ck = c >> 3, e = c & 7;
        if (c < C) {
            double s0 = 0.0, s1 = 0.0;
            for (int k = 0; k < rstep; ++k) { s0 += sh[0][k * CH + ck][e]; s1 += sh[1][k * CH + ck][e]; }
            const int g = c / Cg, cc = c % Cg;
            atomicAdd(sums + g * 2 * Cg + cc, s0);
            atomicAdd(sums + g * 2 * Cg + Cg + cc, s1);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void a2d_bn_act8_kernel(const T* __restrict__ z, const float* __restrict__ prescale, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ res, T* __restrict__ y,
                                                          long M, int C, int ld_z, int ld_y, int CH) {
    const int chunk = threadIdx.x % CH, c0 = chunk * 8;
    float ps[8], sc[8], sf[8];
    par8(prescale, c0, C, 1.f, ps); par8(scale, c0, C, 0.f, sc); par8(shift, c0, C, 0.f, sf);
    const long rstep = (long)gridDim.x * (256 / CH);
    for (long r = (long)blockIdx.x * (256 / CH) + threadIdx.x / CH; r < M; r += rstep) {
        float v[8], rr[8];
        ld8<T>(z + r * ld_z + c0, v);
        if (res) ld8<T>(res + r * ld_y + c0, rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = fmaxf(v[e] * ps[e] * sc[e] + sf[e], 0.f);
            if (c0 + e >= C) o = 0.f;
            v[e] = res ? o + rr[e] : o;
        }
        st8<T>(y + r * ld_y + c0, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void a2d_bn_bwd8_kernel(const T* __restrict__ dy, const T* __restrict__ z, const float* __restrict__ prescale,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const double* __restrict__ sums, T* __restrict__ dz, long M, int C, int ld_dy, int ld_z,
                                                          int Cg, double count, int training, int CH) {
    const int chunk = threadIdx.x % CH, c0 = chunk * 8;
    float ps[8], mu[8], rs[8], sc[8], sf[8], m0[8], m1[8];
    par8(prescale, c0, C, 1.f, ps); par8(mean, c0, C, 0.f, mu); par8(rstd, c0, C, 0.f, rs); par8(scale, c0, C, 0.f, sc); par8(shift, c0, C, 0.f, sf);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        m0[e] = m1[e] = 0.f;
        if (training && c < C) {
            const int gi = c / Cg, cc = c % Cg;
            m0[e] = (float)(sums[gi * 2 * Cg + cc] / count); m1[e] = (float)(sums[gi * 2 * Cg + Cg + cc] / count);
        }
    }
    const long rstep = (long)gridDim.x * (256 / CH);
    for (long r = (long)blockIdx.x * (256 / CH) + threadIdx.x / CH; r < M; r += rstep) {
        float v[8], g[8];
        ld8<T>(z + r * ld_z + c0, v);
        ld8<T>(dy + r * ld_dy + c0, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = v[e] * ps[e];
            const float gg = (x * sc[e] + sf[e] > 0.f) ? g[e] : 0.f;
            const float o = training ? gg - m0[e] - (x - mu[e]) * rs[e] * m1[e] : gg;
            v[e] = (c0 + e < C) ? o * sc[e] * ps[e] : 0.f;
        }
        st8<T>(dz + r * ld_z + c0, v);
    }
}

// ------------------------------------------------------------------ convolution weight packing
// src / grad: nn.Conv2d layout [Co][Ci][3][3] (f32).  Packed matrices are row-major with row stride ld:
//  mode 0 (forward,  y = gather(x) W0^T):  dst[co][j*CiP + ci] = W[co][ci][j]
//  mode 1 (data gradient, dx = gather(dy) W1^T):  dst[ci][j*CoP + co] = W[co][ci][8 - j]     (same row maps, taps mirrored)
//  mode 2 (weight gradient back to the master layout):  W[co][ci][j] += src[co][j*CiP + ci]   (src f32)
template <typename T>
__global__ __launch_bounds__(256) void a2d_pack_w_kernel(const float* __restrict__ src, T* __restrict__ dst, float* __restrict__ grad,
                                                         int Co, int Ci, int CP, int ld, int mode) {
    const int n = Co * Ci * 9;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int j = i % 9, ci = (i / 9) % Ci, co = i / (9 * Ci);
        if (mode == 0) dst[(long)co * ld + j * CP + ci] = from_f32<T>(src[i]);
        else if (mode == 1) dst[(long)ci * ld + (8 - j) * CP + co] = from_f32<T>(src[i]);
        else grad[i] += src[(long)co * ld + j * CP + ci];
    }
}

// ------------------------------------------------------------------ time attention
// One thread per query frame i of a (batch b, head h) plane; keys / values stream through LDS in tiles of KT frames.
// Pass 1: row maximum and normaliser (lse).  Pass 2: P = exp(s - lse), dropout, O += P v.
template <int F> struct Plane {
    // element (t, f) of channel ch of a [M][ld] channels-last tensor, plane of batch b
    static __device__ __forceinline__ long at(int t, int f, int b, int B, int ld, int ch) { return ((long)(t * B + b) * F + f) * ld + ch; }
};

// dot product over F with four independent accumulation chains (a single chain of 20 dependent FMAs halves the VALU rate)
template <int F>
__device__ __forceinline__ float dotF(const float (&a)[F], const float* __restrict__ b) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int f = 0; f + 3 < F; f += 4) { s0 += a[f] * b[f]; s1 += a[f + 1] * b[f + 1]; s2 += a[f + 2] * b[f + 2]; s3 += a[f + 3] * b[f + 3]; }
#pragma unroll
    for (int f = F & ~3; f < F; ++f) s0 += a[f] * b[f];
    return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ uint64_t tdrop_index(int bh, int i, int j, int T, int Tp) { return ((uint64_t)bh * T + i) * Tp + j; }

constexpr int KT = 64;

template <typename T, int F>
__global__ __launch_bounds__(128) void a2d_time_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ cat, float* __restrict__ lse_out,
                                                           int B, int Tn, float p_drop, unsigned long long seed) {
    __shared__ float sk[KT][F], sv[KT][F];
    const int bh = blockIdx.y, b = bh / HEADS, h = bh % HEADS;
    const int i = blockIdx.x * 128 + threadIdx.x;
    const bool live = i < Tn;
    const int ii = live ? i : Tn - 1;
    float q[F], o[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { q[f] = to_f32(qkv[Plane<F>::at(ii, f, b, B, QKV_C, h)]); o[f] = 0.f; }
    const int Tp = (Tn + 3) & ~3;
    const uint32_t th16 = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float inv_keep = 1.f / (1.f - p_drop);
    float m = -INFINITY, l = 0.f, lse = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        for (int j0 = 0; j0 < Tn; j0 += KT) {
            __syncthreads();
            for (int e = threadIdx.x; e < KT * F; e += 128) {
                const int jj = e / F, f = e % F, j = j0 + jj;
                sk[jj][f] = j < Tn ? to_f32(qkv[Plane<F>::at(j, f, b, B, QKV_C, HEADS + h)]) : 0.f;
                if (pass) sv[jj][f] = j < Tn ? to_f32(qkv[Plane<F>::at(j, f, b, B, QKV_C, 2 * HEADS + h)]) : 0.f;
            }
            __syncthreads();
            const int nj = min(KT, Tn - j0);
            u32x2 hq = {0, 0};
#pragma unroll 4
            for (int jj = 0; jj < nj; ++jj) {
                const float s = dotF<F>(q, sk[jj]);
                if (pass == 0) {
                    const float mn = fmaxf(m, s);
                    l = l * __expf(m - mn) + __expf(s - mn);
                    m = mn;
                } else {
                    float p = __expf(s - lse);
                    if (p_drop > 0.f) {
                        const int j = j0 + jj;
                        if ((j & 3) == 0) hq = drop_hash4(seed, tdrop_index(bh, ii, j, Tn, Tp) >> 2);
                        p = drop_field(hq, j & 3) >= th16 ? p * inv_keep : 0.f;
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) o[f] += p * sv[jj][f];
                }
            }
        }
        if (pass == 0) lse = m + __logf(l);
    }
    if (live) {
        lse_out[(long)bh * Tn + i] = lse;
#pragma unroll
        for (int f = 0; f < F; ++f) cat[Plane<F>::at(i, f, b, B, CAT_C, h)] = from_f32<T>(o[f]);
    }
}

// backward, query side: delta_i = dO_i . O_i ; dq_i = sum_j dS_ij k_j, dS = P (dP - delta), dP = mask/(1-p) * (dO_i . v_j)
template <typename T, int F>
__global__ __launch_bounds__(128) void a2d_time_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ cat, const T* __restrict__ dcat,
                                                              const float* __restrict__ lse_in, float* __restrict__ delta_out, T* __restrict__ dqkv,
                                                              int B, int Tn, float p_drop, unsigned long long seed) {
    __shared__ float sk[KT][F], sv[KT][F];
    const int bh = blockIdx.y, b = bh / HEADS, h = bh % HEADS;
    const int i = blockIdx.x * 128 + threadIdx.x;
    const bool live = i < Tn;
    const int ii = live ? i : Tn - 1;
    float q[F], dO[F], dq[F];
    float delta = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        q[f] = to_f32(qkv[Plane<F>::at(ii, f, b, B, QKV_C, h)]);
        dO[f] = to_f32(dcat[Plane<F>::at(ii, f, b, B, CAT_C, h)]);
        delta += dO[f] * to_f32(cat[Plane<F>::at(ii, f, b, B, CAT_C, h)]);
        dq[f] = 0.f;
    }
    const float lse = lse_in[(long)bh * Tn + ii];
    const int Tp = (Tn + 3) & ~3;
    const uint32_t th16 = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float inv_keep = 1.f / (1.f - p_drop);
    for (int j0 = 0; j0 < Tn; j0 += KT) {
        __syncthreads();
        for (int e = threadIdx.x; e < KT * F; e += 128) {
            const int jj = e / F, f = e % F, j = j0 + jj;
            sk[jj][f] = j < Tn ? to_f32(qkv[Plane<F>::at(j, f, b, B, QKV_C, HEADS + h)]) : 0.f;
            sv[jj][f] = j < Tn ? to_f32(qkv[Plane<F>::at(j, f, b, B, QKV_C, 2 * HEADS + h)]) : 0.f;
        }
        __syncthreads();
        const int nj = min(KT, Tn - j0);
        u32x2 hq = {0, 0};
#pragma unroll 4
        for (int jj = 0; jj < nj; ++jj) {
            const float s = dotF<F>(q, sk[jj]);
            float dp = dotF<F>(dO, sv[jj]);
            const float p = __expf(s - lse);
            if (p_drop > 0.f) {
                const int j = j0 + jj;
                if ((j & 3) == 0) hq = drop_hash4(seed, tdrop_index(bh, ii, j, Tn, Tp) >> 2);
                dp = drop_field(hq, j & 3) >= th16 ? dp * inv_keep : 0.f;
            }
            const float ds = p * (dp - delta);
#pragma unroll
            for (int f = 0; f < F; ++f) dq[f] += ds * sk[jj][f];
        }
    }
    if (live) {
        delta_out[(long)bh * Tn + i] = delta;
#pragma unroll
        for (int f = 0; f < F; ++f) dqkv[Plane<F>::at(i, f, b, B, QKV_C, h)] = from_f32<T>(dq[f]);
    }
}

// backward, key side: one thread per key frame j; queries (q, dO, lse, delta) stream through LDS.
template <typename T, int F>
__global__ __launch_bounds__(128) void a2d_time_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dcat, const float* __restrict__ lse_in,
                                                               const float* __restrict__ delta_in, T* __restrict__ dqkv, int B, int Tn,
                                                               float p_drop, unsigned long long seed) {
    __shared__ float sq[KT][F], sdo[KT][F], sl[KT], sd[KT];
    const int bh = blockIdx.y, b = bh / HEADS, h = bh % HEADS;
    const int j = blockIdx.x * 128 + threadIdx.x;
    const bool live = j < Tn;
    const int jc = live ? j : Tn - 1;
    float k[F], v[F], dk[F], dv[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        k[f] = to_f32(qkv[Plane<F>::at(jc, f, b, B, QKV_C, HEADS + h)]);
        v[f] = to_f32(qkv[Plane<F>::at(jc, f, b, B, QKV_C, 2 * HEADS + h)]);
        dk[f] = 0.f; dv[f] = 0.f;
    }
    const int Tp = (Tn + 3) & ~3;
    const uint32_t th = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f);
    const float inv_keep = 1.f / (1.f - p_drop);
    for (int i0 = 0; i0 < Tn; i0 += KT) {
        __syncthreads();
        for (int e = threadIdx.x; e < KT * F; e += 128) {
            const int iq = e / F, f = e % F, i = i0 + iq;
            sq[iq][f] = i < Tn ? to_f32(qkv[Plane<F>::at(i, f, b, B, QKV_C, h)]) : 0.f;
            sdo[iq][f] = i < Tn ? to_f32(dcat[Plane<F>::at(i, f, b, B, CAT_C, h)]) : 0.f;
        }
        if (threadIdx.x < KT) {
            const int i = i0 + threadIdx.x;
            sl[threadIdx.x] = i < Tn ? lse_in[(long)bh * Tn + i] : 0.f;
            sd[threadIdx.x] = i < Tn ? delta_in[(long)bh * Tn + i] : 0.f;
        }
        __syncthreads();
        const int ni = min(KT, Tn - i0);
        for (int iq = 0; iq < ni; ++iq) {
            const float s = dotF<F>(k, sq[iq]);
            float dp = dotF<F>(v, sdo[iq]);
            const float p = __expf(s - sl[iq]);
            float pd = p;
            if (p_drop > 0.f) {
                const bool keep = dropout_keep(seed, tdrop_index(bh, i0 + iq, jc, Tn, Tp), th);
                pd = keep ? p * inv_keep : 0.f;
                dp = keep ? dp * inv_keep : 0.f;
            }
            const float ds = p * (dp - sd[iq]);
#pragma unroll
            for (int f = 0; f < F; ++f) { dv[f] += pd * sdo[iq][f]; dk[f] += ds * sq[iq][f]; }
        }
    }
    if (live) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            dqkv[Plane<F>::at(j, f, b, B, QKV_C, HEADS + h)] = from_f32<T>(dk[f]);
            dqkv[Plane<F>::at(j, f, b, B, QKV_C, 2 * HEADS + h)] = from_f32<T>(dv[f]);
        }
    }
}

// ------------------------------------------------------------------ frequency attention
// One workgroup per (b, h) plane.  S[f1][f2] = sum_t q[t][f1] k[t][f2] (all frames), A = softmax over f2, dropout,
// out[t][f1] = sum_f2 A[f1][f2] v[t][f2].
constexpr int FT = 64;           // frames per staged chunk

// acc[f1][f2] += sum_t X[t][f1] Y[t][f2] over the whole plane; threads f1*F + f2 < F*F own one entry
template <typename T, int F>
__device__ __forceinline__ float plane_outer_sum(const T* __restrict__ X, int ldx, int chx, const T* __restrict__ Y, int ldy, int chy,
                                                 int b, int B, int Tn, float (*sx)[F], float (*sy)[F]) {
    const int f1 = threadIdx.x / F, f2 = threadIdx.x % F;
    float acc = 0.f;
    for (int t0 = 0; t0 < Tn; t0 += FT) {
        __syncthreads();
        for (int e = threadIdx.x; e < FT * F; e += blockDim.x) {
            const int tt = e / F, f = e % F, t = t0 + tt;
            sx[tt][f] = t < Tn ? to_f32(X[Plane<F>::at(t, f, b, B, ldx, chx)]) : 0.f;
            sy[tt][f] = t < Tn ? to_f32(Y[Plane<F>::at(t, f, b, B, ldy, chy)]) : 0.f;
        }
        __syncthreads();
        if (threadIdx.x < F * F) {
#pragma unroll 8
            for (int tt = 0; tt < FT; ++tt) acc += sx[tt][f1] * sy[tt][f2];
        }
    }
    return acc;
}

template <typename T, int F>
__global__ __launch_bounds__(512) void a2d_freq_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ cat, float* __restrict__ A_out,
                                                           int B, int Tn, float p_drop, unsigned long long seed) {
    static_assert(F * F <= 512, "one thread per score entry");
    __shared__ float sx[FT][F], sy[FT][F], sA[F][F + 1];
    const int bh = blockIdx.x, b = bh / HEADS, h = bh % HEADS;
    const float s = plane_outer_sum<T, F>(qkv, QKV_C, h, qkv, QKV_C, HEADS + h, b, B, Tn, sx, sy);
    if (threadIdx.x < F * F) sA[threadIdx.x / F][threadIdx.x % F] = s;
    __syncthreads();
    if (threadIdx.x < F) {
        const int f1 = threadIdx.x;
        float m = -INFINITY, l = 0.f;
        for (int f2 = 0; f2 < F; ++f2) m = fmaxf(m, sA[f1][f2]);
        for (int f2 = 0; f2 < F; ++f2) l += __expf(sA[f1][f2] - m);
        const uint32_t th = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f);
        const float inv_keep = 1.f / (1.f - p_drop);
        for (int f2 = 0; f2 < F; ++f2) {
            const float a = __expf(sA[f1][f2] - m) / l;
            A_out[((long)bh * F + f1) * F + f2] = a;
            sA[f1][f2] = (p_drop > 0.f && !dropout_keep(seed, ((uint64_t)bh * F + f1) * F + f2, th)) ? 0.f : a * (p_drop > 0.f ? inv_keep : 1.f);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < Tn; t += blockDim.x) {
        float v[F];
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = to_f32(qkv[Plane<F>::at(t, f, b, B, QKV_C, 2 * HEADS + h)]);
#pragma unroll 1
        for (int f1 = 0; f1 < F; ++f1) {
            float o = 0.f;
#pragma unroll
            for (int f2 = 0; f2 < F; ++f2) o += sA[f1][f2] * v[f2];
            cat[Plane<F>::at(t, f1, b, B, CAT_C, HEADS + h)] = from_f32<T>(o);
        }
    }
}

// backward: dAd[f1][f2] = sum_t dO[t][f1] v[t][f2];  dA = mask/(1-p) dAd;  dS = A (dA - rowsum(dA A));
// dv[t][f2] = sum_f1 Ad[f1][f2] dO[t][f1];  dq[t][f1] = sum_f2 dS[f1][f2] k[t][f2];  dk[t][f2] = sum_f1 dS[f1][f2] q[t][f1].
// The three results are ADDED to dqkv (the time-attention backward wrote its part first, same stream).
template <typename T, int F>
__global__ __launch_bounds__(512) void a2d_freq_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dcat, const float* __restrict__ A_in,
                                                           T* __restrict__ dqkv, int B, int Tn, float p_drop, unsigned long long seed) {
    __shared__ float sx[FT][F], sy[FT][F], sAd[F][F + 1], sdS[F][F + 1];
    const int bh = blockIdx.x, b = bh / HEADS, h = bh % HEADS;
    const float dAd = plane_outer_sum<T, F>(dcat, CAT_C, HEADS + h, qkv, QKV_C, 2 * HEADS + h, b, B, Tn, sx, sy);
    if (threadIdx.x < F * F) sdS[threadIdx.x / F][threadIdx.x % F] = dAd;
    __syncthreads();
    if (threadIdx.x < F) {
        const int f1 = threadIdx.x;
        const uint32_t th = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f);
        const float inv_keep = 1.f / (1.f - p_drop);
        float a[F], dA[F], rs = 0.f;
#pragma unroll
        for (int f2 = 0; f2 < F; ++f2) {
            a[f2] = A_in[((long)bh * F + f1) * F + f2];
            const bool keep = !(p_drop > 0.f) || dropout_keep(seed, ((uint64_t)bh * F + f1) * F + f2, th);
            const float kf = keep ? (p_drop > 0.f ? inv_keep : 1.f) : 0.f;
            sAd[f1][f2] = a[f2] * kf;
            dA[f2] = sdS[f1][f2] * kf;
            rs += dA[f2] * a[f2];
        }
#pragma unroll
        for (int f2 = 0; f2 < F; ++f2) sdS[f1][f2] = a[f2] * (dA[f2] - rs);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < Tn; t += blockDim.x) {
        float q[F], k[F], dO[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            q[f] = to_f32(qkv[Plane<F>::at(t, f, b, B, QKV_C, h)]);
            k[f] = to_f32(qkv[Plane<F>::at(t, f, b, B, QKV_C, HEADS + h)]);
            dO[f] = to_f32(dcat[Plane<F>::at(t, f, b, B, CAT_C, HEADS + h)]);
        }
#pragma unroll 1
        for (int f = 0; f < F; ++f) {
            float dq = 0.f, dk = 0.f, dv = 0.f;
#pragma unroll
            for (int g = 0; g < F; ++g) { dq += sdS[f][g] * k[g]; dk += sdS[g][f] * q[g]; dv += sAd[g][f] * dO[g]; }
            const long iq = Plane<F>::at(t, f, b, B, QKV_C, h), ik = Plane<F>::at(t, f, b, B, QKV_C, HEADS + h),
                       iv = Plane<F>::at(t, f, b, B, QKV_C, 2 * HEADS + h);
            dqkv[iq] = from_f32<T>(to_f32(dqkv[iq]) + dq);
            dqkv[ik] = from_f32<T>(to_f32(dqkv[ik]) + dk);
            dqkv[iv] = from_f32<T>(to_f32(dqkv[iv]) + dv);
        }
    }
}

// ------------------------------------------------------------------ weight gradient of the two 3x3 convolutions, all nine taps in one pass
//   dW[co][ci][kh][kw] += sum over pixels (t, b, f) of dY[(t,b,f)][co] * X[(t + kh - 1, b, f + kw - 1)][ci]
// (as nine gathered TN products this read both operands nine times and took 2.0 ms per update for the two blocks).
// A workgroup walks units (batch b, TT consecutive frames): X with a one-pixel halo and dY are staged in LDS as f32; a thread owns
// a 4 (co) x 4 (ci) block for the three taps of one kernel row kh (48 accumulators) and, when CO*CI is small, one of PS frame
// subsets.  Accumulators live across units; the workgroup's partial sums go to a workspace that a second kernel reduces into the
// master layout [CO][CI][3][3].
template <typename T, int CO, int CI>
__global__ __launch_bounds__(256) void a2d_conv_wgrad_kernel(const T* __restrict__ dY, int ld_dy, const T* __restrict__ X, int ld_x,
                                                             float* __restrict__ ws, int B, int Tn, int F, int TT, int units) {
    extern __shared__ float a2d_lds[];
    constexpr int NCI4 = CI / 4, NCO4 = CO / 4, NCOMBO = NCI4 * NCO4, PS = 256 / (NCOMBO * 3) > 0 ? 256 / (NCOMBO * 3) : 1;
    static_assert(NCOMBO * 3 <= 256, "one thread per (4x4 block, kernel row)");
    const int FW = F + 2;
    float* sx = a2d_lds;                                   // [(TT+2)][FW][CI]
    float* sdy = a2d_lds + (TT + 2) * FW * CI;             // [TT][F][CO]
    const int tid = threadIdx.x;
    const bool active = tid < NCOMBO * 3 * PS;
    const int ps = tid / (NCOMBO * 3), rem = tid % (NCOMBO * 3), kh = rem / NCOMBO, combo = rem % NCOMBO;
    const int ci4 = combo % NCI4, co4 = combo / NCI4;
    float acc[3][4][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][i][j] = 0.f;
    for (int e = tid; e < (TT + 2) * FW * CI; e += 256) sx[e] = 0.f;          // halo columns stay zero for the whole kernel
    const int nchunk = (Tn + TT - 1) / TT;
    constexpr int E = Elem<T>::PER16;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int b = u / nchunk, t0 = (u % nchunk) * TT;
        __syncthreads();
        // X rows t0-1 .. t0+TT of batch b (zeros outside the plane), channels in 16-byte chunks
        for (int e = tid; e < (TT + 2) * F * (CI / E); e += 256) {
            const int ch = e % (CI / E), f = (e / (CI / E)) % F, tr = e / ((CI / E) * F);
            const int t = t0 - 1 + tr;
            T tmp[E];
            if (t >= 0 && t < Tn) *reinterpret_cast<u32x4*>(tmp) = *reinterpret_cast<const u32x4*>(X + ((size_t)(t * B + b) * F + f) * ld_x + ch * E);
            float* dst = sx + ((size_t)tr * FW + f + 1) * CI + ch * E;
#pragma unroll
            for (int k = 0; k < E; ++k) dst[k] = (t >= 0 && t < Tn) ? to_f32(tmp[k]) : 0.f;
        }
        for (int e = tid; e < TT * F * (CO / E); e += 256) {
            const int ch = e % (CO / E), f = (e / (CO / E)) % F, tt = e / ((CO / E) * F);
            const int t = t0 + tt;
            T tmp[E];
            if (t < Tn) *reinterpret_cast<u32x4*>(tmp) = *reinterpret_cast<const u32x4*>(dY + ((size_t)(t * B + b) * F + f) * ld_dy + ch * E);
            float* dst = sdy + ((size_t)tt * F + f) * CO + ch * E;
#pragma unroll
            for (int k = 0; k < E; ++k) dst[k] = t < Tn ? to_f32(tmp[k]) : 0.f;
        }
        __syncthreads();
        if (active) {
            for (int tt = ps; tt < TT; tt += PS) {                       // pixel subsets split the frames; f runs with a sliding window
                const float* dyr = sdy + (size_t)tt * F * CO + co4 * 4;
                const float* xr = sx + (size_t)(tt + kh) * FW * CI + ci4 * 4;
                f32x4 x0 = *reinterpret_cast<const f32x4*>(xr), x1 = *reinterpret_cast<const f32x4*>(xr + CI);
#pragma unroll 4
                for (int f = 0; f < F; ++f) {
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dyr + (size_t)f * CO);
                    const f32x4 x2 = *reinterpret_cast<const f32x4*>(xr + (size_t)(f + 2) * CI);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[0][i][j] += d4[i] * x0[j];
                            acc[1][i][j] += d4[i] * x1[j];
                            acc[2][i][j] += d4[i] * x2[j];
                        }
                    x0 = x1; x1 = x2;
                }
            }
        }
    }
    // partial sums of this workgroup -> ws[blockIdx.x][CO][CI][9]; PS frame subsets add up through LDS first.  (One f32 atomic per
    // weight per workgroup was 4.7 M atomics on 9 K addresses: 200-280 us of same-address contention, measured.)
    __syncthreads();
    float* red = a2d_lds;                                  // [CO*CI*9] floats (<= 36 KB, fits in the staging area)
    for (int rep = 0; rep < PS; ++rep) {
        if (active && ps == rep) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int idx = ((co4 * 4 + i) * CI + ci4 * 4 + j) * 9 + kh * 3 + kw;
                        red[idx] = (rep == 0 ? 0.f : red[idx]) + acc[kw][i][j];
                    }
        }
        __syncthreads();
    }
    float* dst = ws + (size_t)blockIdx.x * CO * CI * 9;
    for (int e = tid; e < CO * CI * 9; e += 256) dst[e] = red[e];
}

__global__ __launch_bounds__(256) void a2d_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int n_ws, int n_out, int groups) {
    // dW[i] += sum_g ws[g][i] for the first n_out (real output channels come first in the [CO][CI][9] order); blockIdx.y takes a
    // slice of 32 groups (16 atomics per weight in total)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const int g0 = blockIdx.y * 32, g1 = min(groups, g0 + 32);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int g = g0;
    for (; g + 3 < g1; g += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += ws[(size_t)(g + k) * n_ws + i];
    }
    for (; g < g1; ++g) s[0] += ws[(size_t)g * n_ws + i];
    if (g0 < g1) atomicAdd(dW + i, (s[0] + s[1]) + (s[2] + s[3]));
}

// ------------------------------------------------------------------ (batch, head) planes <-> the head-major layout of the MFMA attention kernels
// chl: channels-last [M][ld] over pixel rows (t, b, f); pl: [G][T][B][HEADS*32], plane of head h in columns 32h .. 32h+F-1 (zero beyond F).
// dir 0: pl = planes of channels ch0 + 4g + h of chl;  dir 1: chl channels <- pl.  With these the time attention of a block is
// s2t_attn_fwd / s2t_attn_bwd at head_dim 32, scale 1 (attention.hip) instead of the one-thread-per-query kernels above.
template <typename T>
__global__ __launch_bounds__(256) void a2d_planes_kernel(T* __restrict__ chl, T* __restrict__ pl, int G, int ch0, int ld, int B, int Tn, int F, int dir) {
    const long n = (long)G * Tn * B * HEADS * 32;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int f = (int)(i & 31), h = (int)((i >> 5) & 3);
        const long tb = (i >> 7) % ((long)Tn * B);
        const int g = (int)((i >> 7) / ((long)Tn * B));
        const long src = (tb * F + f) * ld + ch0 + 4 * g + h;
        if (dir == 0) pl[i] = f < F ? chl[src] : from_f32<T>(0.f);
        else if (f < F) chl[src] = pl[i];
    }
}

// rows of 16 / 32 / 64 channels (8-channel chunks, a power-of-two number of them), both tensors 16-byte aligned
inline bool a2d_vec_ok(int ld_a, const void* a, int ld_b, const void* b) {
    const bool shape = (ld_a == 16 || ld_a == 32 || ld_a == 64 || ld_a == 8) && ld_b == ld_a;
    return shape && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

inline int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

#define A2D_DISPATCH_F(F, ...)                      \
    do {                                            \
        if ((F) == 20) { constexpr int FQ = 20; __VA_ARGS__; } \
        else if ((F) == 10) { constexpr int FQ = 10; __VA_ARGS__; } \
        else if ((F) == 21) { constexpr int FQ = 21; __VA_ARGS__; } \
        else return S2T_ENOTSUP;                    \
    } while (0)

}  // namespace

extern "C" int s2t_a2d_chan_stats(int dtype, const void* z, const void* dy, const float* prescale, const float* mean, const float* rstd,
                                  const float* scale, const float* shift, double* sums, long M, int C, int ld_z, int ld_dy, int Cg,
                                  int mode, void* stream) {
    if (M <= 0) return S2T_OK;
    if (!z || !sums || C <= 0 || C > 64 || Cg <= 0 || C % Cg || ld_z < C || (mode && (!dy || !mean || !rstd || !scale || !shift || ld_dy < C)))
        return S2T_EINVAL;
    long rpb = (M + 1023) / 1024; if (rpb < 256) rpb = 256;
    const dim3 grid((unsigned)((M + rpb - 1) / rpb));
    hipStream_t st = (hipStream_t)stream;
    if (a2d_vec_ok(ld_z, z, mode ? ld_dy : ld_z, dy)) {
        const int CH = ld_z / 8;
        A2D_DISPATCH_T(dtype,
            hipLaunchKernelGGL(a2d_stats8_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)z, (const bf16*)dy, prescale, mean, rstd, scale, shift, sums, M, C, ld_z, ld_dy, Cg, mode, (int)rpb, CH),
            hipLaunchKernelGGL(a2d_stats8_kernel<float>, grid, dim3(256), 0, st, (const float*)z, (const float*)dy, prescale, mean, rstd, scale, shift, sums, M, C, ld_z, ld_dy, Cg, mode, (int)rpb, CH));
        S2T_LAUNCH_CHECK();
        return S2T_OK;
    }
    A2D_DISPATCH_T(dtype,
        hipLaunchKernelGGL(a2d_stats_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)z, (const bf16*)dy, prescale, mean, rstd, scale, shift, sums, M, C, ld_z, ld_dy, Cg, mode, (int)rpb),
        hipLaunchKernelGGL(a2d_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)z, (const float*)dy, prescale, mean, rstd, scale, shift, sums, M, C, ld_z, ld_dy, Cg, mode, (int)rpb));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_bn_act(int dtype, const void* z, const float* prescale, const float* scale, const float* shift, const void* res,
                              void* y, long M, int C, int ld_z, int ld_y, void* stream) {
    if (M <= 0) return S2T_OK;
    if (!z || !scale || !shift || !y || C <= 0 || ld_z < C || ld_y < C) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (ld_z == ld_y && a2d_vec_ok(ld_z, z, ld_y, y) && a2d_vec_ok(ld_y, res, ld_y, res)) {
        const int CH = ld_z / 8;
        const int blocks = grid_for(M * CH);
        A2D_DISPATCH_T(dtype,
            hipLaunchKernelGGL(a2d_bn_act8_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)z, prescale, scale, shift, (const bf16*)res, (bf16*)y, M, C, ld_z, ld_y, CH),
            hipLaunchKernelGGL(a2d_bn_act8_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)z, prescale, scale, shift, (const float*)res, (float*)y, M, C, ld_z, ld_y, CH));
        S2T_LAUNCH_CHECK();
        return S2T_OK;
    }
    A2D_DISPATCH_T(dtype,
        hipLaunchKernelGGL(a2d_bn_act_kernel<bf16>, dim3(grid_for(M * C)), dim3(256), 0, st, (const bf16*)z, prescale, scale, shift, (const bf16*)res, (bf16*)y, M, C, ld_z, ld_y),
        hipLaunchKernelGGL(a2d_bn_act_kernel<float>, dim3(grid_for(M * C)), dim3(256), 0, st, (const float*)z, prescale, scale, shift, (const float*)res, (float*)y, M, C, ld_z, ld_y));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_bn_bwd(int dtype, const void* dy, const void* z, const float* prescale, const float* mean, const float* rstd,
                              const float* scale, const float* shift, const double* sums, void* dz, long M, int C, int ld_dy, int ld_z,
                              int Cg, double count, int training, void* stream) {
    if (M <= 0) return S2T_OK;
    if (!dy || !z || !mean || !rstd || !scale || !shift || !sums || !dz || C <= 0 || Cg <= 0 || C % Cg || ld_dy < C || ld_z < C || count <= 0)
        return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (ld_dy == ld_z && a2d_vec_ok(ld_z, z, ld_dy, dy) && a2d_vec_ok(ld_z, dz, ld_z, dz)) {
        const int CH = ld_z / 8;
        const int blocks = grid_for(M * CH);
        A2D_DISPATCH_T(dtype,
            hipLaunchKernelGGL(a2d_bn_bwd8_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)dy, (const bf16*)z, prescale, mean, rstd, scale, shift, sums, (bf16*)dz, M, C, ld_dy, ld_z, Cg, count, training, CH),
            hipLaunchKernelGGL(a2d_bn_bwd8_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (const float*)z, prescale, mean, rstd, scale, shift, sums, (float*)dz, M, C, ld_dy, ld_z, Cg, count, training, CH));
        S2T_LAUNCH_CHECK();
        return S2T_OK;
    }
    A2D_DISPATCH_T(dtype,
        hipLaunchKernelGGL(a2d_bn_bwd_kernel<bf16>, dim3(grid_for(M * C)), dim3(256), 0, st, (const bf16*)dy, (const bf16*)z, prescale, mean, rstd, scale, shift, sums, (bf16*)dz, M, C, ld_dy, ld_z, Cg, count, training),
        hipLaunchKernelGGL(a2d_bn_bwd_kernel<float>, dim3(grid_for(M * C)), dim3(256), 0, st, (const float*)dy, (const float*)z, prescale, mean, rstd, scale, shift, sums, (float*)dz, M, C, ld_dy, ld_z, Cg, count, training));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_param_grads(const double* sums, float* dgamma, float* dbeta, int Cg, void* stream) {
    if (!sums || !dgamma || !dbeta || Cg <= 0 || Cg > 64) return S2T_EINVAL;
    hipLaunchKernelGGL(a2d_param_grads_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, dgamma, dbeta, Cg);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_pack_w(int dst_dtype, const float* src, void* dst, float* grad, int Co, int Ci, int CP, int ld, int mode, void* stream) {
    if (Co <= 0 || Ci <= 0) return S2T_OK;
    if (!src || mode < 0 || mode > 2 || (mode < 2 && !dst) || (mode == 2 && !grad) || CP < (mode == 1 ? Co : Ci) || ld < 9 * CP) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid_for((long)Co * Ci * 9);
    if (mode == 2) hipLaunchKernelGGL(a2d_pack_w_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float*)nullptr, grad, Co, Ci, CP, ld, mode);
    else A2D_DISPATCH_T(dst_dtype,
        hipLaunchKernelGGL(a2d_pack_w_kernel<bf16>, dim3(blocks), dim3(256), 0, st, src, (bf16*)dst, (float*)nullptr, Co, Ci, CP, ld, mode),
        hipLaunchKernelGGL(a2d_pack_w_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float*)dst, (float*)nullptr, Co, Ci, CP, ld, mode));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_time_fwd(int dtype, const void* qkv, void* cat, float* lse, int B, int T, int F, float p_drop,
                                unsigned long long seed, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!qkv || !cat || !lse || p_drop < 0.f || p_drop >= 1.f) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, B * HEADS);
    ProfScope prof("attn2d", st, 4.0 * B * HEADS * (double)T * T * F, 0.0);
    A2D_DISPATCH_F(F, A2D_DISPATCH_T(dtype,
        hipLaunchKernelGGL((a2d_time_fwd_kernel<bf16, FQ>), grid, dim3(128), 0, st, (const bf16*)qkv, (bf16*)cat, lse, B, T, p_drop, seed),
        hipLaunchKernelGGL((a2d_time_fwd_kernel<float, FQ>), grid, dim3(128), 0, st, (const float*)qkv, (float*)cat, lse, B, T, p_drop, seed)));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_time_bwd(int dtype, const void* qkv, const void* cat, const void* dcat, const float* lse, float* delta, void* dqkv,
                                int B, int T, int F, float p_drop, unsigned long long seed, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!qkv || !cat || !dcat || !lse || !delta || !dqkv || p_drop < 0.f || p_drop >= 1.f) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, B * HEADS);
    ProfScope prof("attn2d", st, 10.0 * B * HEADS * (double)T * T * F, 0.0);
    A2D_DISPATCH_F(F, A2D_DISPATCH_T(dtype,
        { hipLaunchKernelGGL((a2d_time_bwd_dq_kernel<bf16, FQ>), grid, dim3(128), 0, st, (const bf16*)qkv, (const bf16*)cat, (const bf16*)dcat, lse, delta, (bf16*)dqkv, B, T, p_drop, seed);
          hipLaunchKernelGGL((a2d_time_bwd_dkv_kernel<bf16, FQ>), grid, dim3(128), 0, st, (const bf16*)qkv, (const bf16*)dcat, lse, delta, (bf16*)dqkv, B, T, p_drop, seed); },
        { hipLaunchKernelGGL((a2d_time_bwd_dq_kernel<float, FQ>), grid, dim3(128), 0, st, (const float*)qkv, (const float*)cat, (const float*)dcat, lse, delta, (float*)dqkv, B, T, p_drop, seed);
          hipLaunchKernelGGL((a2d_time_bwd_dkv_kernel<float, FQ>), grid, dim3(128), 0, st, (const float*)qkv, (const float*)dcat, lse, delta, (float*)dqkv, B, T, p_drop, seed); }));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_freq_fwd(int dtype, const void* qkv, void* cat, float* A, int B, int T, int F, float p_drop,
                                unsigned long long seed, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!qkv || !cat || !A || p_drop < 0.f || p_drop >= 1.f) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    A2D_DISPATCH_F(F, A2D_DISPATCH_T(dtype,
        hipLaunchKernelGGL((a2d_freq_fwd_kernel<bf16, FQ>), dim3(B * HEADS), dim3(512), 0, st, (const bf16*)qkv, (bf16*)cat, A, B, T, p_drop, seed),
        hipLaunchKernelGGL((a2d_freq_fwd_kernel<float, FQ>), dim3(B * HEADS), dim3(512), 0, st, (const float*)qkv, (float*)cat, A, B, T, p_drop, seed)));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_freq_bwd(int dtype, const void* qkv, const void* dcat, const float* A, void* dqkv, int B, int T, int F,
                                float p_drop, unsigned long long seed, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!qkv || !dcat || !A || !dqkv || p_drop < 0.f || p_drop >= 1.f) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    A2D_DISPATCH_F(F, A2D_DISPATCH_T(dtype,
        hipLaunchKernelGGL((a2d_freq_bwd_kernel<bf16, FQ>), dim3(B * HEADS), dim3(512), 0, st, (const bf16*)qkv, (const bf16*)dcat, A, (bf16*)dqkv, B, T, p_drop, seed),
        hipLaunchKernelGGL((a2d_freq_bwd_kernel<float, FQ>), dim3(B * HEADS), dim3(512), 0, st, (const float*)qkv, (const float*)dcat, A, (float*)dqkv, B, T, p_drop, seed)));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_conv_wgrad(int dtype, const void* dY, int ld_dy, const void* X, int ld_x, float* dW, float* ws, int CO, int CI,
                                  int B, int T, int F, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!dY || !X || !dW || !ws || F <= 0 || ld_dy < CO || ld_x < CI) return S2T_EINVAL;
    const int E = dtype == S2T_BF16 ? 8 : 4;
    if ((ld_dy % E) || (ld_x % E) || (((uintptr_t)dY | (uintptr_t)X) & 15)) return S2T_ENOTSUP;
    const int TT = 6;
    const int units = B * ((T + TT - 1) / TT);
    const int grid = units < S2T_A2D_WGRAD_GROUPS ? units : S2T_A2D_WGRAD_GROUPS;
    hipStream_t st = (hipStream_t)stream;
    // (CO padded to 16, CI) = (16, 64): in_proj with its 12 real output channels; (64, 8): out_proj
#define A2D_WGRAD(TT_, COp, CIp, co_real)                                                                                              \
    do {                                                                                                                               \
        const size_t lds = ((size_t)(TT + 2) * (F + 2) * CIp + (size_t)TT * F * COp) * 4;                                              \
        if (lds > 150 * 1024) return S2T_ENOTSUP;                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&a2d_conv_wgrad_kernel<TT_, COp, CIp>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((a2d_conv_wgrad_kernel<TT_, COp, CIp>), dim3(grid), dim3(256), lds, st, (const TT_*)dY, ld_dy, (const TT_*)X, ld_x, ws, B, T, F, TT, units); \
        hipLaunchKernelGGL(a2d_wgrad_reduce_kernel, dim3((co_real * CIp * 9 + 255) / 256, (grid + 31) / 32), dim3(256), 0, st, ws, dW, COp * CIp * 9, co_real * CIp * 9, grid); \
    } while (0)
    if (CO <= 16 && CI == 64 && ld_dy >= 16) { A2D_DISPATCH_T(dtype, A2D_WGRAD(bf16, 16, 64, CO), A2D_WGRAD(float, 16, 64, CO)); }
    else if (CO == 64 && CI == 8) { A2D_DISPATCH_T(dtype, A2D_WGRAD(bf16, 64, 8, CO), A2D_WGRAD(float, 64, 8, CO)); }
    else return S2T_ENOTSUP;
#undef A2D_WGRAD
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_a2d_planes(int dtype, void* chl, void* planes, int G, int ch0, int ld, int B, int T, int F, int dir, void* stream) {
    if (B <= 0 || T <= 0 || G <= 0) return S2T_OK;
    if (!chl || !planes || F <= 0 || F > 32 || ch0 < 0 || ch0 + 4 * G > ld || (dir != 0 && dir != 1)) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid_for((long)G * T * B * HEADS * 32);
    A2D_DISPATCH_T(dtype,
        hipLaunchKernelGGL(a2d_planes_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (bf16*)chl, (bf16*)planes, G, ch0, ld, B, T, F, dir),
        hipLaunchKernelGGL(a2d_planes_kernel<float>, dim3(blocks), dim3(256), 0, st, (float*)chl, (float*)planes, G, ch0, ld, B, T, F, dir));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
