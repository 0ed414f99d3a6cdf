// 2-D convolutional subsampler over (B, T, F=80) log-mel filterbanks: the HBM-bound pieces.
// Reference: examples/speech_recognition/models/conv_transformer.py:202-232 (+ :348-368):
//   2 x [ Conv2d(3x3, stride 2, pad 1) + bias -> act -> BatchNorm2d -> dropout ] -> flatten -> fc3 -> act -> + pos.
// Layout decisions (MI355X-first, not the reference's NCHW):
//   * activations are channels-last:  y1[b][t2][f2][c]  and  z2[t4][b][f4][c]  (c contiguous), so that
//     a wavefront = 64 channels of one pixel: coalesced 128-B stores, input taps broadcast, BatchNorm
//     statistics accumulate thread-locally (lane = channel) with no cross-lane reduction;
//   * conv1 (1 input channel, 9 MACs/output) is a direct VALU kernel bound by the y1 write;
//   * conv2 (64->64, 576 MACs/output) runs as an implicit GEMM on MFMA through s2t_gemm's row-gather
//     path (gemm.hip); this file only provides its BatchNorm / bias / index plumbing;
//   * fc3 consumes z2 directly: its weight is re-ordered once per step from the reference's
//     k = c*F4+f (conv_transformer.py:225-226) to k' = f*C+c (permute kernels below).
#include "common.hpp"

// ------------------------------------------------------------------ conv1 forward (+ BN statistics)
// x [B][T][F] f32 -> y [B][T2][F2][C] T, y = relu(conv(x)+bias); sums[c] += y, sums[C+c] += y^2 (double)
template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y,
                                                        double* __restrict__ sums, int B, int Tin, int F, int T2,
                                                        int F2, int C, int pos_per_block) {
    __shared__ float red[2][256];
    const int c = threadIdx.x % C, slot = threadIdx.x / C, nslot = 256 / C;
    float wr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wr[i] = w[c * 9 + i];
    const float bc = bias[c];
    const long P = (long)B * T2 * F2;
    const long p0 = (long)blockIdx.x * pos_per_block;
    float s1 = 0.f, s2 = 0.f;
    for (int i = slot; i < pos_per_block; i += nslot) {
        const long p = p0 + i;
        if (p >= P) break;
        const int f2 = (int)(p % F2), t2 = (int)((p / F2) % T2), b = (int)(p / ((long)F2 * T2));
        float acc = bc;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = 2 * t2 + kh - 1;
            if (t < 0 || t >= Tin) continue;
            const float* xr = x + ((long)b * Tin + t) * F;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int f = 2 * f2 + kw - 1;
                if (f >= 0 && f < F) acc += xr[f] * wr[kh * 3 + kw];
            }
        }
        acc = fmaxf(acc, 0.f);
        const T o = from_f32<T>(acc);
        y[p * C + c] = o;
        const float r = to_f32(o);
        s1 += r; s2 += r * r;
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (slot == 0) {
        double a = 0.0, q = 0.0;
        for (int s = 0; s < nslot; ++s) { a += red[0][s * C + c]; q += red[1][s * C + c]; }
        atomicAdd(sums + c, a);
        atomicAdd(sums + C + c, q);
    }
}

// ------------------------------------------------------------------ conv1 backward (weights, bias)
// dpre [B][T2][F2][C] T (gradient w.r.t. conv1 + bias, i.e. after the ReLU mask)
template <typename T>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const T* __restrict__ dpre,
                                                        float* __restrict__ dw, float* __restrict__ db, int B,
                                                        int Tin, int F, int T2, int F2, int C, int pos_per_block) {
    __shared__ float red[10][256];
    const int c = threadIdx.x % C, slot = threadIdx.x / C, nslot = 256 / C;
    float a[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) a[i] = 0.f;
    const long P = (long)B * T2 * F2;
    const long p0 = (long)blockIdx.x * pos_per_block;
    for (int i = slot; i < pos_per_block; i += nslot) {
        const long p = p0 + i;
        if (p >= P) break;
        const int f2 = (int)(p % F2), t2 = (int)((p / F2) % T2), b = (int)(p / ((long)F2 * T2));
        const float g = to_f32(dpre[p * C + c]);
        a[9] += g;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = 2 * t2 + kh - 1;
            if (t < 0 || t >= Tin) continue;
            const float* xr = x + ((long)b * Tin + t) * F;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int f = 2 * f2 + kw - 1;
                if (f >= 0 && f < F) a[kh * 3 + kw] += g * xr[f];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) red[i][threadIdx.x] = a[i];
    __syncthreads();
    if (slot == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            float s = 0.f;
            for (int k = 0; k < nslot; ++k) s += red[i][k * C + c];
            if (i < 9) atomicAdd(dw + c * 9 + i, s); else atomicAdd(db + c, s);
        }
    }
}

// ------------------------------------------------------------------ per-channel sums over [P][C]
// mode 0: sums[c] += y, sums[C+c] += y^2            (BatchNorm statistics)
// mode 1: sums[c] += dyn, sums[C+c] += dyn*xhat     (BatchNorm backward: dbeta, dgamma)
template <typename T>
__global__ __launch_bounds__(256) void chan_sums_kernel(const T* __restrict__ y, const T* __restrict__ dyn,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        double* __restrict__ sums, long P, int C, int mode,
                                                        int pos_per_block) {
    __shared__ float red[2][256];
    const int c = threadIdx.x % C, slot = threadIdx.x / C, nslot = 256 / C;
    const long p0 = (long)blockIdx.x * pos_per_block;
    float s1 = 0.f, s2 = 0.f;
    const float mu = mode ? mean[c] : 0.f, rs = mode ? rstd[c] : 0.f;
    for (int i = slot; i < pos_per_block; i += nslot) {
        const long p = p0 + i;
        if (p >= P) break;
        const float v = to_f32(y[p * C + c]);
        if (mode == 0) { s1 += v; s2 += v * v; }
        else { const float d = to_f32(dyn[p * C + c]); s1 += d; s2 += d * (v - mu) * rs; }
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (slot == 0) {
        double a = 0.0, q = 0.0;
        for (int s = 0; s < nslot; ++s) { a += red[0][s * C + c]; q += red[1][s * C + c]; }
        atomicAdd(sums + c, a);
        atomicAdd(sums + C + c, q);
    }
}

// ------------------------------------------------------------------ BatchNorm finalize
// training: batch statistics from the double sums (biased variance normalises; running_var gets the
// unbiased one; momentum update; nn.BatchNorm2d semantics, conv_transformer.py:212,364-368)
// eval: running statistics.  Outputs mean, rstd, scale = gamma*rstd, shift = beta - mean*scale.
__global__ void bn_finalize_kernel(const double* sums, const float* gamma, const float* beta, float* run_mean,
                                   float* run_var, long long* num_batches, float* mean, float* rstd, float* scale,
                                   float* shift, double count, int C, int training, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mu, var;
    if (training) {
        const double m = sums[c] / count;
        double v = sums[C + c] / count - m * m;
        if (v < 0.0) v = 0.0;
        mu = (float)m; var = (float)v;
        const double unb = count > 1.0 ? v * (count / (count - 1.0)) : v;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
        if (c == 0 && num_batches) *num_batches += 1;
    } else { mu = run_mean[c]; var = run_var[c]; }
    const float rs = rsqrtf(var + eps);
    mean[c] = mu; rstd[c] = rs;
    scale[c] = gamma[c] * rs;
    shift[c] = beta[c] - mu * gamma[c] * rs;
}

// yn = y*scale[c] + shift[c]   ([P][C], c contiguous)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, T* __restrict__ yn, long n, int C) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        yn[i] = from_f32<T>(to_f32(y[i]) * scale[c] + shift[c]);
    }
}

// BatchNorm backward (training statistics) fused with the ReLU mask of the preceding activation:
//   dy = gamma*rstd*(dyn - dbeta/N - xhat*dgamma/N);  dpre = dy * (y > 0)
// also accumulates dgamma/dbeta into the f32 parameter gradients (block 0).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dyn, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums,
                                                           T* __restrict__ dpre, float* dgamma, float* dbeta, long n,
                                                           int C, double count, int training) {
    if (blockIdx.x == 0 && threadIdx.x < C) {
        atomicAdd(dbeta + threadIdx.x, (float)sums[threadIdx.x]);
        atomicAdd(dgamma + threadIdx.x, (float)sums[C + threadIdx.x]);
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float v = to_f32(y[i]);
        const float d = to_f32(dyn[i]);
        float r;
        if (training) {
            const float xh = (v - mean[c]) * rstd[c];
            r = gamma[c] * rstd[c] * (d - (float)(sums[c] / count) - xh * (float)(sums[C + c] / count));
        } else r = gamma[c] * rstd[c] * d;
        dpre[i] = from_f32<T>(v > 0.f ? r : 0.f);
    }
}

// ------------------------------------------------------------------ weight re-ordering for channels-last
// mode 0 (fc3 forward):  dst[n][f*C+c]  = src[n][c*F+f]          (src f32 master -> dst T)
// mode 1 (fc3 grads):    dst[n][c*F+f] += src[n][f*C+c]          (src f32 -> dst f32 accumulate)
template <typename TD>
__global__ __launch_bounds__(256) void permute_cf_kernel(const float* __restrict__ src, TD* __restrict__ dst, int N, int C, int F, int mode) {
    const long n_el = (long)N * C * F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_el; i += (long)gridDim.x * 256) {
        const int k = (int)(i % (C * F)); const long n = i / (C * F);
        if (mode == 0) { const int f = k / C, c = k % C; dst[i] = from_f32<TD>(src[n * C * F + c * F + f]); }
        else { const int c = k / F, f = k % F; dst[i] = from_f32<TD>(to_f32(dst[i]) + src[n * C * F + f * C + c]); }
    }
}
// conv2 weight [co][ci][3][3] (f32 master) <-> implicit-GEMM layouts
// mode 0: dst[co][tap*Ci+ci] = src[co][ci][tap]                       (forward B operand, T)
// mode 1: dst[ci][slot(tap)*Co+co] = src[co][ci][tap], class-major tap order (backward-data B operand, T)
// mode 2: dst[co][ci][tap] += src[co][tap*Ci+ci]                      (f32 grads back to the master layout)
__device__ __constant__ int kTapSlot[9] = {5, 3, 6, 1, 0, 2, 7, 4, 8};   // tap (kh*3+kw) -> class-major slot
template <typename TD>
__global__ __launch_bounds__(256) void permute_conv_w_kernel(const float* __restrict__ src, TD* __restrict__ dst, int Co, int Ci, int mode) {
    const long n_el = (long)Co * Ci * 9;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_el; i += (long)gridDim.x * 256) {
        if (mode == 0) {
            const int ci = (int)(i % Ci), tap = (int)((i / Ci) % 9), co = (int)(i / (9L * Ci));
            dst[i] = from_f32<TD>(src[((long)co * Ci + ci) * 9 + tap]);
        } else if (mode == 1) {
            const int co = (int)(i % Co), slot = (int)((i / Co) % 9), ci = (int)(i / (9L * Co));
            int tap = 0;
            for (int t = 0; t < 9; ++t) if (kTapSlot[t] == slot) tap = t;
            dst[i] = from_f32<TD>(src[((long)co * Ci + ci) * 9 + tap]);
        } else {
            const int tap = (int)(i % 9), ci = (int)((i / 9) % Ci), co = (int)(i / (9L * Ci));
            dst[i] = from_f32<TD>(to_f32(dst[i]) + src[(long)co * 9 * Ci + tap * Ci + ci]);
        }
    }
}

// ------------------------------------------------------------------ positional embedding add
// x[t][b][:] += table[(t < len[b]) ? t+1 : 0][:]   (positional_embedding_audio.py:21-27 +
// sinusoidal_positional_embedding.py: positions 1..len, padding row 0 = zeros); table is f32 [>=T+1][D]
template <typename T>
__global__ __launch_bounds__(256) void add_pos_kernel(T* __restrict__ x, const float* __restrict__ table,
                                                      const int* __restrict__ len, int Tn, int B, int D) {
    const long n = (long)Tn * B * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int d = (int)(i % D), b = (int)((i / D) % B), t = (int)(i / ((long)D * B));
        const int pos = (t < len[b]) ? t + 1 : 0;
        x[i] = from_f32<T>(to_f32(x[i]) + table[(long)pos * D + d]);
    }
}

// ------------------------------------------------------------------ C ABI
static inline int nblocks(long n, int cap = 4096) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }
#define DISPATCH_T(dtype, EXPR_BF16, EXPR_F32) \
    if ((dtype) == S2T_BF16) { EXPR_BF16; } else if ((dtype) == S2T_F32) { EXPR_F32; } else return S2T_ENOTSUP;

extern "C" int s2t_conv1_fwd(int dtype, const float* x, const float* w, const float* bias, void* y, double* sums,
                             int B, int T, int F, int C, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!x || !w || !bias || !y || !sums || (C != 64 && C != 128 && C != 32) || F <= 0) return S2T_EINVAL;
    const int T2 = (T + 1) / 2, F2 = (F + 1) / 2;
    const long P = (long)B * T2 * F2;
    const int ppb = (int)((P + 1023) / 1024 < 64 ? 64 : (P + 1023) / 1024);   // <= ~1024 workgroups: every one ends in 2C same-address atomics
    dim3 grid((unsigned)((P + ppb - 1) / ppb));
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(conv1_fwd_kernel<bf16>, grid, dim3(256), 0, st, x, w, bias, (bf16*)y, sums, B, T, F, T2, F2, C, ppb),
        hipLaunchKernelGGL(conv1_fwd_kernel<float>, grid, dim3(256), 0, st, x, w, bias, (float*)y, sums, B, T, F, T2, F2, C, ppb));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_conv1_bwd(int dtype, const float* x, const void* dpre, float* dw, float* db, int B, int T, int F,
                             int C, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!x || !dpre || !dw || !db || (C != 64 && C != 128 && C != 32)) return S2T_EINVAL;
    const int T2 = (T + 1) / 2, F2 = (F + 1) / 2;
    const long P = (long)B * T2 * F2;
    const int ppb = (int)((P + 511) / 512 < 256 ? 256 : (P + 511) / 512);     // <= ~512 workgroups (10C atomics each)
    dim3 grid((unsigned)((P + ppb - 1) / ppb));
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(conv1_bwd_kernel<bf16>, grid, dim3(256), 0, st, x, (const bf16*)dpre, dw, db, B, T, F, T2, F2, C, ppb),
        hipLaunchKernelGGL(conv1_bwd_kernel<float>, grid, dim3(256), 0, st, x, (const float*)dpre, dw, db, B, T, F, T2, F2, C, ppb));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_chan_sums(int dtype, const void* y, const void* dyn, const float* mean, const float* rstd,
                             double* sums, long P, int C, int mode, void* stream) {
    if (P <= 0) return S2T_OK;
    if (!y || !sums || (C != 64 && C != 128 && C != 32) || (mode && (!dyn || !mean || !rstd))) return S2T_EINVAL;
    const int ppb = (int)((P + 511) / 512 < 256 ? 256 : (P + 511) / 512);
    dim3 grid((unsigned)((P + ppb - 1) / ppb));
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(chan_sums_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)y, (const bf16*)dyn, mean, rstd, sums, P, C, mode, ppb),
        hipLaunchKernelGGL(chan_sums_kernel<float>, grid, dim3(256), 0, st, (const float*)y, (const float*)dyn, mean, rstd, sums, P, C, mode, ppb));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_bn_finalize(const double* sums, const float* gamma, const float* beta, float* run_mean,
                               float* run_var, long long* num_batches, float* mean, float* rstd, float* scale,
                               float* shift, double count, int C, int training, float momentum, float eps, void* stream) {
    if (!gamma || !beta || !run_mean || !run_var || !mean || !rstd || !scale || !shift || (training && !sums)) return S2T_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, gamma, beta, run_mean,
                       run_var, num_batches, mean, rstd, scale, shift, count, C, training, momentum, eps);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_bn_apply(int dtype, const void* y, const float* scale, const float* shift, void* yn, long n, int C, void* stream) {
    if (n <= 0) return S2T_OK;
    if (!y || !scale || !shift || !yn) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(nblocks(n)), dim3(256), 0, st, (const bf16*)y, scale, shift, (bf16*)yn, n, C),
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, (const float*)y, scale, shift, (float*)yn, n, C));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_bn_bwd_apply(int dtype, const void* dyn, const void* y, const float* mean, const float* rstd,
                                const float* gamma, const double* sums, void* dpre, float* dgamma, float* dbeta, long n,
                                int C, double count, int training, void* stream) {
    if (n <= 0) return S2T_OK;
    if (!dyn || !y || !mean || !rstd || !gamma || !sums || !dpre || !dgamma || !dbeta || C > 256) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, dim3(nblocks(n)), dim3(256), 0, st, (const bf16*)dyn, (const bf16*)y, mean, rstd, gamma, sums, (bf16*)dpre, dgamma, dbeta, n, C, count, training),
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, (const float*)dyn, (const float*)y, mean, rstd, gamma, sums, (float*)dpre, dgamma, dbeta, n, C, count, training));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_permute_cf(int dst_dtype, const float* src, void* dst, int N, int C, int F, int mode, void* stream) {
    const long n = (long)N * C * F;
    if (n <= 0) return S2T_OK;
    if (!src || !dst || (mode == 1 && dst_dtype != S2T_F32)) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dst_dtype,
        hipLaunchKernelGGL(permute_cf_kernel<bf16>, dim3(nblocks(n)), dim3(256), 0, st, src, (bf16*)dst, N, C, F, mode),
        hipLaunchKernelGGL(permute_cf_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src, (float*)dst, N, C, F, mode));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_permute_conv_w(int dst_dtype, const float* src, void* dst, int Co, int Ci, int mode, void* stream) {
    const long n = (long)Co * Ci * 9;
    if (n <= 0) return S2T_OK;
    if (!src || !dst || (mode == 2 && dst_dtype != S2T_F32) || mode < 0 || mode > 2) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dst_dtype,
        hipLaunchKernelGGL(permute_conv_w_kernel<bf16>, dim3(nblocks(n)), dim3(256), 0, st, src, (bf16*)dst, Co, Ci, mode),
        hipLaunchKernelGGL(permute_conv_w_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src, (float*)dst, Co, Ci, mode));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_add_pos(int dtype, void* x, const float* table, const int* len, int T, int B, int D, void* stream) {
    const long n = (long)T * B * D;
    if (n <= 0) return S2T_OK;
    if (!x || !table || !len) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(add_pos_kernel<bf16>, dim3(nblocks(n)), dim3(256), 0, st, (bf16*)x, table, len, T, B, D),
        hipLaunchKernelGGL(add_pos_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, (float*)x, table, len, T, B, D));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
