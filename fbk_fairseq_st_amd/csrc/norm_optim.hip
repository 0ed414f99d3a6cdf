// HBM-bound row kernels: LayerNorm fwd/bwd (wavefront reductions), flat multi-tensor Adam with fused
// gradient scaling / clipping, global gradient norm, dtype casts.
// Reference semantics: fairseq/modules/layer_norm.py:29-32 (nn.LayerNorm, eps 1e-5);
// fairseq/optim/adam.py:147-202; fairseq/utils.py:253-277 (clip_grad_norm_);
// fairseq/trainer.py:416-443 (multiply_grads -> clip -> step).
#include "common.hpp"
#include "prof.hpp"

// ------------------------------------------------------------------ LayerNorm
// One wavefront per row.  Lane l owns EPL contiguous elements [l*EPL, l*EPL+EPL) (D = 64*EPL, EPL in {4,8,16}):
// 8/16/32-byte loads per lane, fully coalesced, the row lives in registers; statistics by wavefront
// shuffles.  Other widths (D <= 1024) take the strided scalar variant (EPL = 0).
#ifndef S2T_LN_NT
#define S2T_LN_NT 2      // 1 non-temporal row loads in the backward, 2 non-temporal stores there, 4 loads in the forward, 8 stores in the forward
#endif
template <typename T, int EPL, bool NT = false> struct RowIO {
    static __device__ __forceinline__ void load(const T* row, int lane, int D, float (&v)[16]) {
        if constexpr (EPL == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int j = lane + 64 * i; v[i] = (j < D) ? to_f32(row[j]) : 0.f; }
        } else {
            constexpr int VE = 16 / (int)sizeof(T);                    // elements per 16-byte vector
            constexpr int NV = (EPL * (int)sizeof(T) + 15) / 16;       // vectors per lane
            constexpr int PER = EPL < VE ? EPL : VE;                   // elements taken from each vector
            const T* p = row + lane * EPL;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                T tmp[VE];
                if constexpr (EPL * sizeof(T) >= 16) {
                    if constexpr (NT) *reinterpret_cast<u32x4*>(tmp) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + k * VE));
                    else *reinterpret_cast<u32x4*>(tmp) = *reinterpret_cast<const u32x4*>(p + k * VE);
                } else *reinterpret_cast<u32x2*>(tmp) = *reinterpret_cast<const u32x2*>(p);
#pragma unroll
                for (int e = 0; e < PER; ++e) v[k * VE + e] = to_f32(tmp[e]);
            }
        }
    }
    static __device__ __forceinline__ void store(T* row, int lane, int D, const float (&v)[16]) {
        if constexpr (EPL == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int j = lane + 64 * i; if (j < D) row[j] = from_f32<T>(v[i]); }
        } else {
            constexpr int VE = 16 / (int)sizeof(T);
            constexpr int NV = (EPL * (int)sizeof(T) + 15) / 16;
            constexpr int PER = EPL < VE ? EPL : VE;
            T* p = row + lane * EPL;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                T tmp[VE];
#pragma unroll
                for (int e = 0; e < PER; ++e) tmp[e] = from_f32<T>(v[k * VE + e]);
                if constexpr (EPL * sizeof(T) >= 16) {
                    if constexpr (NT) __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(tmp), reinterpret_cast<u32x4*>(p + k * VE));
                    else *reinterpret_cast<u32x4*>(p + k * VE) = *reinterpret_cast<const u32x4*>(tmp);
                } else *reinterpret_cast<u32x2*>(p) = *reinterpret_cast<const u32x2*>(tmp);
            }
        }
    }
    // column index of register slot i of this lane
    static __device__ __forceinline__ int col(int lane, int i) { return EPL == 0 ? lane + 64 * i : lane * EPL + i; }
    static constexpr int N = EPL == 0 ? 16 : EPL;
};

template <typename T, int EPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int M, int D, float eps) {
    typedef RowIO<T, EPL> IO;
    typedef RowIO<T, EPL, (S2T_LN_NT & 4) != 0> ION;
    typedef RowIO<T, EPL, (S2T_LN_NT & 8) != 0> IOSN;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[16];
    ION::load(x + (size_t)row * D, lane, D, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < IO::N; ++i) s += v[i];
    const float mu = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < IO::N; ++i) { const float d = (IO::col(lane, i) < D) ? v[i] - mu : 0.f; ss += d * d; }
    const float rs = rsqrtf(wave_sum(ss) / (float)D + eps);
    if constexpr (EPL != 0) {
        // gamma / beta as 16-byte vectors (a lane's EPL columns are contiguous)
#pragma unroll
        for (int k = 0; k < EPL / 4; ++k) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + lane * EPL + 4 * k);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(beta + lane * EPL + 4 * k);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * k + e] = (v[4 * k + e] - mu) * rs * gv[e] + bv[e];
        }
    } else {
#pragma unroll
        for (int i = 0; i < IO::N; ++i) { const int j = IO::col(lane, i); if (j < D) v[i] = (v[i] - mu) * rs * gamma[j] + beta[j]; }
    }
    IOSN::store(y + (size_t)row * D, lane, D, v);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) [+ dres],  g = dy*gamma;
// dgamma += sum_rows dy*xhat, dbeta += sum_rows dy  (per-lane partials over a grid-stride row loop,
// combined across the workgroup's waves in LDS, then one f32 atomic per column per workgroup).
// NW waves per workgroup, one row per wave per step.  The kernel is a chain load -> two wave reductions -> store per row, so what
// it needs is rows in flight: the NEXT row's vectors are requested before the current row is reduced, and the launch puts 16
// waves on every CU (one 1024-thread workgroup: the same wave count as four 256-thread ones at a quarter of the same-address
// atomics that end each workgroup).  [24000, 512] bf16 with residual and dropout outputs: 27 us, 4.5 TB/s.
template <typename T, int EPL, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int M, int D,
                                                     T* __restrict__ dx_drop, float p_drop, unsigned long long seed) {
    typedef RowIO<T, EPL> IO;
    typedef RowIO<T, EPL, (S2T_LN_NT & 1) != 0> IOL;
    typedef RowIO<T, EPL, (S2T_LN_NT & 2) != 0> IOS;
    extern __shared__ float sh_ln[];                       // [2][NW][D]
    const uint32_t th16 = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float inv_keep = 1.f / (1.f - p_drop);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ag[16], ab[16], gm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { ag[i] = 0.f; ab[i] = 0.f; gm[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < IO::N; ++i) { const int j = IO::col(lane, i); if (j < D) gm[i] = gamma[j]; }
    const int stride = gridDim.x * NW;
    int row = blockIdx.x * NW + w;
    float xn[16], dn[16], rn[16], mun = 0.f, rsn = 0.f;
    if (row < M) {
        IOL::load(x + (size_t)row * D, lane, D, xn);
        IOL::load(dy + (size_t)row * D, lane, D, dn);
        if (dres) IOL::load(dres + (size_t)row * D, lane, D, rn);
        mun = mean[row]; rsn = rstd[row];
    }
    for (; row < M; row += stride) {
        float xv[16], dv[16], rv[16];
        const float mu = mun, rs = rsn;
#pragma unroll
        for (int i = 0; i < IO::N; ++i) { xv[i] = xn[i]; dv[i] = dn[i]; rv[i] = rn[i]; }
        const int nrow = row + stride;
        if (nrow < M) {                                         // wave-uniform
            IOL::load(x + (size_t)nrow * D, lane, D, xn);
            IOL::load(dy + (size_t)nrow * D, lane, D, dn);
            if (dres) IOL::load(dres + (size_t)nrow * D, lane, D, rn);
            mun = mean[nrow]; rsn = rstd[nrow];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < IO::N; ++i) {
            const bool ok = IO::col(lane, i) < D;
            xv[i] = ok ? (xv[i] - mu) * rs : 0.f;             // xhat
            const float d = ok ? dv[i] : 0.f;
            ag[i] += d * xv[i];
            ab[i] += d;
            dv[i] = d * gm[i];                                 // g
            s1 += dv[i];
            s2 += dv[i] * xv[i];
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < IO::N; ++i) {
            float r = rs * (dv[i] - c1 - xv[i] * c2);
            if (dres) r += rv[i];
            dv[i] = r;
        }
        IOS::store(dx + (size_t)row * D, lane, D, dv);
        if (dx_drop) {
            // second output = dropout(dx) with the consumer block's mask (what s2t_dropout would produce from the stored,
            // rounded dx: element index row*D + col, one hash per aligned group of four)
            if constexpr (EPL != 0) {
#pragma unroll
                for (int k = 0; k < EPL / 4; ++k) {
                    const u32x2 h = drop_hash4(seed, (((uint64_t)row * D) >> 2) + (uint64_t)(lane * (EPL / 4) + k));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dv[4 * k + e] = drop_field(h, e) >= th16 ? to_f32(from_f32<T>(dv[4 * k + e])) * inv_keep : 0.f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < IO::N; ++i) {
                    const int j = IO::col(lane, i);
                    if (j < D) dv[i] = drop_field(drop_hash4(seed, ((uint64_t)row * D + j) >> 2), (int)(((uint64_t)row * D + j) & 3)) >= th16
                                           ? to_f32(from_f32<T>(dv[i])) * inv_keep : 0.f;
                }
            }
            IOS::store(dx_drop + (size_t)row * D, lane, D, dv);
        }
    }
    float* sg = sh_ln;
    float* sb = sh_ln + NW * D;
#pragma unroll
    for (int i = 0; i < IO::N; ++i) {
        const int j = IO::col(lane, i);
        if (j < D) { sg[w * D + j] = ag[i]; sb[w * D + j] = ab[i]; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * D; j += NW * 64) {      // first D threads' worth: dgamma columns, then dbeta
        const float* src = j < D ? sg + j : sb + (j - D);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) t += src[k * D];
        // f32 atomics on one address are served one after the other at the memory side (~40 ns each).  For the big activations
        // the 256 workgroups do not finish together and the chain hides behind the stragglers' rows (writing partials + a finishing
        // kernel measured the same 27 us); for small M the launch is capped at 64 workgroups (512 of them spent 19 us on 2560 rows).
        atomicAdd((j < D ? dgamma + j : dbeta + (j - D)), t);
    }
}

// Small M (decoder-sized activations, small batches): at most 64 workgroups x 16 waves, i.e. 3-4 rows per wave, and inside an update every
// row is a round trip to lines the previous launch wrote: with one row of lookahead the launch was a chain of them (11-13 us for 3,000-4,000
// rows against 4.5 us for the forward).  Here a wave requests ALL its rows of a pass (four) before it reduces the first; the rows stay
// raw (16 bytes per operand and lane) until they are used, so the twelve vectors fit the 128-register budget of a 1,024-thread workgroup.
// Same arithmetic per row, same row order per wave: bit-identical to ln_bwd_kernel.  bf16, D = 512 (8 elements per lane).
template <int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_small_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const bf16* __restrict__ dres,
                                                           bf16* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int M, int D,
                                                           bf16* __restrict__ dx_drop, float p_drop, unsigned long long seed) {
    constexpr int EPL = 8, RPW = 3;
    typedef RowIO<bf16, EPL> IO;
    typedef RowIO<bf16, EPL, (S2T_LN_NT & 2) != 0> IOS;
    extern __shared__ float sh_ln[];                       // [2][NW][D]
    const uint32_t th16 = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f) >> 16;
    const float inv_keep = 1.f / (1.f - p_drop);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ag[EPL], ab[EPL], gm[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; gm[i] = gamma[lane * EPL + i]; }
    const int stride = gridDim.x * NW;
    const bf16* rsrc = dres ? dres : dy;                   // no residual: a second read of dy that is never used (no branch around a load)
    for (int row0 = blockIdx.x * NW + w; row0 < M; row0 += RPW * stride) {
        u32x4 xr[RPW], dr[RPW], rr[RPW];
        float mu[RPW], rs[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const size_t r = (size_t)min(row0 + k * stride, M - 1);          // past the end: the last row again, never used
            xr[k] = *reinterpret_cast<const u32x4*>(x + r * D + lane * EPL);
            dr[k] = *reinterpret_cast<const u32x4*>(dy + r * D + lane * EPL);
            rr[k] = *reinterpret_cast<const u32x4*>(rsrc + r * D + lane * EPL);
            mu[k] = mean[r]; rs[k] = rstd[r];
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int row = row0 + k * stride;
            if (row < M) {                                                   // wave-uniform; nothing inside requests memory
                float xv[16], dv[16];
                const bf16* xe = reinterpret_cast<const bf16*>(&xr[k]);
                const bf16* de = reinterpret_cast<const bf16*>(&dr[k]);
                const bf16* re = reinterpret_cast<const bf16*>(&rr[k]);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < EPL; ++i) {
                    xv[i] = (to_f32(xe[i]) - mu[k]) * rs[k];              // xhat
                    const float d = to_f32(de[i]);
                    ag[i] += d * xv[i];
                    ab[i] += d;
                    dv[i] = d * gm[i];                                     // g
                    s1 += dv[i];
                    s2 += dv[i] * xv[i];
                }
                const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
                for (int i = 0; i < EPL; ++i) {
                    float r = rs[k] * (dv[i] - c1 - xv[i] * c2);
                    if (dres) r += to_f32(re[i]);
                    dv[i] = r;
                }
                IOS::store(dx + (size_t)row * D, lane, D, dv);
                if (dx_drop) {
#pragma unroll
                    for (int kk = 0; kk < EPL / 4; ++kk) {
                        const u32x2 h = drop_hash4(seed, (((uint64_t)row * D) >> 2) + (uint64_t)(lane * (EPL / 4) + kk));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            dv[4 * kk + e] = drop_field(h, e) >= th16 ? to_f32(from_f32<bf16>(dv[4 * kk + e])) * inv_keep : 0.f;
                    }
                    IOS::store(dx_drop + (size_t)row * D, lane, D, dv);
                }
            }
        }
    }
    float* sg = sh_ln;
    float* sb = sh_ln + NW * D;
#pragma unroll
    for (int i = 0; i < EPL; ++i) { sg[w * D + lane * EPL + i] = ag[i]; sb[w * D + lane * EPL + i] = ab[i]; }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * D; j += NW * 64) {
        const float* src = j < D ? sg + j : sb + (j - D);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) t += src[k * D];
        atomicAdd((j < D ? dgamma + j : dbeta + (j - D)), t);
    }
}

static int ln_epl(int D, const void* a, const void* b, const void* c, const void* d, const void* e = nullptr, const void* f = nullptr) {
    const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f;
    if ((al & 15) != 0) return 0;
    if (D == 256) return 4;
    if (D == 512) return 8;
    if (D == 1024) return 16;
    return 0;
}

#define LN_DISPATCH(KERNEL, T, epl, grid, ...)                                                              \
    do {                                                                                                    \
        if (epl == 4) hipLaunchKernelGGL((KERNEL<T, 4>), grid, dim3(256), 0, st, __VA_ARGS__);              \
        else if (epl == 8) hipLaunchKernelGGL((KERNEL<T, 8>), grid, dim3(256), 0, st, __VA_ARGS__);         \
        else if (epl == 16) hipLaunchKernelGGL((KERNEL<T, 16>), grid, dim3(256), 0, st, __VA_ARGS__);       \
        else hipLaunchKernelGGL((KERNEL<T, 0>), grid, dim3(256), 0, st, __VA_ARGS__);                       \
    } while (0)

extern "C" int s2t_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                                 float* mean, float* rstd, int M, int D, float eps, void* stream) {
    if (M <= 0) return M < 0 ? S2T_EINVAL : S2T_OK;
    if (D <= 0 || D > 1024) return S2T_ENOTSUP;
    if (!x || !gamma || !beta || !y || !mean || !rstd) return S2T_EINVAL;
    dim3 grid((M + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    const int epl = ln_epl(D, x, y, gamma, beta);
    if (dtype == S2T_BF16) LN_DISPATCH(ln_fwd_kernel, bf16, epl, grid, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M, D, eps);
    else if (dtype == S2T_F32) LN_DISPATCH(ln_fwd_kernel, float, epl, grid, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                                 const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                                 int M, int D, void* dx_drop, float p_drop, unsigned long long seed, void* stream) {
    if (M <= 0) return M < 0 ? S2T_EINVAL : S2T_OK;
    if (D <= 0 || D > 1024) return S2T_ENOTSUP;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta) return S2T_EINVAL;
    if (dx_drop && (p_drop < 0.f || p_drop >= 1.f)) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int epl = ln_epl(D, dy, x, dres, dx, dx_drop);
    // 16 waves per workgroup: one workgroup per CU for the big activations; for small M at most 64 workgroups (each ends in 2 D
    // same-address atomics, ~40 ns apiece in a chain) of 16 waves, i.e. 2-3 rows per wave at the decoder's 2,560 rows -- with 4 waves
    // per workgroup every wave walked 10 rows one memory latency after the other (17 us per launch)
    // 16 elements per lane (D = 1,024) and the element-wise form keep a row's working set near 250 registers: 8 waves per workgroup
    // (a 1,024-thread workgroup allows 128, and those variants spilled ~200 registers), twice the workgroups
    const bool big = M >= 8192;
    const int nw = (epl == 16 || epl == 0) ? 8 : 16;
    int blocks = (M + nw - 1) / nw;
    const int cap = (big ? 256 : 64) * (nw == 8 ? 2 : 1);
    if (blocks > cap) blocks = cap;
    dim3 grid(blocks);
    const size_t lds = (size_t)2 * nw * D * sizeof(float);
#define LN_BWD_LAUNCH(T, EPL_, NW_)                                                                                         \
    do {                                                                                                                    \
        static bool attr = false;                                                                                           \
        if (!attr && lds > 65536) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<T, EPL_, NW_>),  \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NW_ * 1024 * 4); attr = true; } \
        hipLaunchKernelGGL((ln_bwd_kernel<T, EPL_, NW_>), grid, dim3(NW_ * 64), lds, st, (const T*)dy, (const T*)x, mean, rstd, gamma, \
                           (const T*)dres, (T*)dx, dgamma, dbeta, M, D, (T*)dx_drop, p_drop, seed);                         \
    } while (0)
#define LN_BWD_EPL(T)                                                                                                       \
    do {                                                                                                                    \
        if (epl == 4) LN_BWD_LAUNCH(T, 4, 16); else if (epl == 8) LN_BWD_LAUNCH(T, 8, 16);                                  \
        else if (epl == 16) LN_BWD_LAUNCH(T, 16, 8); else LN_BWD_LAUNCH(T, 0, 8);                                           \
    } while (0)
    if (dtype == S2T_BF16 && epl == 8 && !big && g_s2t_opt_ln_small) {
        static bool attr = false;
        if (!attr && lds > 65536) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_small_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * 1024 * 4); attr = true; }
        hipLaunchKernelGGL(ln_bwd_small_kernel<16>, grid, dim3(16 * 64), lds, st, (const bf16*)dy, (const bf16*)x, mean, rstd, gamma,
                           (const bf16*)dres, (bf16*)dx, dgamma, dbeta, M, D, (bf16*)dx_drop, p_drop, seed);
    }
    else if (dtype == S2T_BF16) LN_BWD_EPL(bf16);
    else if (dtype == S2T_F32) LN_BWD_EPL(float);
    else return S2T_ENOTSUP;
#undef LN_BWD_EPL
#undef LN_BWD_LAUNCH
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------ gradient norm, clip, Adam
// sumsq: part[workgroup] = sum g^2 over its share (double), one launch over the flat gradient arena; clip_coef_kernel adds the partials
// (2,048 workgroups ending in one same-address double atomic each arrive faster than the memory side serves them, ~40 ns apiece).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, double* __restrict__ part) {
    __shared__ double shd[4];
    double s = 0.0;
    const size_t n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = g4[i];
        s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += (double)v * v; }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = shd[0] + shd[1] + shd[2] + shd[3];
}

// acc[0] = sum of the partials; out[0] = gnorm = scale*sqrt(acc) ; out[1] = multiplier = scale * min(1, max_norm/(gnorm+1e-6))
// (fairseq/utils.py:268-276 on gradients already multiplied by `scale`, trainer.py:426-436)
// divisor (device, may be NULL): the gradients are additionally divided by max(*divisor, 1) -- the sample size summed over the ranks, which
// a data-parallel update knows only after an all-reduce; reading it here keeps the host out of the update (trainer.py:426-430)
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ part, int nparts, double* __restrict__ acc, float scale,
                                                        float max_norm, float* __restrict__ out, const double* __restrict__ divisor) {
    __shared__ double shd[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double tot = shd[0] + shd[1] + shd[2] + shd[3];
    acc[0] = tot;
    if (divisor) scale = (float)((double)scale / fmax(divisor[0], 1.0));
    const float gn = scale * (float)sqrt(tot);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (gn + 1e-6f), 1.f);
    out[0] = gn;
    out[1] = scale * coef;
}

// Adam over the flat arena (fairseq/optim/adam.py:147-202): g' = g*mult; m,v update; decoupled wd;
// p -= step_size * m / (sqrt(v)+eps); optionally refresh the bf16 shadow used by the MFMA GEMMs.
__device__ __forceinline__ float adam_one(float& pi, float gi, float& mi, float& vi, float mult, float lr, float beta1, float beta2,
                                          float eps, float wd, float step_size) {
    gi *= mult;
    mi = beta1 * mi + (1.f - beta1) * gi;
    vi = beta2 * vi + (1.f - beta2) * gi * gi;
    if (wd != 0.f) pi -= wd * lr * pi;
    pi -= step_size * mi / (sqrtf(vi) + eps);
    return pi;
}
// VEC: all five arrays 16-byte aligned (the arena's are): four parameters per lane and access, 30 bytes per parameter in 16-byte
// (8-byte for the bf16 shadow) accesses; the tail and unaligned sub-ranges (frozen-parameter gaps) take the element-wise form.
template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16* __restrict__ shadow, size_t n, const float* mult_ptr,
                                                   float lr, float beta1, float beta2, float eps, float wd,
                                                   float step_size) {
    const float mult = mult_ptr ? mult_ptr[1] : 1.f;
    size_t done = 0;
    if constexpr (VEC) {
        // two 16-byte groups per lane and step, every access non-temporal: each of the seven streams is touched once per update and
        // 2 GB of them go by, so nothing here is worth a cache line (tools/lab/adam_probe.hip, 74 M parameters back to back:
        // 441 us = 5.0 TB/s as one cached group per step, 376-400 us = 5.6-5.9 TB/s in this form)
        const size_t nv = n / 4, stride = (size_t)gridDim.x * 256;
        f32x4* P4 = reinterpret_cast<f32x4*>(p); const f32x4* G4 = reinterpret_cast<const f32x4*>(g);
        f32x4* M4 = reinterpret_cast<f32x4*>(m); f32x4* V4 = reinterpret_cast<f32x4*>(v);
        for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < nv; i0 += 2 * stride) {
            f32x4 pi[2], mi[2], vi[2], gi[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const size_t i = i0 + u * stride;
                if (i < nv) {
                    pi[u] = __builtin_nontemporal_load(P4 + i); gi[u] = __builtin_nontemporal_load(G4 + i);
                    mi[u] = __builtin_nontemporal_load(M4 + i); vi[u] = __builtin_nontemporal_load(V4 + i);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const size_t i = i0 + u * stride;
                if (i >= nv) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pi[u][e], me = mi[u][e], ve = vi[u][e];
                    adam_one(pe, gi[u][e], me, ve, mult, lr, beta1, beta2, eps, wd, step_size);
                    pi[u][e] = pe; mi[u][e] = me; vi[u][e] = ve;
                }
                __builtin_nontemporal_store(mi[u], M4 + i); __builtin_nontemporal_store(vi[u], V4 + i); __builtin_nontemporal_store(pi[u], P4 + i);
                if (shadow) {
                    bf16 sh[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) sh[e] = (bf16)pi[u][e];
                    __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(sh), reinterpret_cast<u32x2*>(shadow) + i);
                }
            }
        }
        done = nv * 4;
    }
    for (size_t i = done + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, mult, lr, beta1, beta2, eps, wd, step_size);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (shadow) shadow[i] = (bf16)pi;
    }
}

static int grad_norm_clip_impl(const float* g, size_t n, double* acc_ws, float scale, const double* divisor, float max_norm, float* out2,
                               void* stream);
extern "C" int s2t_grad_norm_clip(const float* g, size_t n, double* acc_ws, float scale, float max_norm,
                                  float* out2, void* stream) {
    return grad_norm_clip_impl(g, n, acc_ws, scale, nullptr, max_norm, out2, stream);
}
extern "C" int s2t_grad_norm_clip_div(const float* g, size_t n, double* acc_ws, float scale, const double* divisor_dev, float max_norm,
                                      float* out2, void* stream) {
    if (!divisor_dev) return S2T_EINVAL;
    return grad_norm_clip_impl(g, n, acc_ws, scale, divisor_dev, max_norm, out2, stream);
}
static int grad_norm_clip_impl(const float* g, size_t n, double* acc_ws, float scale, const double* divisor, float max_norm, float* out2,
                               void* stream) {
    if (!g || !acc_ws || !out2) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t se = hipSuccess;                                          // 2,048 partial sums, per (device, stream)
    double* part = (double*)s2t_scratch(S2T_SCRATCH_GNORM, st, 2048 * sizeof(double), &se);
    if (!part) return S2T_EHIP(se);
    int blocks = 0;
    if (n) {
        blocks = (int)((n / 4 + 255) / 256);
        blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
        hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, st, g, n, part);
        S2T_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, part, blocks, acc_ws, scale, max_norm, out2, divisor);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n,
                             const float* mult2, float lr, float beta1, float beta2, float eps, float wd,
                             int step, void* stream) {
    if (n == 0) return S2T_OK;
    if (!p || !g || !m || !v || step < 1) return S2T_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (((uintptr_t)shadow_bf16 & 7) == 0) && n >= 4;
    int blocks = (int)(((vec ? n / 4 : n) + 255) / 256);
    blocks = blocks > (vec ? 16384 : 4096) ? (vec ? 16384 : 4096) : (blocks < 1 ? 1 : blocks);
    if (vec) hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow_bf16, n,
                                mult2, lr, beta1, beta2, eps, wd, step_size);
    else hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow_bf16, n,
                            mult2, lr, beta1, beta2, eps, wd, step_size);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------ casts / scaling
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = from_f32<TD>(to_f32(s[i]));
}
extern "C" int s2t_cast(int src_dtype, int dst_dtype, const void* src, void* dst, size_t n, void* stream) {
    if (n == 0) return S2T_OK;
    if (!src || !dst) return S2T_EINVAL;
    int blocks = (int)((n + 255) / 256);
    blocks = blocks > 4096 ? 4096 : blocks;
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == S2T_F32 && dst_dtype == S2T_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(blocks), dim3(256), 0, st, (const float*)src, (bf16*)dst, n);
    else if (src_dtype == S2T_BF16 && dst_dtype == S2T_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, (float*)dst, n);
    else if (src_dtype == S2T_F32 && dst_dtype == S2T_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else if (src_dtype == S2T_BF16 && dst_dtype == S2T_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, n);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// x *= *scalar (device scalar): upstream loss-gradient scaling without a host sync
template <typename T>
__global__ __launch_bounds__(256) void scale_dev_kernel(T* x, size_t n, const float* s) {
    const float f = *s;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = from_f32<T>(to_f32(x[i]) * f);
}
extern "C" int s2t_scale_by_device_scalar(int dtype, void* x, size_t n, const float* scalar, void* stream) {
    if (n == 0) return S2T_OK;
    if (!x || !scalar) return S2T_EINVAL;
    int blocks = (int)((n + 255) / 256);
    blocks = blocks > 4096 ? 4096 : blocks;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(scale_dev_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16*)x, n, scalar);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(scale_dev_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)x, n, scalar);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
