// Label-smoothed cross entropy fused with its gradient, and the decoder token embedding.
// Reference: fairseq/criterions/label_smoothed_cross_entropy.py:12-29 with lprobs = log_softmax(logits.float())
// (fairseq/models/fairseq_decoder.py:58-79); fairseq/models/transformer.py:720-737 (embed*sqrt(D) + sinusoidal
// positions from utils.make_positions, fairseq/utils.py:192-202).
#include "common.hpp"
#include "prof.hpp"

// One workgroup per target row (B*L rows, V columns).  Never materialises lprobs:
//   lse = logsumexp(row);  nll = lse - x[y];  smooth = V*lse - sum(x)
//   loss += (1-eps)*nll + eps/V*smooth ; nll_sum += nll     (pad rows contribute nothing)
//   dlogits[v] = gscale * (softmax[v] - (1-eps)*[v==y] - eps/V)   (0 on pad rows)
// NC > 0: the row's 16-byte vectors (at most 256 NC of them) stay in registers between the three sweeps -- one read of the logits, and the
// maximum and the plain sum share one pair of barriers; NC = 0: rows of any length, re-read from L2 per sweep.
template <typename T, int NC>
__global__ __launch_bounds__(256) void lsce_kernel(const T* __restrict__ logits, const long long* __restrict__ target,
                                                   T* __restrict__ dlogits, float* __restrict__ part, long rows, int V, int ld,
                                                   float eps, int pad, float gscale) {
    __shared__ float sh[16];
    // A workgroup walks rows blockIdx.x, blockIdx.x + gridDim.x, ... and hands over its loss sums once, as a partial (below).
    // exp through the hardware exp2 (the gradient and the loss move by ~1e-7 relative).
    float acc_loss = 0.f, acc_nll = 0.f;
    constexpr float L2E = 1.44269504088896f;
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int NR = NC > 0 ? NC : 1;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* x = logits + row * ld;
    T* g = dlogits ? dlogits + row * ld : nullptr;
    const long long y = target[row];
    if (y == pad) {
        if (g) for (int v = threadIdx.x; v < V; v += 256) g[v] = from_f32<T>(0.f);
        continue;
    }
    // 16-byte row accesses when the row is aligned (ld multiple of 8 bf16 / 4 f32: kernels.py alloc_rows), scalar tail otherwise
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    const int nv = vec ? V / E : 0;
    u32x4 rv[NR];
    auto vec_at = [&](int k, int c) -> u32x4 {
        if constexpr (NC > 0) return rv[k];
        else return *reinterpret_cast<const u32x4*>(x + c * E);
    };
    if constexpr (NC > 0) {
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = threadIdx.x + 256 * k;
            rv[k] = c < nv ? *reinterpret_cast<const u32x4*>(x + c * E) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
    // the sweeps visit vector c = threadIdx.x + 256 k; with NC > 0 the trip count is the compile-time NC (rows of <= 256 NC vectors)
    auto sweep = [&](auto body) {
        if constexpr (NC > 0) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int c = threadIdx.x + 256 * k;
                if (c < nv) { T t[E]; *reinterpret_cast<u32x4*>(t) = vec_at(k, c); body(c, t); }
            }
        } else {
            for (int c = threadIdx.x; c < nv; c += 256) { T t[E]; *reinterpret_cast<u32x4*>(t) = vec_at(0, c); body(c, t); }
        }
    };
    float m = -INFINITY, sx = 0.f;
    sweep([&](int, const T (&t)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) { const float f = to_f32(t[e]); m = fmaxf(m, f); sx += f; }
    });
    for (int v = nv * E + threadIdx.x; v < V; v += 256) { const float f = to_f32(x[v]); m = fmaxf(m, f); sx += f; }
    {   // maximum and sum over the workgroup behind one pair of barriers
        m = wave_max(m); sx = wave_sum(sx);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = m; sh[4 + (threadIdx.x >> 6)] = sx; }
        __syncthreads();
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        sx = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    }
    const float ml2 = m * L2E;
    float se = 0.f;
    sweep([&](int, const T (&t)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) se += __builtin_amdgcn_exp2f(__builtin_fmaf(to_f32(t[e]), L2E, -ml2));
    });
    for (int v = nv * E + threadIdx.x; v < V; v += 256) se += __builtin_amdgcn_exp2f(__builtin_fmaf(to_f32(x[v]), L2E, -ml2));
    se = block_sum(se, sh);
    const float lse = m + logf(se);
    if (threadIdx.x == 0) {
        const float nll = lse - to_f32(x[y]);
        const float smooth = (float)V * lse - sx;
        acc_loss += (1.f - eps) * nll + (eps / (float)V) * smooth;
        acc_nll += nll;
    }
    if (g) {
        const float ev = eps / (float)V, lsel2 = lse * L2E;
        sweep([&](int c, const T (&t)[E]) {
            T o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float d = __builtin_amdgcn_exp2f(__builtin_fmaf(to_f32(t[e]), L2E, -lsel2)) - ev;
                if (c * E + e == y) d -= (1.f - eps);
                o[e] = from_f32<T>(d * gscale);
            }
            *reinterpret_cast<u32x4*>(g + c * E) = *reinterpret_cast<const u32x4*>(o);
        });
        for (int v = nv * E + threadIdx.x; v < V; v += 256) {
            float d = __builtin_amdgcn_exp2f(__builtin_fmaf(to_f32(x[v]), L2E, -lsel2)) - ev;
            if (v == y) d -= (1.f - eps);
            g[v] = from_f32<T>(d * gscale);
        }
    }
    __syncthreads();                                 // `sh` is reused by the next row's reductions
    }
    // per-workgroup partial sums; lsce_finish_kernel adds them up (1,024 workgroups ending in two same-address atomics each were a
    // ~40 us chain at the memory side: the whole kernel took 43 us for 82 MB)
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = acc_loss; part[2 * blockIdx.x + 1] = acc_nll; }
}
__global__ __launch_bounds__(256) void lsce_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ sums, int two) {
    __shared__ float sh[16];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = block_sum(a, sh); b = block_sum(b, sh);
    if (threadIdx.x == 0) { sums[0] += a; if (two) sums[1] += b; }
}

extern "C" int s2t_lsce(int dtype, const void* logits, const long long* target, void* dlogits, float* sums2, long rows, int V, int ld,
                        float eps, int pad, float grad_scale, void* stream) {
    if (rows <= 0) return S2T_OK;
    if (!logits || !target || !sums2 || V <= 0 || ld < V) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(rows < 1024 ? rows : 1024);       // four workgroups per CU
    hipError_t se = hipSuccess;                                        // 1,024 x 2 partial sums, per (device, stream)
    float* part = (float*)s2t_scratch(S2T_SCRATCH_LSCE, st, 2 * 1024 * sizeof(float), &se);
    if (!part) return S2T_EHIP(se);
#define S2T_LSCE(T_, NC_) hipLaunchKernelGGL((lsce_kernel<T_, NC_>), dim3(grid), dim3(256), 0, st, (const T_*)logits, target, (T_*)dlogits, part, rows, V, ld, eps, pad, grad_scale)
    if (dtype == S2T_BF16) { if (V <= 256 * 4 * 8) S2T_LSCE(bf16, 4); else S2T_LSCE(bf16, 0); }      // vocabularies up to 8,192 units: the row in registers
    else if (dtype == S2T_F32) { if (V <= 256 * 4 * 4) S2T_LSCE(float, 4); else S2T_LSCE(float, 0); }
    else return S2T_ENOTSUP;
#undef S2T_LSCE
    hipLaunchKernelGGL(lsce_finish_kernel, dim3(1), dim3(256), 0, st, part, (int)grid, sums2, 1);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------ decoder embedding
// tokens [B][L] int64 -> out [L][B][D] (time-major): scale*W[tok] + table[pos], pos = pad + #non-pad up to l
// (pad tokens: position pad -> zero row).  One workgroup per batch row; also emits klen[b] = #non-pad? no:
// the decoder self-attention padding mask is derived from the tokens on the host side of the ABI.
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ tokens, const T* __restrict__ W,
                                                        const float* __restrict__ table, T* __restrict__ out, int B, int L,
                                                        int D, float scale, int pad, int pos_offset) {
    // grid (B, L-chunks of 8): every workgroup recounts the non-pad tokens before its chunk (L <= 1024 integer compares) instead of
    // one workgroup per batch row walking all L positions (64 workgroups on 256 CUs: 68 us for 2.6 MB)
    __shared__ int pos[8];
    const int b = blockIdx.x, l0 = blockIdx.y * 8, l1 = min(L, l0 + 8);
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int l = threadIdx.x; l < l0; l += 64) cnt += tokens[(long)b * L + l] != pad;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (threadIdx.x == 0) {
            int c = pos_offset + cnt;            // incremental decoding: tokens before this slice (transformer.py:714-717)
            for (int l = l0; l < l1; ++l) { const bool np = tokens[(long)b * L + l] != pad; c += np; pos[l - l0] = np ? c + pad : pad; }
        }
    }
    __syncthreads();
    for (long i = threadIdx.x; i < (long)(l1 - l0) * D; i += 256) {
        const int l = l0 + (int)(i / D), d = (int)(i % D);
        const long long tok = tokens[(long)b * L + l];
        out[((long)l * B + b) * D + d] = from_f32<T>(scale * to_f32(W[tok * D + d]) + table[(long)pos[l - l0] * D + d]);
    }
}
// dW[tok][:] += scale * dout[l][b][:]   (f32 atomics; the padding row receives no gradient: nn.Embedding padding_idx)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ tokens, const T* __restrict__ dout,
                                                        float* __restrict__ dW, int B, int L, int D, float scale, int pad) {
    const int l = blockIdx.x, b = blockIdx.y;
    const long long tok = tokens[(long)b * L + l];
    if (tok == pad) return;
    for (int d = threadIdx.x; d < D; d += 256) atomicAdd(dW + tok * D + d, scale * to_f32(dout[((long)l * B + b) * D + d]));
}

extern "C" int s2t_embed_fwd(int dtype, const long long* tokens, const void* W, const float* table, void* out, int B,
                             int L, int D, float scale, int pad, int pos_offset, void* stream) {
    if (B <= 0 || L <= 0) return S2T_OK;
    if (!tokens || !W || !table || !out || L > 1024) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(embed_fwd_kernel<bf16>, dim3(B, (L + 7) / 8), dim3(256), 0, st, tokens, (const bf16*)W, table, (bf16*)out, B, L, D, scale, pad, pos_offset);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(B, (L + 7) / 8), dim3(256), 0, st, tokens, (const float*)W, table, (float*)out, B, L, D, scale, pad, pos_offset);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
extern "C" int s2t_embed_bwd(int dtype, const long long* tokens, const void* dout, float* dW, int B, int L, int D,
                             float scale, int pad, void* stream) {
    if (B <= 0 || L <= 0) return S2T_OK;
    if (!tokens || !dout || !dW) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(embed_bwd_kernel<bf16>, dim3(L, B), dim3(256), 0, st, tokens, (const bf16*)dout, dW, B, L, D, scale, pad);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(L, B), dim3(256), 0, st, tokens, (const float*)dout, dW, B, L, D, scale, pad);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------ dropout (Philox, mask regenerated in backward)
// y = x * keep/(1-p) with keep from (seed, element index); backward = the same call on the gradient.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, float p, unsigned long long seed, int vec) {
    const uint32_t th = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    const float inv = 1.f / (1.f - p);
    constexpr int E = 16 / (int)sizeof(T);                      // elements per 16-byte access = E/4 hash quads
    const size_t nv = vec ? n / E : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        T v[E];
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + i * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            const u32x2 h = drop_hash4(seed, i * (E / 4) + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * k + e] = from_f32<T>(drop_field(h, e) >= (th >> 16) ? to_f32(v[4 * k + e]) * inv : 0.f);
        }
        *reinterpret_cast<u32x4*>(y + i * E) = *reinterpret_cast<const u32x4*>(v);
    }
    for (size_t i = nv * E + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = from_f32<T>(dropout_keep(seed, i, th) ? to_f32(x[i]) * inv : 0.f);
}
extern "C" int s2t_dropout(int dtype, const void* x, void* y, size_t n, float p, unsigned long long seed, void* stream) {
    if (n == 0) return S2T_OK;
    if (!x || !y || p < 0.f || p >= 1.f) return S2T_EINVAL;
    const int vec = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const size_t per = dtype == S2T_BF16 ? 8 : 4;
    int blocks = (int)((n / (vec ? per : 1) + 255) / 256);
    blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(dropout_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n, p, seed, vec);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, n, p, seed, vec);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------ activation backward (elementwise) behind a dropout
// out = dropout(dy) * act'(.)   act 1: relu with y = post-activation (mask y > 0); act 2: gelu with y = pre-activation.  The dropout
// (p_drop > 0) is the backward of the one that FOLLOWED the activation in the forward pass (conv_transformer.py:227-232: fc3, act,
// + positions, dropout): the mask of s2t_dropout on the flat element index, applied to dy first, rounded like the separate pass did.
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ out, size_t n, int act,
                                                      float p, unsigned long long seed, int vec) {
    const uint32_t th = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    const float inv = 1.f / (1.f - p);
    constexpr int E = 16 / (int)sizeof(T);
    const size_t nv = vec ? n / E : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        T d[E], v[E];
        *reinterpret_cast<u32x4*>(d) = *reinterpret_cast<const u32x4*>(dy + i * E);
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(y + i * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            u32x2 h = {0u, 0u};
            if (p > 0.f) h = drop_hash4(seed, i * (E / 4) + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = to_f32(d[4 * k + e]);
                if (p > 0.f) g = to_f32(from_f32<T>(drop_field(h, e) >= (th >> 16) ? g * inv : 0.f));
                const float a = to_f32(v[4 * k + e]);
                d[4 * k + e] = from_f32<T>(act == 1 ? (a > 0.f ? g : 0.f) : g * gelu_grad_f(a));
            }
        }
        *reinterpret_cast<u32x4*>(out + i * E) = *reinterpret_cast<const u32x4*>(d);
    }
    for (size_t i = nv * E + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float g = to_f32(dy[i]);
        if (p > 0.f) g = to_f32(from_f32<T>(dropout_keep(seed, i, th) ? g * inv : 0.f));
        const float a = to_f32(y[i]);
        out[i] = from_f32<T>(act == 1 ? (a > 0.f ? g : 0.f) : g * gelu_grad_f(a));
    }
}
extern "C" int s2t_act_bwd(int dtype, const void* dy, const void* y, void* out, size_t n, int act, float p_drop, unsigned long long seed,
                           void* stream) {
    if (n == 0) return S2T_OK;
    if (!dy || !y || !out || (act != 1 && act != 2) || p_drop < 0.f || p_drop >= 1.f) return S2T_EINVAL;
    const int vec = (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) & 15) == 0;
    const size_t per = dtype == S2T_BF16 ? 8 : 4;
    int blocks = (int)((n / (vec ? per : 1) + 255) / 256);
    blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)dy, (const bf16*)y, (bf16*)out, n, act, p_drop, seed, vec);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)out, n, act, p_drop, seed, vec);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------ y += x (gradient merge points)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = from_f32<T>(to_f32(y[i]) + to_f32(x[i]));
}
extern "C" int s2t_add_inplace(int dtype, const void* x, void* y, size_t n, void* stream) {
    if (n == 0) return S2T_OK;
    if (!x || !y) return S2T_EINVAL;
    int blocks = (int)((n + 255) / 256);
    blocks = blocks > 4096 ? 4096 : blocks;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(add_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, n);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// ------------------------------------------------------------------ word-level knowledge distillation loss
// fairseq/criterions/knowledge_distillation.py:44-96, fused with its gradient.  Per non-pad row:
//   kd    = - sum_k softmax(teacher_logits/tau)_k * log_softmax(logits/tau)[idx_k]          (if lambda > 0)
//   truth = - log_softmax(logits)[y]                                                           (if lambda < 1)
//   loss += (1-lambda)*truth + lambda*kd
//   dlogits = gscale * [ (1-lambda)*(softmax(x) - onehot(y)) + (lambda/tau)*(softmax(x/tau) - scatter_k(w_k)) ]
// teacher_idx [rows][Kt] int64, teacher_logits [rows][Kt] f32.
template <typename T>
__global__ __launch_bounds__(256) void kd_kernel(const T* __restrict__ logits, const long long* __restrict__ target,
                                                 const long long* __restrict__ tidx, const float* __restrict__ tlog,
                                                 T* __restrict__ dlogits, float* __restrict__ sums, long rows, int V, int ld, int Kt,
                                                 float lambda, float tau, int pad, float gscale) {
    __shared__ float sh[16];
    __shared__ float wk[64];
    float acc_loss = 0.f;                            // rows per workgroup, one atomic at the end (as lsce_kernel)
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* x = logits + row * ld;
    T* g = dlogits ? dlogits + row * ld : nullptr;
    const long long y = target[row];
    if (y == pad) {
        if (g) for (int v = threadIdx.x; v < V; v += 256) g[v] = from_f32<T>(0.f);
        continue;
    }
    const float it = 1.f / tau;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, to_f32(x[v]));
    m = block_max(m, sh);
    float s1 = 0.f, st = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) { const float d = to_f32(x[v]) - m; s1 += expf(d); st += expf(d * it); }
    s1 = block_sum(s1, sh);
    st = block_sum(st, sh);
    const float lse1 = m + logf(s1), lset = m * it + logf(st);
    // teacher weights softmax(tlog / tau) over Kt <= 64 entries (one wave)
    if (threadIdx.x < 64) {
        const float tv = threadIdx.x < Kt ? tlog[row * Kt + threadIdx.x] * it : -INFINITY;
        const float tm = wave_max(tv);
        const float te = threadIdx.x < Kt ? expf(tv - tm) : 0.f;
        const float ts = wave_sum(te);
        wk[threadIdx.x] = te / ts;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float kd = 0.f;
        if (lambda > 0.f) for (int k = 0; k < Kt; ++k) kd -= wk[k] * (to_f32(x[tidx[row * Kt + k]]) * it - lset);
        const float truth = lambda < 1.f ? lse1 - to_f32(x[y]) : 0.f;
        acc_loss += (1.f - lambda) * truth + lambda * kd;
    }
    if (g) {
        for (int v = threadIdx.x; v < V; v += 256) {
            const float xv = to_f32(x[v]);
            float d = 0.f;
            if (lambda < 1.f) d += (1.f - lambda) * (expf(xv - lse1) - (v == y ? 1.f : 0.f));
            if (lambda > 0.f) d += lambda * it * expf(xv * it - lset);
            g[v] = from_f32<T>(d * gscale);
        }
        __syncthreads();
        if (lambda > 0.f && threadIdx.x == 0)          // few entries, possibly repeated indices: serial fix-up
            for (int k = 0; k < Kt; ++k) {
                const long long c = tidx[row * Kt + k];
                g[c] = from_f32<T>(to_f32(g[c]) - lambda * it * wk[k] * gscale);
            }
    }
    __syncthreads();                                 // `sh` / `wk` are reused by the next row
    }
    if (threadIdx.x == 0) { sums[2 * blockIdx.x] = acc_loss; sums[2 * blockIdx.x + 1] = 0.f; }      // partials, as lsce_kernel: lsce_finish_kernel adds them
}

extern "C" int s2t_kd_loss(int dtype, const void* logits, const long long* target, const long long* teacher_idx,
                           const float* teacher_logits, void* dlogits, float* sum1, long rows, int V, int ld, int Kt,
                           float lambda, float tau, int pad, float grad_scale, void* stream) {
    if (rows <= 0) return S2T_OK;
    if (!logits || !target || !sum1 || V <= 0 || ld < V || tau <= 0.f || lambda < 0.f || lambda > 1.f) return S2T_EINVAL;
    if (lambda > 0.f && (!teacher_idx || !teacher_logits || Kt < 1 || Kt > 64)) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(rows < 1024 ? rows : 1024);
    hipError_t se = hipSuccess;                                        // 1,024 x 2 partial sums, per (device, stream)
    float* part = (float*)s2t_scratch(S2T_SCRATCH_KD, st, 2 * 1024 * sizeof(float), &se);
    if (!part) return S2T_EHIP(se);
    if (dtype == S2T_BF16) hipLaunchKernelGGL(kd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)logits, target, teacher_idx, teacher_logits, (bf16*)dlogits, part, rows, V, ld, Kt, lambda, tau, pad, grad_scale);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(kd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)logits, target, teacher_idx, teacher_logits, (float*)dlogits, part, rows, V, ld, Kt, lambda, tau, pad, grad_scale);
    else return S2T_ENOTSUP;
    hipLaunchKernelGGL(lsce_finish_kernel, dim3(1), dim3(256), 0, st, part, (int)grid, sum1, 0);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}


// ------------------------------------------------------------------ log-softmax rows (generation)
// out[row][v] = log_softmax(logits[row][:] * inv_temperature)[v] in f32
// (sequence_generator.py:711-768 EnsembleModel.forward_decoder -> get_normalized_probs(log_probs=True))
template <typename T, bool LOG = true>
__global__ __launch_bounds__(256) void log_softmax_kernel(const T* __restrict__ logits, float* __restrict__ out, int V, int ld, float it) {
    __shared__ float sh[16];
    const long row = blockIdx.x;
    const T* x = logits + row * ld;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, to_f32(x[v]) * it);
    m = block_max(m, sh);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) s += expf(to_f32(x[v]) * it - m);
    s = block_sum(s, sh);
    const float lse = m + logf(s);
    for (int v = threadIdx.x; v < V; v += 256) {
        const float lp = to_f32(x[v]) * it - lse;
        out[row * V + v] = LOG ? lp : expf(lp);
    }
}
// gradient of log_softmax / softmax rows w.r.t. the logits from the saved output (include/s2t_hip.h: s2t_softmax_bwd)
template <typename T, bool LOG>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout, T* __restrict__ dx,
                                                          int V, int ld, float it) {
    __shared__ float sh[16];
    const long row = blockIdx.x;
    const float* y = out + row * V;
    const float* g = dout + row * V;
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) s += LOG ? g[v] : g[v] * y[v];
    s = block_sum(s, sh);
    for (int v = threadIdx.x; v < V; v += 256) {
        const float pv = LOG ? expf(y[v]) : y[v];
        dx[row * ld + v] = from_f32<T>(it * (LOG ? g[v] - pv * s : pv * (g[v] - s)));
    }
}
extern "C" int s2t_log_softmax(int dtype, const void* logits, float* out, long rows, int V, int ld, float inv_temperature, void* stream) {
    if (rows <= 0) return S2T_OK;
    if (!logits || !out || V <= 0 || ld < V) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(log_softmax_kernel<bf16>, dim3((unsigned)rows), dim3(256), 0, st, (const bf16*)logits, out, V, ld, inv_temperature);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(log_softmax_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, out, V, ld, inv_temperature);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_softmax_probs(int dtype, const void* logits, float* out, long rows, int V, int ld, float inv_temperature, void* stream) {
    if (rows <= 0) return S2T_OK;
    if (!logits || !out || V <= 0 || ld < V) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL((log_softmax_kernel<bf16, false>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16*)logits, out, V, ld, inv_temperature);
    else if (dtype == S2T_F32) hipLaunchKernelGGL((log_softmax_kernel<float, false>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, out, V, ld, inv_temperature);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
extern "C" int s2t_softmax_bwd(int dtype, const float* out, const float* dout, void* dlogits, long rows, int V, int ld, float inv_temperature,
                               int log_probs, void* stream) {
    if (rows <= 0) return S2T_OK;
    if (!out || !dout || !dlogits || V <= 0 || ld < V) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define S2T_SMB(T_, LOG_) hipLaunchKernelGGL((softmax_bwd_kernel<T_, LOG_>), dim3((unsigned)rows), dim3(256), 0, st, out, dout, (T_*)dlogits, V, ld, inv_temperature)
    if (dtype == S2T_BF16) { if (log_probs) S2T_SMB(bf16, true); else S2T_SMB(bf16, false); }
    else if (dtype == S2T_F32) { if (log_probs) S2T_SMB(float, true); else S2T_SMB(float, false); }
    else return S2T_ENOTSUP;
#undef S2T_SMB
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// Ensemble of M models (fairseq/sequence_generator.py:757-768): log of the MEAN probability = logsumexp over the members' log-
// probabilities - log M, element-wise over the [hypotheses, V] rows.  One pass, HBM-bound; members by pointer (M <= 8).
struct EnsPtrs { const float* p[8]; };
__global__ __launch_bounds__(256) void ensemble_lse_kernel(EnsPtrs in, int n, float* __restrict__ out, size_t numel, float log_n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (size_t)gridDim.x * 256) {
        float x[8];
        float m = -INFINITY;
        for (int j = 0; j < n; ++j) { x[j] = in.p[j][i]; m = fmaxf(m, x[j]); }
        float r = -INFINITY;
        if (m > -INFINITY) {
            float s = 0.f;
            for (int j = 0; j < n; ++j) s += expf(x[j] - m);
            r = m + logf(s) - log_n;
        }
        out[i] = r;
    }
}
extern "C" int s2t_ensemble_lse(int n, const float* const* lprobs, float* out, size_t numel, void* stream) {
    if (numel == 0) return S2T_OK;
    if (n < 1 || n > 8 || !lprobs || !out) return S2T_EINVAL;
    EnsPtrs ptrs;
    for (int j = 0; j < 8; ++j) { ptrs.p[j] = j < n ? lprobs[j] : nullptr; if (j < n && !lprobs[j]) return S2T_EINVAL; }
    size_t blocks = (numel + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ensemble_lse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ptrs, n, out, numel, logf((float)n));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}


// ------------------------------------------------------------------ top-k of logit rows (teacher dump for knowledge distillation)
// vals[row][0..K) = the K largest logits of the row in descending order, idx = their columns (ties: lower column first)
// (scripts/generate_topk.py:64-66 torch.topk(net_output[0], k, dim=-1)).  One wavefront per row, K selection rounds: every round
// takes the largest element that comes after the previous pick in (value desc, column asc) order.
template <typename T>
__global__ __launch_bounds__(256) void topk_kernel(const T* __restrict__ x, float* __restrict__ vals, int* __restrict__ idx, long rows, int V,
                                                   int ld, int K) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const T* xr = x + row * ld;
    float pv = INFINITY; int pi = -1;
    for (int k = 0; k < K; ++k) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int c = lane; c < V; c += 64) {
            const float v = to_f32(xr[c]);
            const bool after = v < pv || (v == pv && c > pi);
            if (after && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        pv = bv; pi = bi;
        if (lane == 0) { vals[row * K + k] = bv; idx[row * K + k] = bi; }
    }
}
extern "C" int s2t_topk(int dtype, const void* x, float* vals, int* idx, long rows, int V, int ld, int K, void* stream) {
    if (rows <= 0) return S2T_OK;
    if (!x || !vals || !idx || V <= 0 || ld < V || K <= 0 || K > V) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (dtype == S2T_BF16) hipLaunchKernelGGL(topk_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, vals, idx, rows, V, ld, K);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(topk_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, vals, idx, rows, V, ld, K);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
