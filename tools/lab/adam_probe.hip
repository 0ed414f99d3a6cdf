// Adam over a flat 74 M-parameter arena: which form of the streaming loop gets closest to the HBM rate?  (30 bytes per parameter:
// p, g, m, v read; p, m, v written in f32; the bf16 shadow written.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/adam_probe.hip -o tools/_bin/adam_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16;

__device__ __forceinline__ void adam4(f32x4& p, f32x4 g, f32x4& m, f32x4& v, float mult, float b1, float b2, float eps, float ss) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gi = g[e] * mult;
        m[e] = b1 * m[e] + (1.f - b1) * gi;
        v[e] = b2 * v[e] + (1.f - b2) * gi * gi;
        p[e] -= ss * m[e] / (sqrtf(v[e]) + eps);
    }
}
__device__ __forceinline__ u32x2 pack4(f32x4 p) {
    bf16 s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = (bf16)p[e];
    return *reinterpret_cast<const u32x2*>(s);
}
// MODE 0: grid-stride, one 16-byte group per thread and step (the shipped form).  1: two groups per step.  2: as 1, non-temporal.
// 3: four groups per step, non-temporal.  4: one group, non-temporal.
template <int MODE>
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              bf16* __restrict__ sh, size_t n) {
    constexpr int U = MODE == 0 || MODE == 4 ? 1 : (MODE == 3 ? 4 : 2);
    constexpr bool NT = MODE >= 2;
    const size_t nv = n / 4, stride = (size_t)gridDim.x * 256;
    f32x4* P = reinterpret_cast<f32x4*>(p); const f32x4* G = reinterpret_cast<const f32x4*>(g);
    f32x4* Mv = reinterpret_cast<f32x4*>(m); f32x4* V = reinterpret_cast<f32x4*>(v); u32x2* S = reinterpret_cast<u32x2*>(sh);
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < nv; i0 += stride * U) {
        f32x4 pi[U], gi[U], mi[U], vi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * stride;
            if (i < nv) {
                if constexpr (NT) { pi[u] = __builtin_nontemporal_load(P + i); gi[u] = __builtin_nontemporal_load(G + i);
                                    mi[u] = __builtin_nontemporal_load(Mv + i); vi[u] = __builtin_nontemporal_load(V + i); }
                else { pi[u] = P[i]; gi[u] = G[i]; mi[u] = Mv[i]; vi[u] = V[i]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * stride;
            if (i < nv) {
                adam4(pi[u], gi[u], mi[u], vi[u], 1.f, 0.9f, 0.98f, 1e-8f, 1e-3f);
                if constexpr (NT) { __builtin_nontemporal_store(mi[u], Mv + i); __builtin_nontemporal_store(vi[u], V + i);
                                    __builtin_nontemporal_store(pi[u], P + i); __builtin_nontemporal_store(pack4(pi[u]), S + i); }
                else { Mv[i] = mi[u]; V[i] = vi[u]; P[i] = pi[u]; S[i] = pack4(pi[u]); }
            }
        }
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int MODE> static int run(float* p, float* g, float* m, float* v, bf16* sh, size_t n, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(adam_k<MODE>, dim3(blocks), dim3(256), 0, 0, p, g, m, v, sh, n);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(adam_k<MODE>, dim3(blocks), dim3(256), 0, 0, p, g, m, v, sh, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("mode %d, %5d workgroups: %.1f us = %.2f TB/s\n", MODE, blocks, best * 100, n * 30.0 / (best * 100) / 1e6);
    return 0;
}
int main() {
    const size_t n = 74000000;
    float *p, *g, *m, *v; bf16* sh;
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&sh, n * 2));
    CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        if (run<0>(p, g, m, v, sh, n, blocks)) return 1;
        if (run<1>(p, g, m, v, sh, n, blocks)) return 1;
        if (run<2>(p, g, m, v, sh, n, blocks)) return 1;
        if (run<3>(p, g, m, v, sh, n, blocks)) return 1;
        if (run<4>(p, g, m, v, sh, n, blocks)) return 1;
    }
    return 0;
}
