// How does the MI355X hold its clock under MFMA-dense phases: by the AVERAGE over phases (a kernel that idles a quarter of the time may
// run its dense quarter-hours faster) or phase by phase (the dense phase runs at the same clock whatever surrounds it)?
// Every workgroup (4 waves, one per SIMD, 256 workgroups) alternates a dense phase of `n_dense` back-to-back v_mfma_f32_32x32x16_bf16
// on random operands (register-resident) with an idle phase of `idle` x s_sleep 127, for `reps` rounds; wave 0 of workgroup 0 stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around every dense phase.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lab/dvfs_probe.hip -o tools/_bin/dvfs_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void dvfs_kernel(float* out, unsigned long long* st, int n_dense, int idle, int reps) {
    const int lane = threadIdx.x & 63;
    u32x4 a[4], b[4];
    uint32_t s = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; return __float_as_uint(f) >> 16; };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[i][e] = rnd() | (rnd() << 16); b[i][e] = rnd() | (rnd() << 16); }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const bool stamp = blockIdx.x == 0 && threadIdx.x == 0;
    for (int r = 0; r < reps; ++r) {
        unsigned long long c0, t0, c1, t1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(t0) :: "memory");
        for (int k = 0; k < n_dense; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[(i + k) & 3]), acc[i], 0, 0, 0);
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(t1) :: "memory");
        if (stamp) { st[4 * r] = c0; st[4 * r + 1] = t0; st[4 * r + 2] = c1; st[4 * r + 3] = t1; }
        for (int k = 0; k < idle; ++k) asm volatile("s_sleep 127" ::: "memory");
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][15];
    if (sum == 123.456f) out[lane] = sum;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    float* out; unsigned long long* st;
    const int reps = 400;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&st, reps * 4 * 8));
    std::vector<unsigned long long> h(reps * 4);
    // dense phase: n_dense x 4 MFMAs of 32 cycles per SIMD = 128 n_dense cycles (~25 us at n_dense = 400)
    for (int n_dense : {400, 100}) {
        for (int idle : {0, 1, 3, 8, 24}) {               // idle is scaled below so that the idle phase is idle/8 .. 3x of the dense one
            const int idle_sleeps = idle * n_dense / 16;    // s_sleep 127 ~ 8,128 cycles?  (64 x 127); calibrated by the stamps below
            for (int warm = 0; warm < 2; ++warm) {
                hipLaunchKernelGGL(dvfs_kernel, dim3(256), dim3(256), 0, 0, out, st, n_dense, idle_sleeps, reps);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> ghz, dense_us, period_us;
            for (int r = reps / 2; r < reps - 1; ++r) {
                const double cyc = (double)(h[4 * r + 2] - h[4 * r]), us = (double)(h[4 * r + 3] - h[4 * r + 1]) * 0.01;
                ghz.push_back(cyc / us * 1e-3); dense_us.push_back(us);
                period_us.push_back((double)(h[4 * (r + 1) + 1] - h[4 * r + 1]) * 0.01);
            }
            std::sort(ghz.begin(), ghz.end()); std::sort(dense_us.begin(), dense_us.end()); std::sort(period_us.begin(), period_us.end());
            const double d = dense_us[dense_us.size() / 2], p = period_us[period_us.size() / 2];
            printf("dense phase %4d x 4 MFMA: %6.1f us of every %6.1f us (duty %.2f): clock in the dense phase %.2f GHz (median), %.0f TFLOP/s inside it\n",
                   n_dense, d, p, d / p, ghz[ghz.size() / 2], 256.0 * 4 * n_dense * 4 * 32768.0 / d * 1e-6);
        }
    }
    return 0;
}
