// Does a chip-wide burst of tile stores complete at HBM rate, or do L2 / the Infinity Cache absorb it when the memory system was idle
// before (as it is during a GEMM's main loop)?  Every workgroup (one per CU, 8 waves): spin `gap` cycles, stamp, store `kb` KiB
// (16-byte stores, 64-byte row segments as gemm256's epilogue), s_waitcnt vmcnt(0), stamp.  Prints the median / max completion time.
//   hipcc -O3 --offload-arch=gfx950 tools/store_burst_probe.hip -o /tmp/sbp && /tmp/sbp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// the workgroups' tiles lie as in a GEMM output [rows][2048 bf16]: tile (mt, nt) of 16 * nstore rows x 256 columns, 8 column tiles per row block
__global__ __launch_bounds__(512) void burst(char* out, long ld, int nstore, int rounds, long gap, unsigned long long* times, long region) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long s = wall_clock64();
        while (wall_clock64() - s < (unsigned long long)gap) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        // wave (wr, wc) = (wave / 4, wave % 4): rows wr * 8 * nstore .. of the tile, columns 64 wc .. (two 32-column halves 128 apart, as gemm256)
        char* base = out + (long)r * region + (long)(blockIdx.x / 8) * (16 * nstore) * ld + (blockIdx.x % 8) * 512;
        const int wr = wave >> 2, wc = wave & 3;
        for (int i = 0; i < nstore; ++i) {
            const long row = (long)wr * 8 * nstore + (i >> 1) * 16 + (lane & 15);
            *reinterpret_cast<u32x4*>(base + row * ld + (i & 1) * 256 + wc * 64 + (lane >> 4) * 16) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) times[(long)r * gridDim.x + blockIdx.x] = t1 - t0;
    }
}

int main() {
    char* buf; const size_t bytes = (size_t)6 << 30; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    unsigned long long* times; hipMalloc(&times, 256 * 64 * sizeof(unsigned long long));
    const int rounds = 12;
    for (int grid : {1, 64, 256})
        for (int nstore : {4, 8, 16})                        // per lane: 32 / 64 / 128 KiB per workgroup and burst
            for (long gap_us : {0L, 10L}) {
                const long ld = 4096, region = (long)((grid + 7) / 8) * 16 * nstore * ld;   // one output matrix per round
                if ((size_t)rounds * region > bytes) continue;
                const long gap = gap_us * 100;                                   // wall_clock64 ticks at 100 MHz
                hipLaunchKernelGGL(burst, dim3(grid), dim3(512), 0, 0, buf, ld, nstore, rounds, gap, times, region);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h((size_t)rounds * grid);
                hipMemcpy(h.data(), times, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                std::vector<unsigned long long> late(h.begin() + 4 * grid, h.end());     // skip the first rounds (cold TLB / page faults)
                std::sort(late.begin(), late.end());
                printf("grid %3d, %3d KiB per workgroup and burst (%5.1f MB chip-wide), %2ld us idle before: median %6llu cycles, max %6llu\n",
                       grid, nstore * 8, grid * nstore * 8 / 1024.0, gap_us, late[late.size() / 2], late.back());
            }
    return 0;
}
