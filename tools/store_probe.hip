// How fast can ONE workgroup of 8 waves put a 256 x 256 bf16 tile (128 KiB, 16 x 16-byte stores per lane) into memory, by the shape of
// a store instruction's footprint?  (gemm256's epilogue: 16 rows x 64-byte segments per wave-instruction.)
//   hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// seg = bytes of a row that one instruction covers contiguously (64, 128, 256, 1024); ld = row stride of the output in bytes
template <int SEG, bool NT>
__global__ __launch_bounds__(512) void probe(char* out, long ld, int tiles_per_wg, long tile_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16;                   // lanes per row segment
    constexpr int RPI = 64 / LPR;                   // rows per instruction
    u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    for (int t = 0; t < tiles_per_wg; ++t) {
        char* base = out + ((long)blockIdx.x * tiles_per_wg + t) * tile_stride;
        // the wave owns 16 instructions: a [16 * RPI rows] x [SEG bytes] block per instruction group
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long row = (long)(wave * 16 + i) * RPI + lane / LPR;
            char* p = base + row * ld + (lane % LPR) * 16;
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
            else *reinterpret_cast<u32x4*>(p) = v;
        }
    }
}

template <int SEG, bool NT> float run(char* buf, int grid, int tiles, long ld, long tile_stride, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<SEG, NT>), dim3(grid), dim3(512), 0, 0, buf, ld, tiles, tile_stride);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<SEG, NT>), dim3(grid), dim3(512), 0, 0, buf, ld, tiles, tile_stride);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main() {
    char* buf; const size_t bytes = (size_t)1 << 31; hipMalloc(&buf, bytes);
    const int tiles = 8;                                       // per workgroup and launch: amortises the launch itself
    for (int grid : {1, 256}) {
        printf("grid %d workgroups x %d tiles of 128 KiB per launch\n", grid, tiles);
        const long ts = 131072;                                // tiles are packed: rows of a tile are `ld` apart inside its own 128 KiB x (ld/ (SEG)) region
#define RUN(SEG, NT) { const long ld = 4096; const long tstride = (long)(8 * 16 * (64 / (SEG / 16))) * ld; \
            if ((size_t)grid * tiles * tstride <= bytes) { float us = run<SEG, NT>(buf, grid, tiles, ld, tstride, 20); \
            printf("  segment %4d B%s: %8.1f us per launch = %6.2f us per tile, %6.1f B/clk/CU at 2.1 GHz, %7.2f TB/s aggregate\n", SEG, NT ? " nontemporal" : "            ", \
                   us, us / tiles, 131072.0 / (us / tiles * 2100.0), (double)grid * tiles * 131072.0 / us * 1e-6); } }
        RUN(64, false) RUN(128, false) RUN(256, false) RUN(1024, false) RUN(64, true) RUN(1024, true)
        (void)ts;
    }
    return 0;
}
