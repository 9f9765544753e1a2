// Per-CU store throughput (measured: one workgroup alone 125-133 GB/s = 1.0 us per 128 KiB tile; 64 at once 71 GB/s each =
// 4.5 TB/s in total, the chip's write rate): G workgroups of 512 threads, each writes `iters` distinct 128 KiB tiles (256 rows x 512 B at a row
// stride of `ld` bytes, like a 256x256 bf16 output tile), 16 bytes per lane.  Prints GB/s per workgroup and in total.
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void store_tiles(char* out, long ld, int iters, int tiles_per_row, int mode) {
    const int tid = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const long tile = (long)blockIdx.x * iters + it;
        const long tr = tile / tiles_per_row, tc = tile % tiles_per_row;
        char* base = out + tr * 256 * ld + tc * 512;
        const f32x4 v = {1.f * tid, 2.f, 3.f, (float)it};
        if (mode == 0) {                 // a wave stores 2 rows x 512 B per instruction (32 lanes x 16 B per row)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i * 16 + (tid >> 5), col = (tid & 31) * 16;
                *reinterpret_cast<f32x4*>(base + row * ld + col) = v;
            }
        } else {                         // a wave stores 16 rows x 64 B per instruction (MFMA-fragment shaped)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int wave = tid >> 6, lane = tid & 63;
                const int row = (wave >> 2) * 128 + (i >> 1) * 16 + (lane & 15), col = (wave & 3) * 128 + (i & 1) * 64 + (lane >> 4) * 16;
                *reinterpret_cast<f32x4*>(base + row * ld + col) = v;
            }
        }
    }
}
int main() {
    const long ld = 6144;                       // bytes per output row (N = 3072 bf16)
    const int tiles_per_row = ld / 512;
    const long bytes = 1L << 30;              // tiles written per launch; the buffer also holds the unused end of the last tile row
    char* out;
    if (hipMalloc(&out, bytes + 256 * ld) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int G : {1, 8, 32, 64, 128, 256, 512, 1024}) {
            int iters = (int)(bytes / (128 * 1024) / G); if (iters > 64) iters = 64;
            store_tiles<<<G, 512>>>(out, ld, iters, tiles_per_row, mode);
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            (void)hipEventRecord(e0);
            store_tiles<<<G, 512>>>(out, ld, iters, tiles_per_row, mode);
            (void)hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) return 3;
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            const double gb = (double)G * iters * 128 * 1024 / 1e9;
            printf("mode %d  G %4d  iters %3d  %.3f ms  per-WG %.1f GB/s  total %.0f GB/s  (%.2f us per 128 KiB tile)\n", mode, G, iters, ms,
                   gb / G / (ms * 1e-3), gb / (ms * 1e-3), ms * 1e3 / iters);
        }
    return 0;
}
