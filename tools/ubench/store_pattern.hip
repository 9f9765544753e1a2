// Micro-benchmark: streaming-store bandwidth as a function of the contiguous bytes each wave-instruction writes per row.
// Output matrix [M][N] bf16 (row stride N*2 bytes); a workgroup of 256 threads owns a 128 x 256 tile and writes it with
// 16 bytes per lane; SEG = contiguous bytes per row per instruction (128, 256, 512).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int SEG>
__global__ __launch_bounds__(256) void k(char* out, int M, int N, int tiles_n) {
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int LPR = SEG / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;          // rows per instruction
    const size_t rs = (size_t)N * 2;
    // the tile is 128 rows x 512 bytes; wave w handles column segment(s)
    const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    constexpr int SEGS = 512 / SEG;        // segments per tile row
    for (int sidx = wave; sidx < SEGS * (128 / RPI); sidx += 4) {
        const int seg = sidx % SEGS, rb = sidx / SEGS;
        const int row = tm * 128 + rb * RPI + lane / LPR;
        const size_t off = (size_t)row * rs + (size_t)tn * 512 + seg * SEG + (lane % LPR) * 16;
        *reinterpret_cast<u32x4*>(out + off) = v;
    }
}
int main() {
    const int M = 43776, N = 4096;
    char* out; hipMalloc(&out, (size_t)M * N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles_n = N * 2 / 512, tiles = (M / 128) * tiles_n;
    auto run = [&](auto kern, const char* name) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, out, M, N, tiles_n);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, out, M, N, tiles_n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us  %.2f TB/s\n", name, ms * 100, (double)M * N * 2 / (ms / 10 * 1e-3) / 1e12);
    };
    run(k<128>, "128 B per row per instruction");
    run(k<256>, "256 B per row per instruction");
    run(k<512>, "512 B per row per instruction");
    hipMemset(out, 0, (size_t)M * N * 2);
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipMemsetAsync(out, 1, (size_t)M * N * 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("memset: %.1f us %.2f TB/s\n", ms * 100, (double)M * N * 2 / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
