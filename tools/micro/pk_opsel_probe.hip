// Does a packed fp32 multiply-add that takes the HIGH register of a source pair for its LOW result (op_sel:[0,1,0]) give the
// same bits as two scalar multiply-adds while waves of another kernel share the SIMD?
//
// Victim: every lane repeats the two-instruction sequence of the GEMM epilogue with a folded LayerNorm
//     t = -mean * u + a          (v_pk_fma_f32 ... op_sel_hi:[0,1,1] neg_lo neg_hi, mean = low register of the pair s)
//     t =  t * rstd + c          (v_pk_fma_f32 ... op_sel:[0,1,0],              rstd = high register of the pair s)
// on two interleaved chains and counts, per lane and per half, results that differ from v_fma_f32.  Aggressors run on a
// second stream at the same time: DPP moves, MFMA, exp2, LDS traffic, global streaming, packed converts.
//
//     hipcc --offload-arch=gfx950 -O2 tools/micro/pk_opsel_probe.hip -o tools/exp/pk_opsel_probe && tools/exp/pk_opsel_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                      \
            std::exit(1);                                                                   \
        }                                                                                   \
    } while (0)

template <int FORM>
__global__ __launch_bounds__(256) void victim(const f32x2* __restrict__ stat, unsigned* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    const f32x2 s = stat[(blockIdx.x * 256 + threadIdx.x) & 1023];      // (mean, rstd) from memory, as in the kernel
    f32x2 u0 = {0.5f + lane, 0.25f * lane}, u1 = {1.5f, -0.75f + lane};
    f32x2 a0 = {1.0f * lane, 2.0f + lane}, a1 = {-3.0f, 0.5f * lane};
    const f32x2 c0 = {0.125f, -0.25f}, c1 = {4.0f, 0.0625f};
    unsigned lo = 0, hi = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 t0, t1;
        if (FORM == 0) {        // rstd taken from the high register for both results
            asm volatile(
                "v_pk_fma_f32 %0, %2, %3, %5 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                "v_pk_fma_f32 %1, %2, %4, %6 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                "v_pk_fma_f32 %0, %0, %2, %7 op_sel:[0,1,0]\n"
                "v_pk_fma_f32 %1, %1, %2, %8 op_sel:[0,1,0]\n"
                : "=&v"(t0), "=&v"(t1)
                : "v"(s), "v"(u0), "v"(u1), "v"(a0), "v"(a1), "v"(c0), "v"(c1));
        } else {                // the same arithmetic with the factor broadcast into a pair of its own: no op_sel on it
            const f32x2 r2 = {s[1], s[1]};
            asm volatile(
                "v_pk_fma_f32 %0, %2, %3, %5 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                "v_pk_fma_f32 %1, %2, %4, %6 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                "v_pk_fma_f32 %0, %0, %9, %7\n"
                "v_pk_fma_f32 %1, %1, %9, %8\n"
                : "=&v"(t0), "=&v"(t1)
                : "v"(s), "v"(u0), "v"(u1), "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(r2));
        }
        float r00, r01, r10, r11;
        asm volatile(
            "v_fma_f32 %0, -%4, %5, %9\n"
            "v_fma_f32 %1, -%4, %6, %10\n"
            "v_fma_f32 %2, -%4, %7, %11\n"
            "v_fma_f32 %3, -%4, %8, %12\n"
            : "=&v"(r00), "=&v"(r01), "=&v"(r10), "=&v"(r11)
            : "v"(s[0]), "v"(u0[0]), "v"(u0[1]), "v"(u1[0]), "v"(u1[1]), "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]));
        r00 = __builtin_fmaf(r00, s[1], c0[0]);
        r01 = __builtin_fmaf(r01, s[1], c0[1]);
        r10 = __builtin_fmaf(r10, s[1], c1[0]);
        r11 = __builtin_fmaf(r11, s[1], c1[1]);
        lo += (t0[0] != r00) + (t1[0] != r10);
        hi += (t0[1] != r01) + (t1[1] != r11);
        a0[0] += 1.0f;
        a1[1] -= 0.5f;
        u0[1] += 0.125f;
    }
    if (lo) atomicAdd(&bad[lane], lo);
    if (hi) atomicAdd(&bad[64 + lane], hi);
}

template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* __restrict__ sink, const float* __restrict__ src, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    float v = tid * 0.001f + 1.0f, w = 0.5f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) {
        fa[i] = (__bf16)(0.01f * (tid + i));
        fb[i] = (__bf16)(0.02f * (tid - i));
    }
    lds[tid] = v;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {            // DPP moves inside rows of 16 lanes
            int b = __builtin_bit_cast(int, v);
            b = __builtin_amdgcn_update_dpp(b, b, 0x140, 0xf, 0xf, false);      // row_mirror
            v += __builtin_bit_cast(float, b) * 1e-9f;
            b = __builtin_amdgcn_update_dpp(b, b, 0x4E, 0xf, 0xf, false);       // quad_perm
            w += __builtin_bit_cast(float, b) * 1e-9f;
        } else if (KIND == 1) {     // matrix pipe
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
        } else if (KIND == 2) {     // transcendental
            v = __builtin_amdgcn_exp2f(v * 0.001f) + w;
        } else if (KIND == 3) {     // LDS traffic
            lds[(tid * 17 + it) & 4095] = v;
            v += lds[(tid * 33 + it * 7) & 4095] * 1e-9f;
        } else if (KIND == 4) {     // global streaming
            v += src[((size_t)blockIdx.x * 256 + tid + (size_t)it * 65536) & ((1u << 24) - 1)] * 1e-9f;
        } else if (KIND == 5) {     // packed converts and packed fp32 with swapped halves
            f32x2 p = {v, w};
            asm volatile("v_pk_mul_f32 %0, %0, %0 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(p));
            v = p[0] * 1e-9f + 1.0f;
            w = p[1] * 1e-9f + 0.5f;
        }
    }
    sink[blockIdx.x * 256 + tid] = v + w + acc[0] + acc[1] + acc[2] + acc[3];
}

// One packed instruction per iteration, the form given by F, against scalar arithmetic on the halves op_sel / op_sel_hi name.
#define PK3(ASM, L0, L1, L2, H0, H1, H2)                                                                             \
    {                                                                                                                \
        asm volatile(ASM : "=&v"(t) : "v"(x), "v"(y), "v"(z));                                                      \
        rl = __builtin_fmaf(x[L0], y[L1], z[L2]);                                                                    \
        rh = __builtin_fmaf(x[H0], y[H1], z[H2]);                                                                    \
    }
#define PK2(ASM, OP, L0, L1, H0, H1)                                                                                 \
    {                                                                                                                \
        asm volatile(ASM : "=&v"(t) : "v"(x), "v"(y));                                                              \
        rl = x[L0] OP y[L1];                                                                                         \
        rh = x[H0] OP y[H1];                                                                                         \
    }
template <int F>
__global__ __launch_bounds__(256) void victim_form(const f32x2* __restrict__ stat, unsigned* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    const f32x2 y = stat[(blockIdx.x * 256 + threadIdx.x) & 1023];
    f32x2 x = {0.5f + lane, 0.25f * lane + 1.0f};
    const f32x2 z = {0.125f + lane, -0.25f};
    unsigned lo = 0, hi = 0;
    for (int it = 0; it < iters; ++it) {
#pragma clang fp contract(off)
        f32x2 t;
        float rl, rh;
        if constexpr (F == 0) PK3("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", 0, 1, 0, 1, 1, 1)
        else if constexpr (F == 1) PK3("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", 1, 0, 0, 1, 1, 1)
        else if constexpr (F == 2) PK3("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]", 0, 0, 1, 1, 1, 1)
        else if constexpr (F == 3) PK3("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]", 0, 0, 0, 0, 1, 1)
        else if constexpr (F == 4) PK3("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]", 0, 0, 0, 1, 0, 1)
        else if constexpr (F == 5) PK3("v_pk_fma_f32 %0, %1, %2, %3", 0, 0, 0, 1, 1, 1)
        else if constexpr (F == 6) PK2("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]", *, 1, 0, 0, 1)
        else if constexpr (F == 7) PK2("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]", *, 0, 1, 0, 1)
        else if constexpr (F == 8) PK2("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", +, 0, 1, 1, 0)
        else if constexpr (F == 9) PK2("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]", *, 0, 1, 0, 0)
        else if constexpr (F == 10) PK2("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]", *, 1, 1, 1, 1)
        else PK2("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]", +, 1, 0, 1, 1)
        lo += (t[0] != rl);
        hi += (t[1] != rh);
        x[0] += 1.0f;
        x[1] += 0.125f;
    }
    if (lo) atomicAdd(&bad[lane], lo);
    if (hi) atomicAdd(&bad[64 + lane], hi);
}

// The packed instruction and the MFMA loop inside ONE dispatch.  Workgroup ids go round the eight XCDs, and inside an XCD
// probably round its CUs, so the role is the parity of the bits of id >> 3: however ids are dealt out in powers of two,
// every CU gets workgroups of both kinds, and their waves share SIMDs as they do when two kernels overlap.  Launched
// with 2 * blocks workgroups; workgroup id writes sink[id * 256 ..], and sink holds 2 * blocks * 256 floats.
__global__ __launch_bounds__(256) void one_dispatch(const f32x2* __restrict__ stat, unsigned* __restrict__ bad, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    if (__popc(blockIdx.x >> 3) & 1) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bf16x8 fa, fb;
        for (int i = 0; i < 8; ++i) {
            fa[i] = (__bf16)(0.01f * (threadIdx.x + i));
            fb[i] = (__bf16)(0.02f * (threadIdx.x - i));
        }
        for (int it = 0; it < iters * 4; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
        sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
        return;
    }
    const f32x2 y = stat[(blockIdx.x * 256 + threadIdx.x) & 1023];
    f32x2 x = {0.5f + lane, 0.25f * lane + 1.0f};
    const f32x2 z = {0.125f + lane, -0.25f};
    unsigned lo = 0, hi = 0;
    for (int it = 0; it < iters; ++it) {
#pragma clang fp contract(off)
        f32x2 t;
        float rl, rh;
        PK3("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", 0, 1, 0, 1, 1, 1)
        lo += (t[0] != rl);
        hi += (t[1] != rh);
        x[0] += 1.0f;
        x[1] += 0.125f;
    }
    if (lo) atomicAdd(&bad[lane], lo);
    if (hi) atomicAdd(&bad[64 + lane], hi);
}

// The mixed-precision multiply-adds (fp16 halves of 32-bit registers picked by op_sel / op_sel_hi) under the same conditions.
// x holds two fp16 values in one register; the reference converts the half the modifiers name and uses v_fma_f32.
template <int F>
__global__ __launch_bounds__(256) void victim_mix(const f32x2* __restrict__ stat, unsigned* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    const f32x2 y = stat[(blockIdx.x * 256 + threadIdx.x) & 1023];
    const float z = 0.125f + lane;
    unsigned wrong = 0;
    float hl = 0.5f + lane, hh = 1.25f + 0.5f * lane;           // exactly representable in fp16 for every lane and step below
    for (int it = 0; it < iters; ++it) {
#pragma clang fp contract(off)
        const _Float16 l16 = (_Float16)hl, h16 = (_Float16)hh;
        const unsigned x = (unsigned)__builtin_bit_cast(unsigned short, l16) | ((unsigned)__builtin_bit_cast(unsigned short, h16) << 16);
        float t, r;
        if constexpr (F == 0) {
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=&v"(t) : "v"(x), "v"(y[0]), "v"(z));
            r = __builtin_fmaf((float)l16, y[0], z);
        } else if constexpr (F == 1) {
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(t) : "v"(x), "v"(y[0]), "v"(z));
            r = __builtin_fmaf((float)h16, y[0], z);
        } else if constexpr (F == 2) {
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=&v"(t) : "v"(y[1]), "v"(x), "v"(z));
            r = __builtin_fmaf(y[1], (float)h16, z);
        } else {
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=&v"(t) : "v"(x), "v"(x), "v"(z));
            r = __builtin_fmaf((float)h16, (float)l16, z);
        }
        wrong += (t != r);
        hl += 0.5f;
        if (hl > 512.f) hl = 0.5f + lane;
        hh += 0.25f;
        if (hh > 256.f) hh = 1.25f + 0.5f * lane;
    }
    if (wrong) atomicAdd(&bad[lane], wrong);
}

template <int F>
void run_mix(const char* name, hipStream_t sa, hipStream_t sb, const f32x2* stat, unsigned* bad, float* sink, const float* src, int blocks, int iters) {
    for (int beside = 0; beside < 2; ++beside) {
        CHECK(hipMemset(bad, 0, 128 * sizeof(unsigned)));
        CHECK(hipDeviceSynchronize());
        if (beside) hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, sb, sink, src, iters * 4);
        hipLaunchKernelGGL(victim_mix<F>, dim3(blocks), dim3(256), 0, sa, stat, bad, iters);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        unsigned b[128];
        CHECK(hipMemcpy(b, bad, sizeof(b), hipMemcpyDeviceToHost));
        unsigned long long w[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) w[l / 16] += b[l];
        std::printf("%-58s %s: wrong per 16 lanes %llu %llu %llu %llu\n", name, beside ? "beside MFMA" : "alone      ", w[0], w[1], w[2], w[3]);
        std::fflush(stdout);
    }
}

template <int F>
void run_form(const char* name, hipStream_t sa, hipStream_t sb, const f32x2* stat, unsigned* bad, float* sink, const float* src, int blocks, int iters) {
    for (int beside = 0; beside < 2; ++beside) {
        CHECK(hipMemset(bad, 0, 128 * sizeof(unsigned)));
        CHECK(hipDeviceSynchronize());
        if (beside) hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, sb, sink, src, iters * 4);
        hipLaunchKernelGGL(victim_form<F>, dim3(blocks), dim3(256), 0, sa, stat, bad, iters);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        unsigned b[128];
        CHECK(hipMemcpy(b, bad, sizeof(b), hipMemcpyDeviceToHost));
        unsigned long long lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) {
            lo[l / 16] += b[l];
            hi[l / 16] += b[64 + l];
        }
        std::printf("%-58s %s: low wrong per 16 lanes %llu %llu %llu %llu, high %llu %llu %llu %llu\n", name, beside ? "beside MFMA" : "alone      ",
                    lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
        std::fflush(stdout);
    }
}

template <int KIND>
void launch_aggressor(hipStream_t s, float* sink, const float* src, int blocks, int iters) {
    hipLaunchKernelGGL(aggressor<KIND>, dim3(blocks), dim3(256), 0, s, sink, src, iters);
}

int main() {
    const int blocks = 256 * 4, iters = 200000;
    f32x2* stat;
    unsigned* bad;
    float *sink, *src;
    CHECK(hipMalloc(&stat, 1024 * sizeof(f32x2)));
    CHECK(hipMalloc(&bad, 128 * sizeof(unsigned)));
    CHECK(hipMalloc(&sink, 2 * blocks * 256 * sizeof(float)));
    CHECK(hipMalloc(&src, (size_t)(1u << 24) * sizeof(float)));
    CHECK(hipMemset(src, 0, (size_t)(1u << 24) * sizeof(float)));
    std::vector<f32x2> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = f32x2{3.0f + 0.01f * (i % 16), 1.0f + (i % 7)};
    CHECK(hipMemcpy(stat, h.data(), 1024 * sizeof(f32x2), hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    const char* names[] = {"none", "DPP row moves", "MFMA 16x16x32 bf16", "exp2", "LDS traffic", "global streaming", "packed mul with swapped halves"};
    for (int form = 0; form < 2; ++form) {
        for (int k = -1; k < 6; ++k) {
            CHECK(hipMemset(bad, 0, 128 * sizeof(unsigned)));
            CHECK(hipDeviceSynchronize());
            const int ai = iters * 4;
            switch (k) {
                case 0: launch_aggressor<0>(sb, sink, src, blocks, ai); break;
                case 1: launch_aggressor<1>(sb, sink, src, blocks, ai); break;
                case 2: launch_aggressor<2>(sb, sink, src, blocks, ai); break;
                case 3: launch_aggressor<3>(sb, sink, src, blocks, ai / 4); break;
                case 4: launch_aggressor<4>(sb, sink, src, blocks, ai / 16); break;
                case 5: launch_aggressor<5>(sb, sink, src, blocks, ai); break;
                default: break;
            }
            if (form == 0) hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), 0, sa, stat, bad, iters);
            else hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), 0, sa, stat, bad, iters);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned> b(128);
            CHECK(hipMemcpy(b.data(), bad, 128 * sizeof(unsigned), hipMemcpyDeviceToHost));
            unsigned long long lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) {
                lo[l / 16] += b[l];
                hi[l / 16] += b[64 + l];
            }
            std::printf("%s, beside %-32s: low results wrong per 16-lane group %llu %llu %llu %llu, high %llu %llu %llu %llu\n",
                        form == 0 ? "op_sel:[0,1,0]" : "broadcast pair ", names[k + 1], lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
            std::fflush(stdout);
        }
    }
    const int fi = 20000;
    run_form<0>("v_pk_fma_f32 op_sel:[0,1,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<1>("v_pk_fma_f32 op_sel:[1,0,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<2>("v_pk_fma_f32 op_sel:[0,0,1]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<3>("v_pk_fma_f32 op_sel_hi:[0,1,1]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<4>("v_pk_fma_f32 op_sel_hi:[1,0,1]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<5>("v_pk_fma_f32 (no op_sel)", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<6>("v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<7>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<8>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<9>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<10>("v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,1]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_form<11>("v_pk_add_f32 op_sel:[1,0] op_sel_hi:[1,1]", sa, sb, stat, bad, sink, src, blocks, fi);
    {
        CHECK(hipMemset(bad, 0, 128 * sizeof(unsigned)));
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(one_dispatch, dim3(blocks * 2), dim3(256), 0, sa, stat, bad, sink, fi);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        unsigned b[128];
        CHECK(hipMemcpy(b, bad, sizeof(b), hipMemcpyDeviceToHost));
        unsigned long long lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) {
            lo[l / 16] += b[l];
            hi[l / 16] += b[64 + l];
        }
        std::printf("%-58s %s: low wrong per 16 lanes %llu %llu %llu %llu, high %llu %llu %llu %llu\n", "v_pk_fma_f32 op_sel:[0,1,0]",
                    "beside MFMA waves of the SAME dispatch", lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
        std::fflush(stdout);
    }
    run_mix<0>("v_fma_mix_f32 op_sel_hi:[1,0,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_mix<1>("v_fma_mix_f32 op_sel:[1,0,0] op_sel_hi:[1,0,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_mix<2>("v_fma_mix_f32 op_sel:[0,1,0] op_sel_hi:[0,1,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    run_mix<3>("v_fma_mix_f32 op_sel:[1,0,0] op_sel_hi:[1,1,0]", sa, sb, stat, bad, sink, src, blocks, fi);
    return 0;
}
