// Bandwidth-bound normalisation kernels: row LayerNorm (fp32 -> bf16/fp16) and NHWC GroupNorm (16-bit in and out).
// Every global access is a 16-byte (LayerNorm) or 8/16-byte (GroupNorm) vector per lane; statistics are two-pass in
// registers (LayerNorm) or fp32 partials combined in double (GroupNorm).
#include "common.h"
#include <mutex>

namespace {

// One wave per row, C <= 64 * 4 * VEC4 floats.  C % 4 == 0.  SPLIT (16-bit T): the output row holds the pieces of
// split_scale * y in the form `split_form` (the fp32 mode's GEMM operand, see vda_split3 and store_pieces).
template <int VEC4, typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, T* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ b, int rows,
                                                        int C, float eps, int rows_per_group, int in_group, int in_skip,
                                                        const float* __restrict__ pe, int pe_hw, int pe_T,
                                                        float split_scale = 0.f, int split_form = 0, int* sat_flag = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int grp = row / rows_per_group;
    const size_t in_row = (size_t)grp * in_group + in_skip + (row - grp * rows_per_group);
    const float* xr = x + in_row * C;
    f32x4 v[VEC4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VEC4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    const float* per = pe ? pe + (size_t)((row / pe_hw) % pe_T) * C : nullptr;
    T* orow = out + (size_t)row * (SPLIT ? pieces_width(split_form, C) : C);
#pragma unroll
    for (int i = 0; i < VEC4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
            f32x4 y = (v[i] - mean) * rstd * ww + bb;
            if (per) y += *reinterpret_cast<const f32x4*>(per + c);
            if constexpr (SPLIT) {
                y *= split_scale;
                note_saturation_as(sat_flag, y, T{});
                typename Vec<T>::v4 hi, lo;
                split4<T>(y, hi, lo);
                store_pieces<T>(orow, c, C, split_form, hi, lo);
            } else {
                const typename Vec<T>::v4 o = {(T)y[0], (T)y[1], (T)y[2], (T)y[3]};
                *reinterpret_cast<typename Vec<T>::v4*>(orow + c) = o;
            }
        }
    }
}

// The rows rounded to T, and (mean, rstd) of the rounded values (two-pass, like layernorm_kernel).  A workgroup owns 16
// consecutive rows (four per wave), i.e. one whole 128-byte line of `stat`: on this part every XCD has its own L2, and two
// workgroups that write different bytes of one line may run on different XCDs.
template <int VEC4, typename T>
__global__ __launch_bounds__(256) void round_rowstats_kernel(const float* __restrict__ x, T* __restrict__ out,
                                                             float* __restrict__ stat, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < 4; ++k) {
        const int row = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + k;          // wave-uniform
        if (row >= rows) return;
        const float* xr = x + (size_t)row * C;
        T* orow = out + (size_t)row * C;
        f32x4 v[VEC4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VEC4; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < C) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(xr + c);
                const typename Vec<T>::v4 o = {(T)f[0], (T)f[1], (T)f[2], (T)f[3]};
                *reinterpret_cast<typename Vec<T>::v4*>(orow + c) = o;
                v[i] = f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VEC4; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + eps);
        if (lane == 0) {
            stat[2 * (size_t)row] = mean;
            stat[2 * (size_t)row + 1] = rstd;
        }
    }
}

// One thread per row: the partial (sum, sum of squares) pairs of the row's 64-column blocks, added in block order.
__global__ __launch_bounds__(256) void rowstats_finish_kernel(const float* __restrict__ part, float* __restrict__ stat, int rows,
                                                              int slots, float eps) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float* p = part + (size_t)row * slots * 2;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < slots; ++i) {
        const float2 v = *reinterpret_cast<const float2*>(p + 2 * i);
        s += (double)v.x;
        q += (double)v.y;
    }
    const double n = 64.0 * slots;
    const double mu = s / n;
    double var = q / n - mu * mu;
    if (var < 0) var = 0;
    float2 o;
    o.x = (float)mu;
    o.y = (float)(1.0 / sqrt(var + (double)eps));
    *reinterpret_cast<float2*>(stat + 2 * (size_t)row) = o;
}

// GroupNorm pass 1: grid (frames, splits), 256 threads.  A block walks every splits-th slab of `ppb` pixels of its frame;
// thread t < ppb * C/8 owns the 8 channels [8 (t % (C/8)) ..] of pixel row t / (C/8) of each slab and keeps PER-CHANNEL
// fp32 sums in registers, so no accumulation order depends on scheduling.  The block then reduces them per group in a
// fixed order (double) and stores one (sum, sum of squares) pair per (frame, split, group): pass 2 adds the splits in
// order.  No atomics: the result is bit-identical from run to run.
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const T* __restrict__ x, double* __restrict__ partial, int hw,
                                                              int C, int groups) {
    __shared__ float ch_s[256][9], ch_q[256][9];      // [thread][channel of the thread], padded against bank conflicts
    const int frame = blockIdx.x;
    const int cv = C / 8;                       // 8-channel vectors per pixel (host guarantees cv <= 256)
    const int cpg = C / groups;                 // channels per group
    const int ppb = 256 / cv;                   // pixels per slab
    const int t = threadIdx.x;
    const int cvec = t % cv, prow = t / cv;
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
    if (prow < ppb) {
        const T* xf = x + (size_t)frame * hw * C + cvec * 8;
        for (int p = blockIdx.y * ppb + prow; p < hw; p += gridDim.y * ppb) {
            const typename Vec<T>::v8 v = *reinterpret_cast<const typename Vec<T>::v8*>(xf + (size_t)p * C);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float f = (float)v[k]; s[k] += f; q[k] += f * f; }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { ch_s[t][k] = s[k]; ch_q[t][k] = q[k]; }
    __syncthreads();
    if (t < groups) {
        double gs = 0.0, gq = 0.0;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c)
            for (int r = 0; r < ppb; ++r) {
                gs += (double)ch_s[r * cv + (c >> 3)][c & 7];
                gq += (double)ch_q[r * cv + (c >> 3)][c & 7];
            }
        double* dst = partial + (((size_t)frame * gridDim.y + blockIdx.y) * groups + t) * 2;
        dst[0] = gs;
        dst[1] = gq;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              const double* __restrict__ partial, int hw, int C, int groups,
                                                              float eps) {
    __shared__ float mean_s[64], rstd_s[64];
    const int frame = blockIdx.x;
    const int cpg = C / groups;
    if (threadIdx.x < groups) {
        double gs = 0.0, gq = 0.0;
        for (int sp = 0; sp < (int)gridDim.y; ++sp) {      // same split count as pass 1, added in order
            const double* src = partial + (((size_t)frame * gridDim.y + sp) * groups + threadIdx.x) * 2;
            gs += src[0];
            gq += src[1];
        }
        const double n = (double)hw * cpg;
        const double mu = gs / n;
        double var = gq / n - mu * mu;
        if (var < 0) var = 0;
        mean_s[threadIdx.x] = (float)mu;
        rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cv = C / 8;
    const size_t total = (size_t)hw * cv;
    const T* xf = x + (size_t)frame * hw * C;
    T* of = out + (size_t)frame * hw * C;
    for (size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.y * blockDim.x) {
        const int c0 = (int)(i % cv) * 8;
        const typename Vec<T>::v8 v = *reinterpret_cast<const typename Vec<T>::v8*>(xf + i * 8);
        typename Vec<T>::v8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = (c0 + k) / cpg;
            o[k] = (T)(((float)v[k] - mean_s[g]) * rstd_s[g] * w[c0 + k] + b[c0 + k]);
        }
        *reinterpret_cast<typename Vec<T>::v8*>(of + i * 8) = o;
    }
}

}  // namespace

extern "C" int vda_layernorm(const float* x, void* out, const float* weight, const float* bias, int rows, int C, float eps,
                             int rows_per_group, int in_group, int in_skip, const float* pe, int pe_hw, int pe_T,
                             int out_type, float split_scale, int split_layout, int* sat_flag, void* stream) {
    if (!x || !out || !weight || !bias || rows <= 0 || C <= 0) return VDA_EINVAL;
    if (C % 4 != 0 || C > 2048 || rows_per_group <= 0 || in_group < rows_per_group + in_skip) return VDA_ESHAPE;
    if (pe && (pe_hw <= 0 || pe_T <= 0)) return VDA_EINVAL;
    if (out_type != VDA_T_BF16 && out_type != VDA_T_F16 && out_type != VDA_T_F32) return VDA_EINVAL;
    const bool split = split_scale != 0.f;
    if (split && (!(split_scale > 0.f) || out_type == VDA_T_F32)) return VDA_EINVAL;
    if (split && C % 64 != 0) return VDA_ESHAPE;
    if (split && split_layout != VDA_PIECES_TRIPLE && split_layout != VDA_PIECES_PAIRS) return VDA_EINVAL;
    const int form = pieces_form(split_layout);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((rows + 3) / 4), block(256);
#define LN_LAUNCH(V)                                                                                                       \
    do {                                                                                                                   \
        if (split && out_type == VDA_T_F16)                                                                                \
            hipLaunchKernelGGL((layernorm_kernel<V, f16, true>), grid, block, 0, s, x, reinterpret_cast<f16*>(out), weight, bias, \
                               rows, C, eps, rows_per_group, in_group, in_skip, pe, pe_hw, pe_T, split_scale, form, sat_flag);   \
        else if (split)                                                                                                    \
            hipLaunchKernelGGL((layernorm_kernel<V, bf16, true>), grid, block, 0, s, x, reinterpret_cast<bf16*>(out), weight, bias, \
                               rows, C, eps, rows_per_group, in_group, in_skip, pe, pe_hw, pe_T, split_scale, form, sat_flag);   \
        else if (out_type == VDA_T_F16)                                                                                         \
            hipLaunchKernelGGL((layernorm_kernel<V, f16>), grid, block, 0, s, x, reinterpret_cast<f16*>(out), weight, bias, \
                               rows, C, eps, rows_per_group, in_group, in_skip, pe, pe_hw, pe_T, 0.f, 0, nullptr);            \
        else if (out_type == VDA_T_F32)                                                                                    \
            hipLaunchKernelGGL((layernorm_kernel<V, float>), grid, block, 0, s, x, reinterpret_cast<float*>(out), weight,  \
                               bias, rows, C, eps, rows_per_group, in_group, in_skip, pe, pe_hw, pe_T, 0.f, 0, nullptr);      \
        else                                                                                                               \
            hipLaunchKernelGGL((layernorm_kernel<V, bf16>), grid, block, 0, s, x, reinterpret_cast<bf16*>(out), weight,    \
                               bias, rows, C, eps, rows_per_group, in_group, in_skip, pe, pe_hw, pe_T, 0.f, 0, nullptr);      \
    } while (0)
    if (C <= 256) LN_LAUNCH(1);
    else if (C <= 512) LN_LAUNCH(2);
    else if (C <= 1024) LN_LAUNCH(4);
    else LN_LAUNCH(8);
#undef LN_LAUNCH
    return hip_rc(hipGetLastError());
}

extern "C" int vda_round_rowstats(const float* x, void* out16, float* stat, int rows, int C, float eps, int out_type, void* stream) {
    if (!x || !out16 || !stat || rows <= 0 || C <= 0) return VDA_EINVAL;
    if (C % 4 != 0 || C > 2048) return VDA_ESHAPE;
    if (out_type != VDA_T_BF16 && out_type != VDA_T_F16) return VDA_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((rows + 15) / 16), block(256);
#define RR_LAUNCH(V)                                                                                                          \
    do {                                                                                                                      \
        if (out_type == VDA_T_F16)                                                                                            \
            hipLaunchKernelGGL((round_rowstats_kernel<V, f16>), grid, block, 0, s, x, reinterpret_cast<f16*>(out16), stat, rows, C, eps); \
        else                                                                                                                  \
            hipLaunchKernelGGL((round_rowstats_kernel<V, bf16>), grid, block, 0, s, x, reinterpret_cast<bf16*>(out16), stat, rows, C, eps); \
    } while (0)
    if (C <= 256) RR_LAUNCH(1);
    else if (C <= 512) RR_LAUNCH(2);
    else if (C <= 1024) RR_LAUNCH(4);
    else RR_LAUNCH(8);
#undef RR_LAUNCH
    return hip_rc(hipGetLastError());
}

extern "C" int vda_rowstats_finish(const float* part, float* stat, int rows, int slots, float eps, void* stream) {
    if (!part || !stat || rows <= 0 || slots <= 0) return VDA_EINVAL;
    if (((uintptr_t)part & 7) != 0 || ((uintptr_t)stat & 7) != 0) return VDA_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(rowstats_finish_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, stat, rows, slots, eps);
    return hip_rc(hipGetLastError());
}

// ---- vda_groupnorm_static: the same two passes with the partial sums in a device array of this library ------------------------
// [frame][split][group][2] doubles at the maximum of every extent: 4 MiB per GPU, held from the library's first use of that
// GPU until the process ends.  The array's name is only a handle on the host: kernels get the address that
// hipGetSymbolAddress returns for the CURRENT device, looked up once per device and kept.
constexpr int GN_STATIC_MAX_GROUPS = 64;
static __device__ __align__(256) double gn_scratch[VDA_GROUPNORM_STATIC_MAX_FRAMES * VDA_GROUPNORM_MAX_SPLITS * GN_STATIC_MAX_GROUPS * 2];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex gn_mutex;
double* gn_scratch_ptr[MAX_DEVICES];       // per device: device address of gn_scratch, nullptr until looked up

int gn_scratch_for_current_device(double** ptr) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_rc(e);
    if (dev < 0 || dev >= MAX_DEVICES) return VDA_ESHAPE;
    std::lock_guard<std::mutex> lock(gn_mutex);
    if (!gn_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(gn_scratch));
        if (e != hipSuccess) return hip_rc(e);
        if (!p) return VDA_EINVAL;
        gn_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *ptr = gn_scratch_ptr[dev];
    return VDA_OK;
}
}  // namespace

extern "C" int vda_groupnorm_static(const void* x, void* out, const float* weight, const float* bias, int frames, int hw,
                                    int C, int groups, float eps, int dtype, void* stream) {
    if (frames > VDA_GROUPNORM_STATIC_MAX_FRAMES) return VDA_ESHAPE;       // the scratch array holds 64 frames' partial sums
    double* work = nullptr;
    const int rc = gn_scratch_for_current_device(&work);
    if (rc != VDA_OK) return rc;
    // every partial sum pass 2 reads is written by pass 1 of the same call: nothing has to be cleared between calls
    return vda_groupnorm(x, out, weight, bias, work, frames, hw, C, groups, eps, dtype, stream);
}

extern "C" int vda_groupnorm(const void* x, void* out, const float* weight, const float* bias, void* work, int frames,
                             int hw, int C, int groups, float eps, int dtype, void* stream) {
    if (!x || !out || !weight || !bias || !work || frames <= 0 || hw <= 0) return VDA_EINVAL;
    if (groups <= 0 || groups > 64 || C % groups != 0 || C % 8 != 0 || C > 2048) return VDA_ESHAPE;
    if (((uintptr_t)work & 7) != 0) return VDA_ESHAPE;
    if (dtype != VDA_T_BF16 && dtype != VDA_T_F16 && dtype != VDA_T_F32) return VDA_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* sums = reinterpret_cast<double*>(work);       // [frame][split][group][2], every element written by pass 1
    const size_t total = (size_t)hw * (C / 8);
    int splits = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (splits > VDA_GROUPNORM_MAX_SPLITS) splits = VDA_GROUPNORM_MAX_SPLITS;
    if (splits < 1) splits = 1;
    dim3 grid(frames, splits);
#define GN_LAUNCH(TT)                                                                                                     \
    do {                                                                                                                  \
        hipLaunchKernelGGL(groupnorm_stats_kernel<TT>, grid, dim3(256), 0, s, reinterpret_cast<const TT*>(x), sums, hw, C, groups); \
        hipLaunchKernelGGL(groupnorm_apply_kernel<TT>, grid, dim3(256), 0, s, reinterpret_cast<const TT*>(x),             \
                           reinterpret_cast<TT*>(out), weight, bias, sums, hw, C, groups, eps);                          \
    } while (0)
    if (dtype == VDA_T_F16) GN_LAUNCH(f16);
    else if (dtype == VDA_T_F32) GN_LAUNCH(float);
    else GN_LAUNCH(bf16);
#undef GN_LAUNCH
    return hip_rc(hipGetLastError());
}
