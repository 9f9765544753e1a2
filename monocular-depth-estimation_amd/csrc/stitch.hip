// Window stitching on the device: least-squares sums, 2x2 solve, affine + clamp + cross-fade (see include/vda_hip.h).
#include "common.h"

namespace {

constexpr int LSQ_BLOCKS = VDA_LSQ_BLOCKS;

// Pass 1: a fixed grid of LSQ_BLOCKS workgroups, each writes its four partial sums (double) to its own slot; pass 2 adds the
// slots in order.  No atomics: the sums, and with them scale and shift, are bit-identical from run to run.
__global__ __launch_bounds__(256) void lsq_partial_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                          double* __restrict__ part, long n) {
    __shared__ double red[4][4];
    double a00 = 0, a01 = 0, b0 = 0, b1 = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double pv = p[i], tv = t[i];
        a00 += pv * pv;
        a01 += pv;
        b0 += pv * tv;
        b1 += tv;
    }
    double v[4] = {a00, a01, b0, b1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 4; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        part[blockIdx.x * 4 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

__global__ void lsq_reduce_kernel(double* __restrict__ sums, const double* __restrict__ part, long n) {
    const int k = threadIdx.x;                       // 0..3: sum p*p, sum p, sum p*t, sum t
    if (k < 4) {
        double tot = 0;
        for (int b = 0; b < LSQ_BLOCKS; ++b) tot += part[b * 4 + k];
        // sums layout: {sum p*p, sum p, count, sum p*t, sum t}
        const int slot = k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 3 : 4;
        sums[slot] += tot;
    }
    if (k == 4) sums[2] += (double)n;
}

// The solve repeats the reference's float32 arithmetic on float32-rounded sums (utils/util.py:46-60).
__global__ void solve_kernel(const double* __restrict__ sums, float* __restrict__ ss) {
    const float a00 = (float)sums[0], a01 = (float)sums[1], a11 = (float)sums[2], b0 = (float)sums[3], b1 = (float)sums[4];
    const float det = a00 * a11 - a01 * a01;
    float x0 = 1.f, x1 = 0.f;
    if (det != 0.f) {
        x0 = (a11 * b0 - a01 * b1) / det;
        x1 = (-a01 * b0 + a00 * b1) / det;
    }
    ss[0] = x0;
    ss[1] = x1;
}

__global__ __launch_bounds__(256) void affine_blend_kernel(const float* __restrict__ x, const float* __restrict__ pre,
                                                           float* __restrict__ out, const float* __restrict__ ss, float w_pre,
                                                           float w_post, long n) {
    const float sc = ss[0], sh = ss[1];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i] * sc + sh;
        v = v < 0.f ? 0.f : v;
        v *= w_post;
        if (pre) v += w_pre * pre[i];
        out[i] = v;
    }
}

// Metric post-processing: one depth value per frame at that frame's reference pixel, and a per-frame scale.
__global__ void sample_points_kernel(const float* __restrict__ depth, const int* __restrict__ xy, float* __restrict__ out, int n,
                                     long hw, int w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = depth[(size_t)i * hw + (size_t)xy[2 * i + 1] * w + xy[2 * i]];
}

// grid (frames, slices): 16 bytes per lane; hw % 4 == 0 is not required (scalar tail)
__global__ __launch_bounds__(256) void scale_frames_kernel(const float* __restrict__ depth, const float* __restrict__ scales,
                                                           float* __restrict__ out, long hw) {
    const float s = scales[blockIdx.x];
    const float* src = depth + (size_t)blockIdx.x * hw;
    float* dst = out + (size_t)blockIdx.x * hw;
    const bool vec = (hw & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const long stride = (long)gridDim.y * blockDim.x;
    if (vec) {
        for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < hw / 4; i += stride)
            reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i] * s;
    } else {
        for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < hw; i += stride) dst[i] = src[i] * s;
    }
}

// Depth visualisation: global minimum / maximum (two passes, fixed grid; min and max do not depend on the order), then
// 8-bit quantisation with the reference's float32 arithmetic and a colour table lookup.
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long n) {
    __shared__ float red[2][4];
    float lo = INFINITY, hi = -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
        part[2 * blockIdx.x + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    }
}

__global__ void minmax_final_kernel(float* __restrict__ mm, const float* __restrict__ part) {
    float lo = INFINITY, hi = -INFINITY;
    for (int b = threadIdx.x; b < LSQ_BLOCKS; b += 64) { lo = fminf(lo, part[2 * b]); hi = fmaxf(hi, part[2 * b + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if (threadIdx.x == 0) { mm[0] = lo; mm[1] = hi; }
}

// one pixel per thread: index = (uint8)((d - lo) / (hi - lo) * 255), each step rounded to float32 like numpy's
__global__ __launch_bounds__(256) void colorize_kernel(const float* __restrict__ depth, const float* __restrict__ mm,
                                                       const unsigned char* __restrict__ lut, unsigned char* __restrict__ out, long n) {
    const float lo = mm[0], range = mm[1] - mm[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float t = __fmul_rn(__fdiv_rn(__fsub_rn(depth[i], lo), range), 255.0f);
        const int g = (int)t & 255;             // numpy's float -> uint8 cast truncates; the value is within 0..255
        if (lut) {
            out[3 * i] = lut[3 * g];
            out[3 * i + 1] = lut[3 * g + 1];
            out[3 * i + 2] = lut[3 * g + 2];
        } else {
            out[i] = (unsigned char)g;
        }
    }
}

inline int grid_for(long n) {
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int vda_scale_shift_sums(const float* pred, const float* target, double* sums, long n, void* stream) {
    if (!pred || !target || !sums || n <= 0) return VDA_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* part = sums + 5;                         // LSQ_BLOCKS x 4 partial sums behind the five totals
    hipLaunchKernelGGL(lsq_partial_kernel, dim3(LSQ_BLOCKS), dim3(256), 0, s, pred, target, part, n);
    hipLaunchKernelGGL(lsq_reduce_kernel, dim3(1), dim3(64), 0, s, sums, part, n);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_solve_scale_shift(const double* sums, float* ss, void* stream) {
    if (!sums || !ss) return VDA_EINVAL;
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), sums, ss);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_affine_blend(const float* x, const float* pre, float* out, const float* ss, float w_pre, float w_post,
                                long n, void* stream) {
    if (!x || !out || !ss || n <= 0) return VDA_EINVAL;
    if (w_pre != 0.f && !pre) return VDA_EINVAL;
    hipLaunchKernelGGL(affine_blend_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       w_pre != 0.f ? pre : nullptr, out, ss, w_pre, w_post, n);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_sample_points(const float* depth, const int* xy, float* out, int frames, int h, int w, void* stream) {
    if (!depth || !xy || !out || frames <= 0 || h <= 0 || w <= 0) return VDA_EINVAL;
    hipLaunchKernelGGL(sample_points_kernel, dim3((frames + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), depth,
                       xy, out, frames, (long)h * w, w);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_scale_frames(const float* depth, const float* scales, float* out, int frames, long hw, void* stream) {
    if (!depth || !scales || !out || frames <= 0 || hw <= 0) return VDA_EINVAL;
    long slices = (hw / 4 + 255) / 256;
    if (slices > 64) slices = 64;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(scale_frames_kernel, dim3(frames, (unsigned)slices), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       depth, scales, out, hw);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_minmax_f32(const float* x, float* minmax, long n, void* stream) {
    if (!x || !minmax || n <= 0) return VDA_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(LSQ_BLOCKS), dim3(256), 0, s, x, minmax + 2, n);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, s, minmax, minmax + 2);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_colorize_u8(const float* depth, const float* minmax, const uint8_t* lut, uint8_t* out, long n, void* stream) {
    if (!depth || !minmax || !out || n <= 0) return VDA_EINVAL;
    hipLaunchKernelGGL(colorize_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), depth, minmax, lut,
                       out, n);
    return hip_rc(hipGetLastError());
}

extern "C" const char* vda_version(void) { return "vda_hip 0.2 gfx950"; }
