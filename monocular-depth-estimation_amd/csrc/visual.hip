// Per-pixel post-processing of depth videos behind the reference's driver scripts: the colour-map variants of run1.py,
// run1dist.py and run4.py, and the warp + blend step of the metric driver's temporal refinement (metric_depth/run2.py).
// One pixel per lane; every arithmetic step is rounded on its own (__f*_rn), in float32 or in float64 as numpy would
// evaluate the reference's expression, so that the 8-bit results are the reference's byte for byte.
#include "common.h"

namespace {

inline int grid_for(long n) {
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

constexpr int CMAP_N = 256;      // matplotlib's table size; rows N, N+1, N+2 of a table are its under / over / bad colours

// matplotlib.colors.Colormap.__call__ on a float X in [0, 1]: xa = X * N; xa == N -> N - 1; negative -> under, >= N -> over,
// NaN -> bad; else truncation.  `P` is the precision X arrives in (the multiplication is done in it).
template <typename P>
__device__ __forceinline__ int cmap_index(P x01) {
    if (x01 != x01) return CMAP_N + 2;
    P xa = x01 * (P)CMAP_N;
    if (xa == (P)CMAP_N) xa = (P)(CMAP_N - 1);
    if (xa < (P)0) return CMAP_N;
    if (xa >= (P)CMAP_N) return CMAP_N + 1;
    return (int)xa;
}

// A product or sum that stays one: the build contracts a * b + c into a fused multiply-add wherever it sees one, which numpy's
// element-wise arithmetic never does.  The value is pinned behind an empty asm.
__device__ __forceinline__ float pin(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ double pin(double v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ double div_rn(double a, double b) { return __ddiv_rn(a, b); }

__device__ __forceinline__ void put_rgb(unsigned char* out, long i, const unsigned char* lut, int idx) {
    out[3 * i] = lut[3 * idx];
    out[3 * i + 1] = lut[3 * idx + 1];
    out[3 * i + 2] = lut[3 * idx + 2];
}

// out = lut[cmap_index((clip(x, lo, hi) - sub) / div)], x >= cap -> the table's last colour.  numpy's clip passes NaN through.
template <typename P>
__global__ __launch_bounds__(256) void colormap_kernel(const float* __restrict__ x, const unsigned char* __restrict__ lut,
                                                       unsigned char* __restrict__ out, long n, P sub, P div, P lo, P hi, float cap) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = x[i];
        P v = (P)d;
        if (v == v) v = v < lo ? lo : (v > hi ? hi : v);
        int idx = cmap_index<P>(div_rn(sub_rn(v, sub), div));
        if (d >= cap) idx = CMAP_N - 1;
        put_rgb(out, i, lut, idx);
    }
}

// run1.py:28-60: d <= critical -> lut_a[d / critical]; critical < d <= traffic -> lut_b[(d - critical) / range]; beyond -> grey
// (64, 64, 64); NaN (no zone) -> black.  float32 throughout, as numpy evaluates a float32 map against Python scalars.
__global__ __launch_bounds__(256) void zones_kernel(const float* __restrict__ x, const unsigned char* __restrict__ lut_a,
                                                    const unsigned char* __restrict__ lut_b, unsigned char* __restrict__ out, long n,
                                                    float critical, float traffic, float range) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = x[i];
        if (d <= critical) {
            put_rgb(out, i, lut_a, cmap_index<float>(__fdiv_rn(d, critical)));
        } else if (d <= traffic) {
            put_rgb(out, i, lut_b, cmap_index<float>(__fdiv_rn(__fsub_rn(d, critical), range)));
        } else {
            const unsigned char g = d > traffic ? 64 : 0;
            out[3 * i] = g;
            out[3 * i + 1] = g;
            out[3 * i + 2] = g;
        }
    }
}

// out = clip((x * (x > far ? far_weight : 1) - sub) / div, lo, hi)   (run4.py:29-42: range mask, then contrast stretch)
template <typename P>
__global__ __launch_bounds__(256) void range_kernel(const float* __restrict__ x, float* __restrict__ out, long n, float far,
                                                    float far_weight, P sub, P div, P lo, P hi, int stretch) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float d = x[i];
        d = pin(__fmul_rn(d, d > far ? far_weight : 1.0f));
        if (stretch) {
            P v = div_rn(sub_rn((P)d, sub), div);
            if (v == v) v = v < lo ? lo : (v > hi ? hi : v);
            d = (float)v;
        }
        out[i] = d;
    }
}

// 256-bin histogram of bytes: per-workgroup bins in LDS, then one vector atomic add per bin and workgroup
__global__ __launch_bounds__(256) void histogram_kernel(const unsigned char* __restrict__ x, unsigned int* __restrict__ hist, long n) {
    __shared__ unsigned int bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) atomicAdd(&bins[x[i]], 1u);
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(&hist[threadIdx.x], bins[threadIdx.x]);
}

// g = (uint8)((x - lo) / range * 255) as in the is_depths branch of save_video, then an optional byte-to-byte table
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, long n, float lo,
                                                       float range) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float t = __fmul_rn(__fdiv_rn(__fsub_rn(x[i], lo), range), 255.0f);
        out[i] = (unsigned char)((int)t & 255);
    }
}

// out = lut[cmap_index(remap[g] / 255)] for bytes g (the equalised map of run1dist.py:27-44 through a colour table)
__global__ __launch_bounds__(256) void bytes_colormap_kernel(const unsigned char* __restrict__ g, const unsigned char* __restrict__ remap,
                                                             const unsigned char* __restrict__ lut, unsigned char* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = __fdiv_rn((float)remap[g[i]], 255.0f);
        put_rgb(out, i, lut, cmap_index<float>(v));
    }
}

// OpenCV's remap(INTER_LINEAR, BORDER_REFLECT) of a float32 map, restated: the sampling position is rounded to 1/32 of a pixel
// (sx = round_half_even(32 x)), the four taps are weighted by the products of the 1/32 fractions in float32 and added in the
// order v00 w00 + v01 w01 + v10 w10 + v11 w11; a tap outside the map is mirrored about the edge (..., 1, 0 | 0, 1, ...).
__device__ __forceinline__ int reflect(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p - 1 : 2 * len - p - 1;
    return p;
}

// out = w_cur * cur + w_prev * remap(prev, (x, y) + flow)      (metric_depth/run2.py:28-40, 76)
__global__ __launch_bounds__(256) void warp_blend_kernel(const float* __restrict__ prev, const float* __restrict__ cur,
                                                         const float* __restrict__ flow, float* __restrict__ out, int h, int w,
                                                         float w_cur, float w_prev) {
    const long n = (long)h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        // the maps are (grid + flow) evaluated in float64 and stored as float32 (run2.py:30-35)
        const float mx = (float)((double)x + (double)flow[2 * i]), my = (float)((double)y + (double)flow[2 * i + 1]);
        float warped = 0.f;
        if (mx == mx && my == my && fabsf(mx) < 3.0e7f && fabsf(my) < 3.0e7f) {
            const int sx = (int)rintf(__fmul_rn(mx, 32.0f)), sy = (int)rintf(__fmul_rn(my, 32.0f));
            const int ix = sx >> 5, iy = sy >> 5;
            const float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
            const int x0 = reflect(ix, w), x1 = reflect(ix + 1, w), y0 = reflect(iy, h), y1 = reflect(iy + 1, h);
            const float w00 = pin((1.0f - fy) * (1.0f - fx)), w01 = pin((1.0f - fy) * fx);
            const float w10 = pin(fy * (1.0f - fx)), w11 = pin(fy * fx);
            float acc = pin(prev[(long)y0 * w + x0] * w00);
            acc = pin(acc + pin(prev[(long)y0 * w + x1] * w01));
            acc = pin(acc + pin(prev[(long)y1 * w + x0] * w10));
            acc = pin(acc + pin(prev[(long)y1 * w + x1] * w11));
            warped = acc;
        }
        out[i] = pin(w_cur * cur[i]) + pin(w_prev * warped);
    }
}

}  // namespace

extern "C" int vda_colormap_u8(const float* x, const uint8_t* lut, uint8_t* out, long n, double sub, double div, double clip_lo,
                               double clip_hi, float cap, int f64, void* stream) {
    if (!x || !lut || !out || n <= 0) return VDA_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (f64)
        hipLaunchKernelGGL(colormap_kernel<double>, dim3(grid_for(n)), dim3(256), 0, s, x, lut, out, n, sub, div, clip_lo, clip_hi, cap);
    else
        hipLaunchKernelGGL(colormap_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, x, lut, out, n, (float)sub, (float)div,
                           (float)clip_lo, (float)clip_hi, cap);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_colorize_zones_u8(const float* x, const uint8_t* lut_critical, const uint8_t* lut_traffic, uint8_t* out, long n,
                                     float critical_max, float traffic_max, float traffic_range, void* stream) {
    if (!x || !lut_critical || !lut_traffic || !out || n <= 0) return VDA_EINVAL;
    hipLaunchKernelGGL(zones_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, lut_critical,
                       lut_traffic, out, n, critical_max, traffic_max, traffic_range);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_range_focus_f32(const float* x, float* out, long n, float far, float far_weight, int stretch, double sub,
                                   double div, double clip_lo, double clip_hi, int f64, void* stream) {
    if (!x || !out || n <= 0) return VDA_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (f64)
        hipLaunchKernelGGL(range_kernel<double>, dim3(grid_for(n)), dim3(256), 0, s, x, out, n, far, far_weight, sub, div, clip_lo,
                           clip_hi, stretch);
    else
        hipLaunchKernelGGL(range_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, x, out, n, far, far_weight, (float)sub, (float)div,
                           (float)clip_lo, (float)clip_hi, stretch);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_quantize_u8(const float* x, uint8_t* out, long n, float lo, float range, void* stream) {
    if (!x || !out || n <= 0) return VDA_EINVAL;
    hipLaunchKernelGGL(quantize_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out, n, lo, range);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_histogram_u8(const uint8_t* x, uint32_t* hist, long n, void* stream) {
    if (!x || !hist || n <= 0) return VDA_EINVAL;
    hipLaunchKernelGGL(histogram_kernel, dim3(grid_for(n) > 512 ? 512 : grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       hist, n);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_colormap_bytes_u8(const uint8_t* g, const uint8_t* remap, const uint8_t* lut, uint8_t* out, long n, void* stream) {
    if (!g || !remap || !lut || !out || n <= 0) return VDA_EINVAL;
    hipLaunchKernelGGL(bytes_colormap_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, remap, lut, out,
                       n);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_warp_blend_f32(const float* prev, const float* cur, const float* flow, float* out, int h, int w, float w_cur,
                                  float w_prev, void* stream) {
    if (!prev || !cur || !flow || !out || h <= 0 || w <= 0) return VDA_EINVAL;
    if (out == prev) return VDA_EINVAL;           // every output pixel reads four pixels of prev
    hipLaunchKernelGGL(warp_blend_kernel, dim3(grid_for((long)h * w)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), prev, cur,
                       flow, out, h, w, w_cur, w_prev);
    return hip_rc(hipGetLastError());
}
