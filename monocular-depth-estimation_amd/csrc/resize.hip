// Bandwidth-bound data-movement kernels: bilinear resampling (align_corners=True), patch im2col, token assembly.
#include "common.h"

namespace {

// Source index arithmetic follows ATen's upsample_bilinear2d with align_corners=True in fp32:
// scale = (in-1)/(out-1) (0 when out == 1), src = scale * dst, i0 = (int)src, i1 = i0 + (i0 < in-1), w1 = src - i0.
struct Lerp {
    int i0, i1;
    float w0, w1;
};
// (The build contracts a * b + c wherever it finds one, and finds different ones in different kernels.  The rounded product
// `src` is therefore pinned behind an empty asm, and blend_px below spells its fused multiply-adds out: every kernel that
// resamples then gives the same bits.)
__device__ __forceinline__ Lerp lerp_of(int dst, int in, int out) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float src = scale * (float)dst;
    asm volatile("" : "+v"(src));
    Lerp l;
    l.i0 = (int)src;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

// The four-tap blend w0y (w0x a + w1x b) + w1y (w0x c + w1x d) in a fixed form: three products, three fused multiply-adds.
__device__ __forceinline__ float blend_px(const Lerp& ly, const Lerp& lx, float a, float b, float c, float d) {
    const float top = __builtin_fmaf(lx.w1, b, lx.w0 * a);
    const float bottom = __builtin_fmaf(lx.w1, d, lx.w0 * c);
    return __builtin_fmaf(ly.w1, bottom, ly.w0 * top);
}

// One workgroup per output row (frame, oy): the vertical taps are computed once per workgroup and all per-element index
// arithmetic is 32-bit (a flat 64-bit index split by division costs more instructions than the 4 loads and the store).
template <typename T>
__global__ __launch_bounds__(256) void bilinear_nhwc_kernel(const T* __restrict__ x, T* __restrict__ out, int in_h,
                                                            int in_w, int out_h, int out_w, int C) {
    typedef typename Vec<T>::v8 v8;
    const unsigned cv = C / 8;
    const int item = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);     // neighbouring rows share their source rows
    const int f = item / out_h, oy = item - f * out_h;
    const Lerp ly = lerp_of(oy, in_h, out_h);
    const T* row0 = x + ((size_t)f * in_h + ly.i0) * in_w * C;
    const T* row1 = x + ((size_t)f * in_h + ly.i1) * in_w * C;
    T* orow = out + (size_t)item * out_w * C;
    const unsigned n = out_w * cv;
    for (unsigned i = threadIdx.x; i < n; i += 256) {
        const unsigned ox = i / cv, c = (i - ox * cv) * 8;
        const Lerp lx = lerp_of((int)ox, in_w, out_w);
        const v8 a = *reinterpret_cast<const v8*>(row0 + lx.i0 * C + c);
        const v8 b = *reinterpret_cast<const v8*>(row0 + lx.i1 * C + c);
        const v8 cc = *reinterpret_cast<const v8*>(row1 + lx.i0 * C + c);
        const v8 d = *reinterpret_cast<const v8*>(row1 + lx.i1 * C + c);
        v8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[k] = (T)blend_px(ly, lx, (float)a[k], (float)b[k], (float)cc[k], (float)d[k]);
        }
        *reinterpret_cast<v8*>(orow + (size_t)i * 8) = o;
    }
}

// The same resampling of an fp32 map, written as the 16-bit operand pieces of scale * v (fp32 mode: the input of a convolution
// that runs on the resampled map, which then never exists in fp32).  Four channels per lane and step.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_nhwc_pieces_kernel(const float* __restrict__ x, T* __restrict__ out, int in_h, int in_w,
                                                                   int out_h, int out_w, int C, float scale, int form, int* sat_flag) {
    const unsigned cv = C / 4;
    const int item = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);     // neighbouring rows share their source rows
    const int f = item / out_h, oy = item - f * out_h;
    const Lerp ly = lerp_of(oy, in_h, out_h);
    const float* row0 = x + ((size_t)f * in_h + ly.i0) * in_w * C;
    const float* row1 = x + ((size_t)f * in_h + ly.i1) * in_w * C;
    const int width = pieces_width(form, C);
    T* orow = out + (size_t)item * out_w * width;
    const unsigned n = out_w * cv;
    for (unsigned i = threadIdx.x; i < n; i += 256) {
        const unsigned ox = i / cv, c = (i - ox * cv) * 4;
        const Lerp lx = lerp_of((int)ox, in_w, out_w);
        const f32x4 a = *reinterpret_cast<const f32x4*>(row0 + lx.i0 * C + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(row0 + lx.i1 * C + c);
        const f32x4 cc = *reinterpret_cast<const f32x4*>(row1 + lx.i0 * C + c);
        const f32x4 d = *reinterpret_cast<const f32x4*>(row1 + lx.i1 * C + c);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = blend_px(ly, lx, a[k], b[k], cc[k], d[k]) * scale;
        note_saturation_as(sat_flag, v, T{});
        typename Vec<T>::v4 hi, lo;
        split4<T>(v, hi, lo);
        store_pieces<T>(orow + (size_t)ox * width, (int)c, C, form, hi, lo);
    }
}

__global__ __launch_bounds__(256) void bilinear_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int in_h,
                                                           int in_w, int out_h, int out_w, int relu) {
    const int item = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);     // neighbouring rows share their source rows
    const int f = item / out_h, oy = item - f * out_h;
    const Lerp ly = lerp_of(oy, in_h, out_h);
    const float* row0 = x + ((size_t)f * in_h + ly.i0) * in_w;
    const float* row1 = x + ((size_t)f * in_h + ly.i1) * in_w;
    float* orow = out + (size_t)item * out_w;
    for (int ox = threadIdx.x; ox < out_w; ox += 256) {
        const Lerp lx = lerp_of(ox, in_w, out_w);
        float v = blend_px(ly, lx, row0[lx.i0], row0[lx.i1], row1[lx.i0], row1[lx.i1]);
        if (relu) v = fmaxf(v, 0.f);
        orow[ox] = v;
    }
}

// One thread per 8 output columns of an im2col row.  k = c*196 + ky*14 + kx.
template <typename T>
__global__ __launch_bounds__(256) void im2col_patch14_kernel(const float* __restrict__ x, T* __restrict__ out, int H, int W,
                                                             int hp, int wp, int ldk, size_t total) {
    const int kv = ldk / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k0 = (int)(i % kv) * 8;
        size_t row = i / kv;
        const int px = (int)(row % wp);
        row /= wp;
        const int py = (int)(row % hp);
        const int f = (int)(row / hp);
        typename Vec<T>::v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            float v = 0.f;
            if (k < 588) {
                const int c = k / 196, rem = k - c * 196;
                const int ky = rem / 14, kx = rem - ky * 14;
                v = x[(((size_t)f * 3 + c) * H + py * 14 + ky) * W + px * 14 + kx];
            }
            o[j] = (T)v;
        }
        *reinterpret_cast<typename Vec<T>::v8*>(out + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, float* __restrict__ x, int hw,
                                                              int D, size_t total) {
    const int dv = D / 4, n_tok = hw + 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % dv) * 4;
        const size_t row = i / dv;
        const int t = (int)(row % n_tok);
        const size_t f = row / n_tok;
        f32x4 v = t == 0 ? *reinterpret_cast<const f32x4*>(cls + d)
                         : *reinterpret_cast<const f32x4*>(tok + (f * hw + (t - 1)) * D + d);
        v += *reinterpret_cast<const f32x4*>(pos + (size_t)t * D + d);
        *reinterpret_cast<f32x4*>(x + i * 4) = v;
    }
}


// Frame preprocessing: uint8 RGB HWC -> (x/255) -> bicubic resize -> ImageNet normalisation -> fp32 CHW.
// The resize follows OpenCV's INTER_CUBIC on float32 images: src = (dst + 0.5) * (in/out) - 0.5, Keys kernel with
// A = -0.75 (4th coefficient = 1 - the other three), source indices clamped to the border, horizontal pass then
// vertical pass in float32.  With in == out it is the identity.
__device__ __forceinline__ void cv_cubic(int dst, int in, int out, int& i0, float w[4]) {
    const double scale = (double)in / (double)out;
    float f = (float)((dst + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    const float A = -0.75f;
    w[0] = ((A * (f + 1.f) - 5.f * A) * (f + 1.f) + 8.f * A) * (f + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * f - (A + 3.f)) * f * f + 1.f;
    w[2] = ((A + 2.f) * (1.f - f) - (A + 3.f)) * (1.f - f) * (1.f - f) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
    i0 = s - 1;
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int h0,
                                                         int w0, int H, int W, size_t total) {
    const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t r = i / W;
        const int y = (int)(r % H);
        const size_t f = r / H;
        int ix0, iy0;
        float wx[4], wy[4];
        cv_cubic(x, w0, W, ix0, wx);
        cv_cubic(y, h0, H, iy0, wy);
        const uint8_t* img = src + f * (size_t)h0 * w0 * 3;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int yy = min(max(iy0 + ky, 0), h0 - 1);
            float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int xx = min(max(ix0 + kx, 0), w0 - 1);
                const uint8_t* px = img + ((size_t)yy * w0 + xx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) row[c] += wx[kx] * ((float)px[c] / 255.0f);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += wy[ky] * row[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dst[((f * 3 + c) * H + y) * (size_t)W + x] = (float)(((double)acc[c] - mean[c]) / stdv[c]);
    }
}

// fp32 rows -> 16-bit rows of three segments [hi | lo | hi] (each Cp wide, zero beyond C) of scale * x: the A operand of a
// GEMM whose weight rows are packed [hi | hi | lo], so that one MFMA pass over K' = 3 Cp accumulates hi*hi + lo*hi + hi*lo.
// An activation is applied first (it cannot be applied to hi and lo separately); for VDA_SPLIT_GEGLU the input row holds
// 2C values in blocks [16 values | 16 gates] and output column c is value * GELU(gate).
// One workgroup per 4 rows; 32-bit index arithmetic inside a row.
template <typename T>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, T* __restrict__ out, long rows, int C,
                                                     int ldx, int Cp, int act, float scale, int form, int* sat_flag) {
    const long row0 = (long)blockIdx.x * 4;
    const int cv = Cp / 4;
    for (int i = threadIdx.x; i < 4 * cv; i += 256) {
        const int r = i / cv, c = (i - r * cv) * 4;
        const long row = row0 + r;
        if (row >= rows) break;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {                                                              // C % 4 == 0
            if (act == VDA_SPLIT_GEGLU) {
                const float* s = x + row * ldx + (c >> 4) * 32 + (c & 15);
                v = *reinterpret_cast<const f32x4*>(s);
                const f32x4 g = *reinterpret_cast<const f32x4*>(s + 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= gelu_erf(g[k]);
            } else {
                v = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
            }
        }
        typename Vec<T>::v4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f = v[k];
            if (act == VDA_SPLIT_RELU) f = fmaxf(f, 0.f);
            if (act == VDA_SPLIT_GELU) f = gelu_erf(f);
            v[k] = f * scale;
        }
        note_saturation_as(sat_flag, v, T{});
        split4<T>(v, hi, lo);
        store_pieces<T>(out + row * (long)pieces_width(form, Cp), c, Cp, form, hi, lo);
    }
}

inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int vda_bilinear_nhwc(const void* x, void* out, int frames, int in_h, int in_w, int out_h, int out_w, int C,
                                 int dtype, void* stream) {
    if (!x || !out || frames <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || C <= 0) return VDA_EINVAL;
    if (C % 8 != 0) return VDA_ESHAPE;
    if (dtype != VDA_T_BF16 && dtype != VDA_T_F16 && dtype != VDA_T_F32) return VDA_EINVAL;
    if ((long)frames * out_h > 0x7fffffffL || (long)out_w * (C / 8) > 0x7fffffffL || (long)in_w * C > 0x7fffffffL) return VDA_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(frames * out_h);
    if (dtype == VDA_T_F16)
        hipLaunchKernelGGL(bilinear_nhwc_kernel<f16>, grid, dim3(256), 0, s, reinterpret_cast<const f16*>(x),
                           reinterpret_cast<f16*>(out), in_h, in_w, out_h, out_w, C);
    else if (dtype == VDA_T_F32)
        hipLaunchKernelGGL(bilinear_nhwc_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float*>(x),
                           reinterpret_cast<float*>(out), in_h, in_w, out_h, out_w, C);
    else
        hipLaunchKernelGGL(bilinear_nhwc_kernel<bf16>, grid, dim3(256), 0, s, reinterpret_cast<const bf16*>(x),
                           reinterpret_cast<bf16*>(out), in_h, in_w, out_h, out_w, C);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_bilinear_nhwc_pieces(const float* x, void* out, int frames, int in_h, int in_w, int out_h, int out_w, int C,
                                        int out_type, float scale, int layout, int* sat_flag, void* stream) {
    if (!x || !out || frames <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || C <= 0) return VDA_EINVAL;
    if (C % 64 != 0) return VDA_ESHAPE;
    if ((out_type != VDA_T_BF16 && out_type != VDA_T_F16) || !(scale > 0.f)) return VDA_EINVAL;
    if (layout != VDA_PIECES_TRIPLE && layout != VDA_PIECES_PAIRS) return VDA_EINVAL;
    if ((long)frames * out_h > 0x7fffffffL || (long)out_w * (C / 4) > 0x7fffffffL || (long)in_w * C > 0x7fffffffL) return VDA_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(frames * out_h);
    if (out_type == VDA_T_F16)
        hipLaunchKernelGGL(bilinear_nhwc_pieces_kernel<f16>, grid, dim3(256), 0, s, x, reinterpret_cast<f16*>(out), in_h, in_w, out_h,
                           out_w, C, scale, pieces_form(layout), sat_flag);
    else
        hipLaunchKernelGGL(bilinear_nhwc_pieces_kernel<bf16>, grid, dim3(256), 0, s, x, reinterpret_cast<bf16*>(out), in_h, in_w, out_h,
                           out_w, C, scale, pieces_form(layout), sat_flag);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_bilinear_f32(const float* x, float* out, int frames, int in_h, int in_w, int out_h, int out_w, int relu,
                                void* stream) {
    if (!x || !out || frames <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return VDA_EINVAL;
    if ((long)frames * out_h > 0x7fffffffL) return VDA_ESHAPE;
    hipLaunchKernelGGL(bilinear_f32_kernel, dim3(frames * out_h), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out,
                       in_h, in_w, out_h, out_w, relu);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_im2col_patch14(const float* x, void* out, int frames, int H, int W, int ldk, int out_type, void* stream) {
    if (!x || !out || frames <= 0 || H <= 0 || W <= 0) return VDA_EINVAL;
    if (out_type != VDA_T_BF16 && out_type != VDA_T_F16 && out_type != VDA_T_F32) return VDA_EINVAL;
    if (H % 14 != 0 || W % 14 != 0 || ldk % 64 != 0 || ldk < 588) return VDA_ESHAPE;
    const int hp = H / 14, wp = W / 14;
    const size_t total = (size_t)frames * hp * wp * (ldk / 8);
    if (out_type == VDA_T_F32)
        hipLaunchKernelGGL(im2col_patch14_kernel<float>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           x, reinterpret_cast<float*>(out), H, W, hp, wp, ldk, total);
    else if (out_type == VDA_T_F16)
        hipLaunchKernelGGL(im2col_patch14_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           x, reinterpret_cast<f16*>(out), H, W, hp, wp, ldk, total);
    else
        hipLaunchKernelGGL(im2col_patch14_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           x, reinterpret_cast<bf16*>(out), H, W, hp, wp, ldk, total);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_split3(const float* x, void* out, long rows, int C, int ldx, int Cp, int act, int out_type, float scale,
                          int layout, int* sat_flag, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0) return VDA_EINVAL;
    if (act < VDA_SPLIT_NONE || act > VDA_SPLIT_GEGLU) return VDA_EINVAL;
    if ((out_type != VDA_T_BF16 && out_type != VDA_T_F16) || !(scale > 0.f)) return VDA_EINVAL;
    if (layout != VDA_PIECES_TRIPLE && layout != VDA_PIECES_PAIRS) return VDA_EINVAL;
    const int form = pieces_form(layout);
    const int in_cols = act == VDA_SPLIT_GEGLU ? 2 * C : C;
    if (C % 4 != 0 || Cp % 64 != 0 || Cp < C || ldx < in_cols || ldx % 4 != 0 || (rows + 3) / 4 > 0x7fffffffL) return VDA_ESHAPE;
    if (act == VDA_SPLIT_GEGLU && C % 16 != 0) return VDA_ESHAPE;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (out_type == VDA_T_F16)
        hipLaunchKernelGGL(split3_kernel<f16>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                           reinterpret_cast<f16*>(out), rows, C, ldx, Cp, act, scale, form, sat_flag);
    else
        hipLaunchKernelGGL(split3_kernel<bf16>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                           reinterpret_cast<bf16*>(out), rows, C, ldx, Cp, act, scale, form, sat_flag);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_assemble_tokens(const float* tok, const float* cls, const float* pos, float* x, int frames, int hw, int D,
                                   void* stream) {
    if (!tok || !cls || !pos || !x || frames <= 0 || hw <= 0 || D <= 0) return VDA_EINVAL;
    if (D % 4 != 0) return VDA_ESHAPE;
    const size_t total = (size_t)frames * (hw + 1) * (D / 4);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tok,
                       cls, pos, x, hw, D, total);
    return hip_rc(hipGetLastError());
}

extern "C" int vda_preprocess_u8(const uint8_t* frames_u8, float* out, int frames, int h0, int w0, int H, int W, void* stream) {
    if (!frames_u8 || !out || frames <= 0 || h0 <= 0 || w0 <= 0 || H <= 0 || W <= 0) return VDA_EINVAL;
    const size_t total = (size_t)frames * H * W;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), frames_u8,
                       out, h0, w0, H, W, total);
    return hip_rc(hipGetLastError());
}
