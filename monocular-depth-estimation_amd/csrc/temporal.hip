// Temporal self-attention over the T <= 32 frames of one spatial position and head (see vda_temporal_attention).
//
// One wave per (batch, position, head).  Lane (t = lane & 31, half = lane >> 5) owns query frame t and one half of the
// head dimension: it forms partial dot products over its half for all T keys (K/V rows are LDS broadcasts: every lane of
// a half reads the same address), the two halves are combined with one lane ^ 32 exchange per key, the softmax over the
// T scores is lane-local, and the lane then accumulates its half of the output row.  The arithmetic is tiny (T*T*hd per
// wave); the kernel is bound by the strided gather of the T rows, which it reads exactly once.
#include <type_traits>
#include "common.h"

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2(f16x2 a, f16x2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
__device__ __forceinline__ float dot2(bf16x2 a, bf16x2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
// c + w * (one half of the packed fp16 pair h2), the half converted inside the instruction
template <int HI>
__device__ __forceinline__ float mix_fma_f16(unsigned h2, float w, float c) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(w), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(w), "v"(c));
    return r;
}

template <int HD, typename T16>
__global__ __launch_bounds__(64) void temporal_attn_kernel(const T16* __restrict__ qkv, T16* __restrict__ out, int T,
                                                           int hw, int heads, float scale_log2e) {
    constexpr int HH = HD / 2;
    constexpr int V4 = HH / 4;
    typedef typename Vec<T16>::v4 v4_t;
    __shared__ __attribute__((aligned(16))) T16 ks[32 * HD];
    __shared__ __attribute__((aligned(16))) T16 vs[32 * HD];
    const int lane = threadIdx.x, t = lane & 31, half = lane >> 5;
    const int head = blockIdx.x % heads;
    const int p = (blockIdx.x / heads) % hw;
    const int b = blockIdx.x / (heads * hw);
    const int C = heads * HD, ld = 3 * C;
    const size_t fstride = (size_t)hw * ld;                       // distance between consecutive frames of one position
    const T16* base = qkv + ((size_t)b * T * hw + p) * ld + head * HD;

    // K and V rows of the T frames: 16-byte pieces (8 values of a 16-bit type, 4 of fp32), all loads of a lane issued
    // before the first LDS store so that they are in flight together
    constexpr int PV = 16 / sizeof(T16);                            // values per 16-byte piece
    typedef typename std::conditional<sizeof(T16) == 2, typename Vec<T16>::v8, typename Vec<T16>::v4>::type piece_t;
    constexpr int PPR = HD / PV;                                    // pieces per row
    constexpr int NIT = (32 * PPR + 63) / 64;
    piece_t kp[NIT], vp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        const int j = min(idx / PPR, T - 1), c = (idx % PPR) * PV;
        kp[it] = *reinterpret_cast<const piece_t*>(base + j * fstride + C + c);
        vp[it] = *reinterpret_cast<const piece_t*>(base + j * fstride + 2 * C + c);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        const int j = idx / PPR, c = (idx % PPR) * PV;
        if (j < T) {
            *reinterpret_cast<piece_t*>(&ks[j * HD + c]) = kp[it];
            *reinterpret_cast<piece_t*>(&vs[j * HD + c]) = vp[it];
        }
    }
    // 16-bit types: the dot products run on the packed two-element dot instruction (fp32 accumulation, no conversions) on the
    // rows as stored; the scale is applied to the sums.  fp32: plain fused multiply-adds on pre-scaled q.
    constexpr bool PACKED = sizeof(T16) == 2;
    typedef typename std::conditional<PACKED, T16, _Float16>::type P16;
    typedef P16 pair_t __attribute__((ext_vector_type(2)));
    float q[PACKED ? 1 : HH];
    pair_t qp[PACKED ? HH / 2 : 1];
    {
        const T16* qrow = base + (size_t)min(t, T - 1) * fstride + half * HH;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const v4_t v = *reinterpret_cast<const v4_t*>(qrow + 4 * i);
            if constexpr (PACKED) {
                qp[2 * i] = pair_t{(P16)v[0], (P16)v[1]};
                qp[2 * i + 1] = pair_t{(P16)v[2], (P16)v[3]};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) q[4 * i + k] = (float)v[k] * scale_log2e;
            }
        }
    }
    __syncthreads();

    float s[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float acc = 0.f;
        if (j < T) {
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                const v4_t k4 = *reinterpret_cast<const v4_t*>(&ks[j * HD + half * HH + 4 * i]);
                if constexpr (PACKED) {
                    acc = dot2(qp[2 * i], pair_t{(P16)k4[0], (P16)k4[1]}, acc);
                    acc = dot2(qp[2 * i + 1], pair_t{(P16)k4[2], (P16)k4[3]}, acc);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(q[4 * i + k], (float)k4[k], acc);
                }
            }
            acc += __shfl_xor(acc, 32);
            if constexpr (PACKED) acc *= scale_log2e;
        } else {
            acc = -INFINITY;
        }
        s[j] = acc;
    }
    float m = s[0];
#pragma unroll
    for (int j = 1; j < 32; ++j) m = fmaxf(m, s[j]);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        s[j] = __builtin_amdgcn_exp2f(s[j] - m);
        l += s[j];
    }
    float o[HH];
#pragma unroll
    for (int i = 0; i < HH; ++i) o[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j < T) {
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                const v4_t v4 = *reinterpret_cast<const v4_t*>(&vs[j * HD + half * HH + 4 * i]);
                if constexpr (std::is_same<T16, f16>::value) {       // mixed-precision FMA: the half is converted inside the instruction
                    const u32x2 b = __builtin_bit_cast(u32x2, v4);
                    o[4 * i] = mix_fma_f16<0>(b[0], s[j], o[4 * i]);
                    o[4 * i + 1] = mix_fma_f16<1>(b[0], s[j], o[4 * i + 1]);
                    o[4 * i + 2] = mix_fma_f16<0>(b[1], s[j], o[4 * i + 2]);
                    o[4 * i + 3] = mix_fma_f16<1>(b[1], s[j], o[4 * i + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[4 * i + k] = fmaf(s[j], (float)v4[k], o[4 * i + k]);
                }
            }
        }
    }
    if (t < T) {
        const float inv = 1.f / l;
        T16* orow = out + (((size_t)b * T + t) * hw + p) * C + head * HD + half * HH;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const v4_t v = {(T16)(o[4 * i] * inv), (T16)(o[4 * i + 1] * inv), (T16)(o[4 * i + 2] * inv),
                            (T16)(o[4 * i + 3] * inv)};
            *reinterpret_cast<v4_t*>(orow + 4 * i) = v;
        }
    }
}

}  // namespace

extern "C" int vda_temporal_attention(const void* qkv, void* out, int batch, int T, int hw, int heads, int hd, float scale,
                                      int dtype, void* stream) {
    if (!qkv || !out || batch <= 0 || T <= 0 || hw <= 0 || heads <= 0) return VDA_EINVAL;
    if (T > 32) return VDA_ESHAPE;
    if (dtype != VDA_T_BF16 && dtype != VDA_T_F16 && dtype != VDA_T_F32) return VDA_EINVAL;
    const long blocks = (long)batch * hw * heads;
    if (blocks > 0x7fffffffL) return VDA_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
#define TA_LAUNCH(H)                                                                                                     \
    do {                                                                                                                 \
        if (dtype == VDA_T_F16)                                                                                          \
            hipLaunchKernelGGL((temporal_attn_kernel<H, f16>), dim3((unsigned)blocks), dim3(64), 0, s,                    \
                               reinterpret_cast<const f16*>(qkv), reinterpret_cast<f16*>(out), T, hw, heads, sl);        \
        else if (dtype == VDA_T_F32)                                                                                     \
            hipLaunchKernelGGL((temporal_attn_kernel<H, float>), dim3((unsigned)blocks), dim3(64), 0, s,                  \
                               reinterpret_cast<const float*>(qkv), reinterpret_cast<float*>(out), T, hw, heads, sl);    \
        else                                                                                                             \
            hipLaunchKernelGGL((temporal_attn_kernel<H, bf16>), dim3((unsigned)blocks), dim3(64), 0, s,                   \
                               reinterpret_cast<const bf16*>(qkv), reinterpret_cast<bf16*>(out), T, hw, heads, sl);      \
    } while (0)
    switch (hd) {
        case 8: TA_LAUNCH(8); break;
        case 16: TA_LAUNCH(16); break;
        case 24: TA_LAUNCH(24); break;
        case 32: TA_LAUNCH(32); break;
        case 48: TA_LAUNCH(48); break;
        case 64: TA_LAUNCH(64); break;
        case 96: TA_LAUNCH(96); break;
        case 128: TA_LAUNCH(128); break;
        default: return VDA_ESHAPE;
    }
#undef TA_LAUNCH
    return hip_rc(hipGetLastError());
}
