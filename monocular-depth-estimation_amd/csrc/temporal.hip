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
    float q[HH];
    {
        const T16* qrow = base + (size_t)min(t, T - 1) * fstride + half * HH;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const v4_t v = *reinterpret_cast<const v4_t*>(qrow + 4 * i);
#pragma unroll
            for (int k = 0; k < 4; ++k) q[4 * i + k] = (float)v[k] * scale_log2e;
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
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = fmaf(q[4 * i + k], (float)k4[k], acc);
            }
            acc += __shfl_xor(acc, 32);
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
#pragma unroll
                for (int k = 0; k < 4; ++k) o[4 * i + k] = fmaf(s[j], (float)v4[k], o[4 * i + k]);
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
