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

typedef float f32x16 __attribute__((ext_vector_type(16)));
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

// NW waves per workgroup take NW consecutive heads of one position: their rows are neighbours in memory (HD values each), and
// workgroups are dealt to the XCDs round-robin, so one head per workgroup would make up to eight L2s fetch the same 128-byte
// lines (measured at HD = 32: twice the algorithmic bytes fetched).
template <int HD, typename T16, int NW>
__global__ __launch_bounds__(64 * NW) void temporal_attn_kernel(const T16* __restrict__ qkv, T16* __restrict__ out, int T,
                                                                int hw, int heads, float scale_log2e) {
    constexpr int HH = HD / 2;
    constexpr int V4 = HH / 4;
    typedef typename Vec<T16>::v4 v4_t;
    __shared__ __attribute__((aligned(16))) T16 ks_all[NW * 32 * HD];
    __shared__ __attribute__((aligned(16))) T16 vs_all[NW * 32 * HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = lane & 31, half = lane >> 5;
    T16* ks = ks_all + wave * 32 * HD;
    T16* vs = vs_all + wave * 32 * HD;
    const int groups = heads / NW;
    const int head = (blockIdx.x % groups) * NW + wave;
    const int p = (blockIdx.x / groups) % hw;
    const int b = blockIdx.x / (groups * hw);
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

// The same attention on the matrix cores, for 16-bit types and head dims that are multiples of 32: per wave
//   S^T[key][query] = K Q^T   (HD/16 mfma_32x32x16; lane (r, h) supplies 16 bytes of row r of K and of Q per k-step,
//                              straight from global memory),
//   softmax over the keys of a query: the query sits on the lane, its 32 keys in 16 registers of lanes r and r + 32,
//   O^T[dim][query] = V^T P^T (2 HD/32 products; the probabilities, rounded to the operand type, are the B operand as they
//                              stand in the score registers, which fixes the key order of a k-step to 4h + {0..3, 8..11}
//                              (+ 16 per step); the V^T operand comes from a row-major LDS copy of V by transposed reads
//                              that gather the keys in that order).
// A 32 x 32 x HD problem costs 16 matrix instructions at HD = 128 where the vector form above issues some four thousand.
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int HD, typename T16, int NW>
__global__ __launch_bounds__(64 * NW) void temporal_attn_mfma_kernel(const T16* __restrict__ qkv, T16* __restrict__ out, int T, int hw,
                                                                     int heads, float scale_log2e) {
    static_assert(HD % 32 == 0, "head dim in 32-wide blocks");
    typedef typename Vec<T16>::v8 v8_t;
    typedef typename Vec<T16>::v4 v4_t;
    constexpr int KS = HD / 16, DB = HD / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int groups = heads / NW;
    const int head = (blockIdx.x % groups) * NW + wave;
    const int p = (blockIdx.x / groups) % hw;
    const int b = blockIdx.x / (groups * hw);
    const int C = heads * HD, ld = 3 * C;
    const size_t fstride = (size_t)hw * ld;
    const T16* base = qkv + ((size_t)b * T * hw + p) * ld + head * HD;

    // all loads first: Q and K fragments (row min(r, T-1)), then the V rows in 16-byte pieces (rows past T-1 repeat row
    // T-1: finite values that meet zero probabilities), which go through LDS as [key][64 values] tiles to be read transposed
    __shared__ __attribute__((aligned(16))) char vsm_all[NW * ((HD + 63) / 64) * 4096];
    char* vsm = vsm_all + wave * ((HD + 63) / 64) * 4096;
    const T16* qrow = base + (size_t)min(r, T - 1) * fstride + 8 * h;
    v8_t qf[KS], kf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *reinterpret_cast<const v8_t*>(qrow + 16 * ks);
        kf[ks] = *reinterpret_cast<const v8_t*>(qrow + C + 16 * ks);
    }
    constexpr int PPR = HD / 8, NIT = 32 * PPR / 64;         // pieces per row; pieces per lane
    v8_t vp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane, j = idx / PPR, c8 = idx - j * PPR;
        vp[it] = *reinterpret_cast<const v8_t*>(base + (size_t)min(j, T - 1) * fstride + 2 * C + 8 * c8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane, j = idx / PPR, col = 8 * (idx - j * PPR);
        *reinterpret_cast<v8_t*>(vsm + (col >> 6) * 4096 + v_off(j, col & 63)) = vp[it];
    }
    __syncthreads();
    // V^T fragments (transposed reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its block)
    const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    v8_t vf[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const char* tile = vsm + ((32 * db) >> 6) * 4096;
            const int k0 = 16 * s2 + 4 * h + tq, c = ((32 * db) & 63) + tcol;
            const v4_t lo = read_tr<T16>(tile + v_off(k0, c));
            const v4_t hi = read_tr<T16>(tile + v_off(k0 + 8, c));
            vf[db][s2] = v8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }

    f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = mfma32(kf[ks], qf[ks], s);
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
        s[i] = key < T ? s[i] * scale_log2e : -INFINITY;
        m = fmaxf(m, s[i]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));              // key 0 < T belongs to the h = 0 half: never -inf
    float l = 0.f;
    v8_t pf[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(s[i] - m);
        l += pv;
        pf[i >> 3][i & 7] = (T16)pv;
    }
    l += __shfl_xor(l, 32);

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) o[db] = mfma32(vf[db][s2], pf[s2], o[db]);
    }
    if (r < T) {
        const float inv = 1.f / l;
        T16* orow = out + (((size_t)b * T + r) * hw + p) * C + head * HD + 4 * h;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4_t v = {(T16)(o[db][4 * g] * inv), (T16)(o[db][4 * g + 1] * inv), (T16)(o[db][4 * g + 2] * inv),
                                (T16)(o[db][4 * g + 3] * inv)};
                *reinterpret_cast<v4_t*>(orow + 32 * db + 8 * g) = v;
            }
    }
}

}  // namespace

extern "C" int vda_temporal_attention(const void* qkv, void* out, int batch, int T, int hw, int heads, int hd, float scale,
                                      int dtype, void* stream) {
    if (!qkv || !out || batch <= 0 || T <= 0 || hw <= 0 || heads <= 0) return VDA_EINVAL;
    if (T > 32) return VDA_ESHAPE;
    if (dtype != VDA_T_BF16 && dtype != VDA_T_F16 && dtype != VDA_T_F32) return VDA_EINVAL;
    if ((long)batch * hw * heads > 0x7fffffffL) return VDA_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
    // NW heads per workgroup: as many as cover 512 bytes of a position's row (at most 8), if they divide the head count
#define TA_RUN(KERNEL, H, TY, NW)                                                                                              \
    hipLaunchKernelGGL((KERNEL<H, TY, NW>), dim3((unsigned)((long)batch * hw * (heads / (NW)))), dim3(64 * (NW)), 0, s,       \
                       reinterpret_cast<const TY*>(qkv), reinterpret_cast<TY*>(out), T, hw, heads, sl)
#define TA_TYPED(KERNEL, H, TY)                                                                                                \
    do {                                                                                                                       \
        constexpr int WANT = (H) * sizeof(TY) >= 512 ? 1 : 512 / ((H) * sizeof(TY)) >= 8 ? 8 : 512 / ((H) * sizeof(TY)) >= 4 ? 4 : 2; \
        if (heads % WANT == 0) TA_RUN(KERNEL, H, TY, WANT); else TA_RUN(KERNEL, H, TY, 1);                                     \
    } while (0)
#define TA_LAUNCH(H)                                                                                                           \
    do {                                                                                                                       \
        if (dtype == VDA_T_F16) TA_TYPED(temporal_attn_kernel, H, f16);                                                        \
        else if (dtype == VDA_T_F32) TA_TYPED(temporal_attn_kernel, H, float);                                                 \
        else TA_TYPED(temporal_attn_kernel, H, bf16);                                                                          \
    } while (0)
#define TA_LAUNCH_MFMA(H)                                                                                                      \
    do {                                                                                                                       \
        if (dtype == VDA_T_F16) TA_TYPED(temporal_attn_mfma_kernel, H, f16); else TA_TYPED(temporal_attn_mfma_kernel, H, bf16); \
    } while (0)
    if (dtype != VDA_T_F32 && hd % 32 == 0) {       // 16-bit types, whole 32-wide blocks: the matrix-core form
        switch (hd) {
            case 32: TA_LAUNCH_MFMA(32); break;
            case 64: TA_LAUNCH_MFMA(64); break;
            case 96: TA_LAUNCH_MFMA(96); break;
            case 128: TA_LAUNCH_MFMA(128); break;
            default: return VDA_ESHAPE;
        }
        return hip_rc(hipGetLastError());
    }
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
#undef TA_LAUNCH_MFMA
#undef TA_TYPED
#undef TA_RUN
    return hip_rc(hipGetLastError());
}
