// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution with a fused epilogue (see vda_gemm in include/vda_hip.h).
//
// Structure: one workgroup = 4 waves computes a BM x BN tile of the output, K is walked in steps of 64.
// Both operand tiles are K-contiguous ([rows][64] bf16, 128-byte rows) and are copied global -> LDS by 16-byte
// global_load_lds instructions, double buffered: the copy of K-step t+1 is in flight while step t is multiplied.
// The LDS image is lane-linear (a requirement of global_load_lds); the bank-conflict XOR is applied to the per-lane
// SOURCE address when staging and to the address when reading fragments (same involution, see tile_off()).
// The MFMA is issued as mfma(W fragment, A fragment), i.e. it produces the transposed tile, so each lane ends up with
// 4 CONSECUTIVE output columns of one row: epilogue loads/stores are 8- or 16-byte vectors.
#include "common.h"

namespace {

constexpr int BK = 64;

struct GemmTiles {
    int tiles_m, tiles_n;
};

typedef short short8 __attribute__((ext_vector_type(8)));

template <typename V8>
__device__ __forceinline__ V8 relu8(V8 v) {
    // the float sign bit is the int16 sign bit, so max(int16, 0) is ReLU for bf16 and fp16 (and maps -0.0 to +0.0)
    short8 s = __builtin_bit_cast(short8, v);
    short8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    s = __builtin_elementwise_max(s, z);
    return __builtin_bit_cast(V8, s);
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f32x4 load4(const void* p, int type, size_t idx) {
    if (type == VDA_T_F32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + idx);
    if (type == VDA_T_BF16) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + idx);
        return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
    }
    const f16x4 r = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(p) + idx);
    return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
}

__device__ __forceinline__ void store4(void* out, int out_type, size_t idx, f32x4 v) {
    if (out_type == VDA_T_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + idx) = v;
    } else if (out_type == VDA_T_BF16) {
        bf16x4 b = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(out) + idx) = b;
    } else {
        f16x4 b = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(out) + idx) = b;
    }
}

template <int BM, int BN, int WM, int WN, int AMODE, typename T>
__global__ __launch_bounds__(256) void gemm_kernel(const VdaGemmArgs p, const GemmTiles gt) {
    constexpr int TM = BM / WM / 16;      // 16x16 MFMA tiles per wave along M
    constexpr int TN = BN / WN / 16;
    constexpr int NA = BM / 32;           // 16-byte pieces each thread stages for A per K-step
    constexpr int NB = BN / 32;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    typedef typename Vec<T>::v8 frag_t;
    // the staging below only moves 16-bit patterns, so its pointer arithmetic uses bf16* for either operand type
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- workgroup -> tile: spread over the 8 XCDs so that each XCD works on a contiguous run of tiles, and walk
    //      tiles in bands of 8 row-tiles so that concurrently resident tiles share A and W panels in that XCD's L2.
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int tm, tn;
    {
        constexpr int GM = 8;
        const int per_band = GM * gt.tiles_n;
        const int band = tile / per_band;
        const int first = band * GM;
        const int rows = min(GM, gt.tiles_m - first);
        const int in_band = tile - band * per_band;
        tm = first + in_band % rows;
        tn = in_band / rows;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging sources -------------------------------------------------------------------------
    const int srow = tid >> 3;                               // row inside a 32-row staging slab
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);        // source 16-byte chunk for this thread's LDS slot
    const bf16* a_src[NA];
    int a_iy[NA], a_ix[NA];
    const bf16* w_src[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = min(m0 + i * 32 + srow, p.M - 1);
        if (AMODE == VDA_A_LINEAR) {
            a_src[i] = reinterpret_cast<const bf16*>(p.a) + (size_t)m * p.lda + schunk * 8;
        } else {
            const int hw = p.out_h * p.out_w;
            const int img = m / hw, rem = m - img * hw;
            const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
            a_iy[i] = oy * p.stride - 1;
            a_ix[i] = ox * p.stride - 1;
            a_src[i] = reinterpret_cast<const bf16*>(p.a) + (size_t)img * p.in_h * p.in_w * p.cin + schunk * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = min(n0 + i * 32 + srow, p.N - 1);
        w_src[i] = reinterpret_cast<const bf16*>(p.w) + (size_t)n * p.K + schunk * 8;
    }
    const bf16* zsrc = reinterpret_cast<const bf16*>(p.zeros) + schunk * 8;
    const int cpt = (AMODE == VDA_A_CONV3) ? p.cin / BK : 1;  // K-steps per filter tap

    auto stage = [&](int kt, int buf) {
        char* abuf = smem + buf * (A_BYTES + B_BYTES);
        char* bbuf = abuf + A_BYTES;
        if (AMODE == VDA_A_LINEAR) {
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16(a_src[i] + kt * BK, abuf + (i * 256 + wave * 64) * 16);
        } else {
            const int tap = kt / cpt, cc = kt - tap * cpt;
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
                const bf16* s = ok ? a_src[i] + ((size_t)iy * p.in_w + ix) * p.cin + cc * BK : zsrc;
                glds16(s, abuf + (i * 256 + wave * 64) * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(w_src[i] + kt * BK, bbuf + (i * 256 + wave * 64) * 16);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    const int frow = lane & 15, fq = lane >> 4;
    stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const char* abuf = smem + cur * (A_BYTES + B_BYTES);
        const char* bbuf = abuf + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / WM) + i * 16 + frow;
                af[i] = *reinterpret_cast<const frag_t*>(abuf + tile_off(row, ks * 4 + fq));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / WN) + j * 16 + frow;
                wf[j] = *reinterpret_cast<const frag_t*>(bbuf + tile_off(row, ks * 4 + fq));
            }
            if (p.relu_in) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = relu8(af[i]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        }
        __syncthreads();   // also drains the in-flight copy of step kt+1 (vmcnt(0) precedes the barrier)
    }

    // ---- epilogue: lane holds rows m = .. + (lane & 15), columns n = .. + (lane >> 4) * 4 + {0,1,2,3} ------------
    const int mrow0 = m0 + wm * (BM / WM) + frow;
    const int ncol0 = n0 + wn * (BN / WN) + fq * 4;

    if (p.epi == VDA_EPI_DOT) {
        float part[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) part[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ncol0 + j * 16;
            if (n < p.N) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
                const f32x4 dw = *reinterpret_cast<const f32x4*>(p.dot_w + n);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[i] += fmaxf(acc[i][j][r] + b[r], 0.f) * dw[r];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v = part[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int m = mrow0 + i * 16;
            if (fq == 0 && m < p.M) reinterpret_cast<float*>(p.out)[m] = fmaxf(v + p.dot_b, 0.f);
        }
        return;
    }

    if (p.epi == VDA_EPI_GEGLU) {
        const int half0 = (n0 + wn * (BN / WN)) / 2 + fq * 4;
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
            const int nv = ncol0 + j * 16;            // value columns; gate columns are 16 further
            if (nv + 16 >= p.N) continue;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + nv);
            const f32x4 bg = *reinterpret_cast<const f32x4*>(p.bias + nv + 16);
            const int oc = half0 + (j / 2) * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow0 + i * 16;
                if (m >= p.M) continue;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (acc[i][j][r] + bv[r]) * gelu_erf(acc[i][j + 1][r] + bg[r]);
                store4(p.out, p.out_type, (size_t)m * p.ldc + oc, v);
            }
        }
        return;
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 16;
        if (n >= p.N) continue;
        int bias_n = n;
        size_t col_off = n;
        if (p.epi == VDA_EPI_CONVT) {
            const int tap = n / p.ct_c, co = n - tap * p.ct_c;
            const int ky = tap / p.ct_k, kx = tap - ky * p.ct_k;
            bias_n = co;
            col_off = ((size_t)ky * (p.ct_w * p.ct_k) + kx) * p.ldc + co;
        }
        f32x4 b = {0.f, 0.f, 0.f, 0.f}, g = {1.f, 1.f, 1.f, 1.f};
        if (p.bias) b = *reinterpret_cast<const f32x4*>(p.bias + bias_n);
        if (p.gamma) g = *reinterpret_cast<const f32x4*>(p.gamma + n);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mrow0 + i * 16;
            if (m >= p.M) continue;
            f32x4 v = acc[i][j] + b;
            if (p.act == VDA_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            } else if (p.act == VDA_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v *= g;
            if (p.resid_type != VDA_T_NONE) v += load4(p.resid, p.resid_type, (size_t)m * p.ldr + n);
            if (p.resid2) v += load4(p.resid2, p.ab_type, (size_t)m * p.ldr2 + n);
            size_t idx;
            if (p.epi == VDA_EPI_CONVT) {
                const int iy = m / p.ct_w, x = m - iy * p.ct_w;     // iy = img * ct_h + y
                idx = ((size_t)iy * p.ct_k * (p.ct_w * p.ct_k) + (size_t)x * p.ct_k) * p.ldc + col_off;
            } else {
                idx = (size_t)m * p.ldc + col_off;
            }
            store4(p.out, p.out_type, idx, v);
        }
    }
}

template <int BM, int BN, int WM, int WN, typename T>
int launch_t(const VdaGemmArgs& a, hipStream_t s) {
    GemmTiles gt{(a.M + BM - 1) / BM, (a.N + BN - 1) / BN};
    const int lds = 2 * (BM + BN) * 128;
    dim3 grid(gt.tiles_m * gt.tiles_n);
    if (a.a_mode == VDA_A_CONV3) {
        auto k = gemm_kernel<BM, BN, WM, WN, VDA_A_CONV3, T>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, gt);
    } else {
        auto k = gemm_kernel<BM, BN, WM, WN, VDA_A_LINEAR, T>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, gt);
    }
    return hip_rc(hipGetLastError());
}

template <int BM, int BN, int WM, int WN>
int launch(const VdaGemmArgs& a, hipStream_t s) {
    return a.ab_type == VDA_T_F16 ? launch_t<BM, BN, WM, WN, f16>(a, s) : launch_t<BM, BN, WM, WN, bf16>(a, s);
}

}  // namespace

extern "C" int vda_gemm(const VdaGemmArgs* args, void* stream) {
    if (!args || !args->a || !args->w || !args->out) return VDA_EINVAL;
    const VdaGemmArgs& a = *args;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return VDA_EINVAL;
    if (a.K % BK != 0 || a.N % 4 != 0 || a.ldc % 4 != 0) return VDA_ESHAPE;
    if (a.resid_type != VDA_T_NONE && (!a.resid || a.ldr % 4 != 0)) return VDA_EINVAL;
    if (a.resid2 && a.ldr2 % 4 != 0) return VDA_ESHAPE;
    if (a.a_mode == VDA_A_CONV3) {
        if (!a.zeros || a.cin <= 0 || a.cin % BK != 0 || a.K != 9 * a.cin) return VDA_ESHAPE;
        if (a.stride < 1 || a.out_h <= 0 || a.out_w <= 0 || a.M % (a.out_h * a.out_w) != 0) return VDA_ESHAPE;
        if (a.out_h != (a.in_h - 1) / a.stride + 1 || a.out_w != (a.in_w - 1) / a.stride + 1) return VDA_ESHAPE;
    } else if (a.a_mode == VDA_A_LINEAR) {
        if (a.lda < a.K || a.lda % 8 != 0) return VDA_ESHAPE;
    } else {
        return VDA_EINVAL;
    }
    if (a.epi == VDA_EPI_GEGLU) {
        if (a.N % 32 != 0 || !a.bias || a.resid_type != VDA_T_NONE || a.resid2) return VDA_ESHAPE;
    } else if (a.epi == VDA_EPI_CONVT) {
        if (a.ct_k < 1 || a.ct_c % 4 != 0 || a.N != a.ct_k * a.ct_k * a.ct_c || a.M % (a.ct_h * a.ct_w) != 0) return VDA_ESHAPE;
        if (a.resid_type != VDA_T_NONE || a.resid2 || a.gamma) return VDA_ESHAPE;
    } else if (a.epi == VDA_EPI_DOT) {
        if (a.N > 32 || !a.bias || !a.dot_w || a.out_type != VDA_T_F32) return VDA_ESHAPE;
    } else if (a.epi != VDA_EPI_PLAIN) {
        return VDA_EINVAL;
    }
    if (a.out_type != VDA_T_F32 && a.out_type != VDA_T_BF16 && a.out_type != VDA_T_F16) return VDA_EINVAL;
    if (a.ab_type != VDA_T_BF16 && a.ab_type != VDA_T_F16) return VDA_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a.N <= 32) return launch<128, 32, 4, 1>(a, s);
    if (a.N <= 64) return launch<128, 64, 2, 2>(a, s);
    return launch<128, 128, 2, 2>(a, s);
}
