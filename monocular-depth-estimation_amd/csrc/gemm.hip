// 16-bit MFMA GEMM / implicit-GEMM 3x3 convolution with a fused epilogue (see vda_gemm in include/vda_hip.h):
// argument checks, tile selection, and the plain double-buffered kernel used for narrow or small problems.
//
// Plain kernel: one workgroup computes a BM x BN tile of the output, K is walked in steps of 64.  Both operand tiles are
// K-contiguous ([rows][64] 16-bit values, 128-byte rows) and are copied global -> LDS by 16-byte global_load_lds
// instructions, double buffered: the copy of K-step t+1 is in flight while step t is multiplied.  The LDS image is
// lane-linear (a requirement of global_load_lds); the bank-conflict XOR is applied to the per-lane SOURCE address when
// staging and to the address when reading fragments (same involution, see tile_off()).  The MFMA is issued as
// mfma(W fragment, A fragment), i.e. it produces the transposed tile, so each lane ends up with 4 CONSECUTIVE output
// columns of one row: epilogue loads/stores are 8- or 16-byte vectors.
#include "gemm_common.h"

namespace vda {

// PAIRS (VdaGemmArgs.ab_pairs): a K step of 64 stored columns is [hi 32 | lo 32] of 32 operand columns, for A and W alike;
// the step multiplies W_lo A_hi + W_hi A_lo + W_hi A_hi from one copy and one fragment read of each piece.
template <int BM, int BN, int WM, int WN, int AMODE, int SPEC, typename T, bool PAIRS = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const VdaGemmArgs p, const GemmTiles gt) {
    constexpr int TM = BM / WM / 16;      // 16x16 MFMA tiles per wave along M
    constexpr int TN = BN / WN / 16;
    constexpr int NT = WM * WN * 64;      // threads per workgroup
    constexpr int SR = NT / 8;            // rows covered by one staging instruction of the whole workgroup
    constexpr int NA = BM / SR;           // 16-byte pieces each thread stages for A per K-step
    constexpr int NB = BN / SR;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    typedef typename Vec<T>::v8 frag_t;
    // the staging below only moves 16-bit patterns, so its pointer arithmetic uses bf16* for either operand type
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tm, tn;
    tile_of_workgroup(gt, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging sources -------------------------------------------------------------------------
    const int srow = tid >> 3;                               // row inside an SR-row staging slab
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);        // source 16-byte chunk for this thread's LDS slot
    const bf16* a_src[NA];
    int a_iy[NA], a_ix[NA];
    const bf16* w_src[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = min(m0 + i * SR + srow, p.M - 1);
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
        const int n = min(n0 + i * SR + srow, p.N - 1);
        w_src[i] = reinterpret_cast<const bf16*>(p.w) + (size_t)n * p.K + schunk * 8;
    }
    const bf16* zsrc = reinterpret_cast<const bf16*>(p.zeros) + schunk * 8;
    const int cpt = (AMODE == VDA_A_CONV3) ? p.cin / BK : 1;  // K-steps per filter tap

    auto stage = [&](int kt, int buf) {
        char* abuf = smem + buf * (A_BYTES + B_BYTES);
        char* bbuf = abuf + A_BYTES;
        if (AMODE == VDA_A_LINEAR) {
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16(a_src[i] + kt * BK, abuf + (i * NT + wave * 64) * 16);
        } else {
            const int tap = kt / cpt, cc = kt - tap * cpt;
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
                const bf16* s = ok ? a_src[i] + ((size_t)iy * p.in_w + ix) * p.cin + cc * BK : zsrc;
                glds16(s, abuf + (i * NT + wave * 64) * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(w_src[i] + kt * BK, bbuf + (i * NT + wave * 64) * 16);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    const int frow = lane & 15, fq = lane >> 4;
    // bias + fp32 residual into an fp32 output (the encoder's proj / fc2 at widths the 256x256 kernel does not take): fetch
    // the residual now; it lands during the K loop
    constexpr bool PRE = (SPEC == EPI_SPEC_F32_RES32 || SPEC == EPI_SPEC_F32_RES32_EMIT) && TM * TN <= 16;
    f32x4 pre[PRE ? TM : 1][TN];
    if constexpr (PRE) residual_prefetch<TM, TN>(p, pre, m0 + wm * (BM / WM) + frow, n0 + wn * (BN / WN) + fq * 4);
    // folded LayerNorm, consumer side: (mean, rstd) of this lane's rows, in flight during the K loop
    constexpr bool LN = SPEC == EPI_SPEC_OWN_LN || SPEC == EPI_SPEC_OWN_GELU_LN;
    f32x2 rs[LN ? TM : 1];
    if constexpr (LN) rowstat_prefetch<TM>(p, rs, m0 + wm * (BM / WM) + frow);
    stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const char* abuf = smem + cur * (A_BYTES + B_BYTES);
        const char* bbuf = abuf + A_BYTES;
        if constexpr (PAIRS) {
            frag_t af[2][TM], wf[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const frag_t*>(abuf + tile_off(wm * (BM / WM) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const frag_t*>(bbuf + tile_off(wn * (BN / WN) + j * 16 + frow, ks * 4 + fq));
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)          // the two small terms first, then hi * hi
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma16(wf[term == 0][j], af[term == 1][i], acc[i][j]);
        } else
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

    gemm_epilogue_spec<SPEC, T, TM, TN, LN>(p, acc, m0 + wm * (BM / WM) + frow, n0 + wn * (BN / WN) + fq * 4, fq, smem + wave * STAGE_BYTES,
                                            PRE ? pre : nullptr, rs);
}

namespace {

template <int BM, int BN, int WM, int WN, int AMODE, int SPEC, typename T, bool PAIRS = false>
int launch_k(const VdaGemmArgs& a, hipStream_t s) {
    GemmTiles gt{(a.M + BM - 1) / BM, (a.N + BN - 1) / BN};
    const int lds = 2 * (BM + BN) * 128;
    dim3 grid(gt.tiles_m * gt.tiles_n);
    auto k = gemm_kernel<BM, BN, WM, WN, AMODE, SPEC, T, PAIRS>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), lds, s, a, gt);
    return hip_rc(hipGetLastError());
}

// specialised epilogues (EPI_SPEC_*) for the forms the model uses; anything else takes the general one
template <int BM, int BN, int WM, int WN, typename T>
int launch_t(const VdaGemmArgs& a, hipStream_t s) {
    const int spec = epi_spec_of(a);
    if (a.ab_pairs) {         // the fp32 mode's forms only
#define VDA_CASE_P(MODE, SPEC) case SPEC: return launch_k<BM, BN, WM, WN, MODE, SPEC, T, true>(a, s);
        if (a.a_mode == VDA_A_CONV3) return launch_k<BM, BN, WM, WN, VDA_A_CONV3, EPI_SPEC_GENERAL, T, true>(a, s);
        switch (spec) {
            VDA_CASE_P(VDA_A_LINEAR, EPI_SPEC_F32) VDA_CASE_P(VDA_A_LINEAR, EPI_SPEC_F32_RES32)
            VDA_CASE_P(VDA_A_LINEAR, EPI_SPEC_SPLIT) VDA_CASE_P(VDA_A_LINEAR, EPI_SPEC_SPLIT_GELU)
            default: return launch_k<BM, BN, WM, WN, VDA_A_LINEAR, EPI_SPEC_GENERAL, T, true>(a, s);
        }
#undef VDA_CASE_P
    }
#define VDA_CASE(MODE, SPEC) case SPEC: return launch_k<BM, BN, WM, WN, MODE, SPEC, T>(a, s);
    if (a.a_mode == VDA_A_CONV3) {
        switch (spec) {
            VDA_CASE(VDA_A_CONV3, EPI_SPEC_OWN) VDA_CASE(VDA_A_CONV3, EPI_SPEC_OWN_RES) VDA_CASE(VDA_A_CONV3, EPI_SPEC_OWN_RELU)
            VDA_CASE(VDA_A_CONV3, EPI_SPEC_OWN_RES2)
            default: return launch_k<BM, BN, WM, WN, VDA_A_CONV3, EPI_SPEC_GENERAL, T>(a, s);
        }
    }
    switch (spec) {
        VDA_CASE(VDA_A_LINEAR, EPI_SPEC_OWN) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_OWN_GELU) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_F32_RES32)
        VDA_CASE(VDA_A_LINEAR, EPI_SPEC_OWN_RES) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_OWN_RES32) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_GEGLU)
        VDA_CASE(VDA_A_LINEAR, EPI_SPEC_F32) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_CONVT)
        VDA_CASE(VDA_A_LINEAR, EPI_SPEC_OWN_LN) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_OWN_GELU_LN) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_F32_RES32_EMIT)
        VDA_CASE(VDA_A_LINEAR, EPI_SPEC_SPLIT) VDA_CASE(VDA_A_LINEAR, EPI_SPEC_SPLIT_GELU)
        default: return launch_k<BM, BN, WM, WN, VDA_A_LINEAR, EPI_SPEC_GENERAL, T>(a, s);
    }
#undef VDA_CASE
}

template <int BM, int BN, int WM, int WN>
int launch(const VdaGemmArgs& a, hipStream_t s) {
    return a.ab_type == VDA_T_F16 ? launch_t<BM, BN, WM, WN, f16>(a, s) : launch_t<BM, BN, WM, WN, bf16>(a, s);
}

}  // namespace
}  // namespace vda

namespace {

// argument checks of vda_gemm: VDA_OK or the error code
int check_args(const VdaGemmArgs* args) {
    using namespace vda;
    if (!args || !args->a || !args->w || !args->out) return VDA_EINVAL;
    const VdaGemmArgs& a = *args;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return VDA_EINVAL;
    if (a.K % BK != 0 || a.N % 4 != 0 || a.ldc % 4 != 0) return VDA_ESHAPE;
    if (a.resid_type != VDA_T_NONE && (!a.resid || a.ldr % 4 != 0)) return VDA_EINVAL;
    if (a.resid2 && a.ldr2 % 4 != 0) return VDA_ESHAPE;
    if (a.resid2 && a.resid_type == VDA_T_NONE) return VDA_EINVAL;
    if (a.a_mode == VDA_A_CONV3) {
        if (!a.zeros || a.cin <= 0 || a.cin % BK != 0 || a.K != 9 * a.cin) return VDA_ESHAPE;
        if (a.stride < 1 || a.out_h <= 0 || a.out_w <= 0 || a.M % (a.out_h * a.out_w) != 0) return VDA_ESHAPE;
        if (a.out_h != (a.in_h - 1) / a.stride + 1 || a.out_w != (a.in_w - 1) / a.stride + 1) return VDA_ESHAPE;
        if (a.up_h < 0 || a.up_w < 0 || (a.up_h > 0) != (a.up_w > 0)) return VDA_EINVAL;
        if (a.cin_used < 0 || a.cin_used % 32 != 0 || a.cin_used > a.cin) return VDA_ESHAPE;
    } else if (a.a_mode == VDA_A_LINEAR) {
        if (a.lda < a.K || a.lda % 8 != 0 || a.up_h != 0 || a.up_w != 0 || a.cin_used != 0) return VDA_ESHAPE;
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
    // the epilogue's GELU is accurate to 16-bit storage only; fp32 callers apply the exact one in vda_split3 or ask for a
    // split output, whose GELU is the exact form
    if ((a.act == VDA_ACT_GELU || a.epi == VDA_EPI_GEGLU) && a.out_type == VDA_T_F32 && !a.split_out) return VDA_ESHAPE;
    if (a.split_out) {
        if (a.split_out < 2 || a.split_out > 4) return VDA_EINVAL;
        if (!(a.split_scale > 0.f)) return VDA_EINVAL;
        if (a.a_mode != VDA_A_LINEAR || a.epi != VDA_EPI_PLAIN || a.gamma || a.resid_type != VDA_T_NONE || a.row_stat || a.out16) return VDA_ESHAPE;
        if (a.act == VDA_ACT_RELU) return VDA_ESHAPE;
        if (a.N % 64 != 0 || a.ldc % 8 != 0 || a.ldc < (a.split_out == 3 ? 3 : 2) * a.N || ((uintptr_t)a.out & 15) != 0) return VDA_ESHAPE;
    }
    if (a.ab_pairs != 0 && a.ab_pairs != 1) return VDA_EINVAL;
    if (a.ab_pairs && (a.relu_in || a.row_stat || a.out16 || a.cin_used || a.up_h)) return VDA_ESHAPE;
    if (a.ab_type != VDA_T_BF16 && a.ab_type != VDA_T_F16) return VDA_EINVAL;
    if (a.tile_cfg < 0 || a.tile_cfg > 3) return VDA_EINVAL;
    // folded LayerNorm: consumer side (row_stat / w_colsum) and producer side (out16 / row_part)
    if ((a.row_stat != nullptr) != (a.w_colsum != nullptr)) return VDA_EINVAL;
    if (a.row_stat) {
        if (a.a_mode != VDA_A_LINEAR || a.epi != VDA_EPI_PLAIN || a.resid_type != VDA_T_NONE || a.gamma) return VDA_ESHAPE;
        if (((uintptr_t)a.row_stat & 7) != 0 || ((uintptr_t)a.w_colsum & 15) != 0) return VDA_ESHAPE;
    }
    if ((a.out16 != nullptr) != (a.row_part != nullptr)) return VDA_EINVAL;
    if (a.out16) {
        if (a.a_mode != VDA_A_LINEAR || a.epi != VDA_EPI_PLAIN || a.out_type != VDA_T_F32 || a.gamma) return VDA_ESHAPE;
        if (a.N % 64 != 0 || a.N < 128 || a.ld16 < a.N || a.ld16 % 4 != 0) return VDA_ESHAPE;
        if (((uintptr_t)a.out & 15) != 0 || ((uintptr_t)a.out16 & 7) != 0 || ((uintptr_t)a.row_part & 7) != 0) return VDA_ESHAPE;
        if (a.tile_cfg == 3) return VDA_ESHAPE;
        // the kernels address both arrays with 32-bit byte offsets
        if ((uint64_t)a.M * a.ld16 * 2 >= (1ull << 32) || (uint64_t)a.M * (a.N / 64) * 8 >= (1ull << 32)) return VDA_ESHAPE;
    }
    return VDA_OK;
}

// which kernel vda_gemm runs for checked arguments: a VDA_KERNEL_* code, or a negative error code
int choose_kernel(const VdaGemmArgs& a) {
    using namespace vda;
    // 3x3 convolutions: the direct kernel copies each input pixel once per patch and channel slice instead of once per tap.
    // Narrow outputs (N <= 128) always gain (1.3x at N = 128, 1.8x at N = 32); wide ones only while the 8 x 32 patches
    // cover the image well (148 x 148: 90 % covered, +10 %; 74 x 74: 71 %, -4 % against the 256x256 tile).
    // fused upsampling and partly used pixels exist in the direct kernel only
    if (a.a_mode == VDA_A_CONV3 && (a.up_h > 0 || (a.cin_used != 0 && a.cin_used != a.cin)))
        return ((a.tile_cfg == 0 || a.tile_cfg == 3) && conv_direct_supports(a)) ? VDA_KERNEL_CONV_DIRECT : VDA_ESHAPE;
    if ((a.tile_cfg == 0 || a.tile_cfg == 3) && conv_direct_supports(a)) {
        const long ty = (a.out_h + 7) / 8, tx = (a.out_w + 31) / 32;
        const long patches = (long)(a.M / (a.out_h * a.out_w)) * ty * tx;
        const bool covered = (long)a.out_h * a.out_w * 100 >= ty * 8 * tx * 32 * 88;
        if (a.tile_cfg == 3 || (patches >= 512 && (a.N <= 128 || covered))) return VDA_KERNEL_CONV_DIRECT;
    }
    if (a.N <= 32) return VDA_KERNEL_128x32;
    if (a.N <= 64) return VDA_KERNEL_128x64;
    // The 256x256 pipelined kernel moves half as many bytes global -> LDS per MFMA as the 128x128 one, which is what bounds
    // the latter (about 30 B/clk/CU of copy throughput on either); but it runs one workgroup per CU, so nothing overlaps a
    // tile's prologue and its output stores.  It wins once the K loop amortises those or the grid is many rounds deep
    // (measured, 256x256 vs 128x128 tiles, M = 43808: K = 4096 286 vs 404 us, K = 1024 N = 3072 258 vs 302, N = 256
    // (172 tiles) 29.9 vs 34.6; K = 512 N = 2048 106 vs 110; K = 256: N = 4096 129 vs 154, N = 768 35.5 vs 32.7;
    // M = 175232, K = 256: N = 768 114 vs 137, N = 2048 287 vs 326, N = 256 (685 tiles) equal).
    // Both kernels add the K steps in the same order, so the choice does not change results.
    const long tiles256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    // (the column tiles must not be mostly padding: N = 384 wastes a quarter of two 256-wide tiles; 128x128 tiles then
    // win, 96 vs 105 us at M = 43840, K = 1536)
    const bool wide = a.N >= 192 && 5 * a.N >= 4 * 256 * ((a.N + 255) / 256);
    const bool fills = a.K >= 1024 ? tiles256 >= 160 : a.K >= 512 ? tiles256 >= 1024 : a.K >= 256 && tiles256 >= 1800;
    const bool big = a.tile_cfg == 2 || (a.tile_cfg == 0 && wide && fills);
    return big ? VDA_KERNEL_256x256 : VDA_KERNEL_128x128;
}

}  // namespace

extern "C" int vda_gemm_kernel_choice(const VdaGemmArgs* args) {
    const int rc = check_args(args);
    return rc != VDA_OK ? rc : choose_kernel(*args);
}

extern "C" int vda_gemm(const VdaGemmArgs* args, void* stream) {
    using namespace vda;
    const int rc = check_args(args);
    if (rc != VDA_OK) return rc;
    const VdaGemmArgs& a = *args;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (choose_kernel(a)) {
        case VDA_KERNEL_CONV_DIRECT: return launch_conv_direct(a, s);
        case VDA_KERNEL_128x32: return launch<128, 32, 4, 1>(a, s);
        case VDA_KERNEL_128x64: return launch<128, 64, 2, 2>(a, s);
        case VDA_KERNEL_256x256: return launch_gemm256(a, s);
        case VDA_KERNEL_128x128: return launch<128, 128, 2, 2>(a, s);
        default: return VDA_ESHAPE;
    }
}
