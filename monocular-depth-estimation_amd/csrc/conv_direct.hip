// Direct 3x3 convolution (padding 1, stride 1): N = 32, 64 or a multiple of 128 output channels (blocks of 128 channels are
// separate workgroups, grid.y).
//
// As an implicit GEMM a 3x3 convolution gathers every input pixel nine times; with few output channels there is little
// MFMA work per gathered byte and the global -> LDS copy path, not the matrix pipe, sets the time (128 output channels:
// 63 % of the MFMA rate at best; 32: a quarter).  Here a workgroup owns an 8 x 32 patch of output pixels of one image and
// copies the 10 x 34 input halo of the patch into LDS ONCE per 64-channel slice; the nine taps then read their A fragments
// from shifted positions of that halo (16 consecutive pixels of a halo row are the 16 rows of an MFMA operand).  Only the
// weights are streamed per tap.
//
// 4 waves; wave w owns patch rows 2w, 2w+1 = four 16-pixel MFMA row tiles x all N channels (4 x N/16 accumulator tiles).
// LDS: one halo slice (340 pixels x 128 B) + two weight buffers (one step = TPS taps x N rows x 128 B): <= 78 KiB, so two
// workgroups share a CU and one computes while the other waits for its next halo slice.
// Step loop (a step = TPS taps of one channel slice):  [first step of a slice: issue the halo copies]  issue the weight
// copies of the next step; wait until everything but those has landed (counted vmcnt; copies retire in issue order);
// barrier; MFMAs; barrier (the buffers may be refilled).
//
// LDS images are lane-linear (global_load_lds); halo pixel hp keeps its 16-byte channel chunk c at slot c ^ (hp & 7),
// which is conflict-free for the ds_read_b128 lane groups at any starting pixel of a 16-pixel run; weight rows use the
// common tile_off() swizzle.
#include "gemm_common.h"

namespace vda {

namespace {

constexpr int PT_H = 8, PT_W = 32;                         // output patch of a workgroup
constexpr int HALO_W = PT_W + 2, HALO_H = PT_H + 2, HALO_PX = HALO_W * HALO_H;
constexpr int HALO_ITERS = (HALO_PX * 8 + 255) / 256;      // 16-byte pieces per thread and slice
constexpr int HALO_BYTES = HALO_ITERS * 256 * 16;          // whole iterations: lanes past the last pixel copy zeros into the pad

template <int TN, int TPS, bool RELU_IN, typename T>
__global__ __launch_bounds__(256, 2) void conv3_direct_kernel(const VdaGemmArgs p, const int tiles_x, const int tiles_y) {
    typedef typename Vec<T>::v8 frag_t;
    constexpr int N = TN * 16;
    constexpr int W_ROWS = TPS * N, W_BYTES = W_ROWS * 128, W_ITERS = W_ROWS * 8 / 256;
    constexpr int SPC = 9 / TPS;                            // steps per channel slice
    static_assert(9 % TPS == 0 && (W_ROWS * 8) % 256 == 0, "step shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wbuf = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int tx = blockIdx.x % tiles_x;
    const int ty = (blockIdx.x / tiles_x) % tiles_y;
    const int img = blockIdx.x / (tiles_x * tiles_y);
    const int oy0 = ty * PT_H, ox0 = tx * PT_W;
    const bf16* a_img = reinterpret_cast<const bf16*>(p.a) + (size_t)img * p.in_h * p.in_w * p.cin;
    const bf16* zsrc = reinterpret_cast<const bf16*>(p.zeros);
    const int n0 = blockIdx.y * N;                           // first output channel of this workgroup

    // ---- copy sources (element offsets inside the image; -1 = padding or pad lanes: copy zeros) ------------------------------
    int hoff[HALO_ITERS];
#pragma unroll
    for (int it = 0; it < HALO_ITERS; ++it) {
        const int piece = it * 256 + tid;
        const int hp = piece >> 3, slot = piece & 7;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
        const bool ok = hp < HALO_PX && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
        hoff[it] = ok ? (iy * p.in_w + ix) * p.cin + ((slot ^ (hp & 7)) << 3) : -1;
    }
    int woff[W_ITERS];
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
        const int piece = it * 256 + tid;
        const int row = piece >> 3, slot = piece & 7;        // row = tap-in-step * N + n
        const int t_in = row / N, n = row - t_in * N;
        woff[it] = (n * 9 + t_in) * p.cin + ((slot ^ ((row >> 1) & 7)) << 3);
    }
    auto issue_halo = [&](int cc) {
#pragma unroll
        for (int it = 0; it < HALO_ITERS; ++it) {
            const bf16* s = hoff[it] >= 0 ? a_img + hoff[it] + cc * 64 : zsrc;
            glds16(s, halo + (it * 256 + wave * 64) * 16);
        }
    };
    auto issue_w = [&](int s, int buf) {
        const int cc = s / SPC, tap0 = (s - cc * SPC) * TPS;
        const bf16* wb = reinterpret_cast<const bf16*>(p.w) + ((size_t)n0 * 9 + tap0) * p.cin + cc * 64;
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) glds16(wb + woff[it], wbuf + buf * W_BYTES + (it * 256 + wave * 64) * 16);
    };

    // ---- fragment addressing -------------------------------------------------------------------------------------------------
    int hbase[4];                                            // halo pixel of this lane's row of m-tile i for tap (0, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) hbase[i] = (2 * wave + (i >> 1)) * HALO_W + (i & 1) * 16 + frow;
    int w_off[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[j][ks] = tile_off(j * 16 + frow, ks * 4 + fq);

    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int steps = (p.cin / 64) * SPC;
    issue_w(0, 0);
    for (int s = 0; s < steps; ++s) {
        const int cc = s / SPC, tg = s - cc * SPC;
        if (tg == 0) issue_halo(cc);                         // all reads of the previous slice ended at the last barrier
        if (s + 1 < steps) {
            issue_w(s + 1, (s + 1) & 1);
            if (W_ITERS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (W_ITERS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* wb = wbuf + (s & 1) * W_BYTES;
#pragma unroll
        for (int t_in = 0; t_in < TPS; ++t_in) {
            const int tap = tg * TPS + t_in;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int toff = dy * HALO_W + dx;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag_t wf[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const frag_t*>(wb + t_in * N * 128 + w_off[j][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hp = hbase[i] + toff;
                    frag_t af = *reinterpret_cast<const frag_t*>(halo + hp * 128 + (((ks * 4 + fq) ^ (hp & 7)) << 4));
                    if (RELU_IN) af = relu8(af);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(wf[j], af, acc[i][j]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue: lane (frow, fq) holds pixel frow of each m-tile and channels 16 j + 4 fq + {0..3} ----------------------------
    f32x4 bias[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        bias[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

    if (p.epi == VDA_EPI_DOT) {
        f32x4 dw[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) dw[j] = *reinterpret_cast<const f32x4*>(p.dot_w + j * 16 + fq * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) part += fmaxf(acc[i][j][r] + bias[j][r], 0.f) * dw[j][r];
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            const int oy = oy0 + 2 * wave + (i >> 1), ox = ox0 + (i & 1) * 16 + frow;
            if (fq == 0 && oy < p.out_h && ox < p.out_w)
                reinterpret_cast<float*>(p.out)[((size_t)img * p.out_h + oy) * p.out_w + ox] = fmaxf(part + p.dot_b, 0.f);
        }
        return;
    }

    // plain epilogue: each 16-pixel m-tile is staged through a wave-private LDS block (the step loop is over, LDS is free)
    // and stored as whole pixel rows, 16 bytes per lane
    const int esz = p.out_type == VDA_T_F32 ? 4 : 2;
    const int row_bytes = N * esz, rs = row_bytes + 16;     // padded staging rows
    const int cpr = row_bytes >> 4, rpi = 64 / cpr;          // 16-byte chunks per row, rows per store instruction
    char* stage = smem + wave * (16 * (N * 4 + 16));
    const int c_row = lane / cpr, c_chunk = lane - c_row * cpr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = oy0 + 2 * wave + (i >> 1), oxb = ox0 + (i & 1) * 16;
        const bool mine = oy < p.out_h && oxb + frow < p.out_w;                   // this lane's pixel exists
        const size_t pix = ((size_t)img * p.out_h + oy) * p.out_w + oxb + frow;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = acc[i][j] + bias[j];
            if (p.act == VDA_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (mine && p.resid_type != VDA_T_NONE) {
                v += load4(p.resid, p.resid_type, pix * p.ldr + n0 + j * 16 + fq * 4);
                if (p.resid2) v += load4(p.resid2, p.resid_type, pix * p.ldr2 + n0 + j * 16 + fq * 4);
            }
            store4(stage + frow * rs, p.out_type, j * 16 + fq * 4, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int r0 = 0; r0 < 16; r0 += rpi) {
            const int row = r0 + c_row;
            if (row < 16) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * rs + (c_chunk << 4));
                const int ox = oxb + row;
                if (oy < p.out_h && ox < p.out_w)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.out) +
                                              ((((size_t)img * p.out_h + oy) * p.out_w + ox) * p.ldc + n0) * esz + (c_chunk << 4)) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int TN, int TPS, bool RELU_IN, typename T>
int launch_cfg(const VdaGemmArgs& a, hipStream_t s) {
    const int tiles_x = (a.out_w + PT_W - 1) / PT_W, tiles_y = (a.out_h + PT_H - 1) / PT_H;
    const int imgs = a.M / (a.out_h * a.out_w);
    const int lds = HALO_BYTES + 2 * TPS * TN * 16 * 128;
    auto k = conv3_direct_kernel<TN, TPS, RELU_IN, T>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(imgs * tiles_x * tiles_y), a.N / (TN * 16)), dim3(256), lds, s, a, tiles_x, tiles_y);
    return hip_rc(hipGetLastError());
}

template <bool RELU_IN, typename T>
int launch_r(const VdaGemmArgs& a, hipStream_t s) {
    if (a.N % 128 == 0) return launch_cfg<8, 1, RELU_IN, T>(a, s);
    if (a.N == 64) return launch_cfg<4, 3, RELU_IN, T>(a, s);
    return launch_cfg<2, 3, RELU_IN, T>(a, s);
}

template <typename T>
int launch_t(const VdaGemmArgs& a, hipStream_t s) {
    return a.relu_in ? launch_r<true, T>(a, s) : launch_r<false, T>(a, s);
}

}  // namespace

// true if the direct kernel implements this call (the caller has validated the arguments of vda_gemm)
bool conv_direct_supports(const VdaGemmArgs& a) {
    if (a.a_mode != VDA_A_CONV3 || a.stride != 1 || a.out_h != a.in_h || a.out_w != a.in_w) return false;
    if (a.N != 32 && a.N != 64 && a.N % 128 != 0) return false;
    if (a.gamma) return false;
    if (a.epi == VDA_EPI_DOT) return a.N == 32;
    if (a.epi != VDA_EPI_PLAIN || a.act == VDA_ACT_GELU) return false;
    const int esz = a.out_type == VDA_T_F32 ? 4 : 2;
    if ((a.ldc * esz) % 16 != 0 || ((uintptr_t)a.out & 15) != 0 || a.ldc < a.N) return false;
    if ((long)a.in_h * a.in_w * a.cin > 0x7fffffffL) return false;       // 32-bit element offsets inside one image
    return a.M % (a.out_h * a.out_w) == 0;
}

int launch_conv_direct(const VdaGemmArgs& a, hipStream_t s) {
    return a.ab_type == VDA_T_F16 ? launch_t<f16>(a, s) : launch_t<bf16>(a, s);
}

}  // namespace vda
