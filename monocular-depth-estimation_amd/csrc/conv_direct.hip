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
// LDS: one halo slice (340 pixels x 128 B) + two weight buffers (one step = TPS taps x N rows x 128 B): <= 78 KiB for the
// (N, TPS) pairs the launcher picks, so two workgroups share a CU and one computes while the other waits for its next halo slice.
// Step loop (a step = TPS taps of one channel slice):  [first step of a slice: issue the halo copies]  issue the weight
// copies of the next step; wait until everything but those has landed (counted vmcnt; copies retire in issue order);
// barrier; MFMAs; barrier (the buffers may be refilled).
//
// Fused upsampling (UP): the convolution input is the bilinear (align_corners) resampling of a smaller map.  The halo is
// then not copied but computed: each lane blends the four source pixels of its 16-byte piece in fp32 and writes the piece
// to the same LDS slot, so the resampled map (4 - 12 x the source) is never written to or read from HBM.  The four pixels
// come straight from global memory, or (fp16 maps, scales <= 0.9 x 1) from the source patch that global_load_lds has
// parked in the halo area first.
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

// Mixed-precision FMA: one fp16 half of a packed register times an fp32 weight plus an fp32 addend, with no separate
// conversion instruction; the last step of a chain rounds to fp16 (nearest even) into one half of the destination.
template <int HI>
__device__ __forceinline__ float mix_fma(unsigned h2, float w, float c) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(w), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(w), "v"(c));
    return r;
}
// low half first: it defines the register (the other half is then still unspecified), the high half completes it
__device__ __forceinline__ unsigned mix_fma_to_lo(unsigned h2, float w, float c) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(c));
    return d;
}
__device__ __forceinline__ void mix_fma_to_hi(unsigned& d, unsigned h2, float w, float c) {
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h2), "v"(w), "v"(c));
}

// sum_j w[j] * q[j] over four 16-byte pieces of 8 values of type T, accumulated in fp32 and rounded once
template <typename T>
__device__ __forceinline__ u32x4 blend4(const u32x4 (&q)[4], float w0, float w1, float w2, float w3);
template <>
__device__ __forceinline__ u32x4 blend4<f16>(const u32x4 (&q)[4], float w0, float w1, float w2, float w3) {
    u32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float lo = mix_fma<0>(q[0][r], w0, 0.f), hi = mix_fma<1>(q[0][r], w0, 0.f);
        lo = mix_fma<0>(q[1][r], w1, lo);
        hi = mix_fma<1>(q[1][r], w1, hi);
        lo = mix_fma<0>(q[2][r], w2, lo);
        hi = mix_fma<1>(q[2][r], w2, hi);
        unsigned d = mix_fma_to_lo(q[3][r], w3, lo);
        mix_fma_to_hi(d, q[3][r], w3, hi);
        o[r] = d;
    }
    return o;
}
template <>
__device__ __forceinline__ u32x4 blend4<bf16>(const u32x4 (&q)[4], float w0, float w1, float w2, float w3) {
    bf16x8 v[4], o;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(bf16x8, q[j]);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)(w0 * (float)v[0][k] + w1 * (float)v[1][k] + w2 * (float)v[2][k] + w3 * (float)v[3][k]);
    return __builtin_bit_cast(u32x4, o);
}

// w0 * a + w1 * b over two 16-byte pieces, likewise
template <typename T>
__device__ __forceinline__ u32x4 blend2(const u32x4& a, const u32x4& b, float w0, float w1);
template <>
__device__ __forceinline__ u32x4 blend2<f16>(const u32x4& a, const u32x4& b, float w0, float w1) {
    u32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lo = mix_fma<0>(a[r], w0, 0.f), hi = mix_fma<1>(a[r], w0, 0.f);
        unsigned d = mix_fma_to_lo(b[r], w1, lo);
        mix_fma_to_hi(d, b[r], w1, hi);
        o[r] = d;
    }
    return o;
}
template <>
__device__ __forceinline__ u32x4 blend2<bf16>(const u32x4& a, const u32x4& b, float w0, float w1) {
    const bf16x8 va = __builtin_bit_cast(bf16x8, a), vb = __builtin_bit_cast(bf16x8, b);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)(w0 * (float)va[k] + w1 * (float)vb[k]);
    return __builtin_bit_cast(u32x4, o);
}

// PAIRS (VdaGemmArgs.ab_pairs): a 64-column channel slice is [hi 32 | lo 32] of 32 operand channels, in the map and in the
// weights; a tap multiplies W_lo A_hi + W_hi A_lo + W_hi A_hi from one halo slice and one read of each fragment.
template <int TN, int TPS, bool RELU_IN, bool UP, int SPEC, typename T, bool PAIRS = false>
__global__ __launch_bounds__(256, 2) void conv3_direct_kernel(const VdaGemmArgs pa, const int tiles_x, const int tiles_y, const int parked_arg) {
    // (the parked form of the resampling exists for fp16 maps only; the launcher asks for it for no other type)
    const bool parked = __is_same(T, f16) && parked_arg;
    const VdaGemmArgs p = epi_spec_args<SPEC, T>(pa);          // the epilogue's choices as constants (EPI_SPEC_*)
    typedef typename Vec<T>::v8 frag_t;
    constexpr int N = TN * 16;
    constexpr int W_ROWS = TPS * N, W_BYTES = W_ROWS * 128, W_ITERS = W_ROWS * 8 / 256;
    constexpr int SPC = 9 / TPS;                            // steps per channel slice
    static_assert(9 % TPS == 0 && (W_ROWS * 8) % 256 == 0, "step shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wbuf = smem + HALO_BYTES;
    char* up_tab = wbuf + 2 * W_BYTES;                       // UP: (HALO_H + HALO_W) x 16 bytes, see below

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int tx = blockIdx.x % tiles_x;
    const int ty = (blockIdx.x / tiles_x) % tiles_y;
    const int img = blockIdx.x / (tiles_x * tiles_y);
    const int oy0 = ty * PT_H, ox0 = tx * PT_W;
    const int src_h = UP ? p.up_h : p.in_h, src_w = UP ? p.up_w : p.in_w;     // size of the map in memory
    const bf16* a_img = reinterpret_cast<const bf16*>(p.a) + (size_t)img * src_h * src_w * p.cin;
    const bf16* zsrc = reinterpret_cast<const bf16*>(p.zeros);
    const int n0 = blockIdx.y * N;                           // first output channel of this workgroup

    // ---- copy sources (element offsets inside the image; -1 = padding or pad lanes: copy zeros) ------------------------------
    int hoff[UP ? 1 : HALO_ITERS];
    if (!UP) {
#pragma unroll
        for (int it = 0; it < HALO_ITERS; ++it) {
            const int piece = it * 256 + tid;
            const int hp = piece >> 3, slot = piece & 7;
            const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
            const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            const bool ok = hp < HALO_PX && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
            hoff[UP ? 0 : it] = ok ? (iy * p.in_w + ix) * p.cin + ((slot ^ (hp & 7)) << 3) : -1;
        }
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
            const bf16* s = hoff[UP ? 0 : it] >= 0 ? a_img + hoff[UP ? 0 : it] + cc * 64 : zsrc;
            glds16(s, halo + (it * 256 + wave * 64) * 16);
        }
    };
    // UP: blend the four source pixels of every piece and write it where the copy would have put it.  The source
    // positions are recomputed per piece (a few instructions against four loads) rather than kept in registers.
    const float up_sy = UP && p.in_h > 1 ? (float)(src_h - 1) / (float)(p.in_h - 1) : 0.f;
    const float up_sx = UP && p.in_w > 1 ? (float)(src_w - 1) / (float)(p.in_w - 1) : 0.f;
    auto blend_halo = [&](int cc) {
        constexpr int G = TN >= 8 ? 2 : 4;                   // pieces whose 4 loads each are in flight together (registers)
        const unsigned px_step = p.cin * sizeof(T), row_step = src_w * px_step;
        const char* a_cc = reinterpret_cast<const char*>(a_img) + cc * 128;       // wave-uniform base, 32-bit byte offsets
#pragma unroll 1
        for (int g0 = 0; g0 < HALO_ITERS; g0 += G) {
            u32x4 q[G][4];
            float wx1[G], wy1[G];
            bool okg[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int piece = (g0 + g) * 256 + tid;
                const int hp = piece >> 3, slot = piece & 7;
                const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
                const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                const bool ok = hp < HALO_PX && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
                const float fy = up_sy * (float)iy, fx = up_sx * (float)ix;      // ATen: src = scale * dst, i0 = (int)src
                const int y0 = (int)fy, x0 = (int)fx;
                wy1[g] = fy - (float)y0;
                wx1[g] = fx - (float)x0;
                okg[g] = ok;
                // pieces outside the image read the first pixel (in bounds) and are written as zeros
                const unsigned dx = ok && x0 < src_w - 1 ? px_step : 0, dy = ok && y0 < src_h - 1 ? row_step : 0;
                const unsigned o00 = ok ? (unsigned)(y0 * src_w + x0) * px_step + ((slot ^ (hp & 7)) << 4) : 0;
                q[g][0] = *reinterpret_cast<const u32x4*>(a_cc + o00);
                q[g][1] = *reinterpret_cast<const u32x4*>(a_cc + (o00 + dx));
                q[g][2] = *reinterpret_cast<const u32x4*>(a_cc + (o00 + dy));
                q[g][3] = *reinterpret_cast<const u32x4*>(a_cc + (o00 + dy + dx));
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float w00 = (1.f - wy1[g]) * (1.f - wx1[g]), w01 = (1.f - wy1[g]) * wx1[g];
                const float w10 = wy1[g] * (1.f - wx1[g]), w11 = wy1[g] * wx1[g];
                u32x4 o = blend4<T>(q[g], w00, w01, w10, w11);
                if (!okg[g]) o = u32x4{0u, 0u, 0u, 0u};
                if (g0 + g < HALO_ITERS) *reinterpret_cast<u32x4*>(halo + ((g0 + g) * 256 + tid) * 16) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the writes have landed before this wave reaches the barrier
    };
    // UP, both scales such that the patch's 10 x 34 halo comes from at most 11 source rows x 32 source columns (vertical
    // scale <= 1, horizontal <= 0.9): the source patch itself is copied into the halo area by global_load_lds, one source
    // row of 32 pixels x 8 chunks per instruction -- a third of the loads of the four-tap form, no registers, ONE round of
    // memory latency however few registers the accumulators leave -- and every halo piece is then blended from four LDS
    // reads with the four-tap form's arithmetic (one rounding).  In place: all pieces are formed before any is written.
    auto blend_halo_parked = [&](int cc, bool half) {
        const unsigned px_step = p.cin * sizeof(T), row_step = src_w * px_step;
        const char* a_cc = reinterpret_cast<const char*>(a_img) + cc * 128;
        const int iy_lo = max(oy0 - 1, 0), iy_hi = min(oy0 + PT_H, p.in_h - 1);
        const int ix_lo = max(ox0 - 1, 0), ix_hi = min(ox0 + PT_W, p.in_w - 1);
        const int xs0 = (int)(up_sx * (float)ix_lo), ys0 = (int)(up_sy * (float)iy_lo);
        const int ncol = min((int)(up_sx * (float)ix_hi) + 1, src_w - 1) - xs0 + 1;            // <= 32
        const int nrow = min((int)(up_sy * (float)iy_hi) + 1, src_h - 1) - ys0 + 1;            // <= 11
        {
            const int sc = tid >> 3, slot = tid & 7;
            const unsigned col_off = (unsigned)(xs0 + min(sc, ncol - 1)) * px_step + ((slot ^ (sc & 7)) << 4);
            for (int it = 0; it < nrow; ++it)
                glds16(a_cc + ((unsigned)(ys0 + it) * row_step + col_off), halo + (it * 256 + wave * 64) * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (also the weights of this step, needed next anyway)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // A half slice (only 32 of the slice's 64 channels are used) has four pieces per halo pixel instead of eight.
        const int sh = half ? 2 : 3, iters = half ? (HALO_PX * 4 + 255) / 256 : HALO_ITERS;
        u32x4 res[HALO_ITERS];
        constexpr int G = TN >= 8 ? 1 : 4;                   // pieces whose four reads each are in flight together (registers)
        // What a piece needs besides its data -- the parked rows and columns it blends and their weights -- is separable and
        // the same for every channel slice: it comes from the workgroup's tables (up_tab, built once), one entry per halo row
        // and one per halo column.  Left alone the compiler would hoist the per-piece values out of the slice loop and keep
        // some hundred of them live across the MFMA steps: make the piece index depend on this call.
        int t_id = tid;
        asm volatile("" : "+v"(t_id));
#pragma unroll
        for (int g0 = 0; g0 < HALO_ITERS; g0 += G) {
            // (guards, not breaks: a loop with an early exit is not unrolled into straight code, its index into `res` stays a
            // run-time value and the array lands in scratch memory, one memory round trip per piece on the way back)
            u32x4 q[G][4], re[G], ce[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g0 + g < HALO_ITERS && g0 + g < iters) {
                    const int hp = min(((g0 + g) * 256 + t_id) >> sh, HALO_PX - 1);     // (pad lanes blend the last pixel; not written)
                    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
                    re[g] = *reinterpret_cast<const u32x4*>(up_tab + hy * 16);
                    ce[g] = *reinterpret_cast<const u32x4*>(up_tab + (HALO_H + hx) * 16);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g0 + g < HALO_ITERS && g0 + g < iters) {
                    const int piece = (g0 + g) * 256 + t_id;
                    const int hp = piece >> sh, lo = piece & ((1 << sh) - 1), chunk = half ? lo : lo ^ (hp & 7);
                    const unsigned a0 = ce[g][0] ^ (unsigned)(chunk << 4), a1 = ce[g][1] ^ (unsigned)(chunk << 4);
                    q[g][0] = *reinterpret_cast<const u32x4*>(halo + (re[g][0] + a0));
                    q[g][1] = *reinterpret_cast<const u32x4*>(halo + (re[g][0] + a1));
                    q[g][2] = *reinterpret_cast<const u32x4*>(halo + (re[g][1] + a0));
                    q[g][3] = *reinterpret_cast<const u32x4*>(halo + (re[g][1] + a1));
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g0 + g < HALO_ITERS && g0 + g < iters) {
                    // (through named values: __builtin_bit_cast applied directly to a vector element reads element 0)
                    const unsigned uy0 = re[g][2], uy1 = re[g][3], ux0 = ce[g][2], ux1 = ce[g][3];
                    const float wy0 = __uint_as_float(uy0), wy1 = __uint_as_float(uy1);
                    const float wx0 = __uint_as_float(ux0), wx1 = __uint_as_float(ux1);
                    res[g0 + g] = blend4<T>(q[g], wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);               // keep the groups apart: only G x 4 reads are live at a time
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                                 // every parked value has been read
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < HALO_ITERS; ++it) {
            if (it < iters) {
                const int piece = it * 256 + t_id;
                const int hp = piece >> sh, lo = piece & ((1 << sh) - 1), chunk = half ? lo : lo ^ (hp & 7);
                if (hp < HALO_PX) *reinterpret_cast<u32x4*>(halo + hp * 128 + ((chunk ^ (hp & 7)) << 4)) = res[it];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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

    // cin_used: the last slice may be a half slice (32 channels): its second K step and, in the parked resampling, the
    // pieces of its upper 32 channels are skipped
    const int cin_eff = pa.cin_used > 0 ? pa.cin_used : p.cin;
    const int slices = (cin_eff + 63) / 64;
    const bool tail_half = (cin_eff & 63) == 32;
    const int steps = slices * SPC;
    if (UP && parked) {
        // Tables of the parked resampling: entry hy (10 halo rows) = {byte offset of parked row y0, of row y1, 1 - wy, wy},
        // entry 10 + hx (34 halo columns) = {offset of parked column x0 incl. its chunk swizzle, of column x1, 1 - wx, wx};
        // rows and columns outside the image get zero weights (the piece is then a zero), positions with ATen's arithmetic
        // (src = scale * dst, i0 = (int)src).  A piece's four weights are products of one row and one column weight.
        if (tid < HALO_H + HALO_W) {
            const bool is_row = tid < HALO_H;
            const int h = is_row ? tid : tid - HALO_H;
            const int i_lo = is_row ? max(oy0 - 1, 0) : max(ox0 - 1, 0);
            const int i_hi = is_row ? min(oy0 + PT_H, p.in_h - 1) : min(ox0 + PT_W, p.in_w - 1);
            const int idx = (is_row ? oy0 : ox0) + h - 1, extent = is_row ? p.in_h : p.in_w, src_n = is_row ? src_h : src_w;
            const float sc = is_row ? up_sy : up_sx;
            const int s0 = (int)(sc * (float)i_lo);                                  // first parked row / column
            const float f = sc * (float)min(max(idx, i_lo), i_hi);
            const int i0 = (int)f, i1 = min(i0 + 1, src_n - 1);
            const float w1 = f - (float)i0, w0 = 1.f - w1;
            const bool ok = (unsigned)idx < (unsigned)extent;
            const int e0 = i0 - s0, e1 = i1 - s0;
            u32x4 e;
            e[0] = is_row ? e0 * 32 * 128 : e0 * 128 + ((e0 & 7) << 4);
            e[1] = is_row ? e1 * 32 * 128 : e1 * 128 + ((e1 & 7) << 4);
            e[2] = __builtin_bit_cast(unsigned, ok ? w0 : 0.f);
            e[3] = __builtin_bit_cast(unsigned, ok ? w1 : 0.f);
            *reinterpret_cast<u32x4*>(up_tab + tid * 16) = e;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the first use lies behind the barrier that follows the parking)
    }
    issue_w(0, 0);
    for (int s = 0; s < steps; ++s) {
        const int cc = s / SPC, tg = s - cc * SPC;
        const bool half = tail_half && cc == slices - 1;
        if (tg == 0) {                                       // all reads of the previous slice ended at the last barrier
            if (UP) {
                if (parked) blend_halo_parked(cc, half); else blend_halo(cc);
            } else {
                issue_halo(cc);
            }
        }
        if (s + 1 < steps) {
            issue_w(s + 1, (s + 1) & 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_ITERS) : "memory");      // everything but the copies just issued
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
            if constexpr (PAIRS) {
                constexpr int JH = TN > 4 ? 2 : 1, JN = TN / JH;      // channel tiles in halves: fragments of 8 tiles x 2 pieces would not fit
#pragma unroll
                for (int jh = 0; jh < JH; ++jh) {
                    frag_t wf[2][JN];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int j = 0; j < JN; ++j) wf[ks][j] = *reinterpret_cast<const frag_t*>(wb + t_in * N * 128 + w_off[jh * JN + j][ks]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int hp = hbase[i] + toff;
                        const frag_t a_hi = *reinterpret_cast<const frag_t*>(halo + hp * 128 + ((fq ^ (hp & 7)) << 4));
                        const frag_t a_lo = *reinterpret_cast<const frag_t*>(halo + hp * 128 + (((4 + fq) ^ (hp & 7)) << 4));
#pragma unroll
                        for (int j = 0; j < JN; ++j) acc[i][jh * JN + j] = mfma16(wf[1][j], a_hi, acc[i][jh * JN + j]);
#pragma unroll
                        for (int j = 0; j < JN; ++j) acc[i][jh * JN + j] = mfma16(wf[0][j], a_lo, acc[i][jh * JN + j]);
#pragma unroll
                        for (int j = 0; j < JN; ++j) acc[i][jh * JN + j] = mfma16(wf[0][j], a_hi, acc[i][jh * JN + j]);
                    }
                }
            } else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 1 && half) continue;
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
    scale_acc<4, TN>(p, acc);
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

template <int TN, int TPS, bool RELU_IN, bool UP, int SPEC, typename T, bool PAIRS = false>
int launch_cfg(const VdaGemmArgs& a, hipStream_t s) {
    const int tiles_x = (a.out_w + PT_W - 1) / PT_W, tiles_y = (a.out_h + PT_H - 1) / PT_H;
    const int imgs = a.M / (a.out_h * a.out_w);
    const int lds = HALO_BYTES + 2 * TPS * TN * 16 * 128 + (UP ? (HALO_H + HALO_W) * 16 : 0);
    // The parked form of the resampling needs a patch's halo to come from at most 32 source columns and 11 source rows:
    // horizontal scale <= 0.9, vertical <= 1.  fp16 maps only: the bf16 blend converts every value and needs the registers
    // that form leaves to the four-tap loader.
    const int parked = UP && a.ab_type == VDA_T_F16 && 10L * (a.up_w - 1) <= 9L * (a.in_w - 1) && a.up_h <= a.in_h;
    auto k = conv3_direct_kernel<TN, TPS, RELU_IN, UP, SPEC, T, PAIRS>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(imgs * tiles_x * tiles_y), a.N / (TN * 16)), dim3(256), lds, s, a, tiles_x, tiles_y, parked);
    return hip_rc(hipGetLastError());
}

template <bool RELU_IN, bool UP, int SPEC, typename T, bool PAIRS = false>
int launch_r(const VdaGemmArgs& a, hipStream_t s) {
    if (a.N % 128 == 0) return launch_cfg<8, 1, RELU_IN, UP, SPEC, T, PAIRS>(a, s);
    // Taps per step: as many as leave room for two workgroups per CU (halo 44 KiB + two weight buffers <= 80 KiB).  64 channels
    // with three taps per step (92 KiB) ran one workgroup per CU: +2.4 % on the ViT-S window for one tap per step.  32 channels
    // with one tap per step (three workgroups per CU) is slower than three taps and two workgroups (1773 against 1632 us).
    if (a.N == 64) return launch_cfg<4, 1, RELU_IN, UP, SPEC, T, PAIRS>(a, s);
    return launch_cfg<2, 3, RELU_IN, UP, SPEC, T, PAIRS>(a, s);
}

// the forms the DPT head uses get an epilogue of their own (EPI_SPEC_*); anything else takes the general one
template <typename T>
int launch_t(const VdaGemmArgs& a, hipStream_t s) {
    const int spec = epi_spec_of(a);
    if (a.ab_pairs) {         // fp32 mode: no fused resampling or pre-activation (vda_gemm has checked); fp32 outputs
        if (spec == EPI_SPEC_DOT) return launch_r<false, false, EPI_SPEC_DOT, T, true>(a, s);
        return launch_r<false, false, EPI_SPEC_GENERAL, T, true>(a, s);
    }
    if (a.up_h > 0) {
        if (a.relu_in) return launch_r<true, true, EPI_SPEC_GENERAL, T>(a, s);
        if (spec == EPI_SPEC_OWN) return launch_r<false, true, EPI_SPEC_OWN, T>(a, s);
        if (spec == EPI_SPEC_DOT) return launch_r<false, true, EPI_SPEC_DOT, T>(a, s);
        return launch_r<false, true, EPI_SPEC_GENERAL, T>(a, s);
    }
    if (a.relu_in) {
        if (spec == EPI_SPEC_OWN_RELU) return launch_r<true, false, EPI_SPEC_OWN_RELU, T>(a, s);
        return launch_r<true, false, EPI_SPEC_GENERAL, T>(a, s);
    }
    if (spec == EPI_SPEC_OWN) return launch_r<false, false, EPI_SPEC_OWN, T>(a, s);
    if (spec == EPI_SPEC_OWN_RES) return launch_r<false, false, EPI_SPEC_OWN_RES, T>(a, s);
    if (spec == EPI_SPEC_OWN_RES2) return launch_r<false, false, EPI_SPEC_OWN_RES2, T>(a, s);
    return launch_r<false, false, EPI_SPEC_GENERAL, T>(a, s);
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
    if (a.up_h > 0 && (long)a.up_h * a.up_w * a.cin > 0x7fffffffL) return false;
    return a.M % (a.out_h * a.out_w) == 0;
}

int launch_conv_direct(const VdaGemmArgs& a, hipStream_t s) {
    return a.ab_type == VDA_T_F16 ? launch_t<f16>(a, s) : launch_t<bf16>(a, s);
}

}  // namespace vda
