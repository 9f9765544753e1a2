// 256x256-tile GEMM / implicit-GEMM convolution for large problems: 8 waves (2 along M x 4 along N), each wave owns a
// 128x64 output block = 8x4 mfma_f32_16x16x32 tiles (128 accumulator registers); K step 64; 128 KiB of LDS.
//
// Software pipeline.  A K-tile (256 rows of A, 256 rows of W, 64 deep) is split into four 16 KiB REGIONS by WHEN the
// waves need them, not by where they sit in the tile:
//     A0s = for every wave the first 64 of its 128 A rows      B0s = for every wave the first 32 of its 64 W rows
//     A1s = the other 64 A rows                                B1s = the other 32 W rows
// and a K-tile is multiplied in two phases of 32 MFMAs per wave (two quadrants of the wave's block each):
//     PA: read A0s, B0s, B1s -> quadrants (0,0), (0,1)         PB: read A1s (B0s, B1s kept) -> quadrants (1,1), (1,0)
// Every phase also issues global_load_lds copies for a later K-tile, into regions whose reads ended in the phase before:
//     PA(t): A1s(t+1)  (2 instructions per lane)               PB(t): A0s(t+2), B0s(t+2), B1s(t+2)  (6 per lane)
// followed by a COUNTED wait `s_waitcnt vmcnt(8)` (the copies of this phase and the one before stay in flight), a wait for the
// wave's own LDS reads, and a raw s_barrier; a second barrier follows the MFMAs.  A phase is thus two barrier intervals,
// [reads + copy issue + waits] and [32 MFMAs], and the waves 4-7 (the SIMD partners of waves 0-3) run ONE interval behind
// waves 0-3, so that on each SIMD the matrix pipe of one wave overlaps the LDS reads of the other.  (A first form of this
// kernel had four phases of 16 MFMAs per K-tile: twice the barriers for the same products.)
// Invariants (two LDS buffers, K-tile t lives in buffer t & 1):
//   * a region is re-filled in the phase after its last read (A0s, B0s, B1s: read in PA(t), refilled in PB(t); A1s: read in
//     PB(t), refilled in PA(t+1)).  A wave's reads have RETURNED when it arrives at the first barrier of the reading phase (the
//     lgkmcnt wait in front of it); a wave of the early half issues its refill after the second barrier of that phase, which
//     the late half reaches as its first, and the late half's refills come later still;
//   * a region is read no earlier than the phase after the wait that retired it: copies retire in issue order, a phase issues
//     2 or 6 per lane alternately, so vmcnt(8) at phase p retires everything issued before phase p-1, and every region is issued
//     three phases before the phase that reads it.  Each lane waits for its own copies, the barrier extends that to all lanes:
//     the late half finishes its phase-p wait in the interval just before the early half's phase-(p+1) reads;
//   * both hold for the staggered halves because the distances above are whole phases (two intervals) and the stagger is one.
// In the last two K-tiles, phases that have nothing left to issue wait with vmcnt(2) (only A1s(t+1) in flight) or vmcnt(0).
// Why eight waves and not four with 128 x 128 each (which would halve the LDS reads per product and need no stagger): an
// LDS-DMA instruction costs the issuing wave 60-180 cycles, and a K-tile needs 64 of them per workgroup.  With a partner wave
// on the SIMD that cost runs beside the partner's MFMAs; with one wave per SIMD it is added to the wave's own 128 MFMAs
// (measured on a bit-equal four-wave build of this kernel: 292 against 229 us at M = 43840, N = 3072, K = 1024).  The split of
// the copies over the phases is the measured optimum too: 4 + 4 per lane is 3 % slower, 0 + 8 is 4-7 % slower.
#include "gemm_common.h"

namespace vda {

namespace {

constexpr int REGION = 16384;
// region slots inside a buffer, in issue order
constexpr int R_A0 = 0, R_B0 = 1, R_B1 = 2, R_A1 = 3;

__device__ __forceinline__ void wait_vm8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void phase_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Epilogue of the fp32 read-modify-write form (out = acc + bias + resid, fp32 in and out; the encoder's proj and fc2 products
// add into the residual stream this way) for a wave whose 128 x 64 block lies inside the matrix.
//
// The general epilogue fetches the residual of one 16-row m-tile, waits, adds, stores, and only then fetches the next: eight
// exposed memory round trips per wave, with no registers to spare for fetching ahead (the accumulators hold 128).  Here the
// residual arrives by LDS-DMA instead, which needs no registers: the K loop is over, so each wave owns 16 KiB of the tile
// buffers as a ring of four 4 KiB slots, keeps the residual of up to four m-tiles in flight, adds its accumulators to a slot
// in place and copies the slot out in whole 256-byte rows.  A slot is 16 rows x 16 chunks of 16 bytes, lane-linear as the
// DMA writes it; the chunk stored at position c of row r is column chunk c ^ r (the permutation is applied to the source
// address, to the fragment accesses and to the store address), which keeps the 16-row fragment accesses conflict-free.
// Vector-memory operations complete in order, stores included, so the wait before m-tile i counts what was issued after
// its DMA: with D = the 4 DMA pieces and S = the 4 stores of an m-tile the issue order is D0 D1 D2 D3 S0 D4 S1 D5 S2 D6 S3
// D7 S4 S5 S6 S7.  Every store below is issued by every wave (no bounds test), or those counts would not hold.
//
// EMIT (the producer side of a folded LayerNorm, VdaGemmArgs.out16 / row_part): each 16-byte piece a lane copies out is also
// rounded to the operand type and stored (the 16 lanes of a DPP row cover a row's 64 columns: one 128-byte line), and the
// row's sums over the block are reduced across those lanes; with the 8-byte store of the rounded piece and the 8-byte store
// of the sums (lane c == 0 of every DPP row: never an empty instruction) an m-tile then issues S = 12 stores instead of 4.
template <bool EMIT, typename T>
__device__ __forceinline__ void epilogue_res32_dma(const VdaGemmArgs& p, f32x4 (&acc)[8][4], int m_wave0, int n_wave0, char* slots) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fq = lane >> 4;
    const int c = lane & 15, rr = lane >> 4;
    f32x4 bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bias[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n_wave0 + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* resid = reinterpret_cast<const float*>(p.resid);
    float* out = reinterpret_cast<float*>(p.out);
    // producer-side fields of the folded LayerNorm, fetched from the kernel arguments here (see kernarg_field_now)
    void* out16 = nullptr;
    float* row_part = nullptr;
    unsigned ld16 = 0;
    if (EMIT) {
        out16 = kernarg_field_now<void*>(offsetof(VdaGemmArgs, out16));
        row_part = kernarg_field_now<float*>(offsetof(VdaGemmArgs, row_part));
        ld16 = (unsigned)kernarg_field_now<int>(offsetof(VdaGemmArgs, ld16));
    }
    const unsigned slots_n = (unsigned)(p.N >> 6);
    auto issue = [&](int i) {
        char* slot = slots + (i & 3) * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = q * 4 + rr;
            glds16(resid + (size_t)(m_wave0 + i * 16 + r) * p.ldr + n_wave0 + ((c ^ r) << 2), slot + q * 1024);
        }
    };
    auto tile = [&](int i) {
        char* slot = slots + (i & 3) * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4* ptr = reinterpret_cast<f32x4*>(slot + frow * 256 + (((j * 4 + fq) ^ frow) << 4));
            *ptr = *ptr + (acc[i][j] + bias[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's LDS writes are done before other lanes read them
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = q * 4 + rr;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slot + r * 256 + (c << 4));
            *reinterpret_cast<u32x4*>(out + (size_t)(m_wave0 + i * 16 + r) * p.ldc + n_wave0 + ((c ^ r) << 2)) = v;
            if (EMIT) {
                const unsigned m = (unsigned)(m_wave0 + i * 16 + r);
                emit_rounded_row<T>(out16, row_part, __builtin_bit_cast(f32x4, v), (m * ld16 + (unsigned)(n_wave0 + ((c ^ r) << 2))) * 2u,
                                    (m * slots_n + (unsigned)(n_wave0 >> 6)) * 8u, true, c == 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot has been read before a DMA refills it
    };
    // before m-tile i everything issued after its DMA may still be in flight: D = 4 per later DMA, S per earlier m-tile's stores
    constexpr int S = EMIT ? 12 : 4;
    issue(0); issue(1); issue(2); issue(3);
    wait_vm<12>(); tile(0); issue(4);                  // D1 D2 D3
    wait_vm<12 + S>(); tile(1); issue(5);              // D2 D3 S0 D4
    wait_vm<12 + 2 * S>(); tile(2); issue(6);          // D3 S0 D4 S1 D5
    wait_vm<12 + 3 * S>(); tile(3); issue(7);          // S0 D4 S1 D5 S2 D6
    wait_vm<12 + 3 * S>(); tile(4);                    // S1 D5 S2 D6 S3 D7
    wait_vm<8 + 3 * S>(); tile(5);                     // S2 D6 S3 D7 S4
    wait_vm<4 + 3 * S>(); tile(6);                     // S3 D7 S4 S5
    wait_vm<3 * S>(); tile(7);                         // S4 S5 S6
    static_assert(12 + 3 * S <= 63, "vmcnt is a 6-bit counter");
}

// SPEC: the epilogue specialisation (EPI_SPEC_*, gemm_common.h).  PAIRS (VdaGemmArgs.ab_pairs): the 64 stored columns of a K-tile
// are [hi 32 | lo 32] of 32 operand columns, so the two k-steps' fragments of a phase are the hi and the lo pieces and a quadrant
// issues 24 MFMAs (W_lo A_hi + W_hi A_lo + W_hi A_hi) on the copies and fragment reads that serve 16 in the plain form.
template <int AMODE, bool RELU_IN, int SPEC, typename T, bool PAIRS = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const VdaGemmArgs p, const GemmTiles gt) {
    typedef typename Vec<T>::v8 frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    tile_of_workgroup(gt, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    // ---- staging roles: a region is 128 rows x 8 chunks of 16 bytes = 1024 pieces, two per lane ---------------------------
    const int srow = tid >> 3;                               // 0..63
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);        // source chunk of this lane's (lane-linear) LDS slot
    const bf16* a_src[2][2];                                 // [region a][piece i]
    int a_iy[2][2], a_ix[2][2];
    const bf16* w_src[2][2];                                 // [region b][piece i]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // region-local row i*64 + srow belongs to the waves with wm == i: tile row i*128 + a*64 + srow
            const int m = min(m0 + i * 128 + a * 64 + srow, p.M - 1);
            if (AMODE == VDA_A_LINEAR) {
                a_src[a][i] = reinterpret_cast<const bf16*>(p.a) + (size_t)m * p.lda + schunk * 8;
            } else {
                const int hw = p.out_h * p.out_w;
                const int img = m / hw, rem = m - img * hw;
                const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
                a_iy[a][i] = oy * p.stride - 1;
                a_ix[a][i] = ox * p.stride - 1;
                a_src[a][i] = reinterpret_cast<const bf16*>(p.a) + (size_t)img * p.in_h * p.in_w * p.cin + schunk * 8;
            }
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // region-local row i*64 + srow = wn*32 + r with wn = 2i + (srow >> 5), r = srow & 31: tile row wn*64 + b*32 + r
            const int n = min(n0 + (2 * i + (srow >> 5)) * 64 + b * 32 + (srow & 31), p.N - 1);
            w_src[b][i] = reinterpret_cast<const bf16*>(p.w) + (size_t)n * p.K + schunk * 8;
        }
    const bf16* zsrc = reinterpret_cast<const bf16*>(p.zeros) + schunk * 8;
    const int cpt = (AMODE == VDA_A_CONV3) ? p.cin / BK : 1;

    auto issue_a = [&](int a, int kt, int buf) {
        char* dst = smem + buf * (4 * REGION) + (a == 0 ? R_A0 : R_A1) * REGION;
        if (AMODE == VDA_A_LINEAR) {
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(a_src[a][i] + kt * BK, dst + (i * 512 + wave * 64) * 16);
        } else {
            const int tap = kt / cpt, cc = kt - tap * cpt;
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int iy = a_iy[a][i] + dy, ix = a_ix[a][i] + dx;
                const bool ok = (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
                const bf16* s = ok ? a_src[a][i] + ((size_t)iy * p.in_w + ix) * p.cin + cc * BK : zsrc;
                glds16(s, dst + (i * 512 + wave * 64) * 16);
            }
        }
    };
    auto issue_b = [&](int b, int kt, int buf) {
        char* dst = smem + buf * (4 * REGION) + (b == 0 ? R_B0 : R_B1) * REGION;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(w_src[b][i] + kt * BK, dst + (i * 512 + wave * 64) * 16);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    const int frow = lane & 15, fq = lane >> 4;
    // folded LayerNorm, consumer side: (mean, rstd) of this lane's eight rows, in flight during the K loop
    constexpr bool LN = SPEC == EPI_SPEC_OWN_LN || SPEC == EPI_SPEC_OWN_GELU_LN;
    f32x2 rs[LN ? 8 : 1];
    if constexpr (LN) rowstat_prefetch<8>(p, rs, m0 + wm * 128 + frow);
    // fragment read offsets inside a region (the same for both k-steps up to the chunk index)
    int a_off[4][2], b_off[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i][ks] = tile_off(wm * 64 + i * 16 + frow, ks * 4 + fq);
#pragma unroll
        for (int j = 0; j < 2; ++j) b_off[j][ks] = tile_off(wn * 32 + j * 16 + frow, ks * 4 + fq);
    }

    frag_t af[4][2], bf0[2][2], bf1[2][2];
    auto read_a = [&](const char* region) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                af[i][ks] = *reinterpret_cast<const frag_t*>(region + a_off[i][ks]);
                if (RELU_IN) af[i][ks] = relu8(af[i][ks]);
            }
    };
    auto read_b = [&](const char* region, frag_t (&bf)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf[j][ks] = *reinterpret_cast<const frag_t*>(region + b_off[j][ks]);
    };
    auto quadrant = [&](int a, int b, frag_t (&bf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (PAIRS) {
#pragma unroll
            for (int term = 0; term < 3; ++term)          // the two small terms first, then hi * hi
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[a * 4 + i][b * 2 + j] = mfma16(bf[j][term == 0], af[i][term == 1], acc[a * 4 + i][b * 2 + j]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a * 4 + i][b * 2 + j] = mfma16(bf[j][ks], af[i][ks], acc[a * 4 + i][b * 2 + j]);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // The two waves that share a SIMD are waves w and w + 4, i.e. the two values of wm.  The wm == 1 half runs one barrier
    // interval behind the other half, so that on every SIMD one wave is in its MFMA interval while its partner is in its
    // LDS-read / copy-issue interval.  group1 is wave-uniform by construction (readfirstlane makes that provable).
    const bool group1 = __builtin_amdgcn_readfirstlane(wave) >= 4;

    // ---- prologue: all of K-tile 0, and the first three regions of K-tile 1 (A1(0), issued before them, may still be in flight) ----
    issue_a(0, 0, 0);
    issue_b(0, 0, 0);
    issue_b(1, 0, 0);
    issue_a(1, 0, 0);
    if (nk > 1) {
        issue_a(0, 1, 1);
        issue_b(0, 1, 1);
        issue_b(1, 1, 1);
        wait_vm8();
    } else {
        wait_vm<2>();
    }
    phase_barrier();
    if (group1) phase_barrier();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const char* buf = smem + cur * (4 * REGION);
        const bool next1 = t + 1 < nk, next2 = t + 2 < nk;
        // PA: the upper half of the wave's block
        read_a(buf + R_A0 * REGION);
        read_b(buf + R_B0 * REGION, bf0);
        read_b(buf + R_B1 * REGION, bf1);
        if (next1) { issue_a(1, t + 1, cur ^ 1); wait_vm8(); } else { wait_vm0(); }
        wait_lds();
        phase_barrier();
        quadrant(0, 0, bf0);
        quadrant(0, 1, bf1);
        phase_barrier();
        // PB: the lower half (B0s, B1s kept in registers)
        read_a(buf + R_A1 * REGION);
        if (next2) { issue_a(0, t + 2, cur); issue_b(0, t + 2, cur); issue_b(1, t + 2, cur); wait_vm8(); }
        else if (next1) { wait_vm<2>(); }
        else { wait_vm0(); }
        wait_lds();
        phase_barrier();
        quadrant(1, 1, bf1);
        quadrant(1, 0, bf0);
        phase_barrier();
    }
    if (!group1) phase_barrier();     // every wave executes the same number of barriers

    if (SPEC == EPI_SPEC_F32_RES32 || SPEC == EPI_SPEC_F32_RES32_EMIT) {
        // whole 128 x 64 wave blocks of 16-byte aligned rows take the DMA-fed epilogue (wave-uniform choice)
        const int m_wave0 = m0 + wm * 128, n_wave0 = n0 + wn * 64;
        const bool aligned = ((p.ldr | p.ldc) & 3) == 0 && (((uintptr_t)p.resid | (uintptr_t)p.out) & 15) == 0;
        if (__builtin_amdgcn_readfirstlane(m_wave0 + 128 <= p.M && n_wave0 + 64 <= p.N && aligned)) {
            scale_acc<8, 4>(p, acc);
            epilogue_res32_dma<SPEC == EPI_SPEC_F32_RES32_EMIT, T>(p, acc, m_wave0, n_wave0, smem + wave * 16384);
            return;
        }
    }
    // (the staging area of a wave lies inside its own 16 KiB, so waves of one workgroup may take different epilogues)
    static_assert(STAGE_BYTES <= 16384, "a wave's staging area must fit its 16 KiB share of the tile buffers");
    gemm_epilogue_spec<SPEC, T, 8, 4, LN>(p, acc, m0 + wm * 128 + frow, n0 + wn * 64 + fq * 4, fq, smem + wave * 16384, nullptr, rs);
}

template <typename T>
int launch_t(const VdaGemmArgs& a, hipStream_t s) {
    GemmTiles gt{(a.M + 255) / 256, (a.N + 255) / 256};
    const int lds = 8 * REGION;
    dim3 grid(gt.tiles_m * gt.tiles_n);
#define VDA_LAUNCH256P(MODE, RELU, SPEC, PAIRS)                                                                      \
    do {                                                                                                             \
        auto k = gemm256_kernel<MODE, RELU, SPEC, T, PAIRS>;                                                         \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                  \
        hipLaunchKernelGGL(k, grid, dim3(512), lds, s, a, gt);                                                       \
    } while (0)
#define VDA_LAUNCH256(MODE, RELU, SPEC) VDA_LAUNCH256P(MODE, RELU, SPEC, false)
    // specialised epilogues where the model spends its time; everything else takes the general one
    const int spec = epi_spec_of(a);
#define VDA_CASE256(MODE, RELU, SPEC) case SPEC: VDA_LAUNCH256(MODE, RELU, SPEC); break;
#define VDA_CASE256P(MODE, SPEC) case SPEC: VDA_LAUNCH256P(MODE, false, SPEC, true); break;
    if (a.ab_pairs && a.a_mode == VDA_A_CONV3) {
        VDA_LAUNCH256P(VDA_A_CONV3, false, EPI_SPEC_GENERAL, true);
    } else if (a.ab_pairs) {          // the fp32 mode's forms only
        switch (spec) {
            VDA_CASE256P(VDA_A_LINEAR, EPI_SPEC_F32) VDA_CASE256P(VDA_A_LINEAR, EPI_SPEC_F32_RES32)
            VDA_CASE256P(VDA_A_LINEAR, EPI_SPEC_SPLIT) VDA_CASE256P(VDA_A_LINEAR, EPI_SPEC_SPLIT_GELU)
            default: VDA_LAUNCH256P(VDA_A_LINEAR, false, EPI_SPEC_GENERAL, true);
        }
    } else if (a.a_mode == VDA_A_CONV3 && a.relu_in) {
        switch (spec) {
            VDA_CASE256(VDA_A_CONV3, true, EPI_SPEC_OWN_RELU) VDA_CASE256(VDA_A_CONV3, true, EPI_SPEC_OWN_RES)
            default: VDA_LAUNCH256(VDA_A_CONV3, true, EPI_SPEC_GENERAL);
        }
    } else if (a.a_mode == VDA_A_CONV3) {
        switch (spec) {
            VDA_CASE256(VDA_A_CONV3, false, EPI_SPEC_OWN) VDA_CASE256(VDA_A_CONV3, false, EPI_SPEC_OWN_RES)
            VDA_CASE256(VDA_A_CONV3, false, EPI_SPEC_OWN_RES2)
            default: VDA_LAUNCH256(VDA_A_CONV3, false, EPI_SPEC_GENERAL);
        }
    } else if (a.relu_in) {
        VDA_LAUNCH256(VDA_A_LINEAR, true, EPI_SPEC_GENERAL);
    } else {
        switch (spec) {
            VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_OWN) VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_OWN_GELU)
            VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_F32_RES32) VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_OWN_RES)
            VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_OWN_RES32) VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_GEGLU)
            VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_F32) VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_CONVT)
            VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_OWN_LN) VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_OWN_GELU_LN)
            VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_F32_RES32_EMIT)
            VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_SPLIT) VDA_CASE256(VDA_A_LINEAR, false, EPI_SPEC_SPLIT_GELU)
            default: VDA_LAUNCH256(VDA_A_LINEAR, false, EPI_SPEC_GENERAL);
        }
    }
#undef VDA_CASE256
#undef VDA_CASE256P
#undef VDA_LAUNCH256
#undef VDA_LAUNCH256P
    return hip_rc(hipGetLastError());
}

}  // namespace

int launch_gemm256(const VdaGemmArgs& a, hipStream_t s) {
    return a.ab_type == VDA_T_F16 ? launch_t<f16>(a, s) : launch_t<bf16>(a, s);
}

}  // namespace vda
