// Flash-style multi-head self-attention for head dim 64 (see vda_attention in include/vda_hip.h).
//
// One workgroup = 4 waves = 128 NQB query rows of one (frame, head); each wave owns NQB blocks of 32 queries (with NQB = 2
// every K and V fragment read from LDS feeds two MFMAs: half the LDS traffic per MFMA for more registers).  K/V tiles of 64
// keys are copied global -> LDS by global_load_lds into a double buffer (the copy of tile t+1 is in flight during the math
// of tile t); the fp32 variant stages through registers, where it splits the values.  Scores are computed TRANSPOSED, S^T = K Q^T with mfma_32x32x16, so that the query index
// lives on the lane: every lane owns one query's running max / sum, the softmax needs one cross-lane exchange
// (lane ^ 32) per tile, and the fp32 score registers, converted to bf16, are directly the B operand of the second
// product O^T += V^T P^T (an accumulator tile reused as the next MFMA's operand; its k order is permuted and the V^T
// operand is gathered with the same permutation by ds_read_b64_tr_b16 transposed LDS reads).
#include "common.h"

namespace {



// Scores are base-2 logarithms: the caller folds log2(e) / sqrt(64) into q, so a probability is exp2(score - maximum) and
// v_exp_f32 needs no multiply in front of it.
//
// Growth of the running maximum (in those units) below which the accumulated output is NOT rescaled: probabilities of a tile
// may then reach 2^DEFER instead of 1, harmless for 16-bit operands with fp32 accumulation, and the rescale pass is skipped on
// almost every tile once the maximum has settled.
constexpr float DEFER = 6.0f;

// the two 16-bit operand types of the 32x32x16 products and of the transposed 4 x 16 block read
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <int NQB, typename T>
__global__ __launch_bounds__(256, 2) void attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n_tok, int heads, int q_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // Workgroups are handed to the 8 XCDs round-robin.  Renumber them so that each XCD works through a contiguous range of
    // (frame, head, query block) triples: the query blocks of one (frame, head) then share that XCD's L2 copy of its K and V
    // instead of every XCD fetching its own (q, r: bijective for any workgroup count).
    const int nwg = gridDim.x, xcd = blockIdx.x & 7;
    const int per = nwg >> 3, rem = nwg & 7;
    const int wg = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
    const int fh = wg / q_blocks, qb = wg - fh * q_blocks;
    const int frame = fh / heads, head = fh - frame * heads;
    const int D = heads * 64, ld = 3 * D;
    const T* base = qkv + (size_t)frame * n_tok * ld + head * 64;
    const int q0 = qb * (128 * NQB) + wave * (32 * NQB);

    // Q fragments (B operand of S^T = K Q^T): lane (r, h) holds Q[q0 + r][16 ks + 8 h + 0..7]
    v8 qf[NQB][4];
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const T* qrow = base + (size_t)min(q0 + 32 * b + r, n_tok - 1) * ld + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *reinterpret_cast<const v8*>(qrow + 16 * ks);
    }

    // staging: per tile every thread copies 2 K pieces and 2 V pieces of 16 bytes straight from global memory into LDS
    // (global_load_lds: no staging registers, no LDS store instructions).  The LDS image is lane-linear, so the bank swizzles
    // of tile_off() / v_off() are applied to the SOURCE chunk each lane fetches: the lane that fills position sc of a row
    // fetches chunk sc ^ swizzle(row), which is what those functions' readers expect to find there.
    const int srow = tid >> 3, sc = tid & 7;
    const T* kbase = base + D;
    const T* vbase = base + 2 * D;
    auto stage_tile = [&](int kt, int buf) {
        char* kb = smem + buf * 16384;
        char* vb = kb + 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 32 + srow;
            const size_t g = (size_t)min(kt * 64 + row, n_tok - 1) * ld;
            glds16(kbase + g + ((sc ^ ((row >> 1) & 7)) << 3), kb + i * 4096 + wave * 1024);
            glds16(vbase + g + ((sc ^ (((row >> 1) & 1) << 2)) << 3), vb + i * 4096 + wave * 1024);
        }
    };

    f32x16 o[NQB][2], negm[NQB];
    float m_run[NQB], l_run[NQB];
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[b][0][i] = 0.f; o[b][1][i] = 0.f; }
        m_run[b] = 0.f;
        l_run[b] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[b][i] = 0.f;
    }

    const int nkt = (n_tok + 63) / 64;
    stage_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler also drains the copies in front of a barrier)
    __syncthreads();
    // The Q fragments come from plain global loads.  Redefine them here, once they have landed: otherwise the compiler carries
    // "these registers may still be in flight" around the loop and waits, before the first products of EVERY tile, with counts
    // that also cover the K/V loads it has just issued for the next tile (vector memory operations complete in order).
    // Worth 3 % of the kernel.
#pragma unroll
    for (int b = 0; b < NQB; ++b)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[b][ks]));

    // transposed-read lane roles inside a 16-lane group: lane 4q+p supplies row q, columns 4p..4p+3
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * tp;       // + 32 * db

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage_tile(kt + 1, cur ^ 1);      // lands while this tile is multiplied; read after the barrier below
        const char* kb = smem + cur * 16384;
        const char* vb = kb + 8192;

        // The products are accumulated onto -m_run (negm: sixteen copies of it, the C operand of the first product of each
        // chain), so s comes out as score MINUS running maximum and the exponentials below need no subtraction: one VALU
        // instruction per score less, in a loop whose pace is set by vector instruction issue, not by the matrix pipe.
        f32x16 s[NQB][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const v8 kf = *reinterpret_cast<const v8*>(kb + tile_off(32 * kbk + r, 2 * ks + h));
#pragma unroll
                for (int b = 0; b < NQB; ++b)
                    s[b][kbk] = mfma32(kf, qf[b][ks], ks == 0 ? negm[b] : s[b][kbk]);
            }
        }
        v8 pf[NQB][2][2];
#pragma unroll
        for (int b = 0; b < NQB; ++b) {
            if (kt == nkt - 1) {   // keys past the end of the sequence
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt * 64 + 32 * kbk + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key >= n_tok) s[b][kbk][i] = -INFINITY;
                    }
            }
            float mx = s[b][0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[b][0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[b][1][i]);
            // Everything accumulated so far (o, l_run) is at scale 2^-m_run.  Move to a new maximum only if some query of the
            // block needs it (wave-uniform branch; always on the first tile, which sets the maximum whatever its sign); this
            // tile's scores, already relative to the old maximum, move with it.  A query's keys are split over lanes r and
            // r + 32: the vote sees both halves, so the two only have to agree on the maximum inside the branch.
            if (kt == 0 || __builtin_amdgcn_ballot_w64(mx > DEFER) != 0) {
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
                const float alpha = kt == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
                m_run[b] += delta;
                l_run[b] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    o[b][0][i] *= alpha; o[b][1][i] *= alpha;
                    s[b][0][i] -= delta; s[b][1][i] -= delta;
                    negm[b][i] = -m_run[b];
                }
            }
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pv = __builtin_amdgcn_exp2f(s[b][kbk][i]);
                    ps[i & 3] += pv;
                    pf[b][kbk][i >> 3][i & 7] = (T)pv;
                }
            l_run[b] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }

#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int k0 = 32 * kbk + 16 * s2 + 4 * h + tq;
                    const v4 lo = read_tr<T>(vb + v_off(k0, 32 * db + tcol));
                    const v4 hi = read_tr<T>(vb + v_off(k0 + 8, 32 * db + tcol));
                    const v8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int b = 0; b < NQB; ++b) o[b][db] = mfma32(vf, pf[b][kbk][s2], o[b][db]);
                }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's copies of the next tile have landed
        __syncthreads();
    }

#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32);
        const float inv = 1.0f / l_tot;
        const int q = q0 + 32 * b + r;
        if (q < n_tok) {
            T* orow = out + ((size_t)frame * n_tok + q) * D + head * 64 + 4 * h;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4 v = {(T)(o[b][db][4 * g] * inv), (T)(o[b][db][4 * g + 1] * inv),
                                  (T)(o[b][db][4 * g + 2] * inv), (T)(o[b][db][4 * g + 3] * inv)};
                    *reinterpret_cast<v4*>(orow + 32 * db + 8 * g) = v;
                }
        }
    }
}

// fp32 storage mode: q, k, v and the output are fp32.  K and V tiles are split into IEEE-half hi/lo images while they are
// staged (registers -> LDS), q and the probabilities are split in registers, and every product is the three-term sum
// hi*hi + hi*lo + lo*hi on the fp16 matrix cores with fp32 accumulation.  Values are multiplied by powers of two before
// the split (q, k, v by QKV_SCALE, probabilities, which reach 2^DEFER = 64, by P_SCALE) so that the low halves are normal
// numbers, and the sums are scaled back.  Same structure as attn_kernel<1> otherwise.
constexpr float QKV_SCALE = 16.f, P_SCALE = 256.f;
__device__ __forceinline__ f16x4 read_tr_f16(const char* p) {     // transposed 4 x 16 block read of 16-bit patterns
    return read_tr<f16>(p);
}
__global__ __launch_bounds__(256, 2) void attn_split_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n_tok,
                                                            int heads, int q_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 32768];      // per buffer: K hi, K lo, V hi, V lo (8 KiB each)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7;
    const int per = nwg >> 3, rem = nwg & 7;
    const int wg = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
    const int fh = wg / q_blocks, qb = wg - fh * q_blocks;
    const int frame = fh / heads, head = fh - frame * heads;
    const int D = heads * 64, ld = 3 * D;
    const float* base = qkv + (size_t)frame * n_tok * ld + head * 64;
    const int q0 = qb * 128 + wave * 32;

    auto split8 = [](const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f16 x, y;
            split_f16(a[k] * QKV_SCALE, x, y); hi[k] = x; lo[k] = y;
            split_f16(b[k] * QKV_SCALE, x, y); hi[4 + k] = x; lo[4 + k] = y;
        }
    };

    f16x8 qhi[4], qlo[4];
    {
        const float* qrow = base + (size_t)min(q0 + r, n_tok - 1) * ld + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            split8(*reinterpret_cast<const f32x4*>(qrow + 16 * ks), *reinterpret_cast<const f32x4*>(qrow + 16 * ks + 4), qhi[ks], qlo[ks]);
    }

    const int srow = tid >> 3, sc = tid & 7;
    const float* kbase = base + D + sc * 8;
    const float* vbase = base + 2 * D + sc * 8;
    f32x4 kreg[2][2], vreg[2][2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t row = (size_t)min(kt * 64 + i * 32 + srow, n_tok - 1) * ld;
            kreg[i][0] = *reinterpret_cast<const f32x4*>(kbase + row);
            kreg[i][1] = *reinterpret_cast<const f32x4*>(kbase + row + 4);
            vreg[i][0] = *reinterpret_cast<const f32x4*>(vbase + row);
            vreg[i][1] = *reinterpret_cast<const f32x4*>(vbase + row + 4);
        }
    };
    auto write_tile = [&](int buf) {
        char* b0 = smem + buf * 32768;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 32 + srow;
            f16x8 hi, lo;
            split8(kreg[i][0], kreg[i][1], hi, lo);
            *reinterpret_cast<f16x8*>(b0 + tile_off(row, sc)) = hi;
            *reinterpret_cast<f16x8*>(b0 + 8192 + tile_off(row, sc)) = lo;
            split8(vreg[i][0], vreg[i][1], hi, lo);
            *reinterpret_cast<f16x8*>(b0 + 16384 + v_off(row, sc * 8)) = hi;
            *reinterpret_cast<f16x8*>(b0 + 24576 + v_off(row, sc * 8)) = lo;
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (n_tok + 63) / 64;
    load_tile(0);
    write_tile(0);
    __syncthreads();
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * tp;

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const char* kh = smem + cur * 32768;
        const char* kl = kh + 8192;
        const char* vh = kh + 16384;
        const char* vl = kh + 24576;

        f32x16 s[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kbk][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = tile_off(32 * kbk + r, 2 * ks + h);
                const f16x8 khi = *reinterpret_cast<const f16x8*>(kh + off);
                const f16x8 klo = *reinterpret_cast<const f16x8*>(kl + off);
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(klo, qhi[ks], s[kbk], 0, 0, 0);
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qlo[ks], s[kbk], 0, 0, 0);
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qhi[ks], s[kbk], 0, 0, 0);
            }
        }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) s[kbk] *= 1.0f / (QKV_SCALE * QKV_SCALE);
        if (kt == nkt - 1) {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 64 + 32 * kbk + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= n_tok) s[kbk][i] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
        if (__builtin_amdgcn_ballot_w64(mx - m_run > DEFER) != 0) {      // the vote sees both lanes of a query
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        }
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        f16x8 ph[2][2], pl[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = exp2f(s[kbk][i] - m_run);       // full-precision exp2 in this mode
                ps[i & 3] += pv;
                f16 x, y;
                split_f16(pv * P_SCALE, x, y);
                ph[kbk][i >> 3][i & 7] = x;
                pl[kbk][i >> 3][i & 7] = y;
            }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);

#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int k0 = 32 * kbk + 16 * s2 + 4 * h + tq;
                    const int o0 = v_off(k0, 32 * db + tcol), o1 = v_off(k0 + 8, 32 * db + tcol);
                    const f16x4 a0 = read_tr_f16(vh + o0);
                    const f16x4 a1 = read_tr_f16(vh + o1);
                    const f16x4 b0 = read_tr_f16(vl + o0);
                    const f16x4 b1 = read_tr_f16(vl + o1);
                    const f16x8 vhi = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const f16x8 vlo = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo, ph[kbk][s2], o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, pl[kbk][s2], o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, ph[kbk][s2], o[db], 0, 0, 0);
                }

        if (kt + 1 < nkt) write_tile(cur ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / (l_tot * (P_SCALE * QKV_SCALE));
    const int q = q0 + r;
    if (q < n_tok) {
        float* orow = out + ((size_t)frame * n_tok + q) * D + head * 64 + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(orow + 32 * db + 8 * g) = v;
            }
    }
}


// fp32 storage mode on operands that arrive split.  qkv2: per token the IEEE-half hi pieces of scale * [q | k | v] followed by
// the lo pieces (the QKV projection writes them, VdaGemmArgs.split_out = 2); out3: per token [hi | lo | hi] of scale * output, the
// operand of the output projection.  Structure of attn_kernel<1>: the four 8 KiB images of a 64-key tile (K hi, K lo, V hi,
// V lo) are copied global -> LDS by global_load_lds into a double buffer, the scores are accumulated onto minus the running
// maximum and every product is the three-term sum hi*hi + lo*hi + hi*lo.  Scores carry the factor scale^2 (both operands are
// scaled), so the running maximum is kept in those units and a probability is exp2(s / scale^2 + log2 P_SCALE): one fused
// multiply-add in front of the hardware exponential, whose result is already the scaled probability that is split.
// f16(a - h.lo) into the low half of a register / f16(a - h.hi) into the high half of `d` (the other half is kept): one
// mixed-precision multiply-add each (fp32 arithmetic on a half-precision operand read in place, one rounding), in place of a
// conversion to fp32, a subtraction, a conversion back and a pack.
__device__ __forceinline__ unsigned remainder_to_lo(float a, unsigned h) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(h));
    return d;
}
__device__ __forceinline__ void remainder_to_hi(unsigned& d, float a, unsigned h) {
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(a), "v"(h));
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 3) void attn_pairs_kernel(const f16* __restrict__ qkv, f16* __restrict__ out, int n_tok, int heads,
                                                            int q_blocks, float s2, int out_form) {
    // Tiles of 32 keys: per buffer K hi, K lo, V hi, V lo of 4 KiB each.  With 32 KiB per workgroup three workgroups share a CU
    // (the registers allow three waves per SIMD): a wave's score products, exponentials and output products depend on each
    // other in turn, so it is the other waves of the SIMD that keep the matrix pipe busy meanwhile.
    constexpr int TK = 32, PLANE = TK * 128, BUF = 4 * PLANE;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7;
    const int per = nwg >> 3, rem = nwg & 7;
    const int wg = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
    const int fh = wg / q_blocks, qb = wg - fh * q_blocks;
    const int frame = fh / heads, head = fh - frame * heads;
    const int D = heads * 64, lo_off = 3 * D, ld = 6 * D;
    const f16* base = qkv + (size_t)frame * n_tok * ld + head * 64;
    const int q0 = qb * 128 + wave * 32;
    const float inv_s2 = 1.0f / s2;
    constexpr float LOG2_P_SCALE = 8.f;                                // P_SCALE = 256

    f16x8 qhi[4], qlo[4];
    {
        const f16* qrow = base + (size_t)min(q0 + r, n_tok - 1) * ld + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qhi[ks] = *reinterpret_cast<const f16x8*>(qrow + 16 * ks);
            qlo[ks] = *reinterpret_cast<const f16x8*>(qrow + lo_off + 16 * ks);
        }
    }

    // staging: one 16-byte piece of each plane per thread and tile
    const int srow = tid >> 3, sc = tid & 7;
    const f16* kbase = base + D;
    const f16* vbase = base + 2 * D;
    const int kc = (sc ^ ((srow >> 1) & 7)) << 3, vc = (sc ^ (((srow >> 1) & 1) << 2)) << 3;
    auto stage_tile = [&](int kt, int buf) {
        char* b0 = smem + buf * BUF + wave * 1024;
        const size_t g = (size_t)min(kt * TK + srow, n_tok - 1) * ld;
        glds16(kbase + g + kc, b0);
        glds16(kbase + g + lo_off + kc, b0 + PLANE);
        glds16(vbase + g + vc, b0 + 2 * PLANE);
        glds16(vbase + g + lo_off + vc, b0 + 3 * PLANE);
    };

    f32x16 o[2], negm;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; negm[i] = 0.f; }
    float m_run = 0.f, l_run = 0.f;          // the running maximum in score units times scale^2; the sum of P_SCALE * p

    const int nkt = (n_tok + TK - 1) / TK;
    stage_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qhi[ks]), "+v"(qlo[ks]));      // landed: see attn_kernel

    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * tp;
    const float defer = DEFER * s2;

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage_tile(kt + 1, cur ^ 1);
        const char* kh = smem + cur * BUF;
        const char* kl = kh + PLANE;
        const char* vh = kh + 2 * PLANE;
        const char* vl = kh + 3 * PLANE;

        f32x16 sc_ = negm;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = tile_off(r, 2 * ks + h);
            const f16x8 khi = *reinterpret_cast<const f16x8*>(kh + off);
            const f16x8 klo = *reinterpret_cast<const f16x8*>(kl + off);
            sc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(klo, qhi[ks], sc_, 0, 0, 0);
            sc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qlo[ks], sc_, 0, 0, 0);
            sc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qhi[ks], sc_, 0, 0, 0);
        }
        if (kt == nkt - 1) {               // keys past the end of the sequence
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kt * TK + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key >= n_tok) sc_[i] = -INFINITY;
            }
        }
        float mx = sc_[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sc_[i]);
        if (kt == 0 || __builtin_amdgcn_ballot_w64(mx > defer) != 0) {
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
            const float alpha = kt == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta * inv_s2);
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o[0][i] *= alpha; o[1][i] *= alpha;
                sc_[i] -= delta;
                negm[i] = -m_run;
            }
        }
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        u32x4 ph[2], pl[2];                    // eight halves each: the B operands of the second product
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc_[i], inv_s2, LOG2_P_SCALE));     // P_SCALE * p <= 2^14
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc_[i + 1], inv_s2, LOG2_P_SCALE));
            ps[i & 3] += p0;
            ps[(i + 1) & 3] += p1;
            const f16x2 hi = {(f16)p0, (f16)p1};                     // one packed conversion
            const unsigned hh = __builtin_bit_cast(unsigned, hi);
            unsigned ll = remainder_to_lo(p0, hh);
            remainder_to_hi(ll, p1, hh);
            ph[i >> 3][(i & 7) >> 1] = hh;
            pl[i >> 3][(i & 7) >> 1] = ll;
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);

#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int s2i = 0; s2i < 2; ++s2i) {
                const int k0 = 16 * s2i + 4 * h + tq;
                const int o0 = v_off(k0, 32 * db + tcol), o1 = v_off(k0 + 8, 32 * db + tcol);
                const f16x4 a0 = read_tr_f16(vh + o0);
                const f16x4 a1 = read_tr_f16(vh + o1);
                const f16x4 b0 = read_tr_f16(vl + o0);
                const f16x4 b1 = read_tr_f16(vl + o1);
                const f16x8 vhi = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const f16x8 vlo = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                const f16x8 p_hi = __builtin_bit_cast(f16x8, ph[s2i]), p_lo = __builtin_bit_cast(f16x8, pl[s2i]);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo, p_hi, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, p_lo, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, p_hi, o[db], 0, 0, 0);
            }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's copies of the next tile have landed
        __syncthreads();
    }

    // o = (scale * v) (P_SCALE * p) summed; l_run = sum of P_SCALE * p: their ratio is scale * output, the value to split
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r;
    if (q < n_tok) {
        f16* orow = out + ((size_t)frame * n_tok + q) * pieces_width(out_form, D);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
                f16x4 hi, lo;
                split4<f16>(v, hi, lo);
                store_pieces<f16>(orow, head * 64 + 4 * h + 32 * db + 8 * g, D, out_form, hi, lo);
            }
    }
}

}  // namespace

extern "C" int vda_attention_pairs(const void* qkv2, void* out3, int frames, int n_tok, int heads, float scale, int out_layout,
                                   void* stream) {
    if (!qkv2 || !out3 || frames <= 0 || n_tok <= 0 || heads <= 0 || !(scale > 0.f)) return VDA_EINVAL;
    if (out_layout != VDA_PIECES_TRIPLE && out_layout != VDA_PIECES_PAIRS) return VDA_EINVAL;
    if ((((uintptr_t)qkv2 | (uintptr_t)out3) & 15) != 0) return VDA_ESHAPE;
    const int qb = (n_tok + 127) / 128;
    if ((long)frames * heads * qb > 0x7fffffffL) return VDA_ESHAPE;
    hipLaunchKernelGGL(attn_pairs_kernel, dim3(frames * heads * qb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const f16*>(qkv2), reinterpret_cast<f16*>(out3), n_tok, heads, qb, scale * scale, pieces_form(out_layout));
    return hip_rc(hipGetLastError());
}

extern "C" int vda_attention(const void* qkv, void* out, int frames, int n_tok, int heads, int dtype, void* stream) {
    if (!qkv || !out || frames <= 0 || n_tok <= 0 || heads <= 0) return VDA_EINVAL;
    if (dtype != VDA_T_BF16 && dtype != VDA_T_F16 && dtype != VDA_T_F32) return VDA_EINVAL;
    if (dtype == VDA_T_F32) {
        const int qb = (n_tok + 127) / 128;
        if ((long)frames * heads * qb > 0x7fffffffL) return VDA_ESHAPE;
        hipLaunchKernelGGL(attn_split_kernel, dim3(frames * heads * qb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const float*>(qkv), reinterpret_cast<float*>(out), n_tok, heads, qb);
        return hip_rc(hipGetLastError());
    }
    // 64 queries per wave only for long sequences (measured, 32 frames x 16 heads: 5477 tokens +3 %, 1370 tokens -3 %: there
    // the third wave per SIMD that the smaller register footprint allows is worth more than the halved LDS traffic)
    const int nqb = n_tok >= 4096 ? 2 : 1;
    const int q_blocks = (n_tok + 128 * nqb - 1) / (128 * nqb);
    if ((long)frames * heads * q_blocks > 0x7fffffffL) return VDA_ESHAPE;
    dim3 grid(frames * heads * q_blocks);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VDA_ATTN(NQB, T) hipLaunchKernelGGL((attn_kernel<NQB, T>), grid, dim3(256), 0, st, reinterpret_cast<const T*>(qkv), \
                                            reinterpret_cast<T*>(out), n_tok, heads, q_blocks)
    if (dtype == VDA_T_F16) {
        if (nqb == 2) VDA_ATTN(2, f16); else VDA_ATTN(1, f16);
    } else {
        if (nqb == 2) VDA_ATTN(2, bf16); else VDA_ATTN(1, bf16);
    }
#undef VDA_ATTN
    return hip_rc(hipGetLastError());
}
