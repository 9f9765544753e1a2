// Pieces shared by the GEMM kernels: operand-type dispatch of the MFMA, typed vector loads/stores, the workgroup -> tile
// map and the fused epilogue (see vda_gemm in include/vda_hip.h).
#pragma once
#include "common.h"

namespace vda {

constexpr int BK = 64;   // K step of every GEMM kernel (one LDS row = 64 16-bit values = 128 bytes)

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

__device__ __forceinline__ float sat_f16(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
__device__ __forceinline__ float sat_as(float v, f16) { return sat_f16(v); }
__device__ __forceinline__ float sat_as(float v, bf16) { return v; }

__device__ __forceinline__ void store4(void* out, int out_type, size_t idx, f32x4 v) {
    if (out_type == VDA_T_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + idx) = v;
    } else if (out_type == VDA_T_BF16) {
        bf16x4 b = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(out) + idx) = b;
    } else {
        // IEEE half ends at 65504: a larger sum saturates instead of becoming infinite (one more instruction per value; bf16
        // has fp32's range and needs none)
        f16x4 b = {(f16)sat_f16(v[0]), (f16)sat_f16(v[1]), (f16)sat_f16(v[2]), (f16)sat_f16(v[3])};
        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(out) + idx) = b;
    }
}

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), every lane gets the total.  The four steps pair lane c with
// c^1, c^2 (quad permutes), c^7 (mirror within 8) and c^15 (mirror within 16); after each step the partial sums are uniform
// over the group just completed, so the result is the balanced tree ((0+1)+(2+3))+((4+5)+(6+7)) ... over the lane index.
// Every step is an XOR of the lane index, so if the lanes hold the 16 addends in an XOR-permuted order (addend k on lane
// k ^ r) the same pairs are added in the same tree: the sum does not depend on r, bit for bit.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    return v;
}

// Producer side of a folded LayerNorm (VdaGemmArgs.out16 / row_part).  `v` = four consecutive fp32 output values of a row;
// the 16 lanes of this lane's DPP row hold one 64-column block of that row between them (in any XOR-permuted order).
// Stores the values rounded to T at byte offset off16 of out16 and, from the lane `writer`, the block's (sum, sum of squares)
// of the rounded values at byte offset offp of row_part.  The arithmetic is spelled out operation by operation (no contraction) so that every call site adds the
// same numbers in the same order.
// (vda_gemm checks that both arrays are below 4 GiB, so the byte offsets fit 32 bits and the stores take the
// scalar-base + 32-bit-offset form: no 64-bit address registers beside the 128 accumulators.)
#define VDA_GLOBAL_PTR(T, p) ((T __attribute__((address_space(1)))*)(p))
template <typename T>
__device__ __forceinline__ void emit_rounded_row(void* out16, float* row_part, f32x4 v, unsigned off16, unsigned offp, bool in_range, bool writer) {
#pragma clang fp contract(off)
    const typename Vec<T>::v4 h = {(T)sat_as(v[0], T{}), (T)sat_as(v[1], T{}), (T)sat_as(v[2], T{}), (T)sat_as(v[3], T{})};
    const float r0 = (float)h[0], r1 = (float)h[1], r2 = (float)h[2], r3 = (float)h[3];
    float s = (r0 + r1) + (r2 + r3);
    float q = __builtin_fmaf(r1, r1, r0 * r0) + __builtin_fmaf(r3, r3, r2 * r2);
    s = row16_sum(s);
    q = row16_sum(q);
    if (in_range) {
        *VDA_GLOBAL_PTR(typename Vec<T>::v4, reinterpret_cast<char*>(out16) + off16) = h;
        if (writer) {
            const f32x2 o = {s, q};
            *VDA_GLOBAL_PTR(f32x2, reinterpret_cast<char*>(row_part) + offp) = o;
        }
    }
}

// A field of the kernel's first argument (a VdaGemmArgs passed by value: it starts the kernel-argument segment), loaded
// where this is called rather than at kernel entry.  The compiler loads every argument field it sees at entry and keeps it
// for the whole kernel; fields that only an epilogue needs would take scalar registers from the K loop (and, once those
// run out, end up in vector registers and scratch).  The volatile load pins the scalar load to the call site.
template <typename F>
__device__ __forceinline__ F kernarg_field_now(size_t offset) {
    typedef const volatile F __attribute__((address_space(4))) * src_t;
    const char __attribute__((address_space(4)))* base = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    return *(src_t)(base + offset);
}

// Folded LayerNorm, consumer side: (acc - mean * colsum) * rstd + bias on four values.  The two row scalars are first copied
// into register pairs of their own, behind an empty asm the compiler cannot see through, so that the packed multiply-adds
// take whole pairs.  Left alone the compiler reads mean and rstd straight out of the loaded (mean, rstd) pair with op_sel,
// and a packed fp32 instruction whose LOW result reads the HIGH register of a pair returns wrong values in lanes 48..63
// on gfx950 while other waves issue MFMAs on the same SIMD: the product came out as 0, i.e. the output was
// the bare bias, in one column of a 16-row block here and there, and only when two streams overlapped.
struct LnPairs {
    f32x2 nmean, rstd;
};
__device__ __forceinline__ LnPairs ln_pairs(f32x2 st) {
    LnPairs q{f32x2{-st[0], -st[0]}, f32x2{st[1], st[1]}};
    asm volatile("" : "+v"(q.nmean), "+v"(q.rstd));
    return q;
}
__device__ __forceinline__ f32x4 ln_fold4(const LnPairs& q, f32x4 acc, f32x4 csum, f32x4 bias) {
    const f32x2 lo = __builtin_elementwise_fma(__builtin_elementwise_fma(q.nmean, f32x2{csum[0], csum[1]}, f32x2{acc[0], acc[1]}),
                                               q.rstd, f32x2{bias[0], bias[1]});
    const f32x2 hi = __builtin_elementwise_fma(__builtin_elementwise_fma(q.nmean, f32x2{csum[2], csum[3]}, f32x2{acc[2], acc[3]}),
                                               q.rstd, f32x2{bias[2], bias[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// Consumer side: (mean, rstd) of this lane's row of each m-tile, fetched before the K loop so that the epilogue does not
// wait for them.
template <int TM>
__device__ __forceinline__ void rowstat_prefetch(const VdaGemmArgs& p, f32x2 (&rs)[TM], int mrow0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mrow0 + i * 16;
        rs[i] = m < p.M ? reinterpret_cast<const f32x2*>(p.row_stat)[m] : f32x2{0.f, 0.f};
    }
}

// Workgroup -> output tile.  Workgroups b and b + 8 share an XCD, so workgroup ids are first remapped so that each XCD
// gets a contiguous run of tiles (bijective for any grid size), then tiles are walked in bands of 8 row-tiles x all
// column-tiles so that the tiles resident together in one XCD share A and W panels in its L2.
__device__ __forceinline__ void tile_of_workgroup(const GemmTiles& gt, int& tm, int& tn) {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    constexpr int GM = 8;
    const int per_band = GM * gt.tiles_n;
    const int band = tile / per_band;
    const int first = band * GM;
    const int rows = min(GM, gt.tiles_m - first);
    const int in_band = tile - band * per_band;
    tm = first + in_band % rows;
    tn = in_band / rows;
}

// Fused epilogue.  acc[i][j] is the 16x16 MFMA tile (i along M, j along N) of a wave, computed as mfma(W, A), so this
// lane holds row  m = mrow0 + 16 i  and the 4 consecutive columns  n = ncol0 + 16 j + {0,1,2,3}, where
// mrow0 = (wave's first row) + (lane & 15) and ncol0 = (wave's first column) + 4 (lane >> 4).
//
// `stage` is a wave-private LDS area of at least STAGE_BYTES (all main-loop LDS traffic of the workgroup must have
// finished).  For plain and GEGLU epilogues whose output rows are 16-byte aligned the 16x(16 TN) block of one m-tile is
// first written to `stage` in the output type and then copied out with 16 bytes per lane, so every global store
// instruction writes whole, contiguous row segments (complete 128-byte lines for 64-column bf16 / 32-column fp32 blocks)
// instead of 8-byte pieces scattered over 16 rows.
constexpr int STAGE_BYTES = 16 * (4 * 16 * 4 + 16);     // 16 rows x (up to 64 fp32 columns + 16 bytes of padding)

// RS_PRE: rs_pre[i] holds (mean, rstd) of this lane's row of m-tile i (rowstat_prefetch).  A compile-time flag, not a null
// test of the pointer: a pointer to a private array cannot be compared with null at compile time on this target, and the
// comparison alone would keep the array in scratch memory.
template <int TM, int TN, bool RS_PRE = false>
__device__ __forceinline__ void gemm_epilogue_staged(const VdaGemmArgs& p, f32x4 (&acc)[TM][TN], int mrow0, int ncol0, int fq,
                                                     char* stage, const f32x4 (*pre)[TN] = nullptr, const f32x2* rs_pre = nullptr) {
    const int lane = threadIdx.x & 63, frow = lane & 15;
    // split output (VdaGemmArgs.split_out): the block is staged in fp32 and each 16-byte piece is split while it is copied out
    const bool split = p.split_out != 0;
    const int st_type = split ? VDA_T_F32 : p.out_type;       // element type inside the staging area
    const int esz = st_type == VDA_T_F32 ? 4 : 2;
    const bool geglu = p.epi == VDA_EPI_GEGLU;
    const int cols = geglu ? TN * 8 : TN * 16;               // output columns of the wave's block
    const int row_bytes = cols * esz;
    const int rs = row_bytes + 16;                            // padded LDS row stride
    const int cpr = row_bytes >> 4;                           // 16-byte chunks per row
    const int rpi = 64 / cpr;                                 // rows per copy-out instruction
    const int wave_n0 = ncol0 - fq * 4;
    const int out_n0 = geglu ? wave_n0 / 2 : wave_n0;        // first output column of the block
    const int n_out = geglu ? p.N / 2 : p.N;
    // transposed convolution: the block's columns are channels co0.. of one kernel tap (ky, kx), because ct_c is a
    // multiple of the block width; row m = (iy, x) goes to output pixel (iy ct_k + ky, x ct_k + kx)
    const bool convt = p.epi == VDA_EPI_CONVT;
    int tap_n0 = 0, ct_ky = 0, ct_kx = 0;
    if (convt) {
        const int tap = wave_n0 / p.ct_c;
        tap_n0 = tap * p.ct_c;
        ct_ky = tap / p.ct_k;
        ct_kx = tap - ct_ky * p.ct_k;
    }
    f32x4 bias[TN], gam[TN], csum[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 16;
        bias[j] = (p.bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n - tap_n0) : f32x4{0.f, 0.f, 0.f, 0.f};
        gam[j] = (p.gamma && n < p.N) ? *reinterpret_cast<const f32x4*>(p.gamma + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        csum[j] = (p.row_stat && n < p.N) ? *reinterpret_cast<const f32x4*>(p.w_colsum + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int c_row = lane / cpr, c_chunk = lane - c_row * cpr;
    // producer-side fields of a folded LayerNorm, fetched from the kernel arguments here (see kernarg_field_now; every
    // kernel that reaches this epilogue takes the VdaGemmArgs as its first argument)
    void* out16 = nullptr;
    float* row_part = nullptr;
    unsigned ld16 = 0;
    if (p.out16) {
        out16 = kernarg_field_now<void*>(offsetof(VdaGemmArgs, out16));
        row_part = kernarg_field_now<float*>(offsetof(VdaGemmArgs, row_part));
        ld16 = (unsigned)kernarg_field_now<int>(offsetof(VdaGemmArgs, ld16));
    }
    float split_scale = 1.f;
    int* sat_flag = nullptr;
    if (split) {
        split_scale = kernarg_field_now<float>(offsetof(VdaGemmArgs, split_scale));
        sat_flag = kernarg_field_now<int*>(offsetof(VdaGemmArgs, sat_flag));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mrow0 + i * 16;                          // this lane's row of the m-tile
        char* srow = stage + frow * rs;
        f32x2 st = {0.f, 1.f};                                  // folded LayerNorm: (mean, rstd) of row m of A
        if (p.row_stat) {
            if constexpr (RS_PRE) st = rs_pre[i];
            else st = m < p.M ? reinterpret_cast<const f32x2*>(p.row_stat)[m] : f32x2{0.f, 0.f};
        }
        LnPairs ln{};
        if (p.row_stat) ln = ln_pairs(st);
        if (geglu) {
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j + 1][r] + bias[j + 1][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_poly(v[r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= acc[i][j][r] + bias[j][r];
                store4(srow, st_type, (j / 2) * 16 + fq * 4, v);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = ncol0 + j * 16;
                f32x4 v;
                if (p.row_stat) {       // two fused multiply-adds per value, as vectors (they pair up into packed instructions)
                    v = ln_fold4(ln, acc[i][j], csum[j], bias[j]);
                } else {
                    v = acc[i][j] + bias[j];
                }
                if (p.act == VDA_ACT_GELU) {
                    // (split output: the value keeps fp32 precision, so the erf form good to 1.5e-7, not the 16-bit fit)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = split ? gelu_erf(v[r]) : gelu_poly(v[r]);
                } else if (p.act == VDA_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                v *= gam[j];
                if (pre) {                                  // the residual was fetched before the K loop (see residual_prefetch)
                    v += pre[i][j];
                } else if (m < p.M && n < p.N) {
                    if (p.resid_type != VDA_T_NONE) v += load4(p.resid, p.resid_type, (size_t)m * p.ldr + n);
                    if (p.resid2) v += load4(p.resid2, p.resid_type, (size_t)m * p.ldr2 + n);
                }
                store4(srow, st_type, j * 16 + fq * 4, v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's LDS writes are done before any of its lanes reads
        const int mbase = mrow0 - frow + i * 16;                // first row of the m-tile
        for (int r0 = 0; r0 < 16; r0 += rpi) {
            const int row = r0 + c_row;
            if (row < 16) {
                const int mm = mbase + row;
                const int col = (c_chunk << 4) / esz;          // first output column of this lane's chunk
                const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * rs + (c_chunk << 4));
                if (split) {
                    // four fp32 values -> their scaled 16-bit pieces, stored at the same columns of each segment of the row
                    if (mm < p.M && out_n0 + col < n_out) {
                        const f32x4 f = __builtin_bit_cast(f32x4, v) * split_scale;
                        if (p.ab_type == VDA_T_F16) {
                            note_saturation(sat_flag, f);
                            f16x4 hi, lo;
                            split4<f16>(f, hi, lo);
                            store_pieces<f16, true>(reinterpret_cast<f16*>(p.out) + (size_t)mm * p.ldc, out_n0 + col, p.N, p.split_out, hi, lo);
                        } else {
                            bf16x4 hi, lo;
                            split4<bf16>(f, hi, lo);
                            store_pieces<bf16, true>(reinterpret_cast<bf16*>(p.out) + (size_t)mm * p.ldc, out_n0 + col, p.N, p.split_out, hi, lo);
                        }
                    }
                    continue;
                }
                if (mm < p.M && out_n0 + col < n_out) {
                    size_t pix = mm;
                    if (convt) {
                        const int iy = mm / p.ct_w, x = mm - iy * p.ct_w;     // iy = img * ct_h + y
                        pix = ((size_t)iy * p.ct_k + ct_ky) * (p.ct_w * p.ct_k) + (size_t)x * p.ct_k + ct_kx;
                    }
                    // streaming hint: the tile's lines are not read again by this kernel and would only push the operand panels
                    // other workgroups are still reading out of L2 (+0.6 % on the ViT-L window, neutral on ViT-S)
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.out) + (pix * p.ldc + out_n0 - tap_n0 + col) * esz));
                }
                if (p.out16) {
                    // folded LayerNorm, producer side (64 fp32 columns per wave: the 16 lanes of a DPP row hold one row's block)
                    const bool ok = mm < p.M && out_n0 + col < n_out;
                    const f32x4 f = __builtin_bit_cast(f32x4, v);
                    const unsigned off16 = ((unsigned)mm * ld16 + (unsigned)(out_n0 + col)) * 2u;
                    const unsigned offp = ((unsigned)mm * (unsigned)(p.N >> 6) + (unsigned)(out_n0 >> 6)) * 8u;
                    if (p.ab_type == VDA_T_F16) emit_rounded_row<f16>(out16, row_part, f, off16, offp, ok, c_chunk == 0);
                    else emit_rounded_row<bf16>(out16, row_part, f, off16, offp, ok, c_chunk == 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads returned before the next m-tile overwrites the area
    }
}

template <int TM, int TN, bool RS_PRE = false>
__device__ __forceinline__ void gemm_epilogue(const VdaGemmArgs& p, f32x4 (&acc)[TM][TN], int mrow0, int ncol0, int fq,
                                              char* stage = nullptr, const f32x4 (*pre)[TN] = nullptr, const f32x2* rs_pre = nullptr) {
    if (stage && (p.epi == VDA_EPI_PLAIN || p.epi == VDA_EPI_GEGLU || p.epi == VDA_EPI_CONVT)) {
        const int esz = (p.out_type == VDA_T_F32 || p.split_out) ? 4 : 2;
        const int n_out = p.epi == VDA_EPI_GEGLU ? p.N / 2 : p.N;
        const int cols = p.epi == VDA_EPI_GEGLU ? TN * 8 : TN * 16;
        // 16-byte row alignment of the output, whole chunks inside N, and at most 64 chunks per 16-row block row set;
        // for the transposed convolution a block must not straddle two kernel taps
        if ((p.ldc * esz) % 16 == 0 && (n_out * esz) % 16 == 0 && (cols * esz) % 16 == 0 && (cols * esz) <= 256 &&
            ((uintptr_t)p.out & 15) == 0 && (p.epi != VDA_EPI_CONVT || p.ct_c % cols == 0)) {
            gemm_epilogue_staged<TM, TN, RS_PRE>(p, acc, mrow0, ncol0, fq, stage, pre, rs_pre);
            return;
        }
    }
    if (p.split_out) return;      // vda_gemm only accepts split outputs in shapes that take the staged form above
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
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[i] += fmaxf(acc[i][j][r] + b[r], 0.f) * dw[r];
                    asm volatile("" : "+v"(part[i]));      // one row's sum at a time: no packing of rows with op_sel (see ln_pairs)
                }
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
        const int half0 = (ncol0 - fq * 4) / 2 + fq * 4;
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
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j + 1][r] + bg[r];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_poly(v[r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= acc[i][j][r] + bv[r];
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
        f32x4 b = {0.f, 0.f, 0.f, 0.f}, g = {1.f, 1.f, 1.f, 1.f}, cs = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b = *reinterpret_cast<const f32x4*>(p.bias + bias_n);
        if (p.gamma) g = *reinterpret_cast<const f32x4*>(p.gamma + n);
        if (p.row_stat) cs = *reinterpret_cast<const f32x4*>(p.w_colsum + n);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mrow0 + i * 16;
            if (m >= p.M) continue;
            f32x4 v;
            if (p.row_stat) {          // folded LayerNorm: the same two fused multiply-adds as the staged form
                f32x2 st;
                if constexpr (RS_PRE) st = rs_pre[i];
                else st = reinterpret_cast<const f32x2*>(p.row_stat)[m];
                v = ln_fold4(ln_pairs(st), acc[i][j], cs, b);
            } else {
                v = acc[i][j] + b;
            }
            if (p.act == VDA_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_poly(v[r]);
            } else if (p.act == VDA_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v *= g;
            if (pre) {
                v += pre[i][j];
            } else {
                if (p.resid_type != VDA_T_NONE) v += load4(p.resid, p.resid_type, (size_t)m * p.ldr + n);
                if (p.resid2) v += load4(p.resid2, p.resid_type, (size_t)m * p.ldr2 + n);
            }
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

// Epilogue specialisation.  The epilogue above decides everything at run time (output and residual types, activation,
// LayerScale, epilogue kind); inlined into a kernel whose 128 accumulator registers leave little room, the code it does
// not execute still costs (the encoder's 256x256 GEMMs ran 5 % slower with the general form).  A kernel instantiated with
// SPEC > 0 calls the same function on a copy of its arguments whose choices are compile-time constants, so the compiler
// drops every other branch.  "own" = the 16-bit type of the operands.
enum {
    EPI_SPEC_GENERAL = 0,
    EPI_SPEC_OWN = 1,            // plain, own-type output, bias
    EPI_SPEC_OWN_GELU = 2,       // ... with GELU
    EPI_SPEC_F32_RES32 = 3,      // plain, fp32 output, bias + fp32 residual
    EPI_SPEC_OWN_RES = 4,        // plain, own-type output, bias + own-type residual
    EPI_SPEC_OWN_RES32 = 5,      // plain, own-type output, bias + fp32 residual
    EPI_SPEC_GEGLU = 6,          // gated GELU, own-type output
    EPI_SPEC_F32 = 7,            // plain, fp32 output, bias
    EPI_SPEC_OWN_RELU = 8,       // plain, own-type output, bias, ReLU
    EPI_SPEC_OWN_RES2 = 9,       // plain, own-type output, bias + two own-type residuals
    EPI_SPEC_DOT = 10,           // ReLU, dot with a vector, ReLU: fp32 output (the direct convolution's tail form)
    EPI_SPEC_CONVT = 11,         // transposed-convolution scatter, own-type output, bias
    EPI_SPEC_OWN_LN = 12,        // EPI_SPEC_OWN on rows whose LayerNorm is folded in (row_stat / w_colsum)
    EPI_SPEC_OWN_GELU_LN = 13,   // EPI_SPEC_OWN_GELU, likewise
    EPI_SPEC_F32_RES32_EMIT = 14,// EPI_SPEC_F32_RES32 that also emits the rounded rows and their partial sums (out16 / row_part)
    EPI_SPEC_SPLIT = 15,         // plain, bias, output split into 16-bit pieces (split_out)
    EPI_SPEC_SPLIT_GELU = 16     // ... after the exact GELU
};

// the specialisation that describes this call exactly, or EPI_SPEC_GENERAL
inline int epi_spec_of(const VdaGemmArgs& a) {
    const int own = a.ab_type == VDA_T_F16 ? VDA_T_F16 : VDA_T_BF16;
    if (a.gamma) return EPI_SPEC_GENERAL;
    if (a.split_out)
        return a.resid_type != VDA_T_NONE || a.row_stat || a.out16 ? EPI_SPEC_GENERAL
               : a.act == VDA_ACT_NONE ? EPI_SPEC_SPLIT : a.act == VDA_ACT_GELU ? EPI_SPEC_SPLIT_GELU : EPI_SPEC_GENERAL;
    if (a.row_stat)
        return a.out_type == own && a.resid_type == VDA_T_NONE && !a.out16
                   ? (a.act == VDA_ACT_NONE ? EPI_SPEC_OWN_LN : a.act == VDA_ACT_GELU ? EPI_SPEC_OWN_GELU_LN : EPI_SPEC_GENERAL)
                   : EPI_SPEC_GENERAL;
    if (a.out16)
        return a.out_type == VDA_T_F32 && a.resid_type == VDA_T_F32 && a.act == VDA_ACT_NONE && !a.resid2 ? EPI_SPEC_F32_RES32_EMIT
                                                                                                       : EPI_SPEC_GENERAL;
    if (a.epi == VDA_EPI_GEGLU) return a.out_type == own ? EPI_SPEC_GEGLU : EPI_SPEC_GENERAL;
    if (a.epi == VDA_EPI_DOT) return EPI_SPEC_DOT;
    if (a.epi == VDA_EPI_CONVT) return a.out_type == own && a.act == VDA_ACT_NONE ? EPI_SPEC_CONVT : EPI_SPEC_GENERAL;
    if (a.epi != VDA_EPI_PLAIN) return EPI_SPEC_GENERAL;
    if (a.resid2) return a.out_type == own && a.resid_type == own && a.act == VDA_ACT_NONE ? EPI_SPEC_OWN_RES2 : EPI_SPEC_GENERAL;
    if (a.out_type == own) {
        if (a.resid_type == VDA_T_NONE)
            return a.act == VDA_ACT_NONE ? EPI_SPEC_OWN : a.act == VDA_ACT_GELU ? EPI_SPEC_OWN_GELU : EPI_SPEC_OWN_RELU;
        if (a.act != VDA_ACT_NONE) return EPI_SPEC_GENERAL;
        return a.resid_type == own ? EPI_SPEC_OWN_RES : a.resid_type == VDA_T_F32 ? EPI_SPEC_OWN_RES32 : EPI_SPEC_GENERAL;
    }
    if (a.out_type == VDA_T_F32 && a.act == VDA_ACT_NONE) {
        if (a.resid_type == VDA_T_NONE) return EPI_SPEC_F32;
        if (a.resid_type == VDA_T_F32) return EPI_SPEC_F32_RES32;
    }
    return EPI_SPEC_GENERAL;
}

// the arguments with the choices of SPEC written in as constants
template <int SPEC, typename T>
__device__ __forceinline__ VdaGemmArgs epi_spec_args(const VdaGemmArgs& p) {
    if (SPEC == EPI_SPEC_GENERAL) return p;
    constexpr int own = Vec<T>::is_bf16 ? VDA_T_BF16 : VDA_T_F16;
    VdaGemmArgs q = p;
    q.gamma = nullptr;
    q.epi = SPEC == EPI_SPEC_GEGLU ? VDA_EPI_GEGLU : SPEC == EPI_SPEC_DOT ? VDA_EPI_DOT : SPEC == EPI_SPEC_CONVT ? VDA_EPI_CONVT : VDA_EPI_PLAIN;
    q.act = (SPEC == EPI_SPEC_OWN_GELU || SPEC == EPI_SPEC_OWN_GELU_LN || SPEC == EPI_SPEC_SPLIT_GELU) ? VDA_ACT_GELU
            : SPEC == EPI_SPEC_OWN_RELU ? VDA_ACT_RELU : VDA_ACT_NONE;
    if (SPEC != EPI_SPEC_SPLIT && SPEC != EPI_SPEC_SPLIT_GELU) q.split_out = 0;
    else __builtin_assume(q.split_out != 0);
    constexpr bool res32 = SPEC == EPI_SPEC_F32_RES32 || SPEC == EPI_SPEC_F32_RES32_EMIT;
    q.out_type = (res32 || SPEC == EPI_SPEC_F32 || SPEC == EPI_SPEC_DOT) ? VDA_T_F32 : own;
    q.ab_type = own;
    if (SPEC != EPI_SPEC_OWN_LN && SPEC != EPI_SPEC_OWN_GELU_LN) q.row_stat = q.w_colsum = nullptr;
    else __builtin_assume(q.row_stat != nullptr);
    if (SPEC != EPI_SPEC_F32_RES32_EMIT) { q.out16 = nullptr; q.row_part = nullptr; }
    else __builtin_assume(q.out16 != nullptr);
    q.resid_type = (res32 || SPEC == EPI_SPEC_OWN_RES32) ? VDA_T_F32
                   : (SPEC == EPI_SPEC_OWN_RES || SPEC == EPI_SPEC_OWN_RES2) ? own : VDA_T_NONE;
    if (q.resid_type == VDA_T_NONE) q.resid = nullptr;
    if (SPEC != EPI_SPEC_OWN_RES2) q.resid2 = nullptr;
    return q;
}

// operands that were multiplied by powers of two before they were rounded (the fp32 mode's hi/lo halves): undo it on the sums
template <int TM, int TN>
__device__ __forceinline__ void scale_acc(const VdaGemmArgs& p, f32x4 (&acc)[TM][TN]) {
    if (p.acc_scale != 0.f && p.acc_scale != 1.f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= p.acc_scale;
    }
}

template <int SPEC, typename T, int TM, int TN, bool RS_PRE = false>
__device__ __forceinline__ void gemm_epilogue_spec(const VdaGemmArgs& p, f32x4 (&acc)[TM][TN], int mrow0, int ncol0, int fq, char* stage,
                                                   const f32x4 (*pre)[TN] = nullptr, const f32x2* rs_pre = nullptr) {
    scale_acc<TM, TN>(p, acc);
    const VdaGemmArgs q = epi_spec_args<SPEC, T>(p);
    gemm_epilogue<TM, TN, RS_PRE>(q, acc, mrow0, ncol0, fq, stage, pre, rs_pre);
}

// The fp32 residual of a wave's block (this lane's four columns of each 16x16 tile), fetched into registers BEFORE the K
// loop of a kernel whose accumulators leave the room: the epilogue otherwise fetches it one 16-row block at a time, each
// fetch a full memory latency that nothing overlaps.  Elements outside M x N are zero.  (The residual may alias the output:
// every element is read and later written by the same lane.)
template <int TM, int TN>
__device__ __forceinline__ void residual_prefetch(const VdaGemmArgs& p, f32x4 (&pre)[TM][TN], int mrow0, int ncol0) {
    const float* r = reinterpret_cast<const float*>(p.resid);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int m = mrow0 + i * 16, n = ncol0 + j * 16;
            pre[i][j] = (m < p.M && n < p.N) ? *reinterpret_cast<const f32x4*>(r + (size_t)m * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
}

// 256x256-tile kernel with the 4-phase software pipeline (gemm256.hip)
int launch_gemm256(const VdaGemmArgs& a, hipStream_t s);
bool conv_direct_supports(const VdaGemmArgs& a);              // conv_direct.hip: direct 3x3 convolution, N = 32 / 64 / 128
int launch_conv_direct(const VdaGemmArgs& a, hipStream_t s);

}  // namespace vda
