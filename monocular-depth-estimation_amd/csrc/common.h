// Shared device helpers for the gfx950 kernels (wave = 64 lanes, bf16 MFMA with fp32 accumulation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vda_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

// 16-byte asynchronous global -> LDS copy; the LDS destination is (wave-uniform base) + lane * 16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

// Byte offset of 16-byte chunk `chunk` (0..7) of row `row` inside a [rows][64] bf16 LDS tile (128-byte rows).
// The XOR spreads the rows a ds_read_b128 lane group touches over all 16 slots of the 256-byte bank row, for both
// the 16-row (16x16x32) and the 32-row (32x32x16) MFMA operand fragments.
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Exact-form GELU x * Phi(x) with erf from the Abramowitz-Stegun 7.1.26 rational form (|erf error| <= 1.5e-7, far below
// the 2^-9 rounding of the 16-bit value it is stored as).  One v_rcp, one v_exp and ten FMA-class instructions, against
// the ~40 of libm's erff: the GELU epilogue of the 4D-wide MLP GEMM would otherwise cost as much as its K loop.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float h = 0.5f * poly * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // 0.5 * erfc(z)
    return x * (x >= 0.f ? 1.0f - h : h);
}

// The same function for values that are stored in 16 bits (the GEMM epilogues): Phi(x) = 1/2 + x Q(x^2) with a degree-8 fit of erf(x / sqrt 2) / x
// on |x| <= 4.2426 (beyond it Phi is 0 or 1 to 1.4e-5).  |Phi error| <= 1.3e-5, |GELU error| <= 5.4e-5 absolute, against a
// half-unit of 2e-3 |value| for bf16 and 2.4e-4 |value| for fp16; 12 plain instructions, no reciprocal or exponential
// (which cost four issue slots each), about half of gelu_erf.
__device__ __forceinline__ float gelu_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.2426405f, 4.2426405f);
    const float s = xc * xc;
    float q = fmaf(5.626769913e-11f, s, -5.371870531e-09f);
    q = fmaf(q, s, 2.268296662e-07f);
    q = fmaf(q, s, -5.646215693e-06f);
    q = fmaf(q, s, 9.359063109e-05f);
    q = fmaf(q, s, -1.109400329e-03f);
    q = fmaf(q, s, 9.818119211e-03f);
    q = fmaf(q, s, -6.634692377e-02f);
    q = fmaf(q, s, 3.989031273e-01f);
    return x * fmaf(q, xc, 0.5f);
}
// storage types: vectors of 4 / 8 elements per type (float: the fp32 storage mode)
template <typename T> struct Vec;
template <> struct Vec<bf16> { typedef bf16x4 v4; typedef bf16x8 v8; static constexpr bool is_bf16 = true; };
template <> struct Vec<f16> { typedef f16x4 v4; typedef f16x8 v8; static constexpr bool is_bf16 = false; };
template <> struct Vec<float> { typedef f32x4 v4; typedef f32x8 v8; static constexpr bool is_bf16 = false; };

// hi/lo split of an fp32 value into two bf16: v = hi + lo up to 2^-17 |v|.  Three bf16 MFMA products
// (hi*hi + lo*hi + hi*lo, fp32 accumulation) then reproduce an fp32 product to about 2^-16 relative.
__device__ __forceinline__ void split_bf16(float v, bf16& hi, bf16& lo) {
    hi = (bf16)v;
    lo = (bf16)(v - (float)hi);
}

// The same with two IEEE halves (11 significant bits each): v * scale = hi + lo up to 2^-22 |v * scale| while lo stays a
// normal half, i.e. |v * scale| >= 2^-3; below that the error is at most 2^-25 absolute (half the subnormal step).  The
// power-of-two `scale` moves the values of interest into that range; hi saturates at the largest finite half, so an
// out-of-range value degrades (lo takes up the excess) instead of becoming infinite.  hi*hi + lo*hi + hi*lo then
// reproduces an fp32 product to about 2^-21 relative.
__device__ __forceinline__ void split_f16(float v, f16& hi, f16& lo) {
    hi = (f16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    lo = (f16)__builtin_amdgcn_fmed3f(v - (float)hi, -65504.f, 65504.f);
}
__device__ __forceinline__ void split_any(float v, bf16& hi, bf16& lo) { split_bf16(v, hi, lo); }
__device__ __forceinline__ void split_any(float v, f16& hi, f16& lo) { split_f16(v, hi, lo); }

// Range guard of the IEEE-half pieces: a scaled value beyond the largest finite half saturates in split_f16; the caller's flag
// word is then set (a plain vector store of the same 1 from every lane that sees one).  `v` holds scaled values.
constexpr float F16_MAX = 65504.f;
__device__ __forceinline__ void note_saturation(int* flag, f32x4 v) {
    const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    if (flag && m > F16_MAX) *reinterpret_cast<volatile int*>(flag) = 1;
}
__device__ __forceinline__ void note_saturation_as(int* flag, f32x4 v, f16) { note_saturation(flag, v); }
__device__ __forceinline__ void note_saturation_as(int*, f32x4, bf16) {}          // bf16 pieces have fp32's range

// The [hi | lo | ...] pieces of four scaled values, as two 8-byte vectors.
template <typename T>
__device__ __forceinline__ void split4(f32x4 v, typename Vec<T>::v4& hi, typename Vec<T>::v4& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T h, l;
        split_any(v[k], h, l);
        hi[k] = h;
        lo[k] = l;
    }
}

// Where the pieces of an operand row go.  `form`: 2 = two planes [hi | lo] of Cw columns, 3 = VDA_PIECES_TRIPLE [hi | lo | hi],
// 4 = VDA_PIECES_PAIRS (every 32 columns stored as [hi 32 | lo 32]).  c = first of four consecutive operand columns, c % 4 == 0.
constexpr int PIECES_PLANES = 2, PIECES_TRIPLE = 3, PIECES_PAIRS = 4;
__host__ __device__ __forceinline__ int pieces_form(int layout) { return layout == VDA_PIECES_PAIRS ? PIECES_PAIRS : PIECES_TRIPLE; }
__host__ __device__ __forceinline__ int pieces_width(int form, int Cw) { return form == PIECES_TRIPLE ? 3 * Cw : 2 * Cw; }
template <typename T, bool NT = false>
__device__ __forceinline__ void store_pieces(T* row, int c, int Cw, int form, typename Vec<T>::v4 hi, typename Vec<T>::v4 lo) {
    typedef typename Vec<T>::v4 v4;
    T* p0 = form == PIECES_PAIRS ? row + ((c >> 5) << 6) + (c & 31) : row + c;
    T* p1 = form == PIECES_PAIRS ? p0 + 32 : p0 + Cw;
    if (NT) {
        __builtin_nontemporal_store(hi, reinterpret_cast<v4*>(p0));
        __builtin_nontemporal_store(lo, reinterpret_cast<v4*>(p1));
        if (form == PIECES_TRIPLE) __builtin_nontemporal_store(hi, reinterpret_cast<v4*>(p0 + 2 * Cw));
    } else {
        *reinterpret_cast<v4*>(p0) = hi;
        *reinterpret_cast<v4*>(p1) = lo;
        if (form == PIECES_TRIPLE) *reinterpret_cast<v4*>(p0 + 2 * Cw) = hi;
    }
}

// ---- transposed operand reads of a row-major [key][64 values] tile of 16-bit values in LDS (the V operand of attention) ----------
// byte offset of element (key row, 16-bit column col) in the tile: 128-byte rows, the two 64-byte halves of a row swapped on
// rows 2,3 (mod 4) so the 4 rows of a transposed 4x16 block read fall on distinct banks
__device__ __forceinline__ int v_off(int row, int col) { return row * 128 + ((col * 2) ^ (((row >> 1) & 1) << 6)); }
// ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3; every lane must be active
template <typename T> __device__ __forceinline__ typename Vec<T>::v4 read_tr(const char* p);
template <> __device__ __forceinline__ bf16x4 read_tr<bf16>(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(p));
}
typedef short i16x4 __attribute__((ext_vector_type(4)));
template <> __device__ __forceinline__ f16x4 read_tr<f16>(const char* p) {      // (the builtin's own half type is not _Float16)
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(p)));
}

// Workgroup -> work item for grids whose neighbouring items read the same data (image rows, patches).  Consecutive workgroup
// ids go to different XCDs (id mod 8), each with an L2 of its own, so neighbours dealt in id order fetch their shared
// lines once per XCD.  This permutation of [0, n) hands XCD x a contiguous run of items, in the order it receives its
// workgroups (bijective for any n).
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
