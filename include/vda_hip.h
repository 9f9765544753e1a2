/*
 * libvda_hip.so -- C ABI of the MI355X (gfx950) kernels behind the Video Depth Anything per-window forward pass.
 *
 * The reference (SidPam/Monocular-Depth-Estimation) has no native interface for this path: every operation below is
 * a torch ATen call inside a Python nn.Module.  Each entry point cites the reference call site(s) whose arithmetic
 * it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed here;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work;
 *   - return value: 0 on success, a negative VDA_E* code on a rejected argument, or the positive hipError_t of a
 *     failed launch; nothing throws or aborts;
 *   - "bf16" is the raw 16-bit brain-float pattern (uint16_t); activations are channels-last ("NHWC" = [frame, y, x, c]),
 *     i.e. every feature map is a row-major [rows = frame*h*w + y*w + x, channels] matrix.
 *
 * Two storage modes share these entry points.  The default keeps activations in 16 bits.  The fp32 mode (the
 * reference's fp32=True, video_depth.py:67,136) keeps every activation in fp32 (dtype / out_type VDA_T_F32 below) and
 * still multiplies on the 16-bit matrix cores: an fp32 operand becomes 16-bit pieces hi = round(s v), lo = round(s v - hi)
 * and a product is the three-term sum hi*hi + lo*hi + hi*lo accumulated in fp32, at a third of the 16-bit rate instead of the
 * sixteenth of the fp32 matrix cores.  With IEEE-half pieces (11 significant bits each, operands pre-scaled by powers of two
 * into the half range and the sums scaled back, VdaGemmArgs.acc_scale) a product is good to about 2^-21 relative; with bf16
 * pieces (8 bits each, no scaling needed) to about 2^-16.  Two layouts of the pieces along K (VDA_PIECES_*):
 *   TRIPLE  rows [hi | lo | hi] against weights [hi | hi | lo], each segment K wide: any vda_gemm pass over K' = 3K forms the
 *           three terms without knowing about them;
 *   PAIRS   every 32 columns of the operand are stored as 64: [hi 32 | lo 32], for activations and weights alike, and
 *           VdaGemmArgs.ab_pairs tells the kernels, whose K step of 64 stored columns then holds both pieces of 32 operand
 *           columns: they issue the three products from ONE copy of each piece (two thirds of the global -> LDS and LDS ->
 *           register traffic of the triple form per product, and two thirds of the memory).
 */
#ifndef VDA_HIP_H
#define VDA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDA_OK 0
#define VDA_EINVAL (-1)   /* null pointer / non-positive size                      */
#define VDA_ESHAPE (-2)   /* shape not supported by the kernel (alignment, range)  */

/* Library identification: returns a static string "vda_hip <version> gfx950". */
const char* vda_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution on 16-bit MFMA (operands bf16 or fp16, selected by ab_type; same matrix-core
 * rate) with fp32 accumulation and a fused epilogue.
 *
 *   acc[m][n] = sum_k A[m][k] * W[n][k]                                   (W is [N][K] row-major: torch Linear layout)
 *   v = acc + bias[n];  v = act(v);  v *= gamma[n];  v += resid[m][n];  v += resid2[m][n];  out[m][n] = v
 *
 * a_mode VDA_A_CONV3 makes A an implicit im2col view of an NHWC image [img, in_h, in_w, cin] for a 3x3 convolution
 * with padding 1 and stride `stride`: row m = output pixel (img, oy, ox), k = (tap = ky*3+kx, channel), K = 9*cin,
 * W = [N][9][cin].  `zeros` must point to >= 256 bytes of zeros (source of the padding taps).
 *
 * Replaces: nn.Linear / 1x1 Conv2d / ConvTranspose2d(k == stride) / 3x3 Conv2d call sites of the reference:
 *   patch_embed.py:76 (as GEMM over im2col rows), attention.py:51,60, mlp.py:36-39, layer_scale.py:28, block.py:105-106,
 *   dpt.py:60-90,117-124, util/blocks.py:20-32,78-91,160, motion_module.py:113,120,123,
 *   motion_module/attention.py:81-83,90 (via motion_module.py:242-250,290), attention.py:333,374,383-384.
 * ------------------------------------------------------------------------------------------------------------- */
enum { VDA_A_LINEAR = 0, VDA_A_CONV3 = 1 };
enum { VDA_PIECES_TRIPLE = 0, VDA_PIECES_PAIRS = 1 };   /* layouts of an fp32 operand's hi/lo pieces, see the file comment */
enum { VDA_ACT_NONE = 0, VDA_ACT_GELU = 1, VDA_ACT_RELU = 2 };          /* GELU: the erf form, evaluated to 5.4e-5
                                                                           absolute (a fit of erf; 16-bit out_type only) */
enum { VDA_T_NONE = 0, VDA_T_F32 = 1, VDA_T_BF16 = 2, VDA_T_F16 = 3 };   /* "half" below = BF16 or F16 (IEEE half)  */
enum {
    VDA_EPI_PLAIN = 0,
    VDA_EPI_GEGLU = 1,  /* W rows interleaved in blocks of 16: [val 0..15, gate 0..15, val 16..31, ...]; out has N/2
                           columns: out = (val + b) * gelu(gate + b)   (motion_module/attention.py:382-384);
                           16-bit out_type only, like VDA_ACT_GELU                                                  */
    VDA_EPI_CONVT = 2,  /* ConvTranspose2d with kernel == stride == ct_k: row m = input pixel (img, y, x) of an
                           [*, ct_h, ct_w] grid, n = (ky*ct_k + kx)*ct_c + co; the value lands at output pixel
                           (img, y*ct_k+ky, x*ct_k+kx), channel co of an NHWC image with ct_c channels (dpt.py:71-82) */
    VDA_EPI_DOT = 3     /* out_f32[m] = relu(sum_n relu(acc+bias)[n] * dot_w[n] + dot_b): the 3x3 conv + ReLU + 1x1
                           conv + ReLU tail (dpt.py:119-122); needs N <= 32                                          */
};

typedef struct VdaGemmArgs {
    const void* a;        /* half [M][lda] (LINEAR) or NHWC image (CONV3)                    */
    const void* w;        /* half [N][K]                                                      */
    void*       out;      /* out_type [.][ldc]                                                */
    const float* bias;    /* [N] or NULL                                                      */
    const float* gamma;   /* [N] or NULL                                                      */
    const void* resid;    /* resid_type [M][ldr] or NULL; may alias out                        */
    const void* resid2;   /* resid_type [M][ldr2] or NULL (a second residual needs a first)    */
    const void* zeros;    /* >= 256 B of zeros (CONV3 only)                                    */
    const float* dot_w;   /* [N] (EPI_DOT only)                                                */
    float dot_b;
    int M, N, K;          /* K % 64 == 0                                                      */
    int lda, ldc, ldr, ldr2;
    int a_mode, act, epi, out_type, resid_type;
    int ab_type;          /* VDA_T_BF16 or VDA_T_F16: element type of a and w                  */
    int relu_in;          /* apply ReLU to A while loading (pre-activation conv, util/blocks.py:78) */
    int in_h, in_w, cin, out_h, out_w, stride;   /* CONV3 geometry; cin % 64 == 0, M = imgs*out_h*out_w */
    int ct_h, ct_w, ct_k, ct_c;                  /* EPI_CONVT geometry                                   */
    int tile_cfg;         /* 0 = the library picks the kernel; 1..3 force one where the call allows it (tuning and tests):
                             1 = 128x128 tile / 4 waves, 2 = 256x256 tile / 8 waves, software-pipelined, 3 = direct 3x3
                             convolution (stride 1, N = 32, 64 or a multiple of 128, no gamma); other values are VDA_EINVAL */
    int up_h, up_w;       /* CONV3 only; 0 = off.  Otherwise `a` is the NHWC map [img][up_h][up_w][cin] and the convolution
                             input is its bilinear resampling (align_corners=True, arithmetic of vda_bilinear_nhwc) to
                             in_h x in_w, computed while the input patches are loaded: the resampled map never exists in
                             memory (util/blocks.py:154-158 -> :160 and dpt_temporal.py:88,97-100 -> dpt.py:118-119).
                             Only where the direct 3x3 kernel applies (see tile_cfg 3), else VDA_ESHAPE              */
    float acc_scale;      /* 0 or 1: off.  Otherwise every sum acc[m][n] is multiplied by it before the epilogue: the
                             inverse of the powers of two that a and w were scaled by before rounding (see vda_split3) */
    int cin_used;         /* CONV3 only; 0 = cin.  Otherwise a multiple of 32 <= cin: only channels 0 .. cin_used-1 of every
                             pixel enter the products (the rest of the cin-wide pixel, e.g. the padding of a 32-channel map
                             stored 64 wide, is neither multiplied nor required to hold numbers).  Only where the direct
                             3x3 kernel applies (see tile_cfg 3), else VDA_ESHAPE                                      */
    /* LayerNorm folded into the two products around it (block.py:84,87: norm1 -> attention.py:51 qkv, norm2 -> mlp.py:36
       fc1).  With W' = W diag(ln_weight) rounded to the operand type and colsum[n] = sum_k W'[n][k]:
           LN(x) W^T + b = rstd (x W'^T - mean colsum) + (b + W ln_bias),
       so the product can read the un-normalised rows, rounded to 16 bits once, and normalise its sums.
       Producer side (the product that writes the rows x, i.e. proj / fc2 with their residual): */
    void*  out16;         /* NULL = off.  Otherwise ab_type [M][ld16]: every output value rounded to the operand type, stored
                             beside the fp32 output.  Needs out_type F32, EPI_PLAIN, A_LINEAR, N % 64 == 0, N >= 128,
                             16-byte aligned out rows and 8-byte aligned out16 rows, out16 and row_part each below 4 GiB
                             (M * ld16 * 2 and M * N/64 * 8 bytes), and row_part; else VDA_ESHAPE                     */
    float* row_part;      /* [M][N/64][2]: for every row and every block of 64 columns the sum and the sum of squares of
                             the ROUNDED values, added in an order that does not depend on where the row lies in a tile
                             (vda_rowstats_finish turns them into mean and 1/std)                                      */
    int    ld16;
    /* Consumer side (qkv / fc1 reading the rounded rows as A): */
    const float* row_stat;  /* NULL = off.  Otherwise [M][2] = (mean, 1/sqrt(var + eps)) of each row of A; every sum becomes
                               rstd[m] * (acc[m][n] - mean[m] * w_colsum[n]) before bias / activation.  A_LINEAR, EPI_PLAIN,
                               no residual                                                                              */
    const float* w_colsum;  /* [N], required with row_stat                                                            */
    /* fp32 mode: the output written directly as the next product's 16-bit operand (what vda_split3 would make of the fp32
       result), so that neither the fp32 matrix nor a split pass over it exists. */
    int   split_out;      /* 0 = off.  2, 3 or 4: `out` is ab_type [M][ldc] and receives the pieces of split_scale * v (v = the
                             epilogue's result): 2 = [hi | lo] in two segments of N columns (the form vda_attention_pairs
                             reads), 3 = VDA_PIECES_TRIPLE, 4 = VDA_PIECES_PAIRS (2 N columns); ldc >= the row's width.
                             out_type is ignored.  With act VDA_ACT_GELU the exact erf form is evaluated (as VDA_SPLIT_GELU).
                             Needs A_LINEAR, EPI_PLAIN, no gamma, no residual, N % 64 == 0, 16-byte aligned out, ldc % 8 == 0;
                             else VDA_ESHAPE                                                                           */
    float split_scale;    /* power of two > 0 (see vda_split3's `scale`)                                                */
    int*  sat_flag;       /* NULL or one int in device memory that is set to 1 when a scaled value leaves the IEEE-half range
                             (|split_scale * v| > 65504; the piece saturates); never cleared here                        */
    int   ab_pairs;       /* 1: a and w hold VDA_PIECES_PAIRS (K, lda, cin count STORED columns: twice the operand's); every
                             K step multiplies hi*hi + hi*lo + lo*hi.  No relu_in, row_stat or out16.  0: plain operands    */
} VdaGemmArgs;

int vda_gemm(const VdaGemmArgs* args, void* stream);

/* Which kernel vda_gemm runs for these arguments (it launches nothing): a VDA_KERNEL_* code, or the negative error code
 * vda_gemm would return.  For profiling tools that attribute a launch to its kernel. */
enum { VDA_KERNEL_128x32 = 0, VDA_KERNEL_128x64 = 1, VDA_KERNEL_128x128 = 2, VDA_KERNEL_256x256 = 3, VDA_KERNEL_CONV_DIRECT = 4 };
int vda_gemm_kernel_choice(const VdaGemmArgs* args);

/* ---------------------------------------------------------------------------------------------------------------
 * Multi-head self-attention over the tokens of each frame, head dim 64, no mask, flash-style (scores never leave
 * the CU).  qkv: [frames][n_tok][3*heads*64] laid out [q | k | v] x [head][64] as produced by the fused QKV
 * projection, q already multiplied by log2(e) * 64^-1/2 (VDA_ATTENTION_Q_SCALE): the scores are then base-2 logarithms
 * and the probabilities 2^(score - max), which is what the hardware's exponential computes; out:
 * [frames][n_tok][heads*64]; both of type dtype.
 * dtype VDA_T_BF16 or VDA_T_F16: 16-bit operands of that type (probabilities are rounded to it for the second product).
 * dtype VDA_T_F32: fp32 in and out; q, k, v and the probabilities are split into
 * IEEE-half hi/lo pairs on the fly (after scaling by powers of two) and every product is the three-term hi/lo sum.
 * Replaces attention.py:51-59 of the reference (dinov2_layers).
 * ------------------------------------------------------------------------------------------------------------- */
#define VDA_ATTENTION_Q_SCALE 0.18033688011112042   /* log2(e) / 8 */
int vda_attention(const void* qkv, void* out, int frames, int n_tok, int heads, int dtype, void* stream);

/* The fp32 mode's attention on operands that are already split: qkv2 is IEEE half [frames][n_tok][2][3*heads*64], per token
 * the hi pieces of `scale` * [q | k | v] followed by the lo pieces (VdaGemmArgs.split_out = 2 of the fused QKV projection);
 * out3 holds the pieces of `scale` * output in `out_layout` (VDA_PIECES_TRIPLE: [frames][n_tok][3][heads*64];
 * VDA_PIECES_PAIRS: [frames][n_tok][2*heads*64]): the A operand of the output projection.
 * Same arithmetic as vda_attention with dtype VDA_T_F32 (three-term products, fp32 softmax state), but K and V tiles go
 * global -> LDS without passing through registers and are not split again by every query block.  `scale` is the power of
 * two the producer used (q, k and v share it). */
int vda_attention_pairs(const void* qkv2, void* out3, int frames, int n_tok, int heads, float scale, int out_layout, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Row LayerNorm (biased variance, eps inside the sqrt): fp32 rows in, rows of out_type (BF16, F16 or F32) out.
 * Output row r reads input row (r / rows_per_group) * in_group + in_skip + r % rows_per_group, which drops the class
 * token when taking the encoder taps (dinov2.py:310-312); with rows_per_group == in_group and in_skip == 0 it is the
 * identity map.  If pe != NULL, pe[((r / pe_hw) % pe_T)][c] (fp32 [pe_T][C]) is added after the affine step: the
 * temporal position table of motion_module.py:196-197 applied to rows ordered (frame, position).
 * split_scale != 0 (fp32 mode; out_type BF16 or F16, C % 64 == 0): the output rows are the pieces of split_scale * y in
 * split_layout (VDA_PIECES_TRIPLE: [rows][3*C], VDA_PIECES_PAIRS: [rows][2*C]), what vda_split3 would make of the fp32 result
 * y; sat_flag as in VdaGemmArgs (may be NULL).
 * Replaces block.py:84,87 (norm1/norm2), dinov2.py:310, motion_module.py:166,174.
 * ------------------------------------------------------------------------------------------------------------- */
int vda_layernorm(const float* x, void* out, const float* weight, const float* bias, int rows, int C, float eps,
                  int rows_per_group, int in_group, int in_skip, const float* pe, int pe_hw, int pe_T, int out_type,
                  float split_scale, int split_layout, int* sat_flag, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Row statistics for a LayerNorm that is folded into the product that follows it (VdaGemmArgs.row_stat; block.py:84,87).
 * vda_round_rowstats: fp32 rows [rows][C] -> out16 [rows][C] (out_type BF16 or F16: the rows rounded once) and
 *   stat[rows][2] = (mean, 1/sqrt(var + eps)) of the ROUNDED values (biased variance): the first LayerNorm of the
 *   encoder, whose input no product of this library wrote.  C % 4 == 0, C <= 2048.
 * vda_rowstats_finish: part[rows][slots][2] (sum, sum of squares per 64-column block, as VdaGemmArgs.row_part writes them)
 *   -> stat[rows][2]; C = 64 * slots values per row.
 * ------------------------------------------------------------------------------------------------------------- */
int vda_round_rowstats(const float* x, void* out16, float* stat, int rows, int C, float eps, int out_type, void* stream);
int vda_rowstats_finish(const float* part, float* stat, int rows, int slots, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GroupNorm over NHWC maps (dtype BF16, F16 or F32): statistics per (frame, group) over hw * C/groups values, affine,
 * output in the same type.
 * `work` is caller-provided scratch of frames * VDA_GROUPNORM_MAX_SPLITS * groups * 16 bytes, 8-byte aligned (partial sums
 * per slice of a frame; no atomics are used, so results are bit-identical from run to run).  C <= 2048.
 * Replaces motion_module.py:110.
 * ------------------------------------------------------------------------------------------------------------- */
#define VDA_GROUPNORM_MAX_SPLITS 64
int vda_groupnorm(const void* x, void* out, const float* weight, const float* bias, void* work, int frames, int hw,
                  int C, int groups, float eps, int dtype, void* stream);

/* vda_groupnorm without the `work` argument, for callers that cannot allocate device memory (code under HIP graph capture,
 * serving loops with a fixed memory plan): the partial sums live in a 4 MiB device array inside this library, one copy per
 * GPU, held until the process ends.  No counterpart in the reference.  Same results as vda_groupnorm, bit for bit.
 * frames <= VDA_GROUPNORM_STATIC_MAX_FRAMES, else VDA_ESHAPE.  The device that `stream` belongs to must be the current
 * device.  Calls on one GPU share the array: two calls that may run at the same time (different streams) race; calls on
 * one stream, or otherwise ordered, are safe, and nothing needs clearing between them.  Allocates nothing.  The first call on
 * each device looks the array's address up (hipGetSymbolAddress, which may load the code object): make that call outside
 * any graph capture; every later call only enqueues the two kernels. */
#define VDA_GROUPNORM_STATIC_MAX_FRAMES 64
int vda_groupnorm_static(const void* x, void* out, const float* weight, const float* bias, int frames, int hw, int C,
                         int groups, float eps, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Bilinear resampling with align_corners=True (src = dst * (in-1)/(out-1)).
 * vda_bilinear_nhwc: [frames][in_h][in_w][C] -> [frames][out_h][out_w][C] of dtype BF16, F16 or F32   (util/blocks.py:156-158,
 *   dpt_temporal.py:94-96).  C % 8 == 0.
 * vda_bilinear_f32: fp32 single-channel maps [frames][in_h][in_w] -> [frames][out_h][out_w], optional ReLU
 *   (video_depth.py:63-64, 142).
 * ------------------------------------------------------------------------------------------------------------- */
int vda_bilinear_nhwc(const void* x, void* out, int frames, int in_h, int in_w, int out_h, int out_w, int C, int dtype,
                      void* stream);
int vda_bilinear_f32(const float* x, float* out, int frames, int in_h, int in_w, int out_h, int out_w, int relu, void* stream);
/* vda_bilinear_nhwc on an fp32 map whose result is written as the 16-bit operand pieces of `scale` * v (out_type BF16 or F16,
 * `layout` VDA_PIECES_*, rows of 3C or 2C columns; C % 64 == 0): the fp32 mode's input of a convolution on a resampled map
 * (dpt_temporal.py:93-99), which then never exists in fp32.  sat_flag as in vda_split3. */
int vda_bilinear_nhwc_pieces(const float* x, void* out, int frames, int in_h, int in_w, int out_h, int out_w, int C, int out_type,
                             float scale, int layout, int* sat_flag, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Patch embedding front end.
 * vda_im2col_patch14: fp32 NCHW frames [frames][3][H][W] -> rows [frames*hp*wp][ldk] of out_type BF16, F16 or F32, k = c*196 + ky*14 + kx
 *   (the flattening order of the conv weight [D][3][14][14]), columns 588..ldk-1 zero.  ldk % 64 == 0, ldk >= 588.
 * vda_assemble_tokens: x[f][0] = cls + pos[0]; x[f][1+p] = tok[f*hw+p] + pos[1+p]  (fp32)   (dinov2.py:218-219).
 * Replaces patch_embed.py:76-78 together with vda_gemm.
 * ------------------------------------------------------------------------------------------------------------- */
int vda_im2col_patch14(const float* x, void* out, int frames, int H, int W, int ldk, int out_type, void* stream);
int vda_assemble_tokens(const float* tok, const float* cls, const float* pos, float* x, int frames, int hw, int D, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * fp32 mode operand split: x fp32 [rows][ldx] -> out of out_type (BF16 or F16): layout VDA_PIECES_TRIPLE [rows][3*Cp], segments
 * [hi | lo | hi] of Cp columns each; VDA_PIECES_PAIRS [rows][2*Cp], every 32 columns as [hi 32 | lo 32]
 * (Cp % 64 == 0, columns C..Cp-1 zero, C % 4 == 0), hi = round(scale * v), lo = round(scale * v - hi); `scale` > 0
 * is a power of two chosen by the caller (1 for bf16; for IEEE halves one that keeps |scale * v| below 65504 and the values
 * that matter above 2^-3, so that lo is a normal half; hi saturates instead of overflowing).  `act` is applied to v first,
 * because an activation cannot be applied to hi and lo separately:
 *   VDA_SPLIT_RELU   max(v, 0): the pre-activation of util/blocks.py:78;
 *   VDA_SPLIT_GELU   exact-form GELU (mlp.py:37): in the fp32 mode fc1 stores its pre-activation and fc2's operand is
 *                    split here (vda_gemm's own GELU is accurate to 16-bit storage only);
 *   VDA_SPLIT_GEGLU  the input row holds 2C values in blocks [16 values | 16 gates] (the column order of a GEGLU
 *                    projection packed for VDA_EPI_GEGLU, run with a plain epilogue); v = value * GELU(gate)
 *                    (motion_module/attention.py:382-384); C % 16 == 0, ldx >= 2C.
 * For an NHWC image rows = pixels and the result is an image with 3*Cp channels: the CONV3 operand for weights packed
 * [Cout][9][hi | hi | lo].  sat_flag: NULL, or one int in device memory that is set to 1 when |scale * v| > 65504 for some
 * value (IEEE-half pieces only; the piece saturates).  No counterpart in the reference (its fp32 path multiplies in fp32).
 * ------------------------------------------------------------------------------------------------------------- */
enum { VDA_SPLIT_NONE = 0, VDA_SPLIT_RELU = 1, VDA_SPLIT_GELU = 2, VDA_SPLIT_GEGLU = 3 };
int vda_split3(const float* x, void* out, long rows, int C, int ldx, int Cp, int act, int out_type, float scale, int layout,
               int* sat_flag, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Frame preprocessing: uint8 RGB frames [frames][h0][w0][3] -> fp32 [frames][3][H][W]: x/255, bicubic resize with
 * OpenCV INTER_CUBIC semantics (half-pixel centres, A = -0.75, replicated border; identity when sizes match), ImageNet
 * mean/std normalisation, HWC -> CHW.  Replaces video_depth.py:120 + util/transform.py:109-158 (cv2.resize).
 * ------------------------------------------------------------------------------------------------------------- */
int vda_preprocess_u8(const uint8_t* frames_u8, float* out, int frames, int h0, int w0, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Temporal self-attention: for every (batch, spatial position, head) a softmax(q k^T * scale) v over the T <= 32
 * frames.  qkv: rows ordered (batch, frame, position), each row [q | k | v] x [head][hd]; out: same row order,
 * [head][hd] (dtype BF16, F16 or F32 for both; the arithmetic is fp32 in every case).  Replaces motion_module.py:232,279-284,295 + motion_module/attention.py:93-111,182-211.
 * ------------------------------------------------------------------------------------------------------------- */
int vda_temporal_attention(const void* qkv, void* out, int batch, int T, int hw, int heads, int hd, float scale, int dtype,
                           void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Window stitching (video_depth.py:166-205, utils/util.py:40-74).
 * vda_scale_shift_sums: sums[0..4] += {sum p*p, sum p, count, sum p*t, sum t} over n fp32 values.  `sums` holds
 *   VDA_LSQ_SUMS_DOUBLES doubles: the five totals (zeroed by the caller before the first call of a fit) followed by
 *   scratch for per-workgroup partial sums, which are added in a fixed order (no atomics: bit-identical results).
 * vda_affine_blend: out = w_pre * pre + w_post * max(scale * x + shift, 0)   over n fp32 values; scale/shift are read
 *   from device memory (ss[0], ss[1]) so no host round trip is needed; pre may be NULL when w_pre == 0.
 * vda_solve_scale_shift: one-thread 2x2 solve in fp32 from the five sums: ss = {scale, shift} ({1,0} if det == 0).
 * ------------------------------------------------------------------------------------------------------------- */
#define VDA_LSQ_BLOCKS 256
#define VDA_LSQ_SUMS_DOUBLES (5 + 4 * VDA_LSQ_BLOCKS)
int vda_scale_shift_sums(const float* pred, const float* target, double* sums, long n, void* stream);
int vda_solve_scale_shift(const double* sums, float* ss, void* stream);
int vda_affine_blend(const float* x, const float* pre, float* out, const float* ss, float w_pre, float w_post, long n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Metric post-processing (metric_depth/run2.py:42-66, the same code inline at metric_depth/run.py:233-264): rescale every
 * frame so that the prediction at a reference pixel equals a LiDAR range.
 * vda_sample_points: out[i] = depth[i][xy[2i+1]][xy[2i]] for fp32 maps [frames][h][w]; the caller has already clamped the
 *   coordinates into the map (run2.py:51-53).  The handful of scale factors is then computed and interpolated over time
 *   on the host (run2.py:54-66: N scalars).
 * vda_scale_frames: out[i][p] = depth[i][p] * scales[i]   (run2.py:67), scales fp32 [frames] in device memory; out may
 *   alias depth.
 * ------------------------------------------------------------------------------------------------------------- */
int vda_sample_points(const float* depth, const int* xy, float* out, int frames, int h, int w, void* stream);
int vda_scale_frames(const float* depth, const float* scales, float* out, int frames, long hw, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Depth visualisation, the arithmetic of save_video(..., is_depths=True) (utils/dc_utils.py:72-82): the video's global
 * minimum and maximum, then per pixel  g = (uint8)((d - min) / (max - min) * 255)  in float32 and a colour table lookup.
 * vda_minmax_f32: minmax[0..1] = {min, max} of n fp32 values; `minmax` holds VDA_MINMAX_FLOATS floats (the two results
 *   followed by per-workgroup scratch).
 * vda_colorize_u8: out[3i..3i+2] = lut[3g..3g+2] (lut: 256 RGB byte triples in device memory, e.g. matplotlib's inferno
 *   table times 255, truncated) or, with lut == NULL, out[i] = g (the grayscale option).
 * ------------------------------------------------------------------------------------------------------------- */
#define VDA_MINMAX_FLOATS (2 + 2 * VDA_LSQ_BLOCKS)
int vda_minmax_f32(const float* x, float* minmax, long n, void* stream);
int vda_colorize_u8(const float* depth, const float* minmax, const uint8_t* lut, uint8_t* out, long n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Colour-map variants of the reference's drivers, on device-resident fp32 maps.  A colour table is matplotlib's for the map:
 * 259 RGB byte triples = the 256 entries times 255, truncated, followed by the under, over and bad (NaN) colours; an index
 * is taken from a value v in [0, 1] as matplotlib does (v * 256 truncated, 256 -> 255, negative -> under, larger -> over).
 * The arithmetic runs in float32, or in float64 where `f64` is set: whichever numpy evaluates the reference's expression in
 * for the scalar types at hand (the caller knows them).
 * vda_colormap_u8: out[i] = table[index((clip(x[i], clip_lo, clip_hi) - sub) / div)]; x[i] >= cap takes the last of the 256
 *   colours.  Covers run1dist.py:46-65 (contrast stretching between two percentiles), :67-91 (distance capping) and :173-177.
 * vda_colorize_zones_u8: run1.py:28-60: x <= critical_max -> lut_critical[index(x / critical_max)]; up to traffic_max ->
 *   lut_traffic[index((x - critical_max) / traffic_range)]; beyond -> (64, 64, 64); NaN -> (0, 0, 0).
 * vda_range_focus_f32: run4.py:29-42: out = x * (x > far ? far_weight : 1), then, with `stretch`,
 *   clip((out - sub) / div, clip_lo, clip_hi) (the percentiles sub / div come from the caller).
 * vda_quantize_u8 / vda_histogram_u8 / vda_colormap_bytes_u8: run1dist.py:27-44: g = (uint8)((x - lo) / range * 255); the
 *   256-bin histogram of g (`hist` is added to: clear it first); out = table[index(remap[g] / 255)] for the equalisation table
 *   `remap` the caller derives from the histogram (cv2.equalizeHist's rule; OpenCV is not available to pin it).
 * ------------------------------------------------------------------------------------------------------------- */
int vda_colormap_u8(const float* x, const uint8_t* lut, uint8_t* out, long n, double sub, double div, double clip_lo,
                    double clip_hi, float cap, int f64, void* stream);
int vda_colorize_zones_u8(const float* x, const uint8_t* lut_critical, const uint8_t* lut_traffic, uint8_t* out, long n,
                          float critical_max, float traffic_max, float traffic_range, void* stream);
int vda_range_focus_f32(const float* x, float* out, long n, float far, float far_weight, int stretch, double sub, double div,
                        double clip_lo, double clip_hi, int f64, void* stream);
int vda_quantize_u8(const float* x, uint8_t* out, long n, float lo, float range, void* stream);
int vda_histogram_u8(const uint8_t* x, uint32_t* hist, long n, void* stream);
int vda_colormap_bytes_u8(const uint8_t* g, const uint8_t* remap, const uint8_t* lut, uint8_t* out, long n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Temporal refinement step of the metric driver (metric_depth/run2.py:28-40, 76): out = w_cur * cur + w_prev * warp(prev),
 * where warp samples `prev` at (x + flow[y][x][0], y + flow[y][x][1]) the way cv2.remap(INTER_LINEAR, BORDER_REFLECT) does
 * (position rounded to 1/32 pixel, four taps, mirrored border).  fp32 maps [h][w], flow [h][w][2] (the caller's optical
 * flow: the reference's Farneback estimator is an OpenCV routine and is not part of this library).  out must not alias prev.
 * OpenCV is not available here, so parity of the sampling with cv2.remap is unpinned; the blend is exact.
 * ------------------------------------------------------------------------------------------------------------- */
int vda_warp_blend_f32(const float* prev, const float* cur, const float* flow, float* out, int h, int w, float w_cur, float w_prev,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif
