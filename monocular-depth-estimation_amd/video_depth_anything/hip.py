"""ctypes binding of libvda_hip.so (declarations: include/vda_hip.h) operating on torch CUDA(ROCm) tensors.

PyTorch only supplies device memory and the stream here: every function passes raw device pointers and sizes to
the C ABI and raises on a non-zero return code.  There is no fallback: if the library is missing, loading raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDA_HIP_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("VDA_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libvda_hip.so")

A_LINEAR, A_CONV3 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
T_NONE, T_F32, T_BF16, T_F16 = 0, 1, 2, 3
EPI_PLAIN, EPI_GEGLU, EPI_CONVT, EPI_DOT = 0, 1, 2, 3
GROUPNORM_MAX_SPLITS = 64        # VDA_GROUPNORM_MAX_SPLITS: sizes the scratch of groupnorm()
LSQ_SUMS_DOUBLES = 5 + 4 * 256   # VDA_LSQ_SUMS_DOUBLES: five totals + per-workgroup partials of scale_shift_sums()

EXPORTS = [
    "vda_version", "vda_gemm", "vda_gemm_kernel_choice", "vda_attention", "vda_attention_pairs", "vda_layernorm", "vda_round_rowstats", "vda_rowstats_finish", "vda_groupnorm", "vda_groupnorm_static", "vda_bilinear_nhwc", "vda_bilinear_nhwc_pieces", "vda_bilinear_f32",
    "vda_im2col_patch14", "vda_assemble_tokens", "vda_temporal_attention", "vda_scale_shift_sums", "vda_solve_scale_shift",
    "vda_affine_blend", "vda_preprocess_u8", "vda_split3", "vda_sample_points", "vda_scale_frames", "vda_minmax_f32", "vda_colorize_u8",
    "vda_colormap_u8", "vda_colorize_zones_u8", "vda_range_focus_f32", "vda_quantize_u8", "vda_histogram_u8", "vda_colormap_bytes_u8",
    "vda_warp_blend_f32",
]


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p),
        ("resid", C.c_void_p), ("resid2", C.c_void_p), ("zeros", C.c_void_p), ("dot_w", C.c_void_p), ("dot_b", C.c_float),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int), ("ldr2", C.c_int),
        ("a_mode", C.c_int), ("act", C.c_int), ("epi", C.c_int), ("out_type", C.c_int), ("resid_type", C.c_int),
        ("ab_type", C.c_int), ("relu_in", C.c_int),
        ("in_h", C.c_int), ("in_w", C.c_int), ("cin", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int), ("stride", C.c_int),
        ("ct_h", C.c_int), ("ct_w", C.c_int), ("ct_k", C.c_int), ("ct_c", C.c_int), ("tile_cfg", C.c_int),
        ("up_h", C.c_int), ("up_w", C.c_int), ("acc_scale", C.c_float), ("cin_used", C.c_int),
        ("out16", C.c_void_p), ("row_part", C.c_void_p), ("ld16", C.c_int), ("row_stat", C.c_void_p), ("w_colsum", C.c_void_p),
        ("split_out", C.c_int), ("split_scale", C.c_float), ("sat_flag", C.c_void_p), ("ab_pairs", C.c_int),
    ]


PIECES_TRIPLE, PIECES_PAIRS = 0, 1          # VDA_PIECES_*: layouts of an fp32 operand's 16-bit hi/lo pieces


def pieces_width(layout, K):
    """Stored columns of K operand columns."""
    return 2 * K if layout == PIECES_PAIRS else 3 * K


_lib = None
TILE_CFG = int(os.environ.get("VDA_GEMM_TILE_CFG", "0"))      # 0 = library default; see VdaGemmArgs.tile_cfg


def lib():
    """The loaded library; raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP kernels are not built and there is no fallback path")
        l = C.CDLL(LIB_PATH)
        l.vda_version.restype = C.c_char_p
        l.vda_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
        l.vda_gemm_kernel_choice.argtypes = [C.POINTER(GemmArgs)]
        l.vda_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.vda_attention_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        l.vda_layernorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        l.vda_round_rowstats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        l.vda_rowstats_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        l.vda_groupnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_float, C.c_int, C.c_void_p]
        l.vda_groupnorm_static.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_float, C.c_int, C.c_void_p]
        l.vda_bilinear_nhwc.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
        l.vda_bilinear_f32.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
        l.vda_bilinear_nhwc_pieces.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        l.vda_im2col_patch14.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.vda_split3.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                 C.c_int, C.c_void_p, C.c_void_p]
        l.vda_assemble_tokens.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]
        l.vda_temporal_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                             C.c_int, C.c_void_p]
        l.vda_scale_shift_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        l.vda_solve_scale_shift.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.vda_affine_blend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_long,
                                       C.c_void_p]
        l.vda_preprocess_u8.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
        l.vda_sample_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.vda_scale_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p]
        l.vda_minmax_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        l.vda_colorize_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        l.vda_colormap_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long] + [C.c_double] * 4 + [C.c_float, C.c_int, C.c_void_p]
        l.vda_colorize_zones_u8.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_float, C.c_float, C.c_float, C.c_void_p]
        l.vda_range_focus_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_void_p]
        l.vda_quantize_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_void_p]
        l.vda_histogram_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        l.vda_colormap_bytes_u8.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_void_p]
        l.vda_warp_blend_f32.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        for name in EXPORTS[1:]:
            getattr(l, name).restype = C.c_int
        _lib = l
    return _lib


# The current stream of the current device as a raw handle.  torch's own accessors of the raw handle cost a fraction of a
# microsecond; torch.cuda.current_stream() builds a Stream object and resolves the device index in Python, 8 us per call,
# which at the 300 launches of a ViT-S window is a third of the window's time on the device.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _current_device is not None:
        return C.c_void_p(_raw_stream(_current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


_TYPES = {torch.float32: T_F32, torch.bfloat16: T_BF16, torch.float16: T_F16}


def _type(t):
    return _TYPES[t.dtype]


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libvda_hip operates on device tensors only (got a CPU tensor)")


def _need(what, **tensors):
    """The kernels trust the extents they are given, and a launch that runs past a buffer faults the GPU: refuse, on the
    host, any call whose shape arguments reach beyond the memory behind a tensor (counted from its first element to the end
    of its storage, so views into larger buffers pass).  ``tensors``: name = (tensor or None, elements needed)."""
    for name, (t, elements) in tensors.items():
        if t is None:
            continue
        have = t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
        if elements > have:
            raise ValueError(f"{what}: {name} holds {have} elements from its first one, the call needs {elements}")


_zeros = {}


def zero_page(device):
    z = _zeros.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        # filled on whichever stream is current now, read from then on by launches on every stream: wait for the fill once
        torch.cuda.current_stream(z.device).synchronize()
        _zeros[device] = z
    return z


def gemm(a, w, out, *, M, N, K, lda, ldc, bias=None, gamma=None, resid=None, ldr=0, resid2=None, ldr2=0, act=ACT_NONE,
         epi=EPI_PLAIN, relu_in=False, conv=None, convt=None, dot_w=None, dot_b=0.0, tile_cfg=None, upsample_from=None,
         acc_scale=1.0, cin_used=0, out16=None, row_part=None, row_stat=None, w_colsum=None, split_out=0, split_scale=1.0,
         sat_flag=None, ab_pairs=False):
    """Enqueue one vda_gemm.  ``conv`` = (in_h, in_w, cin, out_h, out_w, stride); ``convt`` = (h, w, k, c);
    ``upsample_from`` = (h, w) of the map ``a`` that is bilinearly resampled to in_h x in_w on the fly;
    ``acc_scale`` multiplies every sum before the epilogue (operands pre-scaled by powers of two); ``cin_used``: only the
    first so many channels of every input pixel are multiplied (direct 3x3 kernel only).
    Folded LayerNorm: ``out16`` [M, >= N] (a's type) and ``row_part`` fp32 [M, N // 64, 2] make the product also emit its
    output rows rounded to 16 bits with their partial sums; ``row_stat`` fp32 [M, 2] and ``w_colsum`` fp32 [N] make it
    normalise its sums with the (mean, rstd) of the rows of ``a`` (see VdaGemmArgs in include/vda_hip.h).
    ``split_out`` (fp32 mode): ``out`` is 16-bit and receives the pieces of ``split_scale`` times the result, 2 = planes [hi | lo]
    (2N columns), 3 = PIECES_TRIPLE (3N), 4 = PIECES_PAIRS (2N); ``sat_flag`` (one int32 on the device) is set when a scaled
    value leaves the half range.  ``ab_pairs``: a and w hold PIECES_PAIRS (K, lda and cin count stored columns)."""
    _dev(a, w, out, bias, gamma, resid, resid2, dot_w, out16, row_part, row_stat, w_colsum, sat_flag)
    g = GemmArgs()
    g.a, g.w, g.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.gamma = gamma.data_ptr() if gamma is not None else None
    g.resid = resid.data_ptr() if resid is not None else None
    g.resid2 = resid2.data_ptr() if resid2 is not None else None
    g.dot_w = dot_w.data_ptr() if dot_w is not None else None
    g.dot_b = float(dot_b)
    g.M, g.N, g.K, g.lda, g.ldc, g.ldr, g.ldr2 = M, N, K, lda, ldc, ldr, ldr2
    g.act, g.epi, g.relu_in = act, epi, int(bool(relu_in))
    g.acc_scale = float(acc_scale)
    g.cin_used = int(cin_used)
    g.ab_pairs = int(bool(ab_pairs))
    g.tile_cfg = TILE_CFG if tile_cfg is None else tile_cfg
    if a.dtype != w.dtype:
        raise TypeError("vda_gemm: a and w must share one 16-bit type")
    if resid2 is not None and (resid is None or resid2.dtype != resid.dtype):
        raise TypeError("vda_gemm: resid2 needs resid, of the same type")
    if out16 is not None:
        if out16.dtype != a.dtype or row_part is None or row_part.dtype != torch.float32:
            raise TypeError("vda_gemm: out16 has the operands' type and needs an fp32 row_part")
        g.out16, g.row_part, g.ld16 = out16.data_ptr(), row_part.data_ptr(), out16.stride(0)
    if row_stat is not None:
        if w_colsum is None or row_stat.dtype != torch.float32 or w_colsum.dtype != torch.float32:
            raise TypeError("vda_gemm: row_stat and w_colsum are fp32 and come together")
        g.row_stat, g.w_colsum = row_stat.data_ptr(), w_colsum.data_ptr()
    if split_out:
        if out.dtype != a.dtype:
            raise TypeError("vda_gemm: a split output has the operands' 16-bit type")
        if sat_flag is not None and sat_flag.dtype != torch.int32:
            raise TypeError("vda_gemm: sat_flag is one int32")
        g.split_out, g.split_scale = int(split_out), float(split_scale)
        g.sat_flag = sat_flag.data_ptr() if sat_flag is not None else None
    g.ab_type = _type(a)
    g.out_type = T_F32 if split_out else _type(out)
    g.resid_type = T_NONE if resid is None else _type(resid)
    n_res = N // 2 if epi == EPI_GEGLU else N
    n_out = {0: 1, 2: 2, 3: 3, 4: 2}[split_out] * n_res
    _need("vda_gemm", sat_flag=(sat_flag, 1), w=(w, N * K), gamma=(gamma, N), dot_w=(dot_w, N), w_colsum=(w_colsum, N), row_stat=(row_stat, 2 * M),
          row_part=(row_part, 2 * M * (N // 64)), out16=(out16, (M - 1) * (out16.stride(0) if out16 is not None else 0) + N),
          a=(a if conv is None else None, (M - 1) * lda + K), bias=(bias if convt is None else None, N),
          out=(out if convt is None else None, M if epi == EPI_DOT else (M - 1) * ldc + n_out),
          resid=(resid, (M - 1) * ldr + n_res), resid2=(resid2, (M - 1) * ldr2 + n_res))
    if conv is not None:
        g.a_mode = A_CONV3
        g.in_h, g.in_w, g.cin, g.out_h, g.out_w, g.stride = conv
        g.zeros = zero_page(a.device).data_ptr()
        if upsample_from is not None:
            g.up_h, g.up_w = upsample_from
    else:
        g.a_mode = A_LINEAR
    if convt is not None:
        g.ct_h, g.ct_w, g.ct_k, g.ct_c = convt
    t = _gemm_timing
    if t is not None and conv is None and epi == EPI_PLAIN and gamma is None:
        # the kernel the library picks for this launch, and the epilogue form (= kernel instantiation) it takes
        kernel = KERNEL_NAMES.get(lib().vda_gemm_kernel_choice(C.byref(g)), "?")
        if resid is not None and resid.dtype == torch.float32 and out.dtype == torch.float32 and act == ACT_NONE and resid2 is None:
            form = "f32_res32_emit" if out16 is not None else "f32_res32"
        elif resid is None and out.dtype == a.dtype:
            form = "own_gelu" if act == ACT_GELU else "own" if act == ACT_NONE else "other"
            if row_stat is not None and form != "other":
                form += "_ln"
        else:
            form = "other"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(lib().vda_gemm(C.byref(g), _stream()), "vda_gemm")
        e1.record()
        t.events.append((e0, e1, 2.0 * M * N * K, (kernel, form, _TYPE_NAMES[g.ab_type])))
        return
    _check(lib().vda_gemm(C.byref(g), _stream()), "vda_gemm")


KERNEL_NAMES = {0: "gemm_kernel<128,32>", 1: "gemm_kernel<128,64>", 2: "gemm_kernel<128,128>", 3: "gemm256_kernel", 4: "conv3_direct_kernel"}
_TYPE_NAMES = {T_BF16: "bf16", T_F16: "f16"}


class GemmTiming:
    """Per-launch HIP-event timing of the plain linear GEMMs, tagged (kernel, epilogue form, operand type): the kernel is the
    one the library reports for the launch (vda_gemm_kernel_choice), the form is the kernel instantiation the epilogue
    takes: "f32_res32" (proj, fc2; "f32_res32_emit" when they also emit the rounded rows for a folded LayerNorm), "own_gelu"
    (fc1), "own" (qkv and others); "_ln" marks the forms that normalise their sums (folded LayerNorm).  The events are recorded on the stream the
    kernel is launched on (torch's current stream); launches on concurrent streams overlap and cannot be told apart this
    way, so time with one stream."""

    def __init__(self):
        self.events = []

    def summary(self, tag=None):
        """(launches, total algorithmic FLOP, total milliseconds) of one tag or of all; call after a device synchronize."""
        ev = [e for e in self.events if tag is None or e[3] == tag]
        return len(ev), sum(e[2] for e in ev), sum(e[0].elapsed_time(e[1]) for e in ev)

    def tags(self):
        return sorted({e[3] for e in self.events})


_gemm_timing = None


def enable_gemm_timing():
    global _gemm_timing
    _gemm_timing = GemmTiming()
    return _gemm_timing


def disable_gemm_timing():
    global _gemm_timing
    _gemm_timing = None


def attention(qkv, out, frames, n_tok, heads):
    _dev(qkv, out)
    if qkv.dtype != out.dtype:
        raise TypeError("vda_attention: qkv and out must share one type")
    _need("vda_attention", qkv=(qkv, frames * n_tok * 3 * heads * 64), out=(out, frames * n_tok * heads * 64))
    _check(lib().vda_attention(_ptr(qkv), _ptr(out), frames, n_tok, heads, _type(qkv), _stream()), "vda_attention")


def attention_pairs(qkv2, out3, frames, n_tok, heads, scale, out_layout=PIECES_TRIPLE):
    """vda_attention_pairs: fp16 [hi | lo] planes of scale * qkv in, fp16 pieces of scale * output out (``out_layout``)."""
    _dev(qkv2, out3)
    if qkv2.dtype != torch.float16 or out3.dtype != torch.float16:
        raise TypeError("vda_attention_pairs: IEEE-half pieces in and out")
    _need("vda_attention_pairs", qkv2=(qkv2, frames * n_tok * 6 * heads * 64),
          out3=(out3, frames * n_tok * pieces_width(out_layout, heads * 64)))
    _check(lib().vda_attention_pairs(_ptr(qkv2), _ptr(out3), frames, n_tok, heads, float(scale), int(out_layout), _stream()),
           "vda_attention_pairs")


SPLIT_NONE, SPLIT_RELU, SPLIT_GELU, SPLIT_GEGLU = 0, 1, 2, 3


def split3(x, out, rows, Cdim, ldx, Cp, relu=False, act=None, scale=1.0, sat_flag=None, layout=PIECES_TRIPLE):
    """fp32 [rows][ldx] -> the 16-bit pieces of ``scale * x`` in ``out``'s type (bf16 or fp16), [rows][3*Cp] as [hi | lo | hi]
    (PIECES_TRIPLE) or [rows][2*Cp] with every 32 columns as [hi 32 | lo 32] (PIECES_PAIRS): the A operand of the fp32 mode's
    GEMMs, after the activation ``act`` (SPLIT_*; ``relu=True`` is SPLIT_RELU)."""
    _dev(x, out, sat_flag)
    if act is None:
        act = SPLIT_RELU if relu else SPLIT_NONE
    if sat_flag is not None and sat_flag.dtype != torch.int32:
        raise TypeError("vda_split3: sat_flag is one int32")
    _need("vda_split3", x=(x, (rows - 1) * ldx + (2 * Cdim if act == SPLIT_GEGLU else Cdim)), out=(out, rows * pieces_width(layout, Cp)))
    _check(lib().vda_split3(_ptr(x), _ptr(out), rows, Cdim, ldx, Cp, int(act), _type(out), float(scale), int(layout), _ptr(sat_flag),
                            _stream()), "vda_split3")


def layernorm(x, out, weight, bias, rows, Cdim, eps, rows_per_group=None, in_group=None, in_skip=0, pe=None, pe_hw=1, pe_T=1,
              split_scale=0.0, sat_flag=None, split_layout=PIECES_TRIPLE):
    """``split_scale`` != 0: ``out`` holds the 16-bit pieces of split_scale * LayerNorm(x) in ``split_layout``."""
    _dev(x, out, weight, bias, pe, sat_flag)
    if sat_flag is not None and sat_flag.dtype != torch.int32:
        raise TypeError("vda_layernorm: sat_flag is one int32")
    if rows_per_group is None:
        rows_per_group = in_group = rows
    groups = -(-rows // rows_per_group)
    last = (groups - 1) * in_group + in_skip + (rows - 1 - (groups - 1) * rows_per_group)       # the last input row read
    _need("vda_layernorm", x=(x, (last + 1) * Cdim), out=(out, rows * (pieces_width(split_layout, Cdim) if split_scale else Cdim)), weight=(weight, Cdim),
          bias=(bias, Cdim), pe=(pe, pe_T * Cdim), sat_flag=(sat_flag, 1))
    _check(lib().vda_layernorm(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), rows, Cdim, eps, rows_per_group, in_group,
                               in_skip, _ptr(pe), pe_hw, pe_T, _type(out), float(split_scale), int(split_layout), _ptr(sat_flag), _stream()),
           "vda_layernorm")


def round_rowstats(x, out16, stat, rows, Cdim, eps):
    """fp32 rows -> the rows rounded to ``out16``'s type and ``stat`` [rows, 2] = (mean, rstd) of the rounded values."""
    _dev(x, out16, stat)
    _need("vda_round_rowstats", x=(x, rows * Cdim), out16=(out16, rows * Cdim), stat=(stat, 2 * rows))
    _check(lib().vda_round_rowstats(_ptr(x), _ptr(out16), _ptr(stat), rows, Cdim, eps, _type(out16), _stream()), "vda_round_rowstats")


def rowstats_finish(part, stat, rows, slots, eps):
    """Partial sums [rows, slots, 2] written by a product with ``row_part`` -> ``stat`` [rows, 2] = (mean, rstd)."""
    _dev(part, stat)
    _need("vda_rowstats_finish", part=(part, 2 * rows * slots), stat=(stat, 2 * rows))
    _check(lib().vda_rowstats_finish(_ptr(part), _ptr(stat), rows, slots, eps, _stream()), "vda_rowstats_finish")


def groupnorm(x, out, weight, bias, work, frames, hw, Cdim, groups, eps):
    _dev(x, out, weight, bias, work)
    _need("vda_groupnorm", x=(x, frames * hw * Cdim), out=(out, frames * hw * Cdim), weight=(weight, Cdim), bias=(bias, Cdim),
          work=(work, frames * GROUPNORM_MAX_SPLITS * groups * 2))
    if work.dtype != torch.float64:
        raise TypeError("vda_groupnorm: work is a float64 buffer")
    _check(lib().vda_groupnorm(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), _ptr(work), frames, hw, Cdim, groups, eps,
                               _type(x), _stream()), "vda_groupnorm")


GROUPNORM_STATIC_MAX_FRAMES = 64     # VDA_GROUPNORM_STATIC_MAX_FRAMES


def groupnorm_static(x, out, weight, bias, frames, hw, Cdim, groups, eps):
    """vda_groupnorm with the library's own scratch array (no workspace, no allocation); the tensors' device must be current."""
    _dev(x, out, weight, bias)
    _need("vda_groupnorm_static", x=(x, frames * hw * Cdim), out=(out, frames * hw * Cdim), weight=(weight, Cdim), bias=(bias, Cdim))
    _check(lib().vda_groupnorm_static(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), frames, hw, Cdim, groups, eps, _type(x),
                                      _stream()), "vda_groupnorm_static")


def bilinear_nhwc(x, out, frames, in_h, in_w, out_h, out_w, Cdim):
    _dev(x, out)
    _need("vda_bilinear_nhwc", x=(x, frames * in_h * in_w * Cdim), out=(out, frames * out_h * out_w * Cdim))
    _check(lib().vda_bilinear_nhwc(_ptr(x), _ptr(out), frames, in_h, in_w, out_h, out_w, Cdim, _type(x), _stream()),
           "vda_bilinear_nhwc")


def bilinear_nhwc_pieces(x, out, frames, in_h, in_w, out_h, out_w, Cdim, scale, layout=PIECES_TRIPLE, sat_flag=None):
    """fp32 NHWC map -> the 16-bit operand pieces of ``scale`` times its bilinear resampling (``out``'s type, ``layout``)."""
    _dev(x, out, sat_flag)
    if x.dtype != torch.float32 or (sat_flag is not None and sat_flag.dtype != torch.int32):
        raise TypeError("vda_bilinear_nhwc_pieces: fp32 map, int32 flag")
    _need("vda_bilinear_nhwc_pieces", x=(x, frames * in_h * in_w * Cdim), out=(out, frames * out_h * out_w * pieces_width(layout, Cdim)),
          sat_flag=(sat_flag, 1))
    _check(lib().vda_bilinear_nhwc_pieces(_ptr(x), _ptr(out), frames, in_h, in_w, out_h, out_w, Cdim, _type(out), float(scale),
                                          int(layout), _ptr(sat_flag), _stream()), "vda_bilinear_nhwc_pieces")


def bilinear_f32(x, out, frames, in_h, in_w, out_h, out_w, relu=False):
    _dev(x, out)
    _need("vda_bilinear_f32", x=(x, frames * in_h * in_w), out=(out, frames * out_h * out_w))
    _check(lib().vda_bilinear_f32(_ptr(x), _ptr(out), frames, in_h, in_w, out_h, out_w, int(relu), _stream()), "vda_bilinear_f32")


def im2col_patch14(x, out, frames, H, W, ldk):
    _dev(x, out)
    _need("vda_im2col_patch14", x=(x, frames * 3 * H * W), out=(out, frames * (H // 14) * (W // 14) * ldk))
    _check(lib().vda_im2col_patch14(_ptr(x), _ptr(out), frames, H, W, ldk, _type(out), _stream()), "vda_im2col_patch14")


def assemble_tokens(tok, cls, pos, x, frames, hw, D):
    _dev(tok, cls, pos, x)
    _need("vda_assemble_tokens", tok=(tok, frames * hw * D), cls=(cls, D), pos=(pos, (hw + 1) * D), x=(x, frames * (hw + 1) * D))
    _check(lib().vda_assemble_tokens(_ptr(tok), _ptr(cls), _ptr(pos), _ptr(x), frames, hw, D, _stream()), "vda_assemble_tokens")


def temporal_attention(qkv, out, batch, T, hw, heads, hd, scale):
    _dev(qkv, out)
    _need("vda_temporal_attention", qkv=(qkv, batch * T * hw * 3 * heads * hd), out=(out, batch * T * hw * heads * hd))
    _check(lib().vda_temporal_attention(_ptr(qkv), _ptr(out), batch, T, hw, heads, hd, scale, _type(qkv), _stream()),
           "vda_temporal_attention")


def scale_shift_sums(pred, target, sums, n):
    _dev(pred, target, sums)
    if sums.dtype != torch.float64:
        raise TypeError("vda_scale_shift_sums: sums is a float64 buffer")
    _need("vda_scale_shift_sums", pred=(pred, n), target=(target, n), sums=(sums, LSQ_SUMS_DOUBLES))
    _check(lib().vda_scale_shift_sums(_ptr(pred), _ptr(target), _ptr(sums), n, _stream()), "vda_scale_shift_sums")


def solve_scale_shift(sums, ss):
    _dev(sums, ss)
    if sums.dtype != torch.float64 or ss.dtype != torch.float32:
        raise TypeError("vda_solve_scale_shift: float64 sums, float32 result")
    _need("vda_solve_scale_shift", sums=(sums, 5), ss=(ss, 2))
    _check(lib().vda_solve_scale_shift(_ptr(sums), _ptr(ss), _stream()), "vda_solve_scale_shift")


def affine_blend(x, pre, out, ss, w_pre, w_post, n):
    _dev(x, pre, out, ss)
    _need("vda_affine_blend", x=(x, n), pre=(pre, n), out=(out, n), ss=(ss, 2))
    _check(lib().vda_affine_blend(_ptr(x), _ptr(pre), _ptr(out), _ptr(ss), w_pre, w_post, n, _stream()), "vda_affine_blend")


def preprocess_u8(frames_u8, out, frames, h0, w0, H, W):
    _dev(frames_u8, out)
    _need("vda_preprocess_u8", frames_u8=(frames_u8, frames * h0 * w0 * 3), out=(out, frames * 3 * H * W))
    _check(lib().vda_preprocess_u8(_ptr(frames_u8), _ptr(out), frames, h0, w0, H, W, _stream()), "vda_preprocess_u8")


def sample_points(depth, xy, out, frames, h, w):
    """out[i] = depth[i, xy[i, 1], xy[i, 0]]; xy int32 [frames, 2] (x, y) on the device, already inside the map."""
    _dev(depth, xy, out)
    if xy.dtype != torch.int32 or depth.dtype != torch.float32:
        raise TypeError("vda_sample_points: fp32 maps and int32 coordinates")
    _need("vda_sample_points", depth=(depth, frames * h * w), xy=(xy, 2 * frames), out=(out, frames))
    _check(lib().vda_sample_points(_ptr(depth), _ptr(xy), _ptr(out), frames, h, w, _stream()), "vda_sample_points")


def scale_frames(depth, scales, out, frames, hw):
    _dev(depth, scales, out)
    _need("vda_scale_frames", depth=(depth, frames * hw), scales=(scales, frames), out=(out, frames * hw))
    _check(lib().vda_scale_frames(_ptr(depth), _ptr(scales), _ptr(out), frames, hw, _stream()), "vda_scale_frames")


MINMAX_FLOATS = 2 + 2 * 256        # VDA_MINMAX_FLOATS


def minmax_f32(x, minmax, n):
    _dev(x, minmax)
    _need("vda_minmax_f32", x=(x, n), minmax=(minmax, MINMAX_FLOATS))
    _check(lib().vda_minmax_f32(_ptr(x), _ptr(minmax), n, _stream()), "vda_minmax_f32")


def colorize_u8(depth, minmax, lut, out, n):
    _dev(depth, minmax, lut, out)
    _need("vda_colorize_u8", depth=(depth, n), minmax=(minmax, 2), lut=(lut, 3 * 256), out=(out, n if lut is None else 3 * n))
    _check(lib().vda_colorize_u8(_ptr(depth), _ptr(minmax), _ptr(lut), _ptr(out), n, _stream()), "vda_colorize_u8")


CMAP_ROWS = 259        # a colour table: matplotlib's 256 entries, then the under, over and bad colours


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError("fp32 maps")


def colormap_u8(x, lut, out, n, sub, div, clip_lo=-float("inf"), clip_hi=float("inf"), cap=float("inf"), f64=False):
    _dev(x, lut, out)
    _f32(x)
    _need("vda_colormap_u8", x=(x, n), lut=(lut, 3 * CMAP_ROWS), out=(out, 3 * n))
    _check(lib().vda_colormap_u8(_ptr(x), _ptr(lut), _ptr(out), n, float(sub), float(div), float(clip_lo), float(clip_hi), float(cap),
                                 int(bool(f64)), _stream()), "vda_colormap_u8")


def colorize_zones_u8(x, lut_critical, lut_traffic, out, n, critical_max, traffic_max, traffic_range):
    _dev(x, lut_critical, lut_traffic, out)
    _f32(x)
    _need("vda_colorize_zones_u8", x=(x, n), lut_critical=(lut_critical, 3 * CMAP_ROWS), lut_traffic=(lut_traffic, 3 * CMAP_ROWS), out=(out, 3 * n))
    _check(lib().vda_colorize_zones_u8(_ptr(x), _ptr(lut_critical), _ptr(lut_traffic), _ptr(out), n, float(critical_max),
                                       float(traffic_max), float(traffic_range), _stream()), "vda_colorize_zones_u8")


def range_focus_f32(x, out, n, far=float("inf"), far_weight=1.0, stretch=False, sub=0.0, div=1.0, clip_lo=0.0, clip_hi=1.0, f64=False):
    _dev(x, out)
    _f32(x, out)
    _need("vda_range_focus_f32", x=(x, n), out=(out, n))
    _check(lib().vda_range_focus_f32(_ptr(x), _ptr(out), n, float(far), float(far_weight), int(bool(stretch)), float(sub), float(div),
                                     float(clip_lo), float(clip_hi), int(bool(f64)), _stream()), "vda_range_focus_f32")


def quantize_u8(x, out, n, lo, rng):
    _dev(x, out)
    _f32(x)
    _need("vda_quantize_u8", x=(x, n), out=(out, n))
    _check(lib().vda_quantize_u8(_ptr(x), _ptr(out), n, float(lo), float(rng), _stream()), "vda_quantize_u8")


def histogram_u8(x, hist, n):
    _dev(x, hist)
    if x.dtype != torch.uint8 or hist.dtype != torch.int32:
        raise TypeError("vda_histogram_u8: bytes in, 256 int32 bins")
    _need("vda_histogram_u8", x=(x, n), hist=(hist, 256))
    _check(lib().vda_histogram_u8(_ptr(x), _ptr(hist), n, _stream()), "vda_histogram_u8")


def colormap_bytes_u8(g, remap, lut, out, n):
    _dev(g, remap, lut, out)
    _need("vda_colormap_bytes_u8", g=(g, n), remap=(remap, 256), lut=(lut, 3 * CMAP_ROWS), out=(out, 3 * n))
    _check(lib().vda_colormap_bytes_u8(_ptr(g), _ptr(remap), _ptr(lut), _ptr(out), n, _stream()), "vda_colormap_bytes_u8")


def warp_blend_f32(prev, cur, flow, out, h, w, w_cur, w_prev):
    _dev(prev, cur, flow, out)
    _f32(prev, cur, flow, out)
    if out.data_ptr() == prev.data_ptr():
        raise ValueError("vda_warp_blend_f32: out must not alias prev")
    _need("vda_warp_blend_f32", prev=(prev, h * w), cur=(cur, h * w), flow=(flow, 2 * h * w), out=(out, h * w))
    _check(lib().vda_warp_blend_f32(_ptr(prev), _ptr(cur), _ptr(flow), _ptr(out), h, w, float(w_cur), float(w_prev), _stream()),
           "vda_warp_blend_f32")
