"""Repack a reference-layout ``state_dict`` into the layouts the HIP kernels read.

Everything here runs once per (weights, device): it is preparation, not the per-window path.  Layout rules:
  * every GEMM weight is bf16 ``[N][K]`` (K contiguous) with K and all channel counts zero-padded to multiples of 64;
  * 3x3 conv weights ``[Cout][Cin][3][3]`` become ``[Cout][tap = ky*3+kx][Cin]``;
  * ConvTranspose weights ``[Cin][Cout][k][k]`` become ``[(ky*k+kx)*Cout + co][Cin]``;
  * the attention scale 64^-1/2 times log2(e) is folded into the q rows of the fused QKV weight and bias (in fp32, before
    the one rounding to the operand type): vda_attention works on base-2 scores;
  * the GEGLU projection rows are interleaved value/gate in blocks of 16 (see VDA_EPI_GEGLU);
  * biases, LayerScale, norm parameters and the temporal position tables stay fp32.
"""
import math
import os

ATTENTION_Q_SCALE = math.log2(math.e) / 8.0      # VDA_ATTENTION_Q_SCALE of the C header

import numpy as np
import torch

from .spec import ENCODERS, POS_GRID, TEMPORAL_HEADS

BF16 = torch.bfloat16


def pad64(n):
    return (n + 63) // 64 * 64


def _cubic_weights(t, a=-0.75):
    """Keys cubic convolution coefficients for the 4 taps at offsets -1, 0, 1, 2 around floor(src)."""
    def near(x):    # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1

    def far(x):     # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a

    return np.stack([far(t + 1), near(t), near(1 - t), far(2 - t)], axis=-1)


def resample_pos_embed(pos_embed, hp, wp):
    """Positional table ``[1 + hp*wp, D]`` (float32 numpy) for an hp x wp patch grid.

    Reference: dinov2.py:179-210.  The stored 37x37 table is used as is for the square 37x37 grid; otherwise its
    patch part is resampled with torch's ``upsample_bicubic2d`` rule: align_corners=False, no antialias, the source
    coordinate taken from the GIVEN scale factor (n + 0.1) / 37, ``src = (dst + 0.5) / scale - 0.5``, Keys kernel
    with A = -0.75, source indices clamped to the border.
    """
    pe = np.asarray(pos_embed, dtype=np.float32).reshape(-1, pos_embed.shape[-1])
    g = POS_GRID
    if hp == g and wp == g:
        return pe.copy()
    grid = pe[1:].reshape(g, g, -1).astype(np.float64)

    def axis_taps(n_out):
        scale = float(n_out + 0.1) / g
        # torch computes the source coordinate in float32 (scale = 1 / scale_factor, cast to float)
        inv = np.float32(1.0 / scale)
        src = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * inv - np.float32(0.5)
        i0 = np.floor(src).astype(np.int64)
        t = (src - i0.astype(np.float32)).astype(np.float64)
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, g - 1)
        return idx, _cubic_weights(t)

    iy, wy = axis_taps(hp)
    ix, wx = axis_taps(wp)
    rows = (grid[iy] * wy[:, :, None, None]).sum(1)             # [hp, g, D]
    out = (rows[:, ix] * wx[None, :, :, None]).sum(2)            # [hp, wp, D]
    return np.concatenate([pe[:1], out.reshape(hp * wp, -1).astype(np.float32)], axis=0)


# 16-bit operand type of the encoder in the 16-bit mode ("bf16" or "fp16"; same MFMA rate, 8 against 11 significand bits)
ENCODER_DTYPE = {"bf16": BF16, "fp16": torch.float16}[os.environ.get("VDA_ENCODER_DTYPE", "bf16")]
SPLIT_DTYPE = torch.float16      # piece type of the fp32 mode's hi/lo operands (torch.bfloat16: 8-bit pieces, no scaling)
ACT_SPLIT_SCALE = 16.0           # power of two every fp32 activation is multiplied by before it is split into halves:
#                                  values down to 2^-7 keep a normal low half, values up to 4094 fit the high half


# layout of an fp32 operand's 16-bit pieces along K (VDA_PIECES_* of the C header): TRIPLE = [hi | lo | hi] against weights
# [hi | hi | lo], a plain product over 3K columns; PAIRS = every 32 columns stored as [hi 32 | lo 32] on both sides, which the
# kernels multiply as hi*hi + hi*lo + lo*hi from one copy of each piece (VdaGemmArgs.ab_pairs)
PIECES_TRIPLE, PIECES_PAIRS = 0, 1
PIECES_LAYOUT = PIECES_TRIPLE if os.environ.get("VDA_PIECES_LAYOUT", "pairs") == "triple" else PIECES_PAIRS
MIN_ACT_SPLIT_SCALE = 2.0 ** -8  # the engine lowers the scale by factors of 16 when a value saturates, down to this (values to 1.6e7)


class SplitWeight:
    """A weight matrix packed for the fp32 mode: ``t`` = the 16-bit pieces of ``scale * w`` along the last axis, [..., 3K] laid
    out [hi | hi | lo] (PIECES_TRIPLE; the activation side is [hi | lo | hi], see vda_split3) or [..., 2K] with every 32
    columns as [hi 32 | lo 32] (PIECES_PAIRS, both sides alike); ``scale`` is the power of two that puts the largest |w| just
    below 2048, so that the low halves are normal IEEE halves."""

    def __init__(self, w, dtype=None, layout=None):
        dtype = dtype or SPLIT_DTYPE
        self.layout = PIECES_LAYOUT if layout is None else layout
        self.scale = 1.0
        if dtype == torch.float16:
            top = float(w.abs().max())
            self.scale = 2.0 ** (10 - math.floor(math.log2(top))) if top > 0 else 1.0
        ws = w * self.scale
        hi = ws.to(dtype)
        lo = (ws - hi.float()).to(dtype)
        if self.layout == PIECES_PAIRS:         # [..., K] -> [..., K/32, (hi, lo), 32] -> [..., 2K]
            K = w.shape[-1]
            if K % 32:
                raise ValueError(f"pairs layout: K = {K} is not a multiple of 32")
            blocks = w.shape[:-1] + (K // 32, 32)
            self.t = torch.stack([hi.reshape(blocks), lo.reshape(blocks)], dim=-2).reshape(w.shape[:-1] + (2 * K,)).contiguous()
        else:
            self.t = torch.cat([hi, hi, lo], dim=-1).contiguous()

    @property
    def shape(self):
        return self.t.shape

    def reshape(self, *shape):
        self.t = self.t.reshape(*shape)
        return self


def split_weight(w):
    return SplitWeight(w)


# precision -> which parts keep fp32 activations and multiply three-term hi/lo products: (encoder, head, output stage).
# "half" and "fp32" are the two modes of the model; the mixed ones exist to measure what each part contributes to the error
# (tools/clamped_parity.py): "enc32" = fp32 encoder + 16-bit head, "head32" = bf16 encoder + fp32 head, "tail32" = 16-bit
# everywhere except output_conv1 / output_conv2 (the reference's own fp32 island, dpt_temporal.py:97-100, and the
# convolution in front of it).
PRECISIONS = {"half": (False, False, False), "fp32": (True, True, True), "enc32": (True, False, False),
              "head32": (False, True, True), "tail32": (False, False, True)}


class PackedWeights:
    """Device-resident kernel-layout weights of one model.  ``precision`` "fp32" packs every matrix as 16-bit hi/lo pieces
    (``SplitWeight``, in ``layout``) for the fp32 storage mode (``head_dtype`` is then ignored); see PRECISIONS for the mixed forms."""

    def __init__(self, sd, encoder, features, out_channels, device, head_dtype=torch.float16, precision="half", layout=None,
                 encoder_dtype=None):
        self.device = device
        self.precision = precision
        self.layout = PIECES_LAYOUT if layout is None else layout        # of the fp32 mode's operand pieces (PIECES_*)
        split_weight = lambda t: SplitWeight(t, layout=self.layout)  # noqa: E731
        self.split_enc, self.split_head, self.split_tail = PRECISIONS[precision]
        self.head_dtype = torch.float32 if self.split_head else head_dtype
        encoder_dtype = encoder_dtype or ENCODER_DTYPE
        if encoder_dtype not in (torch.float16, BF16):
            raise ValueError("encoder_dtype: torch.float16 or torch.bfloat16")
        self.encoder_dtype = torch.float32 if self.split_enc else encoder_dtype
        self.encoder = encoder
        D, depth, heads, taps = ENCODERS[encoder]
        self.D, self.depth, self.heads, self.taps = D, depth, heads, taps
        self.F = features
        self.oc = list(out_channels)
        self.Fp = pad64(features)
        self.F2, self.F2p = features // 2, pad64(features // 2)
        self.ocp = [pad64(c) for c in self.oc]
        self._pos_cache = {}
        g = lambda k: sd[k].detach().to(device=device, dtype=torch.float32)  # noqa: E731
        f32 = lambda t: t.contiguous()  # noqa: E731
        bf = split_weight if self.split_enc else (lambda t: t.to(encoder_dtype).contiguous())
        hd_ = split_weight if self.split_head else (lambda t: t.to(head_dtype).contiguous())
        hd_all = hd_

        def padded(t, rows, cols):
            out = torch.zeros((rows, cols), dtype=torch.float32, device=device)
            out[:t.shape[0], :t.shape[1]] = t
            return out

        def padvec(t, n):
            out = torch.zeros(n, dtype=torch.float32, device=device)
            out[:t.shape[0]] = t
            return out

        def conv3(key, cout_p, cin_p):
            w = g(key)                                       # [Cout, Cin, 3, 3]
            out = torch.zeros((cout_p, 9, cin_p), dtype=torch.float32, device=device)
            out[:w.shape[0], :, :w.shape[1]] = w.permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1])
            return hd_(out).reshape(cout_p, -1)              # the hi/lo split, if any, is per filter tap

        def convt(key, k, cp):
            w = g(key)                                       # [Cin, Cout, k, k]
            out = torch.zeros((k * k, cp, cp), dtype=torch.float32, device=device)     # [tap, co, ci]
            out[:, :w.shape[1], :w.shape[0]] = w.permute(2, 3, 1, 0).reshape(k * k, w.shape[1], w.shape[0])
            return hd_(out.reshape(k * k * cp, cp))

        p = "pretrained."
        self.patch_w = bf(padded(g(p + "patch_embed.proj.weight").reshape(D, 588), D, 640))
        self.patch_b = f32(g(p + "patch_embed.proj.bias"))
        self.cls = f32(g(p + "cls_token").reshape(D))
        self.pos_embed_host = sd[p + "pos_embed"].detach().float().cpu().numpy()
        self.norm_w, self.norm_b = f32(g(p + "norm.weight")), f32(g(p + "norm.bias"))
        self.blocks = []
        for i in range(depth):
            b = f"{p}blocks.{i}."
            wqkv, bqkv = g(b + "attn.qkv.weight").clone(), g(b + "attn.qkv.bias").clone()
            wqkv[:D] *= ATTENTION_Q_SCALE
            bqkv[:D] *= ATTENTION_Q_SCALE
            # LayerScale (layer_scale.py:27-28) is folded into the projection it follows: gamma * (x W^T + b) =
            # x (gamma W)^T + gamma b, rounded to the operand type once, from the fp32 product.  The residual GEMMs then
            # have a plain "+ bias + residual" epilogue.
            ls1, ls2 = g(b + "ls1.gamma"), g(b + "ls2.gamma")
            ln1_w, ln1_b, ln2_w, ln2_b = g(b + "norm1.weight"), g(b + "norm1.bias"), g(b + "norm2.weight"), g(b + "norm2.bias")
            wfc1, bfc1 = g(b + "mlp.fc1.weight"), g(b + "mlp.fc1.bias")
            folded = {}
            if not self.split_enc:
                # LayerNorm folded into the product that follows it (block.py:84,87): with W' = W diag(ln_w) rounded to the
                # operand type,  LN(x) W^T + b = rstd (x W'^T - mean colsum(W')) + (b + W ln_b).  The product reads the rows
                # of x rounded to 16 bits (written, with their sums, by the product before it) and normalises its sums.
                for name, w_, b_, lw, lb in (("qkv", wqkv, bqkv, ln1_w, ln1_b), ("fc1", wfc1, bfc1, ln2_w, ln2_b)):
                    wl = bf(w_ * lw[None, :])
                    folded["w" + name + "_ln"] = wl
                    folded["cs_" + name] = f32(wl.double().sum(1).float())
                    # (an elementwise product and a row sum, not a matrix-vector product: BLAS routines may add in an order
                    # that changes from process to process, and ranks must pack bit-identical weights)
                    folded["b" + name + "_ln"] = f32((b_.double() + (w_.double() * lb.double()[None, :]).sum(1)).float())
            self.blocks.append(dict(
                **folded,
                ln1_w=f32(g(b + "norm1.weight")), ln1_b=f32(g(b + "norm1.bias")),
                wqkv=bf(wqkv), bqkv=f32(bqkv),
                wproj=bf(ls1[:, None] * g(b + "attn.proj.weight")), bproj=f32(ls1 * g(b + "attn.proj.bias")),
                ln2_w=f32(g(b + "norm2.weight")), ln2_b=f32(g(b + "norm2.bias")),
                wfc1=bf(wfc1), bfc1=f32(bfc1),
                wfc2=bf(ls2[:, None] * g(b + "mlp.fc2.weight")), bfc2=f32(ls2 * g(b + "mlp.fc2.bias")),
            ))

        h = "head."
        ocp, Fp, F2p = self.ocp, self.Fp, self.F2p
        self.proj_w = [hd_(padded(g(f"{h}projects.{i}.weight").reshape(self.oc[i], D), ocp[i], D)) for i in range(4)]
        self.proj_b = [padvec(g(f"{h}projects.{i}.bias"), ocp[i]) for i in range(4)]
        self.rs0_w, self.rs0_b = convt(h + "resize_layers.0.weight", 4, ocp[0]), padvec(g(h + "resize_layers.0.bias"), ocp[0])
        self.rs1_w, self.rs1_b = convt(h + "resize_layers.1.weight", 2, ocp[1]), padvec(g(h + "resize_layers.1.bias"), ocp[1])
        self.rs3_w, self.rs3_b = conv3(h + "resize_layers.3.weight", ocp[3], ocp[3]), padvec(g(h + "resize_layers.3.bias"), ocp[3])
        sc = h + "scratch."
        self.rn_w = [conv3(f"{sc}layer{i + 1}_rn.weight", Fp, ocp[i]) for i in range(4)]
        self.refine = {}
        for i in range(1, 5):
            r = f"{sc}refinenet{i}."
            d = dict(out_w=hd_(padded(g(r + "out_conv.weight").reshape(self.F, self.F), Fp, Fp)), out_b=padvec(g(r + "out_conv.bias"), Fp))
            for u in (1, 2):
                for c in (1, 2):
                    d[f"u{u}c{c}_w"] = conv3(f"{r}resConfUnit{u}.conv{c}.weight", Fp, Fp)
                    d[f"u{u}c{c}_b"] = padvec(g(f"{r}resConfUnit{u}.conv{c}.bias"), Fp)
            self.refine[i] = d
        hd_ = split_weight if self.split_tail else hd_all
        self.oc1_w, self.oc1_b = conv3(sc + "output_conv1.weight", F2p, Fp), padvec(g(sc + "output_conv1.bias"), F2p)
        self.oc2_w, self.oc2_b = conv3(sc + "output_conv2.0.weight", 32, F2p), f32(g(sc + "output_conv2.0.bias"))
        self.dot_w = f32(g(sc + "output_conv2.2.weight").reshape(32))
        self.dot_b = float(sd[sc + "output_conv2.2.bias"].detach().float().reshape(-1)[0].item())
        hd_ = hd_all

        self.motion = []
        for i, C in enumerate((self.oc[2], self.oc[3], self.F, self.F)):
            if C % 64 != 0 or C % TEMPORAL_HEADS != 0:
                raise ValueError(f"motion module {i}: channel count {C} must be a multiple of 64")
            t = f"{h}motion_modules.{i}.temporal_transformer."
            tb = t + "transformer_blocks.0."
            m = dict(C=C, gn_w=f32(g(t + "norm.weight")), gn_b=f32(g(t + "norm.bias")),
                     in_w=hd_(g(t + "proj_in.weight")), in_b=f32(g(t + "proj_in.bias")),
                     out_w=hd_(g(t + "proj_out.weight")), out_b=f32(g(t + "proj_out.bias")),
                     ffn_w=f32(g(tb + "ff_norm.weight")), ffn_b=f32(g(tb + "ff_norm.bias")),
                     ff2_w=hd_(g(tb + "ff.net.2.weight")), ff2_b=f32(g(tb + "ff.net.2.bias")), attn=[])
            w1, b1 = g(tb + "ff.net.0.proj.weight"), g(tb + "ff.net.0.proj.bias")          # [8C, C]: value rows then gate rows
            n = 4 * C
            m["ff1_w"] = hd_(torch.stack([w1[:n].reshape(n // 16, 16, C), w1[n:].reshape(n // 16, 16, C)], 1).reshape(8 * C, C))
            m["ff1_b"] = f32(torch.stack([b1[:n].reshape(n // 16, 16), b1[n:].reshape(n // 16, 16)], 1).reshape(8 * C))
            for a in (0, 1):
                ab = f"{tb}attention_blocks.{a}."
                m["attn"].append(dict(
                    ln_w=f32(g(f"{tb}norms.{a}.weight")), ln_b=f32(g(f"{tb}norms.{a}.bias")),
                    wqkv=hd_(torch.cat([g(ab + "to_q.weight"), g(ab + "to_k.weight"), g(ab + "to_v.weight")], 0)),
                    wo=hd_(g(ab + "to_out.0.weight")), bo=f32(g(ab + "to_out.0.bias")),
                    pe=f32(g(ab + "pos_encoder.pe").reshape(-1, C)),
                ))
            self.motion.append(m)

    def pos_table(self, hp, wp):
        key = (hp, wp)
        t = self._pos_cache.get(key)
        if t is None:
            t = torch.from_numpy(resample_pos_embed(self.pos_embed_host, hp, wp)).to(self.device).contiguous()
            self._pos_cache[key] = t
        return t
