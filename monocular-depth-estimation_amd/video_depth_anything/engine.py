"""The per-window forward pass as a sequence of libvda_hip kernel launches on the current HIP stream.

Layouts in HBM (T frames, hp x wp patches, N = hp*wp + 1 tokens, D embed dim):
  * encoder residual stream  fp32 [frame][token][D]; every GEMM operand 16-bit (bf16, or fp16 on request); accumulation fp32;
  * head feature maps        fp16 channels-last [frame][y][x][C] (C padded to a multiple of 64), so a 1x1 convolution or
                             Linear is a plain GEMM on the same buffer and the temporal modules need no transposes;
  * temporal hidden state    fp32 [frame][position][C];
  * depth                    fp32 [frame][H][W].
In the fp32 storage mode (``PackedWeights(precision="fp32")``, the reference's ``fp32=True``) every activation above is
fp32 instead of 16-bit; what a product reads are the operand's IEEE-half pieces hi = f16(s v), lo = f16(s v - hi) (weights
are packed the same way), written by the kernel that produced the values where there is one and by ``vda_split3`` otherwise,
and the kernels accumulate hi*hi + lo*hi + hi*lo (``packing.PIECES_PAIRS``: from one copy of each piece; ``PIECES_TRIPLE``:
as a plain product over K' = 3K).
torch is used for allocation (caching allocator = workspace reuse) and the stream only.
"""
import os
import warnings

import torch

from . import hip
from .packing import ACT_SPLIT_SCALE, MIN_ACT_SPLIT_SCALE, PIECES_PAIRS, SPLIT_DTYPE, SplitWeight
from .spec import TEMPORAL_HEADS

BF16 = torch.bfloat16


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


class Pieces:
    """Marks a 16-bit matrix that already holds the hi/lo pieces of an fp32 operand in the engine's layout (written by the
    kernel that produced the values), so that ``Engine.operand`` does not split it again."""

    def __init__(self, t):
        self.t = t


class SaturationError(FloatingPointError):
    pass


class Engine:
    def __init__(self, packed):
        self.w = packed
        self.dev = packed.device
        # fp32 storage + three-term products, per part of the network (packing.PRECISIONS); ``split`` is the setting of the
        # part whose launches are being enqueued
        self.split_enc, self.split_head, self.split_tail = packed.split_enc, packed.split_head, packed.split_tail
        self.split = self.split_enc
        self.hdt = packed.head_dtype         # storage/operand type of the DPT head and the temporal modules
        self.edt = packed.encoder_dtype      # storage type of the encoder's GEMM operands (fp32: operand pieces instead)
        # fp32 storage: the power of two every activation is multiplied by before it is split into IEEE halves.  16 keeps a
        # normal low half for values down to 2^-7 and holds |values| below 4094; a window in which some scaled value leaves
        # the half range (``sat`` is set by the kernel that saw it) is run again with the scale divided by 16, and the lower
        # scale is kept from then on (see forward)
        self.act_scale = ACT_SPLIT_SCALE if SPLIT_DTYPE == torch.float16 else 1.0
        self.sat = torch.zeros(1, dtype=torch.int32, device=self.dev) if self.dev.type == "cuda" else None
        # fp32 storage: the producers write the 16-bit pieces of their fp32 results themselves (LayerNorm, the QKV and fc1
        # products, attention) instead of an fp32 matrix that a vda_split3 pass then reads; False keeps the separate passes
        self.fused_split = SPLIT_DTYPE == torch.float16
        self.layout = packed.layout                       # of the operand pieces (packing.PIECES_*); the weights are packed alike
        self.pw = 2 if self.layout == PIECES_PAIRS else 3    # stored columns per operand column
        self.encoder_streams = 2      # frame groups of the encoder that run on concurrent streams (see encoder())
        self.side_streams = []
        self.head_streams = 2         # 2: parts of the head run on the first side stream beside the rest (see head())
        self.kept = None              # encoder outputs of the last keyed forward() (see forward)
        self.slot_index = {}          # device index tensors of the carried slots, by slot list (see encoder)
        # 16-bit mode: the encoder blocks' LayerNorms can be folded into the qkv / fc1 products (see encoder_frames).
        # None = where it pays: at ViT-S width (+1.2 % on the window).  At ViT-L width the separate pass already hides under
        # the other stream's matrix-bound kernels (two streams: no gain; one stream +0.7 %) while the residual products,
        # the kernel with the largest share of the window, slow down by the rows they additionally emit.  True / False
        # (VideoDepthAnything.fold_layernorm, or VDA_FOLD_LAYERNORM=1 / 0 when the engine is built) force it.
        env = os.environ.get("VDA_FOLD_LAYERNORM")
        self.fold_default = None if env is None else env != "0"
        self.fold_layernorm = self.fold_default

    # ---- small wrappers -------------------------------------------------------------------------------------------
    def operand(self, a, relu=False, act=None):
        """The MFMA operand of activation matrix ``a`` [rows, K]: itself, or its hi/lo pieces (``self.layout``) in fp32 mode
        (``act``: an activation the split applies first; GEGLU halves the width)."""
        if isinstance(a, Pieces):
            assert act is None and not relu
            return a.t, False
        if not self.split:
            assert act is None
            return a, relu
        rows, K = a.shape
        if act == hip.SPLIT_GEGLU:
            K //= 2
        a3 = _empty((rows, self.pw * K), SPLIT_DTYPE, self.dev)
        hip.split3(a, a3, rows, K, a.stride(0), K, relu=relu, act=act, scale=self.act_scale, sat_flag=self.sat, layout=self.layout)
        return a3, False

    def weight(self, w, kw):
        """The device tensor of a packed weight; for the fp32 mode's scaled hi/lo pieces also sets the GEMM's ``acc_scale``."""
        if isinstance(w, SplitWeight):
            kw["acc_scale"] = 1.0 / (w.scale * self.act_scale)
            kw["ab_pairs"] = w.layout == PIECES_PAIRS
            return w.t
        return w

    def linear(self, a, w, out, bias=None, a_act=None, **kw):
        a, _ = self.operand(a, act=a_act)
        M, K = a.shape[0], a.shape[-1]
        N = kw.pop("N", w.shape[0])
        w = self.weight(w, kw)
        hip.gemm(a, w, out, M=M, N=N, K=K, lda=a.stride(0), ldc=kw.pop("ldc", out.stride(0)), bias=bias, **kw)
        return out

    def conv3(self, x, w, bias, cout, stride=1, relu_in=False, out=None, ldc=None, resized=None, **kw):
        """x: [frames, h, w, cin] -> [frames, oh, ow, cout] (3x3, padding 1), both of the head's storage type.
        ``resized=(h', w')``: the convolution runs on the bilinear resampling of x to h' x w'.  With 16-bit storage the
        kernel resamples while it loads its input patches; with fp32 storage the resampling writes the operand pieces of the
        convolution: either way the resized map is never materialised in the storage type."""
        f, h, wd, cin = x.shape
        pre = None
        if resized is not None and tuple(resized) != (h, wd):
            if self.split and self.fused_split and stride == 1 and not relu_in and cin % 64 == 0:
                sh, sw, (h, wd) = h, wd, resized
                pre = Pieces(_empty((f * h * wd, self.pw * cin), SPLIT_DTYPE, self.dev))
                hip.bilinear_nhwc_pieces(x, pre.t, f, sh, sw, h, wd, cin, self.act_scale, self.layout, self.sat)
            elif self.split or stride != 1:
                x = self.resize(x, *resized)
                h, wd = resized
            else:
                sh, sw, (h, wd) = h, wd, resized
                if out is None:
                    out = _empty((f, h, wd, cout), self.hdt, self.dev)
                hip.gemm(x.reshape(f * sh * sw, cin), self.weight(w, kw), out, M=f * h * wd, N=cout, K=9 * cin, lda=0,
                         ldc=cout if ldc is None else ldc, bias=bias, relu_in=relu_in, conv=(h, wd, cin, h, wd, 1),
                         upsample_from=(sh, sw), **kw)
                return out
        oh, ow = (h - 1) // stride + 1, (wd - 1) // stride + 1
        if out is None:
            out = _empty((f, oh, ow, cout), self.hdt, self.dev)
        a, relu_in = self.operand(pre if pre is not None else x.reshape(f * h * wd, cin), relu_in)
        cin = a.shape[1]
        hip.gemm(a, self.weight(w, kw), out, M=f * oh * ow, N=cout, K=9 * cin, lda=0, ldc=cout if ldc is None else ldc, bias=bias,
                 relu_in=relu_in, conv=(h, wd, cin, oh, ow, stride), **kw)
        return out

    def resize(self, x, oh, ow):
        f, h, wd, c = x.shape
        if (h, wd) == (oh, ow):
            return x
        out = _empty((f, oh, ow, c), self.hdt, self.dev)
        hip.bilinear_nhwc(x, out, f, h, wd, oh, ow, c)
        return out

    # ---- encoder (reference: dinov2.py:212-231, 271-281, 297-321; block.py:82-107) --------------------------------------
    def encoder(self, x, frames, H, W, carried=None):
        """The four tapped, normalised token maps [frames * hw, D].  The encoder treats every frame on its own, so the frames
        are split into ``self.encoder_streams`` groups that run on separate HIP streams: while one group's workgroups drain
        at the end of a kernel (a GEMM grid rarely fills its last round of tiles) or wait on HBM (LayerNorm, residual
        epilogues), the other group's matrix-bound kernels take the idle CUs.  Same kernels on the same rows: same bits (in
        the head the number of frames per launch can change which convolution kernel runs, and with it the order of a sum:
        one stream against two agree to rounding there, see head()).

        ``carried = (previous taps, slots)``: the first ``len(slots)`` frames are the frames at ``slots`` of an earlier call
        whose tap buffers are given; their rows are copied instead of computed (a frame's encoder output does not depend on
        the other frames of its window, and consecutive windows of a video share ten of 32 frames, video_depth.py:157-164)."""
        w, dev = self.w, self.dev
        hw = (H // 14) * (W // 14)
        # (fp32 head: the tap norms write the operand pieces of the head's first products directly, see Pieces)
        self.taps_split = self.split_head and self.fused_split
        taps = [_empty((frames * hw, self.pw * w.D), SPLIT_DTYPE, dev) if self.taps_split else _empty((frames * hw, w.D), self.hdt, dev)
                for _ in w.taps]
        first = 0
        if carried is not None:
            prev, slots = carried
            first = len(slots)
            # (kept per slot list: building a device tensor from a Python list is a blocking copy, which made the host wait
            # for the device once per window; a video's windows carry the same slots over)
            idx = self.slot_index.get(tuple(slots))
            if idx is None:
                idx = self.slot_index[tuple(slots)] = torch.tensor(slots, dtype=torch.int64, device=dev)
            for t, p in zip(taps, prev):
                torch.index_select(p.view(-1, hw * p.shape[1]), 0, idx, out=t[:first * hw].view(first, hw * t.shape[1]))
        if first == frames:
            return taps
        w.pos_table(H // 14, W // 14)             # built (and uploaded) on this stream before any other stream reads it
        groups = min(self.encoder_streams, frames - first)
        if groups <= 1:
            self.encoder_frames(x, first, frames, H, W, taps)
            return taps
        main = torch.cuda.current_stream()
        if len(self.side_streams) < groups - 1:
            self.side_streams += [torch.cuda.Stream(device=dev) for _ in range(groups - 1 - len(self.side_streams))]
        fork = torch.cuda.Event()
        fork.record(main)
        bounds = [first + (frames - first) * g // groups for g in range(groups + 1)]
        joins = []
        for g in range(1, groups):
            side = self.side_streams[g - 1]
            side.wait_event(fork)                     # x and the tap buffers exist
            with torch.cuda.stream(side):
                self.encoder_frames(x, bounds[g], bounds[g + 1], H, W, taps)
                done = torch.cuda.Event()
                done.record(side)
            joins.append(done)
        self.encoder_frames(x, bounds[0], bounds[1], H, W, taps)
        for done in joins:
            main.wait_event(done)
        return taps

    def folds_layernorms(self):
        """Whether the encoder blocks' LayerNorms are folded into qkv / fc1 on this engine."""
        D = self.w.D
        # (by default not with IEEE-half operands: the folded form stores the UN-normalised residual rows in the operand type,
        # and a residual value beyond 65504 would saturate there without notice; bf16 has fp32's range)
        auto = D < 1024 and self.edt != torch.float16
        return (auto if self.fold_layernorm is None else bool(self.fold_layernorm)) and not self.split_enc and D % 64 == 0

    def encoder_frames(self, x, f0, f1, H, W, taps):
        """Frames f0..f1-1 through the encoder on the current stream; writes their rows of the tap buffers."""
        w, dev = self.w, self.dev
        D, heads = w.D, w.heads
        self.split = self.split_enc
        frames = f1 - f0
        x = x[f0:f1]
        hp, wp = H // 14, W // 14
        hw, N = hp * wp, hp * wp + 1
        M = frames * N
        cols = _empty((frames * hw, 640), self.edt, dev)
        hip.im2col_patch14(x, cols, frames, H, W, 640)
        tok = _empty((frames * hw, D), torch.float32, dev)
        self.linear(cols, w.patch_w, tok, w.patch_b)
        xs = _empty((M, D), torch.float32, dev)
        hip.assemble_tokens(tok, w.cls, w.pos_table(hp, wp), xs, frames, hw, D)
        del cols, tok
        pieces = self.split and self.fused_split       # fp32 storage with the operand pieces written by the producers
        S = dict(split_scale=self.act_scale, sat_flag=self.sat)
        LN = dict(split_layout=self.layout, **S)
        tap_kw = LN if self.taps_split else {}
        if pieces:
            pw = self.pw
            y = _empty((M, pw * D), SPLIT_DTYPE, dev)        # the pieces of the normed rows
            qkv = _empty((M, 6 * D), SPLIT_DTYPE, dev)       # [hi | lo] planes of q, k, v
            att = _empty((M, pw * D), SPLIT_DTYPE, dev)
            hid = _empty((M, pw * 4 * D), SPLIT_DTYPE, dev)  # the pieces of GELU(fc1)
        else:
            y = _empty((M, D), self.edt, dev)
            qkv = _empty((M, 3 * D), self.edt, dev)
            att = _empty((M, D), self.edt, dev)
            hid = _empty((M, 4 * D), self.edt, dev)
        tap = 0
        # Folded LayerNorm (block.py:84,87): no normalisation pass over the residual stream.  The product that updates the
        # stream (proj, fc2) also writes the updated rows rounded to 16 bits into y, with partial sums per 64 columns;
        # vda_rowstats_finish turns the sums into (mean, rstd) per row; the next product (fc1, qkv) reads y as it is, with
        # weights that carry the norm's scale, and normalises its sums in the epilogue.  The very first norm reads rows no
        # product wrote: one rounding pass (vda_round_rowstats).
        fold = self.folds_layernorms()
        if fold:
            slots = D // 64
            part = _empty((M, slots, 2), torch.float32, dev)
            stat = _empty((M, 2), torch.float32, dev)
            hip.round_rowstats(xs, y, stat, M, D, 1e-6)
            last = len(w.blocks) - 1
        for i, b in enumerate(w.blocks):
            if fold:
                self.linear(y, b["wqkv_ln"], qkv, b["bqkv_ln"], row_stat=stat, w_colsum=b["cs_qkv"])
                hip.attention(qkv, att, frames, N, heads)
                self.linear(att, b["wproj"], xs, b["bproj"], resid=xs, ldr=D, out16=y, row_part=part)
                hip.rowstats_finish(part, stat, M, slots, 1e-6)
                self.linear(y, b["wfc1_ln"], hid, b["bfc1_ln"], act=hip.ACT_GELU, row_stat=stat, w_colsum=b["cs_fc1"])
                if i == last:       # nothing reads the rounded rows of the last block (the tap norm below reads the fp32 stream)
                    self.linear(hid, b["wfc2"], xs, b["bfc2"], resid=xs, ldr=D)
                else:
                    self.linear(hid, b["wfc2"], xs, b["bfc2"], resid=xs, ldr=D, out16=y, row_part=part)
                    hip.rowstats_finish(part, stat, M, slots, 1e-6)
            elif pieces:
                hip.layernorm(xs, y, b["ln1_w"], b["ln1_b"], M, D, 1e-6, **LN)
                self.linear(Pieces(y), b["wqkv"], qkv, b["bqkv"], split_out=2, **S)
                hip.attention_pairs(qkv, att, frames, N, heads, self.act_scale, out_layout=self.layout)
                self.linear(Pieces(att), b["wproj"], xs, b["bproj"], resid=xs, ldr=D)
                hip.layernorm(xs, y, b["ln2_w"], b["ln2_b"], M, D, 1e-6, **LN)
                self.linear(Pieces(y), b["wfc1"], hid, b["bfc1"], act=hip.ACT_GELU, split_out=4 if pw == 2 else 3, **S)      # the exact GELU
                self.linear(Pieces(hid), b["wfc2"], xs, b["bfc2"], resid=xs, ldr=D)
            else:
                hip.layernorm(xs, y, b["ln1_w"], b["ln1_b"], M, D, 1e-6)
                self.linear(y, b["wqkv"], qkv, b["bqkv"])
                hip.attention(qkv, att, frames, N, heads)
                self.linear(att, b["wproj"], xs, b["bproj"], resid=xs, ldr=D)         # LayerScale is folded into the weights
                hip.layernorm(xs, y, b["ln2_w"], b["ln2_b"], M, D, 1e-6)
                if self.split:      # fp32 storage: fc1 stores its pre-activation, the exact GELU runs in fc2's operand split
                    self.linear(y, b["wfc1"], hid, b["bfc1"])
                    self.linear(hid, b["wfc2"], xs, b["bfc2"], a_act=hip.SPLIT_GELU, resid=xs, ldr=D)
                else:
                    self.linear(y, b["wfc1"], hid, b["bfc1"], act=hip.ACT_GELU)
                    self.linear(hid, b["wfc2"], xs, b["bfc2"], resid=xs, ldr=D)
            if i in w.taps:
                t = taps[tap][f0 * hw:f1 * hw]
                tap += 1
                hip.layernorm(xs, t, w.norm_w, w.norm_b, frames * hw, D, 1e-6, rows_per_group=hw, in_group=N, in_skip=1, **tap_kw)

    # ---- temporal module (reference: motion_module.py:102-126, 164-177, 230-297) ----------------------------------------
    def motion(self, idx, x, B, T):
        m, dev = self.w.motion[idx], self.dev
        f, h, wd, C = x.shape
        hw = h * wd
        M = f * hw
        xf = x.reshape(M, C)
        work = _empty((f * hip.GROUPNORM_MAX_SPLITS * 32 * 2,), torch.float64, dev)
        g = _empty((M, C), self.hdt, dev)
        hip.groupnorm(xf, g, m["gn_w"], m["gn_b"], work, f, hw, C, 32, 1e-6)
        hs = _empty((M, C), torch.float32, dev)
        self.linear(g, m["in_w"], hs, m["in_b"])
        pieces = self.split and self.fused_split
        if pieces:      # fp32 storage: the norms write the operand pieces of the products that follow them
            n = Pieces(_empty((M, self.pw * C), SPLIT_DTYPE, dev))
            ln_kw = dict(split_scale=self.act_scale, sat_flag=self.sat, split_layout=self.layout)
            n_out = n.t
        else:
            n, ln_kw = g, {}                             # reuse as the normed-activation buffer
            n_out = n
        qkv = _empty((M, 3 * C), self.hdt, dev)
        o = _empty((M, C), self.hdt, dev)
        hd = C // TEMPORAL_HEADS
        for a in m["attn"]:
            hip.layernorm(hs, n_out, a["ln_w"], a["ln_b"], M, C, 1e-5, pe=a["pe"], pe_hw=hw, pe_T=T, **ln_kw)
            self.linear(n, a["wqkv"], qkv)
            hip.temporal_attention(qkv, o, B, T, hw, TEMPORAL_HEADS, hd, hd ** -0.5)
            self.linear(o, a["wo"], hs, a["bo"], resid=hs, ldr=C)
        hip.layernorm(hs, n_out, m["ffn_w"], m["ffn_b"], M, C, 1e-5, **ln_kw)
        if self.split:          # fp32 storage: plain projection to [values | gates], the gated GELU runs in ff2's operand split
            ff = _empty((M, 8 * C), self.hdt, dev)
            self.linear(n, m["ff1_w"], ff, m["ff1_b"])
            self.linear(ff, m["ff2_w"], o, m["ff2_b"], a_act=hip.SPLIT_GEGLU, resid=hs, ldr=C)
        else:
            ff = _empty((M, 4 * C), self.hdt, dev)
            self.linear(n, m["ff1_w"], ff, m["ff1_b"], epi=hip.EPI_GEGLU)
            self.linear(ff, m["ff2_w"], o, m["ff2_b"], resid=hs, ldr=C)      # 16-bit out: only proj_out reads it
        out = _empty((f, h, wd, C), self.hdt, dev)
        self.linear(o, m["out_w"], out.reshape(M, C), m["out_b"], resid=xf, ldr=C)
        return out

    # ---- RefineNet fusion (reference: util/blocks.py:68-91, 135-162) -----------------------------------------------------
    def fusion(self, idx, x0, x1, oh, ow, resize=True):
        r, Fp = self.w.refine[idx], self.w.Fp
        y = x0
        if x1 is not None:
            t = self.conv3(x1, r["u1c1_w"], r["u1c1_b"], Fp, relu_in=True, act=hip.ACT_RELU)
            y = self.conv3(t, r["u1c2_w"], r["u1c2_b"], Fp, resid=x1.reshape(-1, Fp), ldr=Fp, resid2=x0.reshape(-1, Fp), ldr2=Fp)
        t = self.conv3(y, r["u2c1_w"], r["u2c1_b"], Fp, relu_in=True, act=hip.ACT_RELU)
        z = self.conv3(t, r["u2c2_w"], r["u2c2_b"], Fp, resid=y.reshape(-1, Fp), ldr=Fp)
        # the 1x1 out_conv is applied BEFORE the bilinear resize: both are linear and the interpolation weights sum to
        # one, so the result is the same function at a quarter of the multiply-adds of the reference order
        f, h, wd, _ = z.shape
        zc = _empty((f, h, wd, Fp), self.hdt, self.dev)
        self.linear(z.reshape(-1, Fp), r["out_w"], zc.reshape(-1, Fp), r["out_b"])
        return self.resize(zc, oh, ow) if resize else zc

    # ---- DPT head (reference: dpt_temporal.py:53-114, dpt.py:60-124) ---------------------------------------------------
    def head(self, taps, B, T, hp, wp):
        w, dev = self.w, self.dev
        self.split = self.split_head
        frames = B * T
        hw = hp * wp
        proj = []
        for i in range(4):
            p = _empty((frames * hw, w.ocp[i]), self.hdt, dev)
            self.linear(Pieces(taps[i]) if self.taps_split else taps[i], w.proj_w[i], p, w.proj_b[i])
            proj.append(p)
        # The four pyramid branches are independent until the fusion blocks join them, the coarsest first.  The coarsest
        # branch (stride-2 conv, temporal module, fusion block, temporal module: 19 x 19 and 37 x 37 maps, grids that fill a
        # fraction of the CUs) stays on the caller's stream; the other three, with their large kernels, run beside it on a
        # side stream until the first fusion block that needs them.  Outputs that cross streams are allocated here, on the
        # caller's stream.
        r1 = _empty((frames, 4 * hp, 4 * wp, w.Fp), self.hdt, dev)
        r2 = _empty((frames, 2 * hp, 2 * wp, w.Fp), self.hdt, dev)
        r3 = _empty((frames, hp, wp, w.Fp), self.hdt, dev)

        def other_branches():
            l1 = _empty((frames, 4 * hp, 4 * wp, w.ocp[0]), self.hdt, dev)
            self.linear(proj[0], w.rs0_w, l1, w.rs0_b, N=16 * w.ocp[0], ldc=w.ocp[0], epi=hip.EPI_CONVT, convt=(hp, wp, 4, w.ocp[0]))
            l2 = _empty((frames, 2 * hp, 2 * wp, w.ocp[1]), self.hdt, dev)
            self.linear(proj[1], w.rs1_w, l2, w.rs1_b, N=4 * w.ocp[1], ldc=w.ocp[1], epi=hip.EPI_CONVT, convt=(hp, wp, 2, w.ocp[1]))
            self.conv3(l1, w.rn_w[0], None, w.Fp, out=r1)
            self.conv3(l2, w.rn_w[1], None, w.Fp, out=r2)
            l3 = self.motion(0, proj[2].reshape(frames, hp, wp, w.ocp[2]), B, T)
            self.conv3(l3, w.rn_w[2], None, w.Fp, out=r3)

        joined = None
        if self.head_streams > 1 and self.encoder_streams > 1 and self.side_streams:
            main, side = torch.cuda.current_stream(), self.side_streams[0]
            fork = torch.cuda.Event()
            fork.record(main)
            side.wait_event(fork)                     # the projections and r1, r2, r3 exist
            with torch.cuda.stream(side):
                other_branches()
                joined = torch.cuda.Event()
                joined.record(side)
        else:
            other_branches()
        l4 = self.conv3(proj[3].reshape(frames, hp, wp, w.ocp[3]), w.rs3_w, w.rs3_b, w.ocp[3], stride=2)
        l4 = self.motion(1, l4, B, T)
        r4 = self.conv3(l4, w.rn_w[3], None, w.Fp)
        p4 = self.fusion(4, r4, None, hp, wp)
        p4 = self.motion(2, p4, B, T)
        if joined is not None:
            torch.cuda.current_stream().wait_event(joined)
        p3 = self.fusion(3, p4, r3, 2 * hp, 2 * wp)
        p3 = self.motion(3, p3, B, T)
        # From here on every stage works frame by frame (two fusion blocks, the output convolutions): two halves of the
        # frames on the two streams, as in the encoder.
        depth = _empty((frames, 14 * hp, 14 * wp), torch.float32, dev)
        narrow = not self.split_tail and w.F2 < w.F2p and w.F2 % 32 == 0

        def tail(f0, f1):
            p2 = self.fusion(2, p3[f0:f1], r2[f0:f1], 4 * hp, 4 * wp)
            # the two largest resamplings (to 8x and 14x the patch grid) are folded into the convolutions that read them
            p1 = self.fusion(1, p2, r1[f0:f1], 8 * hp, 8 * wp, resize=False)
            if self.split_tail and not self.split_head:      # measurement form: only the output stage in fp32 storage
                self.split, self.hdt, p1, hdt16 = True, torch.float32, p1.float(), self.hdt
                try:
                    o1 = self.conv3(p1, w.oc1_w, w.oc1_b, w.F2p, resized=(8 * hp, 8 * wp))
                    self.conv3(o1, w.oc2_w, w.oc2_b, 32, out=depth[f0:f1], ldc=4, epi=hip.EPI_DOT, dot_w=w.dot_w, dot_b=w.dot_b,
                               resized=(14 * hp, 14 * wp))
                finally:
                    self.split, self.hdt = False, hdt16
                return
            if narrow:      # ViT-S: 32 channels in a map stored 64 wide; the padding is neither computed nor read
                o1 = _empty((f1 - f0, 8 * hp, 8 * wp, w.F2p), self.hdt, dev)
                self.conv3(p1, w.oc1_w[:w.F2], w.oc1_b[:w.F2], w.F2, out=o1, ldc=w.F2p, resized=(8 * hp, 8 * wp))
            else:
                o1 = self.conv3(p1, w.oc1_w, w.oc1_b, w.F2p, resized=(8 * hp, 8 * wp))
            self.conv3(o1, w.oc2_w, w.oc2_b, 32, out=depth[f0:f1], ldc=4, epi=hip.EPI_DOT, dot_w=w.dot_w, dot_b=w.dot_b,
                       resized=(14 * hp, 14 * wp), **(dict(cin_used=w.F2) if narrow else {}))

        if joined is not None and frames > 1:
            main, side = torch.cuda.current_stream(), self.side_streams[0]
            half = frames // 2
            fork = torch.cuda.Event()
            fork.record(main)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                tail(half, frames)
                done = torch.cuda.Event()
                done.record(side)
            tail(0, half)
            main.wait_event(done)
        else:
            tail(0, frames)
        return depth

    def forward(self, x, frame_keys=None):
        """x: fp32 [B, T, 3, H, W] on the device -> fp32 [B, T, H, W] (reference: video_depth.py:58-65).
        The reference's trailing resize to (H, W) is the identity (the head already emits 14*hp x 14*wp = H x W with
        align_corners=True) and its ReLU is idempotent after the head's own, so neither is launched.

        ``frame_keys`` (B = 1): one hashable identity per frame.  The encoder outputs of this call are kept, and the leading
        frames whose keys were part of the previous keyed call (at the same size) take their encoder outputs from it."""
        B, T, ch, H, W = x.shape
        hp, wp = H // 14, W // 14
        carried = None
        keyed = frame_keys is not None and B == 1
        if keyed:
            if len(frame_keys) != T:
                raise ValueError(f"frame_keys: {len(frame_keys)} keys for {T} frames")
            kept = self.kept
            if kept is not None and kept["size"] == (H, W):
                n = 0
                while n < T and frame_keys[n] in kept["slot"]:
                    n += 1
                if n:
                    carried = (kept["taps"], [kept["slot"][k] for k in frame_keys[:n]])
        while True:
            taps = self.encoder(x.reshape(B * T, ch, H, W), B * T, H, W, carried)
            depth = self.head(taps, B, T, hp, wp).reshape(B, T, H, W)
            if not (self.split_enc or self.split_head or self.split_tail) or SPLIT_DTYPE != torch.float16:
                break
            # fp32 storage: did some activation, times act_scale, leave the IEEE-half range (its high piece then saturated)?
            # One 4-byte read per window, which waits for the window; this mode's windows take a hundred milliseconds.
            if not int(self.sat.item()):
                break
            self.sat.zero_()
            if self.act_scale <= MIN_ACT_SPLIT_SCALE:
                raise SaturationError(f"fp32 storage mode: an activation exceeds {65504 / self.act_scale:.3g} in magnitude, beyond the "
                                      "range of the hi/lo half pieces at the smallest operand scale")
            self.act_scale /= 16.0
            warnings.warn(f"fp32 storage mode: an activation left the IEEE-half range of the operand pieces; the operand scale "
                          f"is lowered to {self.act_scale:g} (|values| up to {65504 / self.act_scale:.3g}) and the window is "
                          f"computed again", RuntimeWarning, stacklevel=3)
            carried, self.kept = None, None        # kept operand pieces carry the old scale
        if keyed:
            self.kept = dict(size=(H, W), slot={k: i for i, k in enumerate(frame_keys)}, taps=taps)
        return depth
