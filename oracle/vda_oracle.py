"""CPU oracle for the Video Depth Anything per-window forward pass and the window/stitch host logic.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this file; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker
or the timed CPU baseline.

It is a functional, fp32, plain-torch/numpy restatement of the reference algorithm that works directly on a
``state_dict`` (no nn.Module tree).  Each function cites the reference lines it follows.  It is pinned to the
reference by the golden vectors under ``tests/golden/`` which ``oracle/make_goldens.py`` produced by importing
the unmodified reference model in the build container (the reference's own tests hold no fixtures for this
path).  The OpenCV bicubic image resize of the preprocessing step cannot be pinned that way (cv2 is absent everywhere
this runs): ``cv_resize_cubic`` restates OpenCV's published INTER_CUBIC arithmetic and that one step is "parity
unpinned"; the golden vectors feed frames at network size.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

ENCODERS = {"vits": (384, 12, 6, (2, 5, 8, 11)), "vitl": (1024, 24, 16, (4, 11, 17, 23))}

INFER_LEN = 32
OVERLAP = 10
KEYFRAMES = [0, 12, 24, 25, 26, 27, 28, 29, 30, 31]
INTERP_LEN = 8


def _t(sd, key, like):
    v = sd[key]
    if not torch.is_tensor(v):
        v = torch.from_numpy(np.asarray(v))
    return v.to(device=like.device, dtype=torch.float32)


# ----------------------------------------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------------------------------------
def pos_embed_for(pos_embed, hp, wp):
    """Positional table for an ``hp x wp`` patch grid (reference: dinov2.py:179-210).

    The stored table is returned untouched for the square 37x37 grid; otherwise the 37x37 patch part is
    resampled bicubically with an explicit scale factor of ``(n + 0.1) / 37`` per axis (align_corners=False,
    no antialias) and the class entry is passed through.
    """
    n_stored = pos_embed.shape[1] - 1
    g = int(math.sqrt(n_stored))
    if hp * wp == n_stored and hp == wp:
        return pos_embed
    pe = pos_embed.float()
    cls_pe, patch_pe = pe[:, :1], pe[:, 1:]
    dim = pe.shape[-1]
    sy, sx = float(hp + 0.1) / g, float(wp + 0.1) / g
    grid = patch_pe.reshape(1, g, g, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(sy, sx), mode="bicubic", antialias=False)
    assert grid.shape[-2] == hp and grid.shape[-1] == wp
    return torch.cat([cls_pe, grid.permute(0, 2, 3, 1).reshape(1, hp * wp, dim)], dim=1)


def encoder_tokens(sd, x):
    """Patch embedding, class token and positional table (reference: dinov2.py:212-231, patch_embed.py:69-82)."""
    p = "pretrained."
    tok = F.conv2d(x, _t(sd, p + "patch_embed.proj.weight", x), _t(sd, p + "patch_embed.proj.bias", x), stride=14)
    n, d, hp, wp = tok.shape
    tok = tok.flatten(2).transpose(1, 2)
    cls = _t(sd, p + "cls_token", x).expand(n, -1, -1)
    tok = torch.cat([cls, tok], dim=1)
    return tok + pos_embed_for(_t(sd, p + "pos_embed", x), hp, wp)


def encoder_block(sd, i, x, heads):
    """One pre-norm transformer block with LayerScale (reference: block.py:82-107, attention.py:49-62,
    mlp.py:35-41, layer_scale.py:27-28)."""
    b = f"pretrained.blocks.{i}."
    n, s, d = x.shape
    hd = d // heads
    y = F.layer_norm(x, (d,), _t(sd, b + "norm1.weight", x), _t(sd, b + "norm1.bias", x), 1e-6)
    qkv = F.linear(y, _t(sd, b + "attn.qkv.weight", x), _t(sd, b + "attn.qkv.bias", x))
    qkv = qkv.reshape(n, s, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    a = torch.softmax(q @ k.transpose(-2, -1), dim=-1) @ v
    a = a.transpose(1, 2).reshape(n, s, d)
    a = F.linear(a, _t(sd, b + "attn.proj.weight", x), _t(sd, b + "attn.proj.bias", x))
    x = x + a * _t(sd, b + "ls1.gamma", x)
    y = F.layer_norm(x, (d,), _t(sd, b + "norm2.weight", x), _t(sd, b + "norm2.bias", x), 1e-6)
    y = F.gelu(F.linear(y, _t(sd, b + "mlp.fc1.weight", x), _t(sd, b + "mlp.fc1.bias", x)))
    y = F.linear(y, _t(sd, b + "mlp.fc2.weight", x), _t(sd, b + "mlp.fc2.bias", x))
    return x + y * _t(sd, b + "ls2.gamma", x)


def encoder_taps(sd, x, encoder, frame_chunk=8):
    """Four normalised intermediate patch-token maps (reference: dinov2.py:271-281, 297-321).

    Frames are independent in the encoder, so they are processed ``frame_chunk`` at a time to bound the
    size of the materialised attention matrix; the arithmetic per frame is unchanged.
    """
    d, depth, heads, taps = ENCODERS[encoder]
    outs = [[] for _ in taps]
    nw, nb = _t(sd, "pretrained.norm.weight", x), _t(sd, "pretrained.norm.bias", x)
    for s0 in range(0, x.shape[0], frame_chunk):
        t = encoder_tokens(sd, x[s0:s0 + frame_chunk])
        for i in range(depth):
            t = encoder_block(sd, i, t, heads)
            if i in taps:
                y = F.layer_norm(t, (d,), nw, nb, 1e-6)
                outs[taps.index(i)].append(y[:, 1:])          # class token is dropped (use_clstoken=False)
    return [torch.cat(o, dim=0) for o in outs]


# ----------------------------------------------------------------------------------------------------------
# temporal (motion) module
# ----------------------------------------------------------------------------------------------------------
def temporal_attention(sd, pre, h, T, heads=8):
    """Self-attention over the T frames of every spatial position (reference: motion_module.py:230-297,
    attention.py:93-111, 182-211).  ``h``: already layer-normed ``[B*T, hw, C]``."""
    bt, hw, c = h.shape
    b = bt // T
    hd = c // heads
    z = h.reshape(b, T, hw, c).permute(0, 2, 1, 3).reshape(b * hw, T, c)
    z = z + _t(sd, pre + "pos_encoder.pe", h)[:, :T]
    q = F.linear(z, _t(sd, pre + "to_q.weight", h))
    k = F.linear(z, _t(sd, pre + "to_k.weight", h))
    v = F.linear(z, _t(sd, pre + "to_v.weight", h))

    def split(u):
        return u.reshape(b * hw, T, heads, hd).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    p = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(b * hw, T, c)
    o = F.linear(o, _t(sd, pre + "to_out.0.weight", h), _t(sd, pre + "to_out.0.bias", h))
    return o.reshape(b, hw, T, c).permute(0, 2, 1, 3).reshape(bt, hw, c)


def temporal_module(sd, idx, x, T):
    """GroupNorm -> proj_in -> 2x(LN, temporal attention, +res) -> LN, GEGLU feed-forward, +res -> proj_out ->
    + input (reference: motion_module.py:102-126, 164-177; attention.py:335-338, 382-384).  ``x``: [B*T, C, h, w]."""
    pre = f"head.motion_modules.{idx}.temporal_transformer."
    bt, c, hh, ww = x.shape
    y = F.group_norm(x, 32, _t(sd, pre + "norm.weight", x), _t(sd, pre + "norm.bias", x), 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(bt, hh * ww, c)
    y = F.linear(y, _t(sd, pre + "proj_in.weight", x), _t(sd, pre + "proj_in.bias", x))
    tb = pre + "transformer_blocks.0."
    for a in (0, 1):
        n = F.layer_norm(y, (c,), _t(sd, f"{tb}norms.{a}.weight", x), _t(sd, f"{tb}norms.{a}.bias", x), 1e-5)
        y = temporal_attention(sd, f"{tb}attention_blocks.{a}.", n, T) + y
    n = F.layer_norm(y, (c,), _t(sd, tb + "ff_norm.weight", x), _t(sd, tb + "ff_norm.bias", x), 1e-5)
    g = F.linear(n, _t(sd, tb + "ff.net.0.proj.weight", x), _t(sd, tb + "ff.net.0.proj.bias", x))
    val, gate = g.chunk(2, dim=-1)
    f = F.linear(val * F.gelu(gate), _t(sd, tb + "ff.net.2.weight", x), _t(sd, tb + "ff.net.2.bias", x))
    y = f + y
    y = F.linear(y, _t(sd, pre + "proj_out.weight", x), _t(sd, pre + "proj_out.bias", x))
    return y.reshape(bt, hh, ww, c).permute(0, 3, 1, 2) + x


# ----------------------------------------------------------------------------------------------------------
# DPT head
# ----------------------------------------------------------------------------------------------------------
def residual_conv_unit(sd, pre, x):
    """conv2(relu(conv1(relu(x)))) + x (reference: util/blocks.py:68-91)."""
    y = F.conv2d(F.relu(x), _t(sd, pre + "conv1.weight", x), _t(sd, pre + "conv1.bias", x), padding=1)
    y = F.conv2d(F.relu(y), _t(sd, pre + "conv2.weight", x), _t(sd, pre + "conv2.bias", x), padding=1)
    return y + x


def fusion_block(sd, idx, x0, x1=None, size=None):
    """RefineNet fusion (reference: util/blocks.py:135-162)."""
    pre = f"head.scratch.refinenet{idx}."
    y = x0
    if x1 is not None:
        y = y + residual_conv_unit(sd, pre + "resConfUnit1.", x1)
    y = residual_conv_unit(sd, pre + "resConfUnit2.", y)
    if size is None:
        y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        y = F.interpolate(y, size=tuple(size), mode="bilinear", align_corners=True)
    return F.conv2d(y, _t(sd, pre + "out_conv.weight", x0), _t(sd, pre + "out_conv.bias", x0))


def head_forward(sd, taps, hp, wp, T):
    """Temporal DPT head (reference: dpt_temporal.py:53-114, dpt.py:60-124).  ``taps``: 4 x [B*T, hp*wp, D].
    Returns ``[B*T, 1, 14hp, 14wp]``."""
    h = "head."
    x0 = taps[0]
    layers = []
    for i, tok in enumerate(taps):
        m = tok.permute(0, 2, 1).reshape(tok.shape[0], tok.shape[-1], hp, wp)
        m = F.conv2d(m, _t(sd, f"{h}projects.{i}.weight", x0), _t(sd, f"{h}projects.{i}.bias", x0))
        if i == 0:
            m = F.conv_transpose2d(m, _t(sd, h + "resize_layers.0.weight", x0), _t(sd, h + "resize_layers.0.bias", x0), stride=4)
        elif i == 1:
            m = F.conv_transpose2d(m, _t(sd, h + "resize_layers.1.weight", x0), _t(sd, h + "resize_layers.1.bias", x0), stride=2)
        elif i == 3:
            m = F.conv2d(m, _t(sd, h + "resize_layers.3.weight", x0), _t(sd, h + "resize_layers.3.bias", x0), stride=2, padding=1)
        layers.append(m)
    l1, l2, l3, l4 = layers
    l3 = temporal_module(sd, 0, l3, T)
    l4 = temporal_module(sd, 1, l4, T)
    sc = h + "scratch."
    r1 = F.conv2d(l1, _t(sd, sc + "layer1_rn.weight", x0), padding=1)
    r2 = F.conv2d(l2, _t(sd, sc + "layer2_rn.weight", x0), padding=1)
    r3 = F.conv2d(l3, _t(sd, sc + "layer3_rn.weight", x0), padding=1)
    r4 = F.conv2d(l4, _t(sd, sc + "layer4_rn.weight", x0), padding=1)
    p4 = fusion_block(sd, 4, r4, None, size=r3.shape[2:])
    p4 = temporal_module(sd, 2, p4, T)
    p3 = fusion_block(sd, 3, p4, r3, size=r2.shape[2:])
    p3 = temporal_module(sd, 3, p3, T)
    # the two finest stages run in micro-batches of 4 frames exactly when the reference does (dpt_temporal.py:88-114);
    # every op there is per-sample, so this only bounds the size of the largest intermediate tensor
    n = p3.shape[0]
    mb = n if (n <= 4 or n % 4 != 0) else 4
    outs = []
    for i in range(0, n, mb):
        p2 = fusion_block(sd, 2, p3[i:i + mb], r2[i:i + mb], size=r1.shape[2:])
        p1 = fusion_block(sd, 1, p2, r1[i:i + mb], size=None)
        o = F.conv2d(p1, _t(sd, sc + "output_conv1.weight", x0), _t(sd, sc + "output_conv1.bias", x0), padding=1)
        o = F.interpolate(o, (14 * hp, 14 * wp), mode="bilinear", align_corners=True)
        o = F.relu(F.conv2d(o, _t(sd, sc + "output_conv2.0.weight", x0), _t(sd, sc + "output_conv2.0.bias", x0), padding=1))
        o = F.relu(F.conv2d(o, _t(sd, sc + "output_conv2.2.weight", x0), _t(sd, sc + "output_conv2.2.bias", x0)))
        outs.append(o)
    return torch.cat(outs, dim=0)


@torch.no_grad()
def forward(sd, x, encoder):
    """``x[B,T,3,H,W]`` -> depth ``[B,T,H,W]`` (reference: video_depth.py:58-65)."""
    B, T, _, H, W = x.shape
    hp, wp = H // 14, W // 14
    x = x.float()
    taps = encoder_taps(sd, x.flatten(0, 1), encoder)
    d = head_forward(sd, taps, hp, wp, T)
    d = F.relu(F.interpolate(d, size=(H, W), mode="bilinear", align_corners=True))
    return d.squeeze(1).unflatten(0, (B, T))


# ----------------------------------------------------------------------------------------------------------
# host logic: preprocessing size rule, window assembly, stitch
# ----------------------------------------------------------------------------------------------------------
def network_size(frame_h, frame_w, input_size=518):
    """Network input (H, W) for a frame size (reference: video_depth.py:68-72, util/transform.py:51-107 with
    keep_aspect_ratio, lower_bound, multiple of 14).  The reference's 'cuda' 420-pixel clamp
    (video_depth.py:75-78) is not part of this CPU oracle."""
    ratio = max(frame_h, frame_w) / min(frame_h, frame_w)
    if ratio > 1.78:
        input_size = int(input_size * 1.777 / ratio)
        input_size = round(input_size / 14) * 14
    scale = max(input_size / frame_w, input_size / frame_h)

    def snap(v):
        y = int(np.round(v / 14) * 14)
        if y < input_size:
            y = int(np.ceil(v / 14) * 14)
        return y

    return snap(scale * frame_h), snap(scale * frame_w)


def normalize_frame(frame_u8):
    """uint8 HWC frame at network size -> float32 CHW (reference: video_depth.py:120, util/transform.py:134,147-148).
    The arithmetic is float32/255 then float64 mean/std, cast to float32 last, as numpy does with python lists."""
    img = frame_u8.astype(np.float32) / 255.0
    img = (img - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]
    return np.ascontiguousarray(np.transpose(img, (2, 0, 1))).astype(np.float32)


def cv_resize_cubic(img, out_h, out_w):
    """float32 HWC image -> (out_h, out_w) with the arithmetic OpenCV documents for ``cv2.resize(..., INTER_CUBIC)`` on
    float images, which is what the reference calls (util/transform.py:109-125 via video_depth.py:80-92): source
    coordinate (dst + 0.5) * in/out - 0.5, the four Keys taps with A = -0.75 (the fourth weight is 1 minus the other
    three), source indices clamped to the border, horizontal pass then vertical pass in float32.
    PARITY UNPINNED: cv2 is absent here and on the GPU box and the reference holds no resized fixture, so this restates
    the published algorithm instead of checking against OpenCV itself."""
    img = np.asarray(img, dtype=np.float32)
    in_h, in_w = img.shape[:2]

    def taps(n_in, n_out):
        scale = float(n_in) / float(n_out)
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        A = np.float32(-0.75)
        one = np.float32(1.0)
        w0 = ((A * (f + one) - np.float32(5.0) * A) * (f + one) + np.float32(8.0) * A) * (f + one) - np.float32(4.0) * A
        w1 = ((A + np.float32(2.0)) * f - (A + np.float32(3.0))) * f * f + one
        g = one - f
        w2 = ((A + np.float32(2.0)) * g - (A + np.float32(3.0))) * g * g + one
        w3 = one - w0 - w1 - w2
        idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
        return idx, np.stack([w0, w1, w2, w3], 1).astype(np.float32)

    ix, wx = taps(in_w, out_w)
    iy, wy = taps(in_h, out_h)
    rows = np.zeros((in_h, out_w) + img.shape[2:], dtype=np.float32)
    for k in range(4):                                   # horizontal pass, taps added in order
        rows = rows + wx[:, k].reshape((1, out_w) + (1,) * (img.ndim - 2)) * img[:, ix[:, k]]
    out = np.zeros((out_h, out_w) + img.shape[2:], dtype=np.float32)
    for k in range(4):
        out = out + wy[:, k].reshape((out_h, 1) + (1,) * (img.ndim - 2)) * rows[iy[:, k]]
    return out.astype(np.float32)


def preprocess_frame(frame_u8, H, W):
    """uint8 HWC frame of any size -> float32 CHW at network size (H, W): /255, bicubic resize, ImageNet normalisation
    (reference: video_depth.py:80-92,120; util/transform.py:109-158)."""
    img = frame_u8.astype(np.float32) / np.float32(255.0)
    if img.shape[:2] != (H, W):
        img = cv_resize_cubic(img, H, W)
    img = (img - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]
    return np.ascontiguousarray(np.transpose(img, (2, 0, 1))).astype(np.float32)


def scale_shift(pred, target):
    """Closed-form least squares ``target ~ s*pred + t`` with an all-ones mask, in float32
    (reference: utils/util.py:40-62)."""
    pred = pred.astype(np.float32)
    target = target.astype(np.float32)
    a00 = np.sum(pred * pred)
    a01 = np.sum(pred)
    a11 = np.float32(pred.size)
    b0 = np.sum(pred * target)
    b1 = np.sum(target)
    det = a00 * a11 - a01 * a01
    if det != 0:
        return (a11 * b0 - a01 * b1) / det, (-a01 * b0 + a00 * b1) / det
    return 1, 0


def blend(pre, post):
    """Linear cross-fade, weight of ``post`` rising 0 -> 1 (reference: utils/util.py:65-74)."""
    n = len(pre)
    step = 1.0 / (n - 1)
    w = [0.0] + [i * step for i in range(1, n - 1)] + [1.0]
    return [pre[i] * (1 - w[i]) + post[i] * w[i] for i in range(n)]


def window_plan(n_frames, batch_size=32):
    """Per window: the list of (source kind, index) that fill its slots (reference: video_depth.py:97-103,
    112-132).  ``("new", j)`` is padded video frame j; ``("prev", k)`` is slot k of the previous window's input."""
    L = min(INFER_LEN, batch_size)
    O = min(OVERLAP, L // 2)
    step = L - O
    pad = (step - n_frames % step) % step + (L - step)
    plan = []
    for w, f0 in enumerate(range(0, n_frames, step)):
        slots = [("new", min(f0 + i, n_frames - 1)) for i in range(L)]
        if w > 0 and O > 0:
            kf = [k for k in KEYFRAMES if k < L][:O]
            for i, k in enumerate(kf):
                slots[i] = ("prev", k)
        plan.append(slots)
    return plan, L, O, n_frames + pad


def stitch(depth_list, n_frames, batch_size=32):
    """Affine alignment of consecutive windows on two key frames and an 8-frame cross-fade
    (reference: video_depth.py:166-209).  ``depth_list``: per-slot float32 maps of all windows, concatenated."""
    L = min(INFER_LEN, batch_size)
    if L < INFER_LEN or batch_size < INFER_LEN:
        return np.stack(depth_list[:n_frames], axis=0)
    out, ref = [], []
    align_len = OVERLAP - INTERP_LEN
    kf = KEYFRAMES[:align_len]
    for f0 in range(0, len(depth_list), INFER_LEN):
        if not out:
            out += depth_list[:INFER_LEN]
            ref = [depth_list[f0 + k] for k in kf]
            continue
        cur = [depth_list[f0 + i] for i in range(len(kf))]
        s, t = scale_shift(np.concatenate(cur), np.concatenate(ref))
        post = []
        for d in depth_list[f0 + align_len:f0 + OVERLAP]:
            d = d * s + t
            d[d < 0] = 0
            post.append(d)
        out[-INTERP_LEN:] = blend(out[-INTERP_LEN:], post)
        for i in range(OVERLAP, INFER_LEN):
            d = depth_list[f0 + i] * s + t
            d[d < 0] = 0
            out.append(d)
        ref = ref[:1]
        for k in kf[1:]:
            d = depth_list[f0 + k] * s + t
            d[d < 0] = 0
            ref.append(d)
    return np.stack(out[:n_frames], axis=0)


@torch.no_grad()
def infer_video_depth(sd, frames, encoder, batch_size=32, forward_fn=None, input_size=None):
    """Whole-video inference (reference: video_depth.py:67-209).  With ``input_size`` None the frames must already be at
    network size; otherwise they are resized to ``network_size(h0, w0, input_size)`` first and the depth is resampled
    back to the frame size.  Returns float32 ``[N, h0, w0]``."""
    n = len(frames)
    H, W = frames[0].shape[:2]
    Hn, Wn = (H, W) if input_size is None else network_size(H, W, input_size)
    plan, L, O, _ = window_plan(n, batch_size)
    fwd = forward_fn or (lambda inp: forward(sd, inp, encoder))
    depth_list, prev = [], None
    for slots in plan:
        cur = [prev[k] if kind == "prev" else torch.from_numpy(preprocess_frame(frames[k], Hn, Wn)) for kind, k in slots]
        inp = torch.stack(cur, 0).unsqueeze(0)
        d = fwd(inp).float()
        d = F.interpolate(d.flatten(0, 1).unsqueeze(1), size=(H, W), mode="bilinear", align_corners=True)
        depth_list += [d[i, 0].cpu().numpy() for i in range(d.shape[0])]
        prev = cur
    return stitch(depth_list, n, batch_size)


# ----------------------------------------------------------------------------------------------------------
# metric post-processing
# ----------------------------------------------------------------------------------------------------------
def apply_lidar_scaling(depths, lidar_refs):
    """LiDAR reference rescaling of a list / array of depth maps (reference: metric_depth/run2.py:42-67, identical inline at
    metric_depth/run.py:233-264).  Returns (list of scaled maps, scale factors)."""
    from scipy.interpolate import interp1d
    scale_factors = np.ones(len(depths))
    for i, depth in enumerate(depths):
        ref = lidar_refs.get(i, lidar_refs.get("global", None))
        if ref:
            x, y, lidar_depth = ref
            if x is None or y is None:
                h, w = depth.shape
                x, y = w // 2, h // 2
            x = min(int(x), depth.shape[1] - 1)
            y = min(int(y), depth.shape[0] - 1)
            pred = depth[y, x]
            if pred > 0:
                scale_factors[i] = lidar_depth / pred
    valid = np.where(scale_factors != 1.0)[0]
    if len(valid) >= 2:
        f = interp1d(valid, scale_factors[valid], kind="linear", fill_value="extrapolate")
        for i in range(len(scale_factors)):
            if scale_factors[i] == 1.0:
                scale_factors[i] = f(i)
    elif len(valid) == 1:
        scale_factors = np.ones(len(depths)) * scale_factors[valid[0]]
    return [d * scale_factors[i] for i, d in enumerate(depths)], scale_factors


def colorize_depth(frames, table_u8=None):
    """8-bit visualisation of a depth video (reference: utils/dc_utils.py:75-81): ``table_u8`` = 256 RGB byte triples
    (inferno) or None for grayscale."""
    frames = np.asarray(frames)
    d_min, d_max = frames.min(), frames.max()
    out = []
    for i in range(frames.shape[0]):
        g = ((frames[i] - d_min) / (d_max - d_min) * 255).astype(np.uint8)
        out.append(table_u8[g] if table_u8 is not None else g)
    return np.stack(out)


# ----------------------------------------------------------------------------------------------------------
# error metrics
# ----------------------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------------------
# colour variants of the reference's drivers and the metric driver's temporal refinement (numpy restatements)
# ----------------------------------------------------------------------------------------------------------
def cmap_lookup(x01, table_u8):
    """matplotlib's ``Colormap.__call__`` on floats in [0, 1] followed by ``(rgb * 255).astype(uint8)``, with the table already in
    bytes: ``table_u8`` [259, 3] = the 256 entries, then the under / over / bad colours.  Index = x * 256 in the array's own
    precision, truncated; exactly 256 -> 255; negative -> under; >= 256 -> over; NaN -> bad."""
    xa = np.array(x01, copy=True)
    xa *= 256
    xa[xa == 256] = 255
    under, over, bad = xa < 0, xa >= 256, np.isnan(xa)
    with np.errstate(invalid="ignore"):
        idx = xa.astype(int)
    idx[under], idx[over], idx[bad] = 256, 257, 258
    return table_u8[idx]


def apply_traffic_zones(depth, reds_u8, viridis_u8, critical_max=10, traffic_max=30):
    """Reference: run1.py:28-60."""
    out = np.zeros(depth.shape + (3,), np.uint8)
    m = depth <= critical_max
    out[m] = cmap_lookup(depth[m] / critical_max, reds_u8)
    m = (depth > critical_max) & (depth <= traffic_max)
    out[m] = cmap_lookup((depth[m] - critical_max) / (traffic_max - critical_max), viridis_u8)
    out[depth > traffic_max] = [64, 64, 64]
    return out


def apply_colormap(depth, vmin, vmax, table_u8):
    """Reference: run1dist.py:173-177."""
    return cmap_lookup((depth - vmin) / (vmax - vmin + 1e-8), table_u8)


def apply_contrast_stretching(depth, table_u8, percentile_low=2, percentile_high=98):
    """Reference: run1dist.py:46-65."""
    flat = depth.flatten()
    flat = flat[~np.isnan(flat)]
    lo, hi = np.percentile(flat, percentile_low), np.percentile(flat, percentile_high)
    return cmap_lookup((np.clip(depth, lo, hi) - lo) / (hi - lo), table_u8)


def apply_distance_capping(depth, vmin, vmax, cap, table_u8):
    """Reference: run1dist.py:67-91."""
    out = cmap_lookup((np.clip(depth, vmin, cap) - vmin) / (cap - vmin + 1e-8), table_u8)
    out[depth >= cap] = table_u8[255]
    return out


def focus_range(depth, min_range=0.0, max_range=50.0):
    """Reference: run4.py:29-42 as applied at run4.py:105-111 (mask, then contrast stretch of the product, stored as float32)."""
    mask = np.ones_like(depth)
    mask[depth > max_range] = 0.1
    d = depth * mask
    in_range = (d >= min_range) & (d <= max_range)
    if np.any(in_range):
        p2, p98 = np.percentile(d[in_range], [2, 98])
        d = np.clip((d - p2) / (p98 - p2 + 1e-8), 0, 1)
    return d.astype(np.float32)


def equalize_hist(g):
    """cv2.equalizeHist on a uint8 image, restated from OpenCV's documented algorithm (parity unpinned: OpenCV is not
    importable): lut[first occupied bin] = 0, lut[i] = round_half_even(255 / (total - h[first]) * cumulative count after it)."""
    hist = np.bincount(g.reshape(-1), minlength=256)
    total, first = int(hist.sum()), int(np.argmax(hist > 0))
    lut = np.zeros(256, np.uint8)
    if hist[first] == total:
        lut[:] = first
    else:
        scale = np.float32(255.0) / np.float32(total - hist[first])
        acc = 0
        for i in range(first + 1, 256):
            acc += int(hist[i])
            lut[i] = min(max(int(np.rint(np.float32(acc) * scale)), 0), 255)
    return lut[g]


def apply_histogram_equalization(depth, vmin, vmax, table_u8):
    """Reference: run1dist.py:27-44 (with ``equalize_hist`` standing in for cv2.equalizeHist)."""
    g = ((depth - vmin) / (vmax - vmin) * 255).astype(np.uint8)
    return cmap_lookup(equalize_hist(g).astype(np.float32) / 255.0, table_u8)


def warp_depth(prev_depth, flow):
    """Reference: metric_depth/run2.py:28-40 with cv2.remap(INTER_LINEAR, BORDER_REFLECT) restated from OpenCV's documented
    behaviour (parity unpinned): positions rounded to 1/32 pixel, four taps with float32 weights, border mirrored about the edge."""
    h, w = flow.shape[:2]
    fm = np.stack(np.meshgrid(np.arange(w), np.arange(h)), axis=-1) + flow
    mx, my = fm[..., 0].astype(np.float32), fm[..., 1].astype(np.float32)
    sx, sy = np.rint(mx * np.float32(32)).astype(np.int64), np.rint(my * np.float32(32)).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    fx, fy = (sx & 31).astype(np.float32) / np.float32(32), (sy & 31).astype(np.float32) / np.float32(32)

    def reflect(p, n):
        p = np.array(p, copy=True)
        if n == 1:
            return np.zeros_like(p)
        for _ in range(64):
            bad = (p < 0) | (p >= n)
            if not bad.any():
                break
            p = np.where(p < 0, -p - 1, np.where(p >= n, 2 * n - p - 1, p))
        return p

    x0, x1, y0, y1 = reflect(ix, w), reflect(ix + 1, w), reflect(iy, h), reflect(iy + 1, h)
    one = np.float32(1)
    src = prev_depth.astype(np.float32)
    acc = src[y0, x0] * ((one - fy) * (one - fx))
    acc = acc + src[y0, x1] * ((one - fy) * fx)
    acc = acc + src[y1, x0] * (fy * (one - fx))
    acc = acc + src[y1, x1] * (fy * fx)
    return acc.astype(np.float32)


def temporal_refine(scaled_depths, flows, prev_depth=None):
    """Reference: metric_depth/run2.py:68-77 with the optical flow given instead of computed."""
    refined = [scaled_depths[0]]
    for i in range(1, len(scaled_depths)):
        warped = prev_depth if (i == 1 and prev_depth is not None) else warp_depth(refined[-1], flows[i - 1])
        refined.append(0.7 * scaled_depths[i] + 0.3 * warped)
    return refined


def max_rel_err(a, b):
    """max |a-b| / max(|b|, 1e-3 max|b|): relative error with a floor, because the depth is ReLU-clamped."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    floor = 1e-3 * np.abs(b).max()
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def abs_rel(pred, gt):
    """mean |gt - pred| / gt over valid pixels (reference: benchmark/eval/metric.py:3-13); valid = gt above
    1e-3 of its maximum."""
    pred = np.asarray(pred, dtype=np.float64)
    gt = np.asarray(gt, dtype=np.float64)
    m = gt > 1e-3 * gt.max()
    return float((np.abs(gt[m] - pred[m]) / gt[m]).mean())
