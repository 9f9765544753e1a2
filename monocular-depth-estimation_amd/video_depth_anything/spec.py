"""Parameter inventory of the Video Depth Anything network and a seeded synthetic weight generator.

The inventory reproduces, key for key and shape for shape, the ``state_dict`` of the reference model
(reference: video_depth_anything/video_depth.py:36-56, dinov2.py:106-168, dpt.py:60-124,
dpt_temporal.py:42-51, motion_module/motion_module.py:84-100,141-161,194, motion_module/attention.py:81-91,
327-333,374) so that ``load_state_dict(strict=True)`` accepts real checkpoints.

The generator draws every tensor from its own ``numpy.random.RandomState`` stream so that any machine regenerates the same
weights from a seed without depending on torch's RNG stream.
"""
from collections import OrderedDict
import math
import zlib

import numpy as np

ENCODERS = {
    # name: (embed dim, depth, heads, tapped block indices)   reference: dinov2.py:339-378, video_depth.py:48-51
    "vits": (384, 12, 6, (2, 5, 8, 11)),
    "vitl": (1024, 24, 16, (4, 11, 17, 23)),
}

MODEL_CONFIGS = {
    # reference: run.py:71-74
    "vits": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384]),
    "vitl": dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024]),
}

PATCH = 14
POS_GRID = 37          # pos_embed is stored for a 518/14 = 37 x 37 grid (dinov2.py:407)
TEMPORAL_HEADS = 8     # dpt_temporal.py:35
# output_conv2.2.bias of the "clamped" synthetic weight profile: the mixed-sign weights sum to about -3.8 on the hidden
# channels of the ViT-S head (median -4.3, deviation 1.35 at ViT-L width), so these put 20-45 % / about 30 % of the pixels at 0
CLAMPED_PROFILE_BIAS = {"vits": 4.2, "vitl": 5.0}
OUTLIER_POS_EMBED = (300.0, -500.0, 800.0, -1200.0, 2000.0, -3000.0)      # "outliers" profile: added to six channels of pos_embed
OUTLIER_HEAD_BIAS = 5.5e4                                                 # ... and one channel of head.projects.2.bias


def state_dict_spec(encoder="vitl", features=256, out_channels=(256, 512, 1024, 1024), num_frames=32):
    """Ordered ``{key: (shape, kind)}``; ``kind`` selects the synthetic distribution in :func:`synth_state_dict`."""
    D, depth, _heads, _taps = ENCODERS[encoder]
    F = features
    oc = list(out_channels)
    s = OrderedDict()

    def add(key, shape, kind):
        s[key] = (tuple(shape), kind)

    p = "pretrained."
    add(p + "cls_token", (1, 1, D), "token")
    add(p + "pos_embed", (1, POS_GRID * POS_GRID + 1, D), "token")
    add(p + "mask_token", (1, D), "token")
    add(p + "patch_embed.proj.weight", (D, 3, PATCH, PATCH), "w")
    add(p + "patch_embed.proj.bias", (D,), "b")
    for i in range(depth):
        b = f"{p}blocks.{i}."
        add(b + "norm1.weight", (D,), "nw")
        add(b + "norm1.bias", (D,), "nb")
        add(b + "attn.qkv.weight", (3 * D, D), "w")
        add(b + "attn.qkv.bias", (3 * D,), "b")
        add(b + "attn.proj.weight", (D, D), "w")
        add(b + "attn.proj.bias", (D,), "b")
        add(b + "ls1.gamma", (D,), "ls")
        add(b + "norm2.weight", (D,), "nw")
        add(b + "norm2.bias", (D,), "nb")
        add(b + "mlp.fc1.weight", (4 * D, D), "w")
        add(b + "mlp.fc1.bias", (4 * D,), "b")
        add(b + "mlp.fc2.weight", (D, 4 * D), "w")
        add(b + "mlp.fc2.bias", (D,), "b")
        add(b + "ls2.gamma", (D,), "ls")
    add(p + "norm.weight", (D,), "nw")
    add(p + "norm.bias", (D,), "nb")

    h = "head."
    for i in range(4):
        add(f"{h}projects.{i}.weight", (oc[i], D, 1, 1), "w")
        add(f"{h}projects.{i}.bias", (oc[i],), "b")
    add(h + "resize_layers.0.weight", (oc[0], oc[0], 4, 4), "wt")      # ConvTranspose2d: [Cin, Cout, k, k]
    add(h + "resize_layers.0.bias", (oc[0],), "b")
    add(h + "resize_layers.1.weight", (oc[1], oc[1], 2, 2), "wt")
    add(h + "resize_layers.1.bias", (oc[1],), "b")
    add(h + "resize_layers.3.weight", (oc[3], oc[3], 3, 3), "w")
    add(h + "resize_layers.3.bias", (oc[3],), "b")
    sc = h + "scratch."
    for i in range(4):
        add(f"{sc}layer{i + 1}_rn.weight", (F, oc[i], 3, 3), "w")
    for i in range(1, 5):
        r = f"{sc}refinenet{i}."
        add(r + "out_conv.weight", (F, F, 1, 1), "w")
        add(r + "out_conv.bias", (F,), "b")
        for u in (1, 2):
            for c in (1, 2):
                add(f"{r}resConfUnit{u}.conv{c}.weight", (F, F, 3, 3), "w")
                add(f"{r}resConfUnit{u}.conv{c}.bias", (F,), "b")
    add(sc + "output_conv1.weight", (F // 2, F, 3, 3), "w")
    add(sc + "output_conv1.bias", (F // 2,), "b")
    add(sc + "output_conv2.0.weight", (32, F // 2, 3, 3), "w")
    add(sc + "output_conv2.0.bias", (32,), "b_out0")
    add(sc + "output_conv2.2.weight", (1, 32, 1, 1), "w_out2")
    add(sc + "output_conv2.2.bias", (1,), "b_out2")
    for i, C in enumerate((oc[2], oc[3], F, F)):
        t = f"{h}motion_modules.{i}.temporal_transformer."
        add(t + "norm.weight", (C,), "nw")
        add(t + "norm.bias", (C,), "nb")
        add(t + "proj_in.weight", (C, C), "w")
        add(t + "proj_in.bias", (C,), "b")
        tb = t + "transformer_blocks.0."
        for a in (0, 1):
            ab = f"{tb}attention_blocks.{a}."
            add(ab + "to_q.weight", (C, C), "w")
            add(ab + "to_k.weight", (C, C), "w")
            add(ab + "to_v.weight", (C, C), "w")
            add(ab + "to_out.0.weight", (C, C), "w")
            add(ab + "to_out.0.bias", (C,), "b")
            add(ab + "pos_encoder.pe", (1, num_frames, C), "pe")
        for a in (0, 1):
            add(f"{tb}norms.{a}.weight", (C,), "nw")
            add(f"{tb}norms.{a}.bias", (C,), "nb")
        add(tb + "ff.net.0.proj.weight", (8 * C, C), "w")
        add(tb + "ff.net.0.proj.bias", (8 * C,), "b")
        add(tb + "ff.net.2.weight", (C, 4 * C), "w")
        add(tb + "ff.net.2.bias", (C,), "b")
        add(tb + "ff_norm.weight", (C,), "nw")
        add(tb + "ff_norm.bias", (C,), "nb")
        add(t + "proj_out.weight", (C, C), "w_small")
        add(t + "proj_out.bias", (C,), "b")
    return s


def sinusoid_table(T, C):
    """Sinusoidal temporal position table ``[1, T, C]`` (reference: motion_module/motion_module.py:189-193)."""
    pos = np.arange(T, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, C, 2, dtype=np.float32) * np.float32(-math.log(10000.0) / C))
    pe = np.zeros((1, T, C), dtype=np.float32)
    pe[0, :, 0::2] = np.sin(pos * div)
    pe[0, :, 1::2] = np.cos(pos * div)
    return pe


def synth_state_dict(encoder="vitl", features=256, out_channels=(256, 512, 1024, 1024), num_frames=32, seed=0,
                     only=None, profile="smooth"):
    """Seeded synthetic weights as ``{key: np.float32 array}``; ``only`` restricts to keys with those prefixes.

    Scales are chosen so activations stay O(1) through every layer and the four motion modules contribute
    (their ``proj_out`` is *not* zero, unlike a freshly constructed reference module, motion_module.py:57-58).
    ``profile`` selects the last 1x1 convolution (``output_conv2.2``), every other tensor is the same:
      "smooth"   positive mixing weights and bias: the depth is positive with a moderate dynamic range, no pixel is
                 near the final ReLU clamp;
      "clamped"  mixed-sign mixing weights and a negative bias: a large share of the pixels is clamped to exactly 0
                 and the positive ones reach down to the clamp (what a trained checkpoint gives: sky = 0), so errors
                 are measured where the 32 hidden channels cancel;
      "outliers" "smooth" plus the statistics trained vision transformers are known for and O(1) synthetic weights lack:
                 six channels of the positional table at +-300 .. 3000 (a few residual-stream channels hundreds of times
                 larger than the rest, in every token of every layer), LayerScale factors of up to 10 on four channels
                 per block, and one head channel near 5.5e4 (a bias of ``projects.2``), close to the end of the IEEE-half
                 range and far beyond what the fp32 mode's default operand scaling holds.
    """
    if profile not in ("smooth", "clamped", "outliers"):
        raise ValueError(profile)
    out = OrderedDict()
    for key, (shape, kind) in state_dict_spec(encoder, features, out_channels, num_frames).items():
        if only is not None and not any(key.startswith(p) for p in only):
            continue
        # one independent stream per tensor, so any subset of keys can be generated alone
        rs = np.random.RandomState((zlib.crc32(key.encode()) + 1000003 * seed) & 0xFFFFFFFF)
        n = int(np.prod(shape))
        if kind == "w":            # [out, in, ...]: unit-gain fan-in init
            fan_in = n // shape[0]
            v = rs.standard_normal(n) * (1.0 / math.sqrt(fan_in))
        elif kind == "wt":         # ConvTranspose [in, out, k, k]: each output pixel sums over `in` only
            v = rs.standard_normal(n) * (1.0 / math.sqrt(shape[0]))
        elif kind == "w_small":
            fan_in = n // shape[0]
            v = rs.standard_normal(n) * (0.5 / math.sqrt(fan_in))
        elif kind == "b":
            v = rs.standard_normal(n) * 0.05
        elif kind == "nw":
            v = 1.0 + 0.1 * rs.standard_normal(n)
        elif kind == "nb":
            v = 0.05 * rs.standard_normal(n)
        elif kind == "ls":
            v = 0.2 + 0.05 * rs.standard_normal(n)
        elif kind == "token":
            v = rs.standard_normal(n) * 0.1
        elif kind == "pe":
            v = sinusoid_table(shape[1], shape[2]).reshape(-1)
        elif kind == "b_out0":
            v = 0.5 + 0.1 * rs.standard_normal(n)
        elif kind == "w_out2" and profile == "clamped":
            v = rs.standard_normal(n) * (1.0 / math.sqrt(n))
        elif kind == "b_out2" and profile == "clamped":
            v = np.full(n, CLAMPED_PROFILE_BIAS[encoder])
        elif kind == "w_out2":     # positive mixing weights keep the depth away from the ReLU clamp
            v = np.abs(rs.standard_normal(n)) * (1.0 / math.sqrt(n)) + 0.02
        elif kind == "b_out2":
            v = np.full(n, 0.25)
        else:
            raise KeyError(kind)
        v = np.asarray(v, dtype=np.float32).reshape(shape)
        if profile == "outliers":
            pick = np.random.RandomState((zlib.crc32(("outliers/" + key).encode()) + 1000003 * seed) & 0xFFFFFFFF)
            if key == "pretrained.pos_embed":
                for c, a in zip(pick.choice(shape[2], 6, replace=False), OUTLIER_POS_EMBED):
                    v[:, :, c] += a
            elif kind == "ls":
                v[pick.choice(n, 4, replace=False)] = pick.uniform(4.0, 10.0, 4).astype(np.float32)
            elif key == "head.projects.2.bias":
                v[pick.randint(n)] = OUTLIER_HEAD_BIAS
        out[key] = v
    return out
