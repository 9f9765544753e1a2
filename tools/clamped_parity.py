"""Parity of both precision modes on the "clamped" weight profile (depth maps with exact zeros and pixels next to the
clamp) against the reference's golden outputs: prints the floored max-relative error, AbsRel and the absolute error as a
fraction of the map's maximum.    python tools/clamped_parity.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vda_oracle as O
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything

cfg = spec.MODEL_CONFIGS["vits"]
sd = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**cfg, seed=1, profile="clamped").items()}
m = VideoDepthAnything(**cfg); m.load_state_dict(sd, strict=True); m = m.cuda().eval()
g4 = np.load(os.path.join(ROOT, "tests/golden/g4b_forward_clamped.npz"))
g5 = np.load(os.path.join(ROOT, "tests/golden/g5b_video_clamped.npz"))


def line(name, got, want):
    err = np.abs(got.astype(np.float64) - want)
    print(f"{name:34s} max_rel(floor 1e-3)={O.max_rel_err(got, want):.3e} abs_rel={O.abs_rel(got, want):.3e} "
          f"max|err|/max={err.max() / want.max():.3e} zeros want/got={np.mean(want == 0):.3f}/{np.mean(got == 0):.3f} "
          f"flipped={np.mean((want == 0) != (got == 0)):.4f}")


for mode in ("half", "fp32"):
    m.precision = mode
    for name, shape, seed in (("T32", (1, 32, 3, 70, 98), 41), ("T4", (1, 4, 3, 70, 98), 42), ("B2T3", (2, 3, 3, 56, 42), 43)):
        x = torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32)).cuda()
        line(f"{mode} forward {name}", m(x).cpu().numpy(), g4[name])
    m.precision = "half"
    for n, bs in ((33, 32), (60, 32), (33, 16)):
        frames = np.random.RandomState(700 + n).randint(0, 256, (n, 28, 42, 3)).astype(np.uint8)
        out, _ = m.infer_video_depth(frames, 24, input_size=28, device="cuda", fp32=(mode == "fp32"), batch_size=bs)
        line(f"{mode} video N{n} bs{bs}", out, g5[f"N{n}_bs{bs}"])
