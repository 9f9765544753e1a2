"""Both precision modes on the "outliers" weight profile against the reference's goldens (G4c); the fp32 mode with and
without producer-written operand pieces; the fp32 mode at a lowered operand scale on the smooth goldens.
    python tools/outlier_probe.py"""
import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vda_oracle as O
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything


def build(profile):
    cfg = spec.MODEL_CONFIGS["vits"]
    sd = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**cfg, seed=1, profile=profile).items()}
    m = VideoDepthAnything(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def line(name, got, want):
    err = np.abs(got.astype(np.float64) - want)
    print(f"{name:50s} max_rel={O.max_rel_err(got, want):.3e} abs_rel={O.abs_rel(got, want):.3e} max|err|/max={err.max() / want.max():.3e} "
          f"finite={np.isfinite(got).all()}", flush=True)


CASES = (("T32", (1, 32, 3, 70, 98), 41), ("B2T3", (2, 3, 3, 56, 42), 43))
m = build("outliers")
g = np.load(os.path.join(ROOT, "tests/golden/g4c_forward_outliers.npz"))
for mode in ("half", "fp32"):
    m.precision = mode
    for name, shape, seed in CASES:
        x = torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32)).cuda()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = m(x).cpu().numpy()
        line(f"outliers {mode} {name} (warnings: {len(w)}, scale {m._ensure_engine().act_scale})", out, g[name])
m = build("smooth")
g = np.load(os.path.join(ROOT, "tests/golden/g4_forward.npz"))
m.precision = "fp32"
x = torch.from_numpy(np.random.RandomState(41).standard_normal((1, 32, 3, 70, 98)).astype(np.float32)).cuda()
eng = m._ensure_engine()
for fused in (True, False):
    eng.fused_split = fused
    for scale in (16.0, 1.0, 1 / 16):
        eng.act_scale = scale
        line(f"smooth fp32 T32 fused={fused} act_scale={scale}", m(x).cpu().numpy(), g["T32"])
