"""Edge-shape sweep of the forward pass against the CPU oracle, in both storage modes: python tools/edge_check.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vda_oracle as O
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything
sd = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**spec.MODEL_CONFIGS["vits"], seed=1).items()}
m = VideoDepthAnything(**spec.MODEL_CONFIGS["vits"]); m.load_state_dict(sd, strict=True); m = m.to("cuda").eval()
for shape in [(1, 1, 3, 14, 14), (1, 2, 3, 14, 14), (1, 8, 3, 14, 14), (1, 1, 3, 28, 28), (1, 3, 3, 14, 70), (2, 2, 3, 28, 14), (1, 5, 3, 42, 154),
              (3, 1, 3, 56, 56), (1, 32, 3, 14, 28)]:
    x = torch.from_numpy(np.random.RandomState(sum(shape)).standard_normal(shape).astype(np.float32))
    want = O.forward(sd, x, "vits").numpy()
    res = []
    for prec in ("half", "fp32"):
        m.precision = prec
        got = m(x.cuda()).cpu().numpy()
        res.append((O.max_rel_err(got, want), O.abs_rel(got, want)))
    m.precision = "half"
    print(shape, "half max_rel %.2e abs_rel %.2e | fp32 max_rel %.2e abs_rel %.2e" % (res[0] + res[1]), "max depth %.3f" % want.max())
