"""Run the same window twice and report where the two results differ."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import spec, hip
from video_depth_anything.video_depth import VideoDepthAnything
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 518
cfg = spec.MODEL_CONFIGS[enc]
m = VideoDepthAnything(**cfg).to("cuda").eval()
x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, 32, 3, H, H)).astype(np.float32)).cuda()
a = m(x).clone()
for rep in range(3):
    b = m(x)
    d = (a - b).abs()
    print(f"rep {rep}: differing elements {(d > 0).sum().item()} of {d.numel()}, max abs diff {d.max().item():.3e}, max value {a.abs().max().item():.3e}")
    if (d > 0).any():
        idx = (d > 0).nonzero()
        print("  first differing indices", idx[:5].tolist(), "frames touched", sorted(set(idx[:, 1].tolist()))[:40])
# stage by stage
eng = m._engine if hasattr(m, "_engine") else None
print("engine attrs:", [k for k in dir(eng) if not k.startswith("__")][:40] if eng is not None else None)
