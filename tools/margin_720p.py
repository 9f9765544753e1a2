"""Where the error of the 720p ViT-L video (the GPU test with the thinnest margin under the 1e-2 bound) comes from: the same
video through every precision form (packing.PRECISIONS), worst per-frame floored max-rel, the value of the reference at the
worst pixel relative to the frame's maximum, and the error there relative to the maximum.
    python tools/margin_720p.py [forms ...]          (VDA_ENCODER_DTYPE=fp16|bf16 selects the 16-bit encoder type)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vda_oracle as O
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
cfg = spec.MODEL_CONFIGS["vitl"]
sd = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**cfg, seed=1).items()}
m = VideoDepthAnything(**cfg)
m.load_state_dict(sd, strict=True)
m = m.cuda().eval()
sd_dev = {k: v.cuda() for k, v in sd.items()}

from test_model_gpu import synth_frames  # noqa: E402  (the frames of the test itself)
frames = synth_frames(34, 720, 1280, 4242)
want = O.infer_video_depth(None, list(frames), "vitl", batch_size=32, input_size=518,
                           forward_fn=lambda inp: O.forward(sd_dev, inp.cuda(), "vitl"))
for mode in (sys.argv[1:] or ("half", "tail32", "head32", "enc32", "fp32")):
    m.precision = mode
    got, _ = m.infer_video_depth(list(frames), 30, input_size=518, device="cuda", batch_size=32)
    per = [O.max_rel_err(got[t], want[t]) for t in range(34)]
    t = int(np.argmax(per))
    w, g = want[t].astype(np.float64), got[t].astype(np.float64)
    rel = np.abs(g - w) / np.maximum(np.abs(w), 1e-3 * w.max())
    y, x = np.unravel_index(int(np.argmax(rel)), rel.shape)
    print(f"{mode:7s} worst per-frame max_rel={max(per):.3e} (frame {t}, pixel {y},{x}: reference {w[y, x] / w.max():.3e} of the "
          f"maximum, error {abs(g[y, x] - w[y, x]) / w.max():.3e} of the maximum)  median per-frame {np.median(per):.3e}  "
          f"abs_rel={O.abs_rel(got, want):.3e}  min/max of the reference {w.min() / w.max():.3e}", flush=True)
    m._engines.pop(mode, None)
    torch.cuda.empty_cache()
