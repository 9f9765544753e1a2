"""Soak run of infer_video_depth on a long synthetic video: python tools/long_video.py [frames] [encoder] [h0] [w0]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
enc = sys.argv[2] if len(sys.argv) > 2 else "vitl"
h0 = int(sys.argv[3]) if len(sys.argv) > 3 else 518
w0 = int(sys.argv[4]) if len(sys.argv) > 4 else 518
sd = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**spec.MODEL_CONFIGS[enc], seed=1).items()}
m = VideoDepthAnything(**spec.MODEL_CONFIGS[enc]); m.load_state_dict(sd, strict=True); m = m.to("cuda").eval()
rs = np.random.RandomState(0)
base = rs.randint(0, 256, (64, h0, w0, 3), dtype=np.uint8)
frames = [base[i % 64] for i in range(n)]                       # a list of frames, as metric_depth/run2.py passes
m.infer_video_depth(frames[:40], 30, input_size=518, device="cuda", batch_size=32)
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
d, _ = m.infer_video_depth(frames, 30, input_size=518, device="cuda", batch_size=32)
dt = time.perf_counter() - t0
print(f"{enc} {n} frames {h0}x{w0}: {n / dt:.1f} video frames/s, depth {d.shape} finite={np.isfinite(d).all()} "
      f"min {d.min():.3f} max {d.max():.3f}, peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
# frames 0..63 repeat with period 64: the same input frame must get a similar depth wherever it appears (windows are aligned
# to one scale), a loose sanity check of the stitching over many windows
a, b = d[70], d[70 + 64 * ((n - 71) // 64)]
print("same frame, early vs late window: mean rel diff %.3e" % (np.abs(a - b).mean() / np.abs(a).mean()))
