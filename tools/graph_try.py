"""Does replaying a window as a HIP graph beat eager launches?  python tools/graph_try.py [vits|vitl] [streams]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything
enc = sys.argv[1] if len(sys.argv) > 1 else "vits"
m = VideoDepthAnything(**spec.MODEL_CONFIGS[enc]).cuda().eval()
m.encoder_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x = torch.randn(1, 32, 3, 518, 518, device="cuda")
for _ in range(3): y = m(x)
torch.cuda.synchronize()
def bench(fn, n=20):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print(enc, "eager ms/window", bench(lambda: m(x)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m(x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    yg = m(x)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph output equals eager:", torch.equal(yg, y))
print(enc, "graph ms/window", bench(g.replay))
