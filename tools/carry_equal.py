"""Is a video the same, bit for bit, with and without the carry-over of encoder outputs between windows, and with the encoder's
frames on one stream or two?    python tools/carry_equal.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything

for enc, n, h, w, size in (("vits", 77, 42, 56, 42), ("vits", 60, 518, 518, 518), ("vitl", 40, 280, 392, 280)):
    cfg = spec.MODEL_CONFIGS[enc]
    m = VideoDepthAnything(**cfg).cuda().eval()
    frames = np.random.RandomState(1).randint(0, 256, (n, h, w, 3), dtype=np.uint8)
    for fp32 in (False, True):
        outs = {}
        for reuse in (True, False):
            for streams in (2, 1):
                m.reuse_encoder_features, m.encoder_streams = reuse, streams
                outs[(reuse, streams)], _ = m.infer_video_depth(frames, 30, input_size=size, device="cuda", fp32=fp32, batch_size=32)
        ref = outs[(False, 1)]
        for k, o in outs.items():
            print(f"{enc} {h}x{w} fp32={fp32} reuse={k[0]} streams={k[1]}: equal={np.array_equal(o, ref)} "
                  f"max diff / max = {np.abs(o - ref).max() / np.abs(ref).max():.2e}", flush=True)
    del m
    torch.cuda.empty_cache()
