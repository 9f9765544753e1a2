"""Time of vda_attention_pairs at the ViT-L window shape (32 frames x 16 heads, 1370 tokens): HIP events around 20 launches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
frames, n_tok, heads = 32, 1370, 16
D = heads * 64
g = torch.Generator().manual_seed(0)
qkv2 = (torch.randn(frames * n_tok, 6 * D, generator=g) * 8).half().cuda()
out = torch.empty(frames * n_tok, 2 * D, dtype=torch.float16, device="cuda")
for _ in range(3):
    hip.attention_pairs(qkv2, out, frames, n_tok, heads, 16.0, out_layout=hip.PIECES_PAIRS)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    hip.attention_pairs(qkv2, out, frames, n_tok, heads, 16.0, out_layout=hip.PIECES_PAIRS)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"vda_attention_pairs {frames}x{heads}x{n_tok}: {us:.1f} us per launch, {3 * 4 * n_tok * n_tok * 64 * frames * heads / us / 1e6:.0f} TFLOP/s of matrix work")
