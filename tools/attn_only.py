"""One shape of vda_attention in a loop (for counter passes): python tools/attn_only.py [frames heads n_tok iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
frames, heads, n_tok, iters = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 16, 1370, 10)))
D = heads * 64
qkv = torch.randn(frames, n_tok, 3 * D, device="cuda")
qkv[..., :D] *= 0.125          # the model folds the 1/8 score scale into q
qkv = qkv.bfloat16()
out = torch.empty(frames, n_tok, D, device="cuda", dtype=torch.bfloat16)
for _ in range(3): hip.attention(qkv, out, frames, n_tok, heads)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): hip.attention(qkv, out, frames, n_tok, heads)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters
print(f"attention {frames}x{heads}x{n_tok}: {t*1e3:.1f} us  {4.0*frames*heads*n_tok*n_tok*64/t/1e9:.1f} TF/s")
