"""Launch time of the 16-bit attention kernel at the encoder's shapes (32 frames): python tools/attn_time.py
VDA_ATTN_WIDE=0 keeps the four-wave workgroups only (A/B runs in separate processes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip

res = []
for name, frames, n_tok, heads in (("vitl518", 32, 1370, 16), ("vits518", 32, 1370, 6), ("vitl518x924", 32, 2443, 16), ("vitl1036", 8, 5477, 16)):
    D = heads * 64
    qkv = (torch.randn(frames, n_tok, 3 * D, device="cuda") * 0.5).bfloat16()
    out = torch.empty(frames, n_tok, D, device="cuda", dtype=torch.bfloat16)
    best = []
    for r in range(3):
        for _ in range(3):
            hip.attention(qkv, out, frames, n_tok, heads)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.attention(qkv, out, frames, n_tok, heads)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20 * 1e3)
    res.append(f"{name} {min(best):.1f} us (sum {out.double().abs().sum().item():.9e})")
print("wide" if os.environ.get("VDA_ATTN_WIDE", "1") != "0" else "four-wave", "; ".join(res), flush=True)
