"""Diagnostic (not shipped, not a test): where does the bf16 error of the forward pass come from?
Runs on the GPU box:  python tools/err_budget.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vda_oracle as O
from video_depth_anything import spec
from video_depth_anything.video_depth import VideoDepthAnything
import torch.nn.functional as F

enc = sys.argv[1] if len(sys.argv) > 1 else "vits"
H, W, T = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (70, 98, 32)
cfg = spec.MODEL_CONFIGS[enc]
sd = {k: torch.from_numpy(v).cuda() for k, v in spec.synth_state_dict(**cfg, seed=1).items()}
m = VideoDepthAnything(**cfg); m.load_state_dict(sd, strict=True); m = m.cuda().eval()
x = torch.from_numpy(np.random.RandomState(41).standard_normal((1, T, 3, H, W)).astype(np.float32)).cuda()
hp, wp = H // 14, W // 14
def fin(d): return F.relu(F.interpolate(d, size=(H, W), mode="bilinear", align_corners=True)).squeeze(1).cpu().numpy()
with torch.no_grad():
    eng = m._ensure_engine()
    ref_taps = O.encoder_taps(sd, x.flatten(0, 1), enc)
    want = fin(O.head_forward(sd, ref_taps, hp, wp, T))
    my_taps = eng.encoder(x.reshape(T, 3, H, W).contiguous(), T, H, W)
    for i in range(4):
        a, b = my_taps[i].float().reshape(ref_taps[i].shape), ref_taps[i]
        print(f"tap{i}: rel rms err {((a-b).pow(2).mean().sqrt()/b.pow(2).mean().sqrt()).item():.3e}  (bf16 rounding alone {((b.bfloat16().float()-b).pow(2).mean().sqrt()/b.pow(2).mean().sqrt()).item():.3e})")
    got = eng.forward(x).cpu().numpy()[0]
    print(f"full          : max_rel {O.max_rel_err(got, want):.3e} abs_rel {O.abs_rel(got, want):.3e}")
    enc_only = fin(O.head_forward(sd, [t.float().reshape(r.shape) for t, r in zip(my_taps, ref_taps)], hp, wp, T))
    print(f"encoder only  : max_rel {O.max_rel_err(enc_only, want):.3e} abs_rel {O.abs_rel(enc_only, want):.3e}")
    head_only = eng.head([t.to(eng.hdt).reshape(-1, t.shape[-1]).contiguous() for t in ref_taps], 1, T, hp, wp).reshape(T, H, W).cpu().numpy()
    print(f"head only     : max_rel {O.max_rel_err(head_only, want):.3e} abs_rel {O.abs_rel(head_only, want):.3e}")
    r16 = fin(O.head_forward(sd, [t.bfloat16().float() for t in ref_taps], hp, wp, T))
    print(f"taps rounded  : max_rel {O.max_rel_err(r16, want):.3e} abs_rel {O.abs_rel(r16, want):.3e}")
    e = np.abs(head_only - want) / np.maximum(np.abs(want), 1e-3 * np.abs(want).max())
    idx = np.argwhere(e > 0.02)
    print("bad pixels:", len(idx), "of", e.size)
    if len(idx):
        print("frames:", np.unique(idx[:, 0])[:40]); print("rows:", np.unique(idx[:, 1])[:60]); print("cols:", np.unique(idx[:, 2])[:60])
        for f, y, x_ in idx[:10]: print(f, y, x_, head_only[f, y, x_], want[f, y, x_])
