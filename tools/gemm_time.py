"""Launch times of the encoder's four product shapes (ViT-L 518x518, 32 frames) on the 256x256 kernel, for A/B runs of two
library builds (VDA_HIP_LIB selects the library).    python tools/gemm_time.py [tile_cfg]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = 43840
res = []
for name, N, K, form in (("qkv", 3072, 1024, "plain"), ("fc1", 4096, 1024, "gelu"), ("proj", 1024, 1024, "resid"), ("fc2", 1024, 4096, "resid")):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if form == "resid" else torch.bfloat16)
    kw = dict(resid=out, ldr=N) if form == "resid" else dict(act=hip.ACT_GELU if form == "gelu" else hip.ACT_NONE)
    best = []
    for r in range(3):
        for _ in range(5):
            hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg, **kw)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 30 * 1e3)
    res.append(f"{name} {min(best):.1f}")
print(os.path.basename(hip.LIB_PATH), " ".join(res), "us", flush=True)
