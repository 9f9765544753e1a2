"""Launch times of the ViT-S encoder's product shapes (518x518, 32 frames) on the 128x128 (tile_cfg 1) and 256x256 (2) kernels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
M = 43840
for name, N, K, form in (("qkv", 1152, 384, "plain"), ("fc1", 1536, 384, "gelu"), ("proj", 384, 384, "resid"), ("fc2", 384, 1536, "resid")):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if form == "resid" else torch.bfloat16)
    kw = dict(resid=out, ldr=N) if form == "resid" else dict(act=hip.ACT_GELU if form == "gelu" else hip.ACT_NONE)
    res = {}
    for cfg in (1, 2, 1, 2):
        for _ in range(5):
            hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg, **kw)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(cfg, []).append(round(e0.elapsed_time(e1) / 30 * 1e3, 1))
    gf = 2.0 * M * N * K / 1e9
    print(f"{name} N={N} K={K}: 128x128 {res[1]} us ({gf / min(res[1]) * 1e3:.0f} TFLOP/s), 256x256 {res[2]} us ({gf / min(res[2]) * 1e3:.0f} TFLOP/s)", flush=True)
