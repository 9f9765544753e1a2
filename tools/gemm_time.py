"""Time vda_gemm on given shapes and tile configurations (HIP events, interleaved rounds).
    python tools/gemm_time.py M [cfg ...]     shapes: the four encoder linears of ViT-S (K = 384) and ViT-L (K = 1024)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
M = int(sys.argv[1]) if len(sys.argv) > 1 else 43840
cfgs = [int(v) for v in sys.argv[2:]] or [0, 1, 2]
D = int(os.environ.get("D", "384"))
shapes = [("qkv", 3 * D, D, "own"), ("proj", D, D, "res"), ("fc1", 4 * D, D, "gelu"), ("fc2", D, 4 * D, "res")]
bf = torch.bfloat16
for name, N, K, kind in shapes:
    a = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") * 0.03).to(bf)
    bias = torch.randn(N, device="cuda")
    if kind == "res":
        out = torch.randn(M, N, device="cuda"); kw = dict(resid=out, ldr=N)
    else:
        out = torch.empty(M, N, device="cuda", dtype=bf); kw = dict(act=hip.ACT_GELU) if kind == "gelu" else {}
    res = {}
    for rnd in range(3):
        for c in cfgs:
            for _ in range(3): hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=c, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=c, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(c, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(name, f"M={M} N={N} K={K}", {c: round(min(v), 1) for c, v in res.items()}, "us")
