"""Do the 128x128 and the 256x256 tile kernels give the same bits?  (They add the K steps in the same order.)  Shapes of the
encoder's products at ViT-L and ViT-S width, forms with and without the fp32 residual.    python tools/tile_equal.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip

g = torch.Generator().manual_seed(0)
for M, N, K, resid in ((5480, 1024, 4096, True), (5480, 1024, 1024, True), (10960, 3072, 1024, False), (5480, 4096, 1024, False),
                       (2740, 384, 1536, True), (43840, 1024, 4096, True), (700, 1024, 4096, True)):
    a = (torch.randn(M, K, generator=g)).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if resid else None
    outs = []
    for cfg in (1, 2):
        out = torch.empty(M, N, dtype=torch.float32 if resid else torch.bfloat16, device="cuda")
        hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=b, resid=r, ldr=N, tile_cfg=cfg)
        outs.append(out.float())
    d = (outs[0] - outs[1]).abs()
    print(f"M={M} N={N} K={K} resid={resid}: equal={torch.equal(outs[0], outs[1])} max diff {d.max().item():.3e} "
          f"differing {int((d > 0).sum())} of {d.numel()}", flush=True)
