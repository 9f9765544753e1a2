"""Launch times and output checksums of the DPT head's two fused-upsampling convolutions (ViT-L and ViT-S shapes of a 518x518
window, 32 frames), for A/B runs of two library builds (VDA_HIP_LIB selects the library).    python tools/conv_up_time.py"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip

B = 32
res = []
for name, sh, sw, h, w_, cin, used, cout, dot in (("vitl_oc1", 148, 148, 296, 296, 256, 0, 128, False), ("vitl_oc2", 296, 296, 518, 518, 128, 0, 32, True),
                                                  ("vits_oc1", 148, 148, 296, 296, 64, 0, 32, False), ("vits_oc2", 296, 296, 518, 518, 64, 32, 32, True)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, sh, sw, cin), generator=g).half().cuda()
    wt = (torch.randn((cout, 9 * cin), generator=g) / math.sqrt(9 * cin)).half().cuda()
    bias = (torch.randn((cout,), generator=g) * 0.1).cuda()
    kw = dict(M=B * h * w_, N=cout, K=9 * cin, lda=0, bias=bias, conv=(h, w_, cin, h, w_, 1), upsample_from=(sh, sw), cin_used=used)
    if dot:
        out = torch.empty((B, h, w_), dtype=torch.float32, device="cuda")
        kw.update(ldc=4, epi=hip.EPI_DOT, dot_w=(torch.randn((32,), generator=g) * 0.3).cuda(), dot_b=0.05)
    else:
        out = torch.empty((B, h, w_, cout), dtype=torch.float16, device="cuda")
        kw.update(ldc=cout)
    best = []
    for r in range(3):
        for _ in range(2):
            hip.gemm(x, wt, out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.gemm(x, wt, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 5 * 1e3)
    chk = out.double().abs().sum().item()
    res.append(f"{name} {min(best):.0f} us (sum {chk:.9e})")
print(os.path.basename(hip.LIB_PATH), "; ".join(res), flush=True)
