"""3x3 convolution kernels on the head's narrow layers: python tools/conv_bench.py [cfgs]   (cfg: 1 implicit GEMM, 3 direct)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,3").split(",")]
dev, dt = "cuda", torch.float16
cfgs_wide = [2 if c == 1 else c for c in cfgs]
for name, B, h, w, cin, cout, dot in (("output_conv1", 32, 296, 296, 256, 128, False), ("output_conv2 + tail", 32, 518, 518, 128, 32, True),
                                      ("refinenet1 unit conv", 32, 148, 148, 256, 256, False), ("refinenet2 unit conv", 32, 74, 74, 256, 256, False)):
    x = torch.randn(B, h, w, cin, device=dev).to(dt)
    wg = (torch.randn(cout, 9 * cin, device=dev) * 0.02).to(dt)
    bias = torch.randn(cout, device=dev)
    res = {}
    for rnd in range(3):
        for c in (cfgs_wide if cout >= 192 else cfgs):
            if dot:
                out = torch.empty(B, h, w, device=dev); dw = torch.rand(cout, device=dev)
                kw = dict(ldc=4, epi=hip.EPI_DOT, dot_w=dw, dot_b=0.1)
            else:
                out = torch.empty(B, h, w, cout, device=dev, dtype=dt); kw = dict(ldc=cout)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            hip.gemm(x, wg, out, M=B * h * w, N=cout, K=9 * cin, lda=0, bias=bias, conv=(h, w, cin, h, w, 1), tile_cfg=c, **kw)
            e0.record()
            for _ in range(3): hip.gemm(x, wg, out, M=B * h * w, N=cout, K=9 * cin, lda=0, bias=bias, conv=(h, w, cin, h, w, 1), tile_cfg=c, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(c, []).append(e0.elapsed_time(e1) / 3)
    for c, ts in res.items():
        t = min(ts); print(f"{name:22s} cfg={c}: {t*1e3:8.1f} us  {2.0*B*h*w*cout*9*cin/t/1e9:8.1f} TF/s")
