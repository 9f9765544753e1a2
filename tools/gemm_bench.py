"""Micro-benchmark of vda_gemm tile configurations on the encoder's shapes (random bf16 data, interleaved rounds)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4").split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 43840
# NxK or NxKxR; R = 1: fp32 output updated in place with gamma * (acc + bias) + residual; R = 2: the same without gamma
# (the encoder's form, LayerScale folded into the weights)
shapes = [tuple(int(v) for v in x.split("x")) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(3072, 1024), (1024, 1024, 1), (4096, 1024), (1024, 4096, 1)]
shapes = [(s + (0,))[:3] for s in shapes]
dev = "cuda"
res = {}
for N, K, R in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    if R:
        x0 = torch.randn(M, N, device=dev); out = x0.clone(); gamma = torch.rand(N, device=dev) * 1e-3
        kw = dict(bias=bias, gamma=gamma, resid=out, ldr=N) if R == 1 else dict(bias=bias, resid=out, ldr=N)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); kw = dict(bias=bias)
    ref = None
    for c in cfgs:
        if R: out.copy_(x0)
        hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, tile_cfg=c, **kw)
        if ref is None: ref = out.clone()
        else: assert torch.equal(ref, out), (N, K, c, (ref.float() - out.float()).abs().max())
    torch.cuda.synchronize()
    for rnd in range(3):
        for c in cfgs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, tile_cfg=c, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((N, K, R, c), []).append(e0.elapsed_time(e1) / 5)
for (N, K, R, c), ts in res.items():
    t = min(ts); print(f"N={N:5d} K={K:5d} resid={R} cfg={c}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:8.1f} TF/s  (median {sorted(ts)[1]*1e3:.1f} us)")
