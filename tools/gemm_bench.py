"""Micro-benchmark of vda_gemm tile configurations on the encoder's shapes (random bf16 data, interleaved rounds)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4").split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 43840
shapes = [tuple(int(v) for v in x.split("x")) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]
dev = "cuda"
res = {}
for N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.randn(N, device=dev)
    ref = None
    for c in cfgs:
        hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=c)
        if ref is None: ref = out.clone()
        else: assert torch.equal(ref, out), (N, K, c, (ref.float() - out.float()).abs().max())
    torch.cuda.synchronize()
    for rnd in range(3):
        for c in cfgs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=c)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((N, K, c), []).append(e0.elapsed_time(e1) / 5)
for (N, K, c), ts in res.items():
    t = min(ts); print(f"N={N:5d} K={K:5d} cfg={c}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:8.1f} TF/s  (median {sorted(ts)[1]*1e3:.1f} us)")
