"""Run one vda_gemm shape/config a few times (for rocprofv3 counter passes).  args: cfg N K [M] [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
cfg, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 43840
it = int(sys.argv[5]) if len(sys.argv) > 5 else 3
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); bias = torch.randn(N, device="cuda")
for _ in range(it): hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg)
torch.cuda.synchronize()
