"""The encoder's four product shapes on the vendor BLAS behind torch.matmul (bf16 in, bf16 out, no bias / activation /
residual: less work than the library's fused products) beside this library's kernels, interleaved.
    python tools/blas_compare.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip

M = 43840


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K, form in (("qkv", 3072, 1024, "plain"), ("fc1", 4096, 1024, "gelu"), ("proj", 1024, 1024, "resid"), ("fc2", 1024, 4096, "resid")):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if form == "resid" else torch.bfloat16)
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(resid=out, ldr=N) if form == "resid" else dict(act=hip.ACT_GELU if form == "gelu" else hip.ACT_NONE)
    ours, blas = [], []
    for r in range(3):
        ours.append(timed(lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, **kw)))
        blas.append(timed(lambda: torch.matmul(a, w.t(), out=ob)))
    gf = 2.0 * M * N * K / 1e9
    print(f"{name}: this library (fused epilogue) {min(ours):.1f} us = {gf / min(ours) * 1e3:.0f} TFLOP/s; "
          f"torch.matmul (bare product) {min(blas):.1f} us = {gf / min(blas) * 1e3:.0f} TFLOP/s", flush=True)
