"""Isolated launch times of the encoder's four products with and without the folded-LayerNorm forms (same operands, same
buffers): separates the cost of the epilogue code from what the surrounding kernels leave in the caches.
    python tools/fold_time.py [frames]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "monocular-depth-estimation_amd"))
import torch  # noqa: E402

from video_depth_anything import hip  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M, D = frames * 1370, 1024
dev = "cuda"
g = torch.Generator().manual_seed(0)
bf = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(torch.bfloat16).to(dev)  # noqa: E731
f32 = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
y, att, hid = bf(M, D), bf(M, D), bf(M, 4 * D)
xs = f32(M, D)
stat = torch.stack([torch.zeros(M), torch.ones(M)], 1).contiguous().to(dev)
part = torch.empty((M, D // 64, 2), dtype=torch.float32, device=dev)
y2 = torch.empty((M, D), dtype=torch.bfloat16, device=dev)


def timed(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {1e3 * e0.elapsed_time(e1) / n:8.1f} us", flush=True)


for name, N, K, a, act in (("qkv", 3 * D, D, y, hip.ACT_NONE), ("fc1", 4 * D, D, y, hip.ACT_GELU)):
    w, b, cs = bf(N, K), f32(N), f32(N)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for rep in range(2):
        timed(name + " plain", lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=b, act=act))
        timed(name + " folded norm", lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=b, act=act, row_stat=stat, w_colsum=cs))
for name, K, a in (("proj", D, att), ("fc2", 4 * D, hid)):
    w, b = bf(D, K), f32(D)
    for rep in range(2):
        timed(name + " plain", lambda: hip.gemm(a, w, xs, M=M, N=D, K=K, lda=K, ldc=D, bias=b, resid=xs, ldr=D))
        timed(name + " emitting", lambda: hip.gemm(a, w, xs, M=M, N=D, K=K, lda=K, ldc=D, bias=b, resid=xs, ldr=D, out16=y2, row_part=part))
