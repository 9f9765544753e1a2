"""The one-wave-per-SIMD 256x256 GEMM kernel (tile_cfg 4) against the eight-wave one (tile_cfg 2): equal bits on the encoder's
product forms (plain, GELU, fp32 residual; interior and cut tiles), then launch times, interleaved.
    python tools/w4_check.py [rounds]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def run(cfg, a, w, bias, N, K, form, M):
    if form == "resid":
        out = torch.arange(M * N, device="cuda", dtype=torch.float32).reshape(M, N).remainder_(7.0)
        hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, resid=out, ldr=N, tile_cfg=cfg)
    else:
        out = torch.full((M, N), float("nan"), device="cuda", dtype=a.dtype)
        hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, act=hip.ACT_GELU if form == "gelu" else hip.ACT_NONE, tile_cfg=cfg)
    return out


ok = True
for dt in (torch.bfloat16, torch.float16):
    for M, N, K, form in ((2740, 3072, 1024, "plain"), (2740, 4096, 1024, "gelu"), (2740, 1024, 4096, "resid"), (2740, 1024, 1024, "resid"),
                          (700, 768, 1024, "plain"), (512, 512, 64, "plain"), (256, 256, 128, "resid"), (43840, 1024, 4096, "resid")):
        a = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(dt)
        bias = torch.randn(N, device="cuda")
        ref, got = run(2, a, w, bias, N, K, form, M), run(4, a, w, bias, N, K, form, M)
        torch.cuda.synchronize()
        eq = torch.equal(ref, got)
        ok &= eq
        print(f"{dt} M={M} N={N} K={K} {form}: equal={eq} max diff {(ref.float() - got.float()).abs().max().item():.3e}", flush=True)
if not ok:
    sys.exit("results differ")

M = 43840
for name, N, K, form in (("qkv", 3072, 1024, "plain"), ("fc1+gelu", 4096, 1024, "gelu"), ("proj+resid", 1024, 1024, "resid"), ("fc2+resid", 1024, 4096, "resid")):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if form == "resid" else torch.bfloat16)
    kw = dict(resid=out, ldr=N) if form == "resid" else dict(act=hip.ACT_GELU if form == "gelu" else hip.ACT_NONE)
    times = {2: [], 4: []}
    for r in range(rounds + 1):
        for cfg in (2, 4):
            for _ in range(3):
                hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg, **kw)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[cfg].append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
    print(name, "eight waves", times[2], "four waves", times[4], "us per launch", flush=True)
