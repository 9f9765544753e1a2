"""Time-stamp breakdown of the persistent 256x256 GEMM (needs a library built with -DVDA_PROFILE256; VDA_HIP_LIB=path)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip
M = 43840
shapes = [tuple(x.split("x")) for x in (sys.argv[1] if len(sys.argv) > 1 else "3072x1024x0,1024x1024x1,1024x4096x1").split(",")]
lib = hip.lib()
R, S = 12, 8
names = ["top->loop", "loop", "params", "setup+prologue", "epilogue", "final wait"]
for N, K, res in shapes:
    N, K, res = int(N), int(K), int(res)
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    if res:
        out = torch.empty(M, N, device="cuda", dtype=torch.float32); r = torch.randn(M, N, device="cuda"); g = torch.rand(N, device="cuda")
        kw = dict(bias=bias, gamma=g, resid=r, ldr=N)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); kw = dict(bias=bias)
    for _ in range(3): hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, tile_cfg=2, **kw)
    torch.cuda.synchronize(); assert lib.vda_prof256_clear() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, tile_cfg=2, **kw); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * R * S, dtype=np.int64)
    assert lib.vda_prof256_read(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(256, 8, R, S).astype(np.float64) / 100.0      # us
    print(f"== N={N} K={K} resid={res}: {e0.elapsed_time(e1)*1e3:.1f} us total")
    t0 = t[:, :, 0, 0].min()
    full = t[:, :, :, 5] > 0
    for rd in range(R):
        if not full[:, :, rd].any(): break
        sel = full[:, :, rd]
        row = []
        for k in range(6):
            if k == 5 and not (t[:, :, rd, 6] > 0).any(): row.append("    -"); continue
            d = (t[:, :, rd, k + 1] - t[:, :, rd, k])[sel & ((t[:, :, rd, k + 1] > 0))]
            row.append(f"{d.mean():5.1f}")
        st = t[:, :, rd, 0][sel] - t0
        print(f" round {rd}: n_waves={sel.sum():4d} start {st.min():6.1f}..{st.max():6.1f} | " + " ".join(f"{n}={v}" for n, v in zip(names, row)))
    cyc = t[:, :, 1, 7] * 100.0; us = t[:, :, 1, 2] - t[:, :, 1, 1]
    print(f"   round1 loop: {cyc.mean():.0f} shader cycles in {us.mean():.2f} us -> {cyc.mean()/us.mean()/1e3:.3f} GHz")
    # per wave-group view of round 1
    for grp, sl in (("waves0-3", slice(0, 4)), ("waves4-7", slice(4, 8))):
        d = t[:, sl, 1, 1:7] - t[:, sl, 1, 0:6]
        print(f"   round1 {grp}: " + " ".join(f"{n}={d[..., k].mean():.1f}" for k, n in enumerate(names)))
