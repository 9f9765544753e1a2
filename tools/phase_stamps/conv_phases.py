"""Per-phase cycle counts of the fused-upsampling convolutions (the DPT head's output_conv1 / output_conv2 shapes, ViT-L and ViT-S),
from the time-stamped build of tools/phase_stamps/build.sh (VDA_HIP_LIB must point at it).  Ticks are shader-clock cycles of the
workgroup's own CU; the counters of different XCDs have different origins, so only differences inside a workgroup mean something."""
import ctypes, math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))
from video_depth_anything import hip

L = hip.lib()
L.vda_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
B = 32
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
    for _ in range(3):
        hip.gemm(x, wt, out, **kw)
    torch.cuda.synchronize()
    assert L.vda_stamps_clear() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.gemm(x, wt, out, **kw); e1.record(); torch.cuda.synchronize()
    nwg = B * ((h + 7) // 8) * ((w_ + 31) // 32)
    buf = np.zeros((40000, 32), dtype=np.uint64)
    assert L.vda_stamps_read(buf.ctypes.data, buf.nbytes) == 0
    t = buf[:nwg].astype(np.int64)
    slices = max(((used or cin) + 63) // 64, 1)
    life = t[:, 31] - t[:, 0]
    print(f"{name}: launch {e0.elapsed_time(e1) * 1e3:.0f} us (stamps included), {nwg} workgroups; lifetime ticks mean {life.mean():.0f} "
          f"p10 {np.percentile(life, 10):.0f} p90 {np.percentile(life, 90):.0f}")
    names = ["slice start", "parked + barrier", "blended", "written", "product steps done"]
    prev = t[:, 0]
    for cc in range(slices):
        for k in range(5):
            cur = t[:, 1 + cc * 5 + k]
            d = cur - prev
            print(f"   slice {cc} {names[k]:18s} +{d.mean():8.0f} ticks (p10 {np.percentile(d, 10):.0f}, p90 {np.percentile(d, 90):.0f})")
            prev = cur
    print(f"   epilogue                   +{(t[:, 31] - prev).mean():8.0f} ticks")
