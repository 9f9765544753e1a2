"""Each HIP kernel, called through the C ABI, against a plain PyTorch fp32 statement of the same operation.

Inputs are rounded to bf16 first so both sides see identical operands; the tolerance then only covers fp32
accumulation order and the final bf16 rounding of the kernel's output (2^-9 relative).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from video_depth_anything import hip  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


HALVES = [torch.bfloat16, torch.float16]
# workgroup tiles of vda_gemm: library's choice, forced 128x128, forced 256x256 pipelined (small shapes exercise its M/N tails)
TILE_CFGS = [0, 1, 2, 3]      # 3 = direct 3x3 convolution kernel where the call allows it, else the library's choice


def rnd(shape, seed, scale=1.0, dtype=BF):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def close(got, want, rtol, name=""):
    got, want = got.float(), want.float()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= rtol * ref, f"{name}: max err {err:.3e} vs max ref {ref:.3e} (rtol {rtol})"


BF_OUT = 6e-3      # bf16 output rounding (2^-8 worst case incl. accumulation noise)
F32_OUT = 2e-5
STORAGE = HALVES + [torch.float32]          # activation storage types: 16-bit modes and the fp32 mode
OUT_TOL = {torch.bfloat16: BF_OUT, torch.float16: BF_OUT, torch.float32: F32_OUT}
# three-term hi/lo product, relative to the largest output: bf16 pieces 2^-16 per operand pair; IEEE-half pieces 2^-21,
# where the bound is set by the fp32 accumulation over K
SPLIT_TOL = {torch.bfloat16: 6e-5, torch.float16: 2e-6}
LAYOUTS = [hip.PIECES_TRIPLE, hip.PIECES_PAIRS]


def hi_lo_of(pieces, K, layout):
    """(hi, lo) [rows, K] of a matrix of operand pieces in either layout."""
    if layout == hip.PIECES_PAIRS:
        b = pieces.reshape(pieces.shape[0], K // 32, 2, 32)
        return b[:, :, 0].reshape(-1, K), b[:, :, 1].reshape(-1, K)
    return pieces[:, :K], pieces[:, K:2 * K]




def test_library_loads_and_reports_version():
    assert b"gfx950" in hip.lib().vda_version()


@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (1370 * 2, 1152, 384), (77, 48 * 4, 1536), (513, 64, 64), (200, 32, 192)])
def test_gemm_plain_exact_integers(M, N, K, cfg):
    """Small-integer operands make every product and sum exact: catches any fragment/layout mix-up bit for bit."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), generator=g).to(BF).to(DEV)
    w = torch.randint(-3, 4, (N, K), generator=g).to(BF).to(DEV)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, tile_cfg=cfg)
    want = a.float() @ w.float().t()
    assert torch.equal(out, want)


@pytest.mark.parametrize("M,N,K,ldr", [(1370, 1024, 1024, 1024), (700, 256, 4096, 256), (256, 512, 1024, 640), (3, 256, 1536, 256)])
def test_gemm_fp32_residual_on_the_large_tile_exact_integers(M, N, K, ldr):
    """fp32 residual + bias on the 256x256 tile, the form of the encoder's residual updates (LayerScale is folded into the
    weights).  Integer data makes the expected result exact; rows cut by M and a strided residual included."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), generator=g).to(BF).to(DEV)
    w = torch.randint(-3, 4, (N, K), generator=g).to(BF).to(DEV)
    bias = torch.randint(-8, 9, (N,), generator=g).float().to(DEV)
    resid = torch.randint(-1000, 1001, (M, ldr), generator=g).float().to(DEV)
    want = a.float() @ w.float().t() + bias + resid[:, :N]
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, resid=resid, ldr=ldr, tile_cfg=2)
    assert torch.equal(out, want)
    if ldr == N:                                                        # in place, as the encoder's residual stream is updated
        x = resid.clone()
        hip.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, resid=x, ldr=N, tile_cfg=2)
        assert torch.equal(x, want)


@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("M,N,K", [(1370, 1024, 1024), (999, 384, 1536)])
def test_gemm_epilogue_bias_gelu_gamma_residual(M, N, K, cfg):
    a, w = rnd((M, K), 1), rnd((N, K), 2, 1 / math.sqrt(K))
    bias, gamma = rnd((N,), 3, 0.1, torch.float32), rnd((N,), 4, 1.0, torch.float32)
    resid = rnd((M, N), 5, 1.0, torch.float32)
    acc = a.float() @ w.float().t() + bias
    out = torch.empty((M, N), dtype=BF, device=DEV)
    hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, act=hip.ACT_GELU, tile_cfg=cfg)
    close(out, F.gelu(acc), BF_OUT, "gelu")
    x = resid.clone()
    hip.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, gamma=gamma, resid=x, ldr=N, tile_cfg=cfg)     # in place on the residual
    close(x, resid + gamma * acc, F32_OUT, "layerscale residual")
    x = resid.clone()      # no gamma: the encoder's residual update once LayerScale is folded into the weights
    hip.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, resid=x, ldr=N, tile_cfg=cfg)
    close(x, resid + acc, F32_OUT, "plain residual")
    r16, r2 = rnd((M, N), 6), rnd((M, N), 7)
    hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, act=hip.ACT_RELU, resid=r16, ldr=N, resid2=r2, ldr2=N, tile_cfg=cfg)
    close(out, F.relu(acc) + r16.float() + r2.float(), BF_OUT, "relu + 2 residuals")


@pytest.mark.parametrize("cfg", [1, 2])
@pytest.mark.parametrize("dt", HALVES)
@pytest.mark.parametrize("form", ["own", "own_relu", "own_res", "own_res32", "own_res2", "f32", "f32_res32", "f32_res16"])
def test_gemm_every_epilogue_form(form, dt, cfg):
    """Each epilogue form the kernels are instantiated for (and one that falls back to the general epilogue, f32_res16),
    on both tile kernels, M and N ragged against the tiles: bias, activation, residual(s) and output type combinations."""
    M, N, K = 333, 320, 1024
    a, w = rnd((M, K), 1, 1.0, dt), rnd((N, K), 2, 1 / math.sqrt(K), dt)
    bias = rnd((N,), 3, 0.1, torch.float32)
    want = a.float() @ w.float().t() + bias
    kw = {}
    out_dt = torch.float32 if form.startswith("f32") else dt
    if form == "own_relu":
        kw["act"] = hip.ACT_RELU
        want = F.relu(want)
    if form in ("own_res", "own_res2", "f32_res16"):
        r = rnd((M, N), 5, 1.0, dt)
        kw.update(resid=r, ldr=N)
        want = want + r.float()
    if form in ("own_res32", "f32_res32"):
        r = rnd((M, N), 5, 1.0, torch.float32)
        kw.update(resid=r, ldr=N)
        want = want + r
    if form == "own_res2":
        r2 = rnd((M, N), 6, 1.0, dt)
        kw.update(resid2=r2, ldr2=N)
        want = want + r2.float()
    out = torch.full((M, N), float("nan"), dtype=out_dt, device=DEV)
    hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, tile_cfg=cfg, **kw)
    close(out, want, OUT_TOL[out_dt] if out_dt != torch.float32 else 1e-4, form)


@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("dt", HALVES)
def test_gemm_geglu(dt, cfg):
    M, C = 700, 192
    a, w = rnd((M, C), 1, 1.0, dt), rnd((8 * C, C), 2, 1 / math.sqrt(C), dt)
    bias = rnd((8 * C,), 3, 0.1, torch.float32)
    proj = a.float() @ w.float().t() + bias
    val, gate = proj.chunk(2, dim=-1)
    want = val * F.gelu(gate)
    # interleave value/gate rows in blocks of 16
    n = 4 * C
    wv, wg = w[:n].reshape(n // 16, 16, C), w[n:].reshape(n // 16, 16, C)
    wi = torch.stack([wv, wg], 1).reshape(8 * C, C).contiguous()
    bi = torch.stack([bias[:n].reshape(n // 16, 16), bias[n:].reshape(n // 16, 16)], 1).reshape(-1).contiguous()
    out = torch.empty((M, n), dtype=dt, device=DEV)
    hip.gemm(a, wi, out, M=M, N=8 * C, K=C, lda=C, ldc=n, bias=bi, epi=hip.EPI_GEGLU, tile_cfg=cfg)
    close(out, want, BF_OUT, "geglu")


@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("k,C,Co", [(4, 64, 64), (2, 128, 128), (2, 64, 96)])   # 96: a 64-column block straddles two taps (unstaged stores)
def test_gemm_conv_transpose(k, C, Co, cfg):
    B, h, w_ = 3, 5, 7
    x = rnd((B, C, h, w_), 1)
    wt = rnd((C, Co, k, k), 2, 1 / math.sqrt(C))          # [Cin, Cout, k, k]
    bias = rnd((Co,), 3, 0.1, torch.float32)
    want = F.conv_transpose2d(x.float(), wt.float(), bias, stride=k).permute(0, 2, 3, 1)
    a = x.permute(0, 2, 3, 1).reshape(B * h * w_, C).contiguous()
    wg = wt.permute(2, 3, 1, 0).reshape(k * k * Co, C).contiguous()       # n = (ky*k+kx)*Co + co, k-dim = ci
    out = torch.full((B, h * k, w_ * k, Co), float("nan"), dtype=BF, device=DEV)
    hip.gemm(a, wg, out, M=B * h * w_, N=k * k * Co, K=C, lda=C, ldc=Co, bias=bias, epi=hip.EPI_CONVT, convt=(h, w_, k, Co), tile_cfg=cfg)
    close(out, want, BF_OUT, "convT")


@pytest.mark.parametrize("dt", HALVES)
@pytest.mark.parametrize("stride,cin,cout,relu_in", [(1, 64, 64, True), (1, 128, 256, False), (2, 192, 192, False), (1, 64, 32, False)])
@pytest.mark.parametrize("cfg", TILE_CFGS)
def test_gemm_conv3x3(stride, cin, cout, relu_in, dt, cfg):
    B, h, w_ = 2, 19, 13
    x = rnd((B, cin, h, w_), 1, 1.0, dt)
    wt = rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin), dt)
    bias = rnd((cout,), 3, 0.1, torch.float32)
    xin = F.relu(x.float()) if relu_in else x.float()
    want = F.conv2d(xin, wt.float(), bias, stride=stride, padding=1).permute(0, 2, 3, 1)
    oh, ow = want.shape[1], want.shape[2]
    a = x.permute(0, 2, 3, 1).contiguous()
    wg = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
    out = torch.full((B, oh, ow, cout), float("nan"), dtype=dt, device=DEV)
    hip.gemm(a, wg, out, M=B * oh * ow, N=cout, K=9 * cin, lda=0, ldc=cout, bias=bias, relu_in=relu_in,
             conv=(h, w_, cin, oh, ow, stride), tile_cfg=cfg)
    close(out, want, BF_OUT, "conv3x3")


@pytest.mark.parametrize("cfg", [1, 3])
@pytest.mark.parametrize("B,h,w_,cin", [(2, 20, 30, 64), (1, 37, 70, 128)])
def test_gemm_conv3x3_dot_tail(B, h, w_, cin, cfg):
    x = rnd((B, cin, h, w_), 1)
    wt = rnd((32, cin, 3, 3), 2, 1 / math.sqrt(9 * cin))
    bias = rnd((32,), 3, 0.3, torch.float32)
    w2 = rnd((32,), 4, 0.3, torch.float32).abs()
    want = F.relu(F.conv2d(F.relu(F.conv2d(x.float(), wt.float(), bias, padding=1)), w2.reshape(1, 32, 1, 1), torch.tensor([0.1], device=DEV)))
    a = x.permute(0, 2, 3, 1).contiguous()
    wg = wt.permute(0, 2, 3, 1).reshape(32, 9 * cin).contiguous()
    out = torch.full((B * h * w_,), float("nan"), dtype=torch.float32, device=DEV)
    hip.gemm(a, wg, out, M=B * h * w_, N=32, K=9 * cin, lda=0, ldc=4, bias=bias, epi=hip.EPI_DOT, dot_w=w2, dot_b=0.1,
             conv=(h, w_, cin, h, w_, 1), tile_cfg=cfg)
    close(out.reshape(B, 1, h, w_), want, 1e-4, "conv+dot tail")


@pytest.mark.parametrize("out_dt", [torch.float16, torch.float32])
@pytest.mark.parametrize("B,h,w_,cin,cout,act", [(2, 19, 45, 64, 128, False), (1, 40, 33, 256, 128, True), (3, 8, 32, 128, 64, True),
                                                 (1, 9, 70, 192, 32, False)])
def test_conv3x3_direct_kernel_matches_implicit_gemm_and_torch(B, h, w_, cin, cout, act, out_dt):
    """Patches that are cut by the right and bottom image borders, several 64-channel slices, every supported width."""
    dt = torch.float16
    x = rnd((B, cin, h, w_), 1, 1.0, dt)
    wt = rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin), dt)
    bias = rnd((cout,), 3, 0.1, torch.float32)
    want = F.conv2d(x.float(), wt.float(), bias, padding=1)
    want = (F.relu(want) if act else want).permute(0, 2, 3, 1)
    a = x.permute(0, 2, 3, 1).contiguous()
    wg = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
    outs = []
    for cfg in (1, 3):
        out = torch.full((B, h, w_, cout), float("nan"), dtype=out_dt, device=DEV)
        hip.gemm(a, wg, out, M=B * h * w_, N=cout, K=9 * cin, lda=0, ldc=cout, bias=bias, act=hip.ACT_RELU if act else hip.ACT_NONE,
                 conv=(h, w_, cin, h, w_, 1), tile_cfg=cfg)
        close(out, want, OUT_TOL[out_dt] if out_dt != torch.float32 else 1e-4, f"conv3x3 cfg {cfg}")
        outs.append(out)
    close(outs[1], outs[0], OUT_TOL[out_dt] if out_dt != torch.float32 else 1e-5, "direct vs implicit GEMM")


@pytest.mark.parametrize("B,h,w_,C", [(2, 19, 45, 256), (1, 40, 33, 128)])
def test_conv3x3_direct_kernel_residual_unit(B, h, w_, C):
    """The second convolution of a residual unit: ReLU on the input, two residual maps added (util/blocks.py:78-91, 147-152)."""
    dt = torch.float16
    x, r1, r2 = rnd((B, h, w_, C), 1, 1.0, dt), rnd((B, h, w_, C), 4, 1.0, dt), rnd((B, h, w_, C), 5, 1.0, dt)
    wt = rnd((C, C, 3, 3), 2, 1 / math.sqrt(9 * C), dt)
    bias = rnd((C,), 3, 0.1, torch.float32)
    want = F.conv2d(F.relu(x.float().permute(0, 3, 1, 2)), wt.float(), bias, padding=1).permute(0, 2, 3, 1) + r1.float() + r2.float()
    wg = wt.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
    outs = []
    for cfg in (1, 3):
        out = torch.full((B, h, w_, C), float("nan"), dtype=dt, device=DEV)
        hip.gemm(x, wg, out, M=B * h * w_, N=C, K=9 * C, lda=0, ldc=C, bias=bias, relu_in=True, resid=r1.reshape(-1, C), ldr=C,
                 resid2=r2.reshape(-1, C), ldr2=C, conv=(h, w_, C, h, w_, 1), tile_cfg=cfg)
        close(out, want, BF_OUT, f"residual unit conv cfg {cfg}")
        outs.append(out)
    close(outs[1], outs[0], 1e-3, "direct vs implicit GEMM")      # same products; the fp32 accumulation order differs


@pytest.mark.parametrize("dt", HALVES)
@pytest.mark.parametrize("B,sh,sw,h,w_,cin,cout,mode", [
    (2, 10, 23, 20, 46, 128, 128, "relu"),        # x2 as in the last fusion block, patches cut by both borders
    (1, 21, 17, 37, 30, 128, 32, "dot"),          # x1.75 as before the output tail, fused 1x1 tail
    (2, 9, 9, 33, 65, 64, 64, "plain"),           # 64 output channels, one channel slice, large ratios
    (1, 12, 40, 12, 80, 192, 256, "plain"),       # one axis unchanged; two 128-channel workgroup columns; three slices
    (1, 1, 1, 5, 7, 64, 32, "relu_in"),           # single source pixel: every sample is that pixel
    (1, 8, 70, 17, 72, 64, 32, "plain"),          # width ratio above 0.9: the four-tap loader from global memory instead of the parked source patch
    (1, 20, 50, 21, 100, 128, 128, "plain"),      # height ratio near 1 with the parked source patch (11 source rows)
])
def test_conv3x3_direct_kernel_with_fused_bilinear_input(B, sh, sw, h, w_, cin, cout, mode, dt):
    """conv3x3(bilinear(x, align_corners=True)) with the resampling done inside the convolution's patch loader
    (util/blocks.py:154-160, dpt_temporal.py:88-100): against torch in fp32 on the same 16-bit inputs, and against this
    library's own two-kernel sequence, whose intermediate map is rounded to 16 bits exactly like the fused patch."""
    x = rnd((B, sh, sw, cin), 1, 1.0, dt)
    wt = rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin), dt)
    bias = rnd((cout,), 3, 0.1, torch.float32)
    wg = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
    up = F.interpolate(x.float().permute(0, 3, 1, 2), size=(h, w_), mode="bilinear", align_corners=True)
    up16 = torch.empty((B, h, w_, cin), dtype=dt, device=DEV)
    hip.bilinear_nhwc(x, up16, B, sh, sw, h, w_, cin)
    close(up16, up.permute(0, 2, 3, 1), 5e-3, "bilinear")
    kw = dict(M=B * h * w_, N=cout, K=9 * cin, lda=0, bias=bias)
    if mode == "dot":
        dw, db = rnd((32,), 4, 0.3, torch.float32), 0.05
        want = F.relu((F.relu(F.conv2d(up16.float().permute(0, 3, 1, 2), wt.float(), bias, padding=1)) * dw.view(1, -1, 1, 1)).sum(1) + db)
        fused = torch.full((B, h, w_), float("nan"), dtype=torch.float32, device=DEV)
        two = torch.full_like(fused, float("nan"))
        kw.update(ldc=4, epi=hip.EPI_DOT, dot_w=dw, dot_b=db)
        tol = 2e-3
    else:
        xin = up16.float().permute(0, 3, 1, 2)
        want = F.conv2d(F.relu(xin) if mode == "relu_in" else xin, wt.float(), bias, padding=1)
        want = (F.relu(want) if mode == "relu" else want).permute(0, 2, 3, 1)
        fused = torch.full((B, h, w_, cout), float("nan"), dtype=dt, device=DEV)
        two = torch.full_like(fused, float("nan"))
        kw.update(ldc=cout, act=hip.ACT_RELU if mode == "relu" else hip.ACT_NONE, relu_in=mode == "relu_in")
        tol = BF_OUT
    hip.gemm(x, wg, fused, conv=(h, w_, cin, h, w_, 1), upsample_from=(sh, sw), **kw)
    hip.gemm(up16, wg, two, conv=(h, w_, cin, h, w_, 1), tile_cfg=3, **kw)
    close(fused, want, tol, "fused vs torch")
    # the two paths may round a resampled value to different neighbours (fp32 blend order), which can move an output
    # by one unit of its 16-bit rounding, nothing more
    close(fused, two, tol, "fused vs bilinear kernel + direct convolution")


@pytest.mark.parametrize("dt", [torch.float16, BF])
@pytest.mark.parametrize("B,sh,sw,h,w_,cin,used,cout,mode", [
    (2, 11, 19, 21, 37, 64, 32, 32, "dot"),        # the ViT-S output tail: a 32-channel map stored 64 wide, resampled, 1x1 tail
    (1, 9, 30, 17, 60, 128, 96, 64, "plain"),      # one and a half slices
    (2, 14, 35, 14, 35, 64, 32, 32, "same"),       # no resampling: the plain halo copy with a half slice
    (1, 8, 70, 17, 72, 64, 32, 32, "plain"),       # width ratio above 0.9: four-tap loader from global memory, half slice
])
def test_conv3x3_direct_kernel_uses_only_the_first_channels_of_a_pixel(B, sh, sw, h, w_, cin, used, cout, mode, dt):
    """cin_used: channels from `used` on are not multiplied and need not hold numbers (they are NaN here, and so are the
    weights that would meet them)."""
    x = rnd((B, sh, sw, cin), 1, 1.0, dt)
    wt = rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * used), dt)
    bias = rnd((cout,), 3, 0.1, torch.float32)
    xs = x[..., :used].float().permute(0, 3, 1, 2)
    if mode != "same":
        xs = F.interpolate(xs, size=(h, w_), mode="bilinear", align_corners=True)
        xs = xs.to(dt).float()                             # the fused patch is rounded to the storage type
    want = F.conv2d(xs, wt[:, :used].float(), bias, padding=1)
    x[..., used:] = float("nan")
    wt[:, used:] = float("nan")
    wg = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
    kw = dict(M=B * h * w_, N=cout, K=9 * cin, lda=0, bias=bias, conv=(h, w_, cin, h, w_, 1), cin_used=used)
    if mode != "same":
        kw["upsample_from"] = (sh, sw)
    if mode == "dot":
        dw, db = rnd((32,), 4, 0.3, torch.float32), 0.05
        want = F.relu((F.relu(want) * dw.view(1, -1, 1, 1)).sum(1) + db)
        out = torch.full((B, h, w_), float("nan"), dtype=torch.float32, device=DEV)
        hip.gemm(x, wg, out, ldc=4, epi=hip.EPI_DOT, dot_w=dw, dot_b=db, **kw)
        close(out, want, 4e-3 if dt == BF else 2e-3, "half slice, dot tail")
    else:
        out = torch.full((B, h, w_, cout), float("nan"), dtype=dt, device=DEV)
        hip.gemm(x, wg, out, ldc=cout, **kw)
        close(out, want.permute(0, 2, 3, 1), BF_OUT, "half slice")
    # not a multiple of 32, more than cin, or a kernel that cannot skip channels: refused
    scratch = torch.empty((B, h, w_, cout), dtype=dt, device=DEV)
    for bad in (24, cin + 32):
        with pytest.raises(RuntimeError):
            hip.gemm(x, wg, scratch, ldc=cout, **{**kw, "cin_used": bad})
    wg48 = rnd((48, 9 * cin), 5, 0.1, dt)
    with pytest.raises(RuntimeError):                                   # 48 output channels: implicit GEMM, which reads whole pixels
        hip.gemm(x, wg48, torch.empty((B, h, w_, 48), dtype=dt, device=DEV), ldc=48, **{**kw, "N": 48, "bias": None})


def test_fused_bilinear_input_is_refused_where_the_direct_kernel_does_not_apply():
    x = rnd((1, 5, 5, 64), 1, 1.0, torch.float16)
    wg = rnd((48, 9 * 64), 2, 0.1, torch.float16)
    out = torch.empty((1, 10, 10, 48), dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError):                                   # 48 output channels: no direct kernel, no silent fallback
        hip.gemm(x, wg, out, M=100, N=48, K=9 * 64, lda=0, ldc=48, conv=(10, 10, 64, 10, 10, 1), upsample_from=(5, 5))


def softmax2(scores):
    """The kernel's scores are base-2 logarithms (log2(e) is folded into q with the 1/8): probabilities 2^s / sum 2^s."""
    return torch.softmax(scores * math.log(2.0), -1)


ATTN_TYPES = [BF, torch.float16]
ATTN_TOL = {BF: 1.2e-2, torch.float16: 2e-3}


@pytest.mark.parametrize("frames,n_tok,heads", [(2, 1370, 6), (3, 64, 2), (1, 129, 16), (2, 36, 1), (1, 4100, 2), (1, 5477, 1)])
@pytest.mark.parametrize("dt", ATTN_TYPES)
def test_attention(frames, n_tok, heads, dt):
    D = heads * 64
    qkv = rnd((frames, n_tok, 3 * D), 1, 1.0, dt)
    out = torch.full((frames, n_tok, D), float("nan"), dtype=dt, device=DEV)
    hip.attention(qkv, out, frames, n_tok, heads)
    q, k, v = qkv.float().reshape(frames, n_tok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (softmax2(q @ k.transpose(-1, -2)) @ v).transpose(1, 2).reshape(frames, n_tok, D)
    close(out, want, ATTN_TOL[dt], "attention")       # probabilities are rounded to the operand type before the second product


@pytest.mark.parametrize("n_tok", [200, 4200])       # 32 and 64 queries per wave
@pytest.mark.parametrize("dt", ATTN_TYPES)
def test_attention_sharp_softmax_rescale(n_tok, dt):
    """One key dominates from a later K/V tile on: forces the running-max rescale branch."""
    frames, heads = 1, 1
    qkv = rnd((frames, n_tok, 192), 2, 0.5, dt)
    qkv[0, 150, 64:128] = qkv[0, 10, 0:64] * 12          # key 150 aligned with query 10
    qkv[0, n_tok - 7, 64:128] = qkv[0, 45, 0:64] * 12    # and one in the last tile, for a query of the second block
    out = torch.empty((frames, n_tok, 64), dtype=dt, device=DEV)
    hip.attention(qkv, out, frames, n_tok, heads)
    q, k, v = qkv.float().reshape(frames, n_tok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (softmax2(q @ k.transpose(-1, -2)) @ v).transpose(1, 2).reshape(frames, n_tok, 64)
    close(out, want, ATTN_TOL[dt], "attention rescale")


@pytest.mark.parametrize("n_tok", [130, 4200])
@pytest.mark.parametrize("offset", [-300.0, 300.0])
@pytest.mark.parametrize("dt", ATTN_TYPES)
def test_attention_scores_far_from_zero(n_tok, offset, dt):
    """Every score of a query is near `offset`: the first tile must set the running maximum whatever its sign (2^-300 and
    2^300 are not fp32 numbers), and later tiles stay relative to it."""
    frames, heads = 1, 1
    qkv = rnd((frames, n_tok, 192), 5, 0.5, dt)
    qkv[..., 63] = 16.0                                   # q[63] = 16
    qkv[..., 127] = offset / 16.0                         # k[63]: adds `offset` to every score (exact in both operand types)
    out = torch.full((frames, n_tok, 64), float("nan"), dtype=dt, device=DEV)
    hip.attention(qkv, out, frames, n_tok, heads)
    q, k, v = qkv.double().reshape(frames, n_tok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (softmax2(q @ k.transpose(-1, -2)) @ v).transpose(1, 2).reshape(frames, n_tok, 64).float()
    close(out, want, ATTN_TOL[dt], "attention, shifted scores")


@pytest.mark.parametrize("rows,C", [(1370 * 2, 1024), (100, 384), (37, 64), (50, 192)])
def test_layernorm(rows, C):
    x = rnd((rows, C), 1, 2.0, torch.float32) + 0.5
    w, b = rnd((C,), 2, 1.0, torch.float32), rnd((C,), 3, 1.0, torch.float32)
    for dt, tol in ((BF, BF_OUT), (torch.float16, 1e-3), (torch.float32, F32_OUT)):
        out = torch.empty((rows, C), dtype=dt, device=DEV)
        hip.layernorm(x, out, w, b, rows, C, 1e-6)
        close(out, F.layer_norm(x, (C,), w, b, 1e-6), tol, "layernorm")


def test_layernorm_drop_class_token_and_position_table():
    frames, n_tok, C = 3, 36, 384
    x = rnd((frames, n_tok, C), 1, 1.0, torch.float32)
    w, b = rnd((C,), 2, 1.0, torch.float32), rnd((C,), 3, 1.0, torch.float32)
    out = torch.empty((frames, n_tok - 1, C), dtype=BF, device=DEV)
    hip.layernorm(x, out, w, b, frames * (n_tok - 1), C, 1e-6, rows_per_group=n_tok - 1, in_group=n_tok, in_skip=1)
    close(out, F.layer_norm(x, (C,), w, b, 1e-6)[:, 1:], BF_OUT, "tap layernorm")
    B, T, hw = 2, 5, 12
    x = rnd((B * T, hw, C), 4, 1.0, torch.float32)
    pe = rnd((32, C), 5, 1.0, torch.float32)
    out = torch.empty((B * T, hw, C), dtype=BF, device=DEV)
    hip.layernorm(x, out, w, b, B * T * hw, C, 1e-5, pe=pe, pe_hw=hw, pe_T=T)
    want = F.layer_norm(x, (C,), w, b, 1e-5).reshape(B, T, hw, C) + pe[:T].reshape(1, T, 1, C)
    close(out, want.reshape(B * T, hw, C), BF_OUT, "layernorm + pe")


# ---- LayerNorm folded into the products around it (VdaGemmArgs.out16 / row_part / row_stat / w_colsum) --------------------------
def _stats_of(rounded):
    v = rounded.double()
    mean = v.mean(1)
    return mean, 1.0 / torch.sqrt(v.var(1, unbiased=False) + 1e-6)


@pytest.mark.parametrize("dt", HALVES)
@pytest.mark.parametrize("rows,C", [(2740, 1024), (100, 384), (37, 64), (5, 192)])
def test_round_rowstats(rows, C, dt):
    x = rnd((rows, C), 1, 2.0, torch.float32) + 0.5
    out = torch.empty((rows, C), dtype=dt, device=DEV)
    stat = torch.full((rows, 2), float("nan"), dtype=torch.float32, device=DEV)
    hip.round_rowstats(x, out, stat, rows, C, 1e-6)
    assert torch.equal(out, x.to(dt))                       # one rounding, nothing else
    mean, rstd = _stats_of(out)
    close(stat[:, 0].double(), mean, 2e-6, "mean of the rounded rows")
    close(stat[:, 1].double(), rstd, 2e-6, "rstd of the rounded rows")


@pytest.mark.parametrize("cfg", [1, 2])
@pytest.mark.parametrize("dt", HALVES)
@pytest.mark.parametrize("M,N,K", [(1370, 1024, 1024), (333, 384, 1536), (256, 128, 64), (700, 1024, 4096)])
def test_gemm_emits_rounded_rows_and_partial_sums(M, N, K, dt, cfg):
    """Producer side: the fp32 output is unchanged, out16 is that output rounded once, row_part holds the sums of the ROUNDED
    values per 64-column block, and vda_rowstats_finish turns them into the rows' mean and 1/std.  Interior tiles take the
    DMA epilogue of the 256x256 kernel, the rows cut by M its staged one; the 128x128 kernel always the staged one."""
    a, w = rnd((M, K), 1, 1.0, dt), rnd((N, K), 2, 1 / math.sqrt(K), dt)
    bias = rnd((N,), 3, 0.1, torch.float32)
    resid = rnd((M, N), 5, 1.0, torch.float32) + 0.25
    plain = resid.clone()
    hip.gemm(a, w, plain, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, resid=plain, ldr=N, tile_cfg=cfg)
    x = resid.clone()
    x16 = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
    part = torch.full((M, N // 64, 2), float("nan"), dtype=torch.float32, device=DEV)
    hip.gemm(a, w, x, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, resid=x, ldr=N, out16=x16, row_part=part, tile_cfg=cfg)
    assert torch.equal(x, plain)                             # the emitting form computes the same fp32 rows, bit for bit
    assert torch.equal(x16, x.to(dt))
    blocks = x16.double().reshape(M, N // 64, 64)
    close(part[..., 0].double(), blocks.sum(2), 1e-6, "block sums")
    close(part[..., 1].double(), (blocks * blocks).sum(2), 1e-6, "block sums of squares")
    stat = torch.full((M, 2), float("nan"), dtype=torch.float32, device=DEV)
    hip.rowstats_finish(part, stat, M, N // 64, 1e-6)
    mean, rstd = _stats_of(x16)
    close(stat[:, 0].double(), mean, 1e-5, "mean")
    close(stat[:, 1].double(), rstd, 1e-5, "rstd")


def test_emitted_partial_sums_do_not_depend_on_the_row_position():
    """The same rows give the same sums, bit for bit, wherever they lie: in an interior 256-row tile (DMA epilogue), in the tile
    cut by M (staged epilogue), on the 128x128 kernel.  A frame's result must not depend on the batch it is computed in."""
    N, K = 1024, 1024
    a, w = rnd((600, K), 1), rnd((N, K), 2, 1 / math.sqrt(K))
    resid = rnd((600, N), 5, 1.0, torch.float32)

    def run(rows, cfg):
        M = len(rows)
        x = resid[rows].clone()
        x16 = torch.empty((M, N), dtype=BF, device=DEV)
        part = torch.empty((M, N // 64, 2), dtype=torch.float32, device=DEV)
        hip.gemm(a[rows].contiguous(), w, x, M=M, N=N, K=K, lda=K, ldc=N, resid=x, ldr=N, out16=x16, row_part=part, tile_cfg=cfg)
        return x16, part

    all_rows = torch.arange(600, device=DEV)
    x16, part = run(all_rows, 2)                      # rows 0..511 interior, 512..599 in the cut tile
    for rows, cfg in ((all_rows[500:600], 2), (all_rows[0:100], 2), (all_rows[200:333], 1), (all_rows[512:600], 1)):
        x16_b, part_b = run(rows, cfg)
        assert torch.equal(x16_b, x16[rows]) and torch.equal(part_b, part[rows]), (int(rows[0]), cfg)


@pytest.mark.parametrize("cfg", [1, 2])
@pytest.mark.parametrize("dt", HALVES)
@pytest.mark.parametrize("M,N,K,gelu", [(1370, 3072, 1024, False), (333, 1536, 384, True), (300, 4096, 1024, True), (40, 192, 64, False)])
def test_gemm_normalises_its_sums_like_a_layernorm_in_front(M, N, K, gelu, dt, cfg):
    """Consumer side: with W' = W diag(g) and colsum(W'), rstd (x W'^T - mean colsum) + (b + W beta) is LayerNorm(x) W^T + b."""
    x = rnd((M, K), 1, 2.0, torch.float32) + 0.3
    lw, lb = rnd((K,), 2, 0.3, torch.float32) + 1.0, rnd((K,), 3, 0.2, torch.float32)
    wgt, bias = rnd((N, K), 4, 1 / math.sqrt(K), torch.float32), rnd((N,), 6, 0.1, torch.float32)
    x16 = torch.empty((M, K), dtype=dt, device=DEV)
    stat = torch.empty((M, 2), dtype=torch.float32, device=DEV)
    hip.round_rowstats(x, x16, stat, M, K, 1e-6)
    wl = (wgt * lw[None, :]).to(dt)
    cs = wl.double().sum(1).float()
    bl = (bias.double() + wgt.double() @ lb.double()).float()
    out = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
    hip.gemm(x16, wl, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bl, act=hip.ACT_GELU if gelu else hip.ACT_NONE,
             row_stat=stat, w_colsum=cs, tile_cfg=cfg)
    want = F.layer_norm(x16.float(), (K,), lw, lb, 1e-6) @ wgt.t() + bias          # the fold is exact for the ROUNDED rows ...
    if gelu:
        want = F.gelu(want)
    close(out, want, 1.2e-2 if dt == BF else 2e-3, "folded layernorm")            # ... up to the rounding of W' and the output


@pytest.mark.parametrize("dt", STORAGE)
@pytest.mark.parametrize("frames,hw,C", [(4, 37 * 37, 256), (3, 100, 192), (2, 361, 384), (2, 50, 64)])
def test_groupnorm(frames, hw, C, dt):
    x = rnd((frames, hw, C), 1, 1.5, dt) + 0.3
    w, b = rnd((C,), 2, 1.0, torch.float32), rnd((C,), 3, 1.0, torch.float32)
    out = torch.empty_like(x)
    work = torch.empty(frames * hip.GROUPNORM_MAX_SPLITS * 32 * 2, dtype=torch.float64, device=DEV)
    hip.groupnorm(x, out, w, b, work, frames, hw, C, 32, 1e-6)
    first = out.clone()
    hip.groupnorm(x, out, w, b, work, frames, hw, C, 32, 1e-6)
    assert torch.equal(first, out)          # no atomics: bit-identical from run to run
    want = F.group_norm(x.float().permute(0, 2, 1), 32, w, b, 1e-6).permute(0, 2, 1)
    close(out, want, OUT_TOL[dt], "groupnorm")


@pytest.mark.parametrize("dt", STORAGE)
@pytest.mark.parametrize("ih,iw,oh,ow", [(19, 19, 37, 37), (37, 66, 74, 132), (5, 7, 10, 14), (8, 8, 14, 14), (6, 6, 6, 6)])
def test_bilinear(ih, iw, oh, ow, dt):
    frames, C = 2, 64
    x = rnd((frames, ih, iw, C), 1, 1.0, dt)
    out = torch.empty((frames, oh, ow, C), dtype=dt, device=DEV)
    hip.bilinear_nhwc(x, out, frames, ih, iw, oh, ow, C)
    want = F.interpolate(x.float().permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    close(out, want, OUT_TOL[dt], "bilinear nhwc")
    xf = rnd((frames, ih, iw), 2, 1.0, torch.float32)
    of = torch.empty((frames, oh, ow), dtype=torch.float32, device=DEV)
    hip.bilinear_f32(xf, of, frames, ih, iw, oh, ow, relu=True)
    want = F.relu(F.interpolate(xf.unsqueeze(1), size=(oh, ow), mode="bilinear", align_corners=True)).squeeze(1)
    close(of, want, 1e-5, "bilinear f32")


def test_patch_embed_front_end():
    frames, H, W, D = 2, 70, 98, 384
    x = rnd((frames, 3, H, W), 1, 1.0, torch.float32)
    wt = rnd((D, 3, 14, 14), 2, 1 / math.sqrt(588))
    bias = rnd((D,), 3, 0.1, torch.float32)
    hp, wp = H // 14, W // 14
    cols32 = torch.empty((frames * hp * wp, 640), dtype=torch.float32, device=DEV)
    hip.im2col_patch14(x, cols32, frames, H, W, 640)
    cols = torch.empty((frames * hp * wp, 640), dtype=BF, device=DEV)
    hip.im2col_patch14(x, cols, frames, H, W, 640)
    assert torch.equal(cols32.to(BF), cols)
    want_cols = F.unfold(x, 14, stride=14).transpose(1, 2).reshape(frames * hp * wp, 588)
    assert torch.equal(cols[:, :588].float(), want_cols.to(BF).float())
    assert torch.count_nonzero(cols[:, 588:]) == 0
    wp_ = torch.zeros((D, 640), dtype=BF, device=DEV)
    wp_[:, :588] = wt.reshape(D, 588)
    tok = torch.empty((frames * hp * wp, D), dtype=torch.float32, device=DEV)
    hip.gemm(cols, wp_, tok, M=frames * hp * wp, N=D, K=640, lda=640, ldc=D, bias=bias)
    want = F.conv2d(x.to(BF).float(), wt.float(), bias, stride=14).flatten(2).transpose(1, 2)
    close(tok.reshape(frames, hp * wp, D), want, F32_OUT * 5, "patch embed")
    cls, pos = rnd((D,), 4, 1.0, torch.float32), rnd((hp * wp + 1, D), 5, 1.0, torch.float32)
    xt = torch.empty((frames, hp * wp + 1, D), dtype=torch.float32, device=DEV)
    hip.assemble_tokens(tok, cls, pos, xt, frames, hp * wp, D)
    want = torch.cat([cls.expand(frames, 1, D), tok.reshape(frames, hp * wp, D)], 1) + pos
    assert torch.equal(xt, want)


@pytest.mark.parametrize("dt", STORAGE)
@pytest.mark.parametrize("B,T,hw,heads,hd", [(1, 32, 12, 8, 24), (2, 5, 7, 8, 48), (1, 32, 9, 8, 8), (1, 32, 10, 8, 128), (1, 4, 6, 8, 32), (1, 1, 5, 8, 8),
                                             (2, 7, 5, 8, 128), (1, 32, 3, 4, 64), (1, 13, 4, 2, 96), (1, 1, 3, 8, 32), (1, 32, 50, 8, 32),
                                             (1, 6, 4, 3, 32), (1, 32, 5, 6, 8), (2, 9, 3, 16, 16)])
def test_temporal_attention(B, T, hw, heads, hd, dt):
    C = heads * hd
    qkv = rnd((B * T * hw, 3 * C), 1, 1.0, dt)
    out = torch.full((B * T * hw, C), float("nan"), dtype=dt, device=DEV)
    hip.temporal_attention(qkv, out, B, T, hw, heads, hd, hd ** -0.5)
    q, k, v = qkv.float().reshape(B, T, hw, 3, heads, hd).permute(3, 0, 2, 4, 1, 5)      # [B, hw, heads, T, hd]
    want = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v
    want = want.permute(0, 3, 1, 2, 4).reshape(B * T * hw, C)
    close(out, want, OUT_TOL[dt], "temporal attention")


@pytest.mark.parametrize("h0,w0,H,W", [(48, 86, 70, 126), (120, 200, 56, 98), (37, 53, 42, 56), (28, 42, 28, 42)])
def test_preprocess_u8_against_cubic_restatement(h0, w0, H, W):
    """uint8 frames -> /255 -> bicubic resize (OpenCV INTER_CUBIC arithmetic) -> normalisation, against the oracle's numpy
    restatement of the same published arithmetic (parity with OpenCV itself is unpinned: cv2 is not available)."""
    import vda_oracle as O
    frames = np.random.RandomState(h0 + W).randint(0, 256, (3, h0, w0, 3), dtype=np.uint8)
    out = torch.full((3, 3, H, W), float("nan"), dtype=torch.float32, device=DEV)
    hip.preprocess_u8(torch.from_numpy(frames).to(DEV), out, 3, h0, w0, H, W)
    want = np.stack([O.preprocess_frame(f, H, W) for f in frames])
    err = np.abs(out.cpu().numpy() - want).max()
    assert err <= 2e-5, f"max abs err {err:.3e}"          # values are O(1); both sides are fp32 sums of 16 taps


def test_stitch_kernels():
    n = 300 * 200 * 2
    p = rnd((n,), 1, 1.0, torch.float32).abs() * 3 + 1
    t = 1.7 * p + 0.3 + rnd((n,), 2, 0.01, torch.float32)
    sums_all = torch.zeros(hip.LSQ_SUMS_DOUBLES, dtype=torch.float64, device=DEV)
    hip.scale_shift_sums(p, t, sums_all, n)
    again = torch.zeros_like(sums_all)
    hip.scale_shift_sums(p, t, again, n)
    assert torch.equal(sums_all[:5], again[:5])             # fixed summation order: bit-identical
    sums = sums_all[:5]
    pd, td = p.double(), t.double()
    want = torch.stack([(pd * pd).sum(), pd.sum(), torch.tensor(float(n), device=DEV, dtype=torch.float64), (pd * td).sum(), td.sum()])
    assert torch.allclose(sums, want, rtol=1e-12)
    ss = torch.empty(2, dtype=torch.float32, device=DEV)
    hip.solve_scale_shift(sums, ss)
    assert abs(ss[0].item() - 1.7) < 1e-3 and abs(ss[1].item() - 0.3) < 3e-3
    x, pre = rnd((n,), 3, 1.0, torch.float32), rnd((n,), 4, 1.0, torch.float32)
    out = torch.empty_like(x)
    hip.affine_blend(x, pre, out, ss, 0.25, 0.75, n)
    want = 0.25 * pre + 0.75 * torch.clamp(x * ss[0] + ss[1], min=0)
    assert torch.allclose(out, want, rtol=1e-6, atol=1e-6)
    hip.affine_blend(x, None, out, ss, 0.0, 1.0, n)
    assert torch.allclose(out, torch.clamp(x * ss[0] + ss[1], min=0), rtol=1e-6, atol=1e-6)


def split_w(w, piece, layout=0):
    from video_depth_anything.packing import SplitWeight
    return SplitWeight(w, piece, layout)


ACT_SCALE = {torch.bfloat16: 1.0, torch.float16: 16.0}     # what the engine multiplies activations by before the split


@pytest.mark.parametrize("piece", HALVES)
@pytest.mark.parametrize("rows,C,Cp,relu", [(1000, 256, 256, False), (37, 48, 64, True), (5, 640, 640, False)])
def test_split3_layout_and_accuracy(rows, C, Cp, relu, piece):
    x = rnd((rows, C + 4), 1, 3.0, torch.float32)[:, :C]                  # row stride larger than C
    out = torch.full((rows, 3 * Cp), float("nan"), dtype=piece, device=DEV)
    sc = ACT_SCALE[piece]
    hip.split3(x, out, rows, C, x.stride(0), Cp, relu=relu, scale=sc)
    ref = (F.relu(x) if relu else x) * sc
    hi, lo, hi2 = out[:, :C].float(), out[:, Cp:Cp + C].float(), out[:, 2 * Cp:2 * Cp + C].float()
    assert torch.equal(hi, ref.to(piece).float()) and torch.equal(hi, hi2)
    assert torch.equal(lo, (ref - hi).to(piece).float())
    if piece == BF:
        assert ((hi + lo) - ref).abs().max() <= 2.0 ** -16 * ref.abs().max()
    else:       # 2^-22 relative while the low half is a normal number, 2^-25 absolute below
        assert (((hi + lo) - ref).abs() <= 2.0 ** -22 * ref.abs() + 2.0 ** -25).all()
    for seg in range(3):
        assert torch.count_nonzero(out[:, seg * Cp + C:(seg + 1) * Cp]) == 0


def test_split3_half_pieces_saturate_instead_of_overflowing():
    x = torch.tensor([[1e6, -1e6, 70000.0, 3.0]], dtype=torch.float32, device=DEV)
    out = torch.empty((1, 3 * 64), dtype=torch.float16, device=DEV)
    hip.split3(x, out, 1, 4, 4, 64, scale=1.0)
    assert torch.isfinite(out.float()).all()
    assert out[0, 0].item() == 65504.0 and out[0, 1].item() == -65504.0
    assert out[0, 2].item() + out[0, 64 + 2].item() == 70000.0        # the low half takes up the excess


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("piece", HALVES)
@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("M,N,K", [(1370, 1024, 1024), (300, 192, 384), (500, 64, 64), (260, 32, 128)])
def test_gemm_fp32_mode_three_term_product(M, N, K, cfg, piece, layout):
    """fp32 operands through vda_split3 / split weights: close to an fp64 product, ~100x (bf16 pieces) or ~3000x (IEEE-half
    pieces, operands scaled by powers of two and the sums scaled back) closer than one 16-bit pass."""
    a, w = rnd((M, K), 1, 1.0, torch.float32), rnd((N, K), 2, 1 / math.sqrt(K), torch.float32)
    bias = rnd((N,), 3, 0.1, torch.float32)
    Ks = hip.pieces_width(layout, K)
    a3 = torch.empty((M, Ks), dtype=piece, device=DEV)
    hip.split3(a, a3, M, K, K, K, scale=ACT_SCALE[piece], layout=layout)
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    sw = split_w(w, piece, layout)
    kw = dict(M=M, N=N, K=Ks, lda=Ks, ldc=N, bias=bias, tile_cfg=cfg, acc_scale=1.0 / (sw.scale * ACT_SCALE[piece]),
              ab_pairs=layout == hip.PIECES_PAIRS)
    hip.gemm(a3, sw.t, out, **kw)
    want = (a.double() @ w.double().t() + bias.double()).float()
    close(out, want, SPLIT_TOL[piece], "three-term gemm")
    r = rnd((M, N), 4, 1.0, torch.float32)
    res = r.clone()
    hip.gemm(a3, sw.t, res, resid=res, ldr=N, **kw)                      # the residual-stream form, in place
    close(res, want + r, SPLIT_TOL[piece], "three-term gemm with an fp32 residual")
    with pytest.raises(RuntimeError):            # the epilogue's GELU is for 16-bit outputs; fp32 callers use vda_split3's
        hip.gemm(a3, sw.t, out, act=hip.ACT_GELU, **kw)


def test_split3_gelu_and_geglu_are_exact_form():
    """The activations the fp32 mode applies while splitting: erf-form GELU (mlp.py:37) and value * GELU(gate) on rows laid
    out in blocks [16 values | 16 gates] (motion_module/attention.py:382-384), to fp32 accuracy."""
    rows, C = 333, 192
    x = rnd((rows, C), 1, 2.0, torch.float32)
    out = torch.full((rows, 3 * C), float("nan"), dtype=BF, device=DEV)
    hip.split3(x, out, rows, C, C, C, act=hip.SPLIT_GELU)
    got = out[:, :C].float() + out[:, C:2 * C].float()
    want = F.gelu(x.double()).float()
    assert (got - want).abs().max() <= 2e-6 * want.abs().max() + 2.0 ** -16 * want.abs().max()
    assert torch.equal(out[:, :C], out[:, 2 * C:])
    half = torch.empty((rows, 3 * C), dtype=torch.float16, device=DEV)
    hip.split3(x, half, rows, C, C, C, act=hip.SPLIT_GELU, scale=16.0)
    got = (half[:, :C].float() + half[:, C:2 * C].float()) / 16.0
    assert (got - want).abs().max() <= 1e-6 * want.abs().max()
    xg = rnd((rows, 2 * C), 2, 2.0, torch.float32)
    hip.split3(xg, out, rows, C, 2 * C, C, act=hip.SPLIT_GEGLU)
    blocks = xg.reshape(rows, C // 16, 2, 16)
    want = (blocks[:, :, 0].double() * F.gelu(blocks[:, :, 1].double())).reshape(rows, C).float()
    got = out[:, :C].float() + out[:, C:2 * C].float()
    assert (got - want).abs().max() <= 2e-6 * want.abs().max() + 2.0 ** -16 * want.abs().max()


def test_gemm_gelu_polynomial_accuracy():
    """The epilogue's GELU over the whole range that matters, through a one-hot product that passes the inputs unchanged:
    against the erf form in fp64, within the stated 5.4e-5 absolute (before the 16-bit rounding of the output)."""
    K = 64
    xs = torch.linspace(-9.0, 9.0, 128 * K).reshape(128, K)                     # fp16 keeps these to 2^-11 relative
    a = xs.to(torch.float16).to(DEV)
    for c0 in range(0, K, 64):
        w = torch.eye(K, dtype=torch.float16, device=DEV)
        out = torch.empty((128, K), dtype=torch.float16, device=DEV)
        hip.gemm(a, w, out, M=128, N=K, K=K, lda=K, ldc=K, act=hip.ACT_GELU)
        want = F.gelu(a.double())
        err = (out.double() - want).abs()
        assert (err <= 5.4e-5 + 2.0 ** -11 * want.abs() + 1e-7).all(), err.max()


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("cout", [128, 64, 32])
@pytest.mark.parametrize("piece", HALVES)
def test_conv3x3_fp32_mode(piece, cout, cfg, layout):
    B, h, w_, cin = 2, 19, 45, 64
    x = rnd((B, cin, h, w_), 1, 1.0, torch.float32)
    wt = rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin), torch.float32)
    bias = rnd((cout,), 3, 0.1, torch.float32)
    want = F.conv2d(F.relu(x).double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1).float()
    a = x.permute(0, 2, 3, 1).contiguous().reshape(B * h * w_, cin)
    cs = hip.pieces_width(layout, cin)
    a3 = torch.empty((B * h * w_, cs), dtype=piece, device=DEV)
    hip.split3(a, a3, B * h * w_, cin, cin, cin, relu=True, scale=ACT_SCALE[piece], layout=layout)
    wg = split_w(wt.permute(0, 2, 3, 1).contiguous(), piece, layout).reshape(cout, 9 * cs)
    out = torch.empty((B, h, w_, cout), dtype=torch.float32, device=DEV)
    hip.gemm(a3, wg.t, out, M=B * h * w_, N=cout, K=9 * cs, lda=0, ldc=cout, bias=bias, conv=(h, w_, cs, h, w_, 1),
             acc_scale=1.0 / (wg.scale * ACT_SCALE[piece]), tile_cfg=cfg, ab_pairs=layout == hip.PIECES_PAIRS)
    close(out, want, SPLIT_TOL[piece], "three-term conv3x3")


@pytest.mark.parametrize("frames,n_tok,heads", [(2, 1370, 2), (1, 129, 3), (2, 36, 1)])
def test_attention_fp32_mode(frames, n_tok, heads):
    D = heads * 64
    qkv = rnd((frames, n_tok, 3 * D), 1, 1.0, torch.float32)
    qkv[..., :D] *= 0.35
    out = torch.full((frames, n_tok, D), float("nan"), dtype=torch.float32, device=DEV)
    hip.attention(qkv, out, frames, n_tok, heads)
    q, k, v = qkv.double().reshape(frames, n_tok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (softmax2(q @ k.transpose(-1, -2)) @ v).transpose(1, 2).reshape(frames, n_tok, D).float()
    close(out, want, 3e-6, "attention, fp32 mode")


# ---- fp32 mode: the producers write the operand pieces themselves ------------------------------------------------------------
def pieces_of(ref, piece=torch.float16):
    hi = ref.clamp(-65504.0, 65504.0).to(piece) if piece == torch.float16 else ref.to(piece)
    lo = (ref - hi.float()).to(piece)
    return hi, lo


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("piece", HALVES)
@pytest.mark.parametrize("rows,C", [(2740, 1024), (100, 384), (37, 64)])
def test_layernorm_writes_operand_pieces(rows, C, piece, layout):
    """vda_layernorm with split_scale: the rows vda_split3 would make of the fp32 result, bit for bit."""
    x = rnd((rows, C), 1, 2.0, torch.float32) + 0.3
    w, b = rnd((C,), 2, 0.3, torch.float32) + 1.0, rnd((C,), 3, 0.2, torch.float32)
    y = torch.empty((rows, C), dtype=torch.float32, device=DEV)
    hip.layernorm(x, y, w, b, rows, C, 1e-6)
    width = hip.pieces_width(layout, C)
    want = torch.empty((rows, width), dtype=piece, device=DEV)
    hip.split3(y, want, rows, C, C, C, scale=ACT_SCALE[piece], layout=layout)
    hi, lo = hi_lo_of(want, C, layout)
    assert torch.equal(hi.float(), (y * ACT_SCALE[piece]).to(piece).float()) and torch.equal(lo.float(), (y * ACT_SCALE[piece] - hi.float()).to(piece).float())
    got = torch.full((rows, width), float("nan"), dtype=piece, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.layernorm(x, got, w, b, rows, C, 1e-6, split_scale=ACT_SCALE[piece], sat_flag=flag, split_layout=layout)
    assert torch.equal(got, want) and int(flag.item()) == 0
    hip.layernorm(x, got, w * 4000.0, b, rows, C, 1e-6, split_scale=ACT_SCALE[piece], sat_flag=flag, split_layout=layout)      # |y| up to ~1.6e4
    assert int(flag.item()) == (1 if piece == torch.float16 else 0) and bool(torch.isfinite(got.float()).all())


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("cfg", [1, 2])
@pytest.mark.parametrize("piece", HALVES)
@pytest.mark.parametrize("M,N,K,segs,gelu", [(1370, 3072, 1024, 2, False), (333, 1536, 384, 3, True), (300, 4096, 1024, 3, True),
                                              (40, 192, 64, 3, False), (700, 64, 128, 2, False)])
def test_gemm_writes_its_output_as_operand_pieces(M, N, K, segs, gelu, piece, cfg, layout):
    """VdaGemmArgs.split_out: the three-term product's result (with the exact GELU where asked) stored as the [hi | lo] or
    [hi | lo | hi] pieces of split_scale * v instead of an fp32 matrix: what vda_split3 makes of the fp32 output of the same
    product, up to the summation-order-free arithmetic both share (bit for bit without GELU; with it, both evaluate the same
    erf form on the same sums).  ``layout``: of the operands AND of the output (segs = 2 always writes the two planes that
    vda_attention_pairs reads); with PIECES_PAIRS the product runs in the kernels' three-products-per-K-step form."""
    if cfg == 2 and N < 192:
        pytest.skip("the 256x256 tile is not offered for narrow outputs")
    sc = ACT_SCALE[piece]
    pairs = layout == hip.PIECES_PAIRS
    a, w = rnd((M, K), 1, 1.0, torch.float32), rnd((N, K), 2, 1 / math.sqrt(K), torch.float32)
    bias = rnd((N,), 3, 0.1, torch.float32)
    Ks = hip.pieces_width(layout, K)
    a3 = torch.empty((M, Ks), dtype=piece, device=DEV)
    hip.split3(a, a3, M, K, K, K, scale=sc, layout=layout)
    sw = split_w(w, piece, layout)
    kw = dict(M=M, N=N, K=Ks, lda=Ks, bias=bias, tile_cfg=cfg, acc_scale=1.0 / (sw.scale * sc), ab_pairs=pairs)
    plain = torch.empty((M, N), dtype=torch.float32, device=DEV)
    hip.gemm(a3, sw.t, plain, ldc=N, **kw)
    form = 2 if segs == 2 else (4 if pairs else 3)
    out_layout = hip.PIECES_TRIPLE if segs == 2 else layout       # (the first two segments of a triple are the two planes)
    width = 2 * N if form != 3 else 3 * N
    want = torch.empty((M, hip.pieces_width(out_layout, N)), dtype=piece, device=DEV)
    hip.split3(plain, want, M, N, N, N, act=hip.SPLIT_GELU if gelu else hip.SPLIT_NONE, scale=sc, layout=out_layout)
    ld = width + 64                                         # a row stride beyond the pieces
    got = torch.full((M, ld), float("nan"), dtype=piece, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.gemm(a3, sw.t, got, ldc=ld, act=hip.ACT_GELU if gelu else hip.ACT_NONE, split_out=form, split_scale=sc, sat_flag=flag, **kw)
    g_hi, g_lo = hi_lo_of(got[:, :width], N, out_layout)
    w_hi, w_lo = hi_lo_of(want, N, out_layout)
    if gelu:        # the same erf form on the same sums, but compiled into two kernels: equal up to a fused multiply-add or two
        sum_got, sum_want = g_hi.float() + g_lo.float(), w_hi.float() + w_lo.float()
        assert (sum_got - sum_want).abs().max() <= SPLIT_TOL[piece] * sum_want.abs().max()
        if form == 3:
            assert torch.equal(got[:, :N], got[:, 2 * N:3 * N])
    else:
        assert torch.equal(got[:, :width], want[:, :width])
    assert bool(torch.isnan(got[:, width:].float()).all()) and int(flag.item()) == 0
    exact = a.double() @ w.double().t() + bias.double()
    close(plain, exact.float(), SPLIT_TOL[piece], "three-term product")
    exact = F.gelu(exact) if gelu else exact
    close((g_hi.float() + g_lo.float()) / sc, exact.float(), SPLIT_TOL[piece], "pieces of the product")
    if piece == torch.float16:      # a scaled value beyond the half range sets the flag, and only then
        hip.gemm(a3, sw.t, got, ldc=ld, split_out=form, split_scale=sc * 2.0 ** 16, sat_flag=flag, **kw)
        assert int(flag.item()) == 1 and bool(torch.isfinite(got[:, :width].float()).all())
    with pytest.raises(RuntimeError):            # no residual form
        hip.gemm(a3, sw.t, got, ldc=ld, split_out=form, split_scale=sc, resid=plain, ldr=N, **kw)


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("frames,n_tok,heads", [(2, 1370, 2), (1, 129, 3), (2, 36, 1), (1, 700, 16)])
def test_attention_on_operand_pieces(frames, n_tok, heads, layout):
    """vda_attention_pairs: [hi | lo] pieces of 16 * qkv in, [hi | lo | hi] pieces of 16 * output out, against fp64."""
    D, sc = heads * 64, 16.0
    qkv = rnd((frames, n_tok, 3 * D), 1, 1.0, torch.float32)
    qkv[..., :D] *= 0.35
    hi, lo = pieces_of(qkv * sc)
    qkv2 = torch.cat([hi, lo], dim=-1).contiguous()
    out3 = torch.full((frames * n_tok, hip.pieces_width(layout, D)), float("nan"), dtype=torch.float16, device=DEV)
    hip.attention_pairs(qkv2, out3, frames, n_tok, heads, sc, out_layout=layout)
    q, k, v = qkv.double().reshape(frames, n_tok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (softmax2(q @ k.transpose(-1, -2)) @ v).transpose(1, 2).reshape(frames, n_tok, D).float()
    hi, lo = hi_lo_of(out3, D, layout)
    got = ((hi.float() + lo.float()) / sc).reshape(frames, n_tok, D)
    close(got, want, 3e-6, "attention on pieces")
    assert bool(torch.isfinite(out3.float()).all())
    if layout == hip.PIECES_TRIPLE:
        assert torch.equal(out3[:, :D], out3[:, 2 * D:])
    old = torch.empty((frames, n_tok, D), dtype=torch.float32, device=DEV)
    hip.attention(qkv, old, frames, n_tok, heads)
    close(got, old, 2e-6, "against the fp32-in / fp32-out form")


def test_attention_on_operand_pieces_with_scores_far_from_zero():
    """Sharp and shifted softmax rows: the running maximum moves (the deferred rescale branch), in units of scale^2."""
    frames, n_tok, heads, sc = 1, 300, 2, 16.0
    D = heads * 64
    qkv = rnd((frames, n_tok, 3 * D), 5, 1.0, torch.float32)
    qkv[..., :D] *= 3.0
    qkv[:, :, D:2 * D] += 2.0 * torch.arange(n_tok, device=DEV, dtype=torch.float32)[None, :, None] / n_tok      # later keys score higher
    hi, lo = pieces_of(qkv * sc)
    out3 = torch.empty((frames, n_tok, 3 * D), dtype=torch.float16, device=DEV)
    hip.attention_pairs(torch.cat([hi, lo], dim=-1).contiguous(), out3, frames, n_tok, heads, sc)
    q, k, v = qkv.double().reshape(frames, n_tok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (softmax2(q @ k.transpose(-1, -2)) @ v).transpose(1, 2).reshape(frames, n_tok, D).float()
    close((out3[..., :D].float() + out3[..., D:2 * D].float()) / sc, want, 1e-5, "attention on pieces, sharp rows")


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("piece", HALVES)
@pytest.mark.parametrize("ih,iw,oh,ow,C", [(19, 19, 37, 37, 256), (10, 14, 35, 49, 64), (6, 6, 6, 6, 128)])
def test_bilinear_writes_operand_pieces(ih, iw, oh, ow, C, piece, layout):
    """vda_bilinear_nhwc_pieces: the pieces vda_split3 makes of vda_bilinear_nhwc's fp32 result, bit for bit."""
    frames, sc = 3, ACT_SCALE[piece]
    x = rnd((frames, ih, iw, C), 1, 2.0, torch.float32)
    y = torch.empty((frames, oh, ow, C), dtype=torch.float32, device=DEV)
    hip.bilinear_nhwc(x, y, frames, ih, iw, oh, ow, C)
    rows, width = frames * oh * ow, hip.pieces_width(layout, C)
    want = torch.empty((rows, width), dtype=piece, device=DEV)
    hip.split3(y.reshape(rows, C), want, rows, C, C, C, scale=sc, layout=layout)
    got = torch.full((rows, width), float("nan"), dtype=piece, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.bilinear_nhwc_pieces(x, got, frames, ih, iw, oh, ow, C, sc, layout, flag)
    assert torch.equal(got, want) and int(flag.item()) == 0
    hip.bilinear_nhwc_pieces(x * 1e4, got, frames, ih, iw, oh, ow, C, sc, layout, flag)
    assert int(flag.item()) == (1 if piece == torch.float16 else 0)


def test_split3_reports_saturation():
    x = rnd((64, 128), 1, 1.0, torch.float32)
    out = torch.empty((64, 3 * 128), dtype=torch.float16, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.split3(x, out, 64, 128, 128, 128, scale=16.0, sat_flag=flag)
    assert int(flag.item()) == 0
    x[17, 5] = -5000.0                                        # 16 * 5000 > 65504
    hip.split3(x, out, 64, 128, 128, 128, scale=16.0, sat_flag=flag)
    assert int(flag.item()) == 1 and float(out[17, 5]) == -65504.0
    flag.zero_()
    hip.split3(x, out, 64, 128, 128, 128, scale=1.0, sat_flag=flag)
    assert int(flag.item()) == 0
    bf = torch.empty((64, 3 * 128), dtype=BF, device=DEV)
    hip.split3(x * 1e6, bf, 64, 128, 128, 128, scale=1.0, sat_flag=flag)             # bf16 pieces have fp32's range
    assert int(flag.item()) == 0


@pytest.mark.parametrize("cfg", [1, 2])
def test_gemm_fp16_outputs_saturate_instead_of_overflowing(cfg):
    """A sum beyond the IEEE-half range (a real checkpoint's activation outlier) is stored as +-65504, not as infinity: the
    reference's own fp16 autocast would give inf here, and one inf turns a whole map into NaN two layers later."""
    M, N, K = 300, 256, 1024
    a = torch.full((M, K), 16.0, dtype=torch.float16, device=DEV)
    w = torch.full((N, K), 8.0, dtype=torch.float16, device=DEV)
    w[1::2] = -8.0                                             # sums of +-131072
    w[:, :8] = 0.0
    w[5] = 0.001                                               # one column in range
    out = torch.empty((M, N), dtype=torch.float16, device=DEV)
    hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, tile_cfg=cfg)
    assert bool(torch.isfinite(out).all())
    want = (a.float() @ w.float().t()).clamp(-65504.0, 65504.0)
    assert torch.equal(out[:, 0::2][:, :2], want[:, 0::2][:, :2].half()) and float(out.max()) == 65504.0 and float(out.min()) == -65504.0
    close(out[:, 5], want[:, 5], 2e-3, "in-range column")
    r = torch.zeros((M, N), dtype=torch.float32, device=DEV)      # the residual form's rounded copy saturates too
    y16 = torch.empty((M, N), dtype=torch.float16, device=DEV)
    part = torch.empty((M, N // 64, 2), dtype=torch.float32, device=DEV)
    hip.gemm(a, w, r, M=M, N=N, K=K, lda=K, ldc=N, resid=r, ldr=N, out16=y16, row_part=part, tile_cfg=cfg)
    assert bool(torch.isfinite(y16).all()) and float(y16.max()) == 65504.0 and float(r.max()) > 65504.0


def test_rejected_arguments_return_error_codes():
    a = torch.zeros((128, 96), dtype=BF, device=DEV)
    w = torch.zeros((128, 96), dtype=BF, device=DEV)
    out = torch.zeros((128, 128), dtype=BF, device=DEV)
    with pytest.raises(RuntimeError):
        hip.gemm(a, w, out, M=128, N=128, K=96, lda=96, ldc=128)        # K not a multiple of 64
    with pytest.raises(RuntimeError):
        hip.gemm(a.cpu(), w, out, M=128, N=128, K=64, lda=96, ldc=128)  # host pointer
    with pytest.raises(RuntimeError):
        hip.temporal_attention(a, out, 1, 33, 1, 8, 8, 1.0)              # T > 32
