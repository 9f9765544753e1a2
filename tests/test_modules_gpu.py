"""The stages of the forward pass one by one on the GPU against the reference's module-level goldens (the same fixtures that
pin the CPU oracle): encoder taps (G1), the four temporal modules (G2), RefineNet fusion blocks (G3).  The whole-forward
tests say THAT something is off; these say WHERE.  Both storage modes; bounds are relative to the largest golden value of
the tensor (16-bit mode: rounding of bf16 / fp16 operands and stored activations through the stage; fp32 mode: the
three-term half products)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from video_depth_anything import hip, spec  # noqa: E402
from video_depth_anything.video_depth import VideoDepthAnything  # noqa: E402

# measured: encoder taps 2.6-2.9e-3 (bf16), head stages 4.5-8.4e-4 (fp16), fp32 mode 3.5e-7 .. 1.1e-6
HALF_ENCODER, HALF_HEAD, FP32_MODE = 8e-3, 2.5e-3, 5e-6


@pytest.fixture(scope="module")
def engines():
    sd = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**spec.MODEL_CONFIGS["vits"], seed=1).items()}
    m = VideoDepthAnything(**spec.MODEL_CONFIGS["vits"])
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    out = {}
    for mode in ("half", "fp32"):
        m.precision = mode
        out[mode] = m._ensure_engine()
    return out


def rel(got, want):
    return float(np.abs(got - want).max() / np.abs(want).max())


def nhwc(eng, a):
    """NCHW fp32 numpy -> the head's channels-last storage on the device (channel counts here are multiples of 64)."""
    return torch.from_numpy(a).cuda().permute(0, 2, 3, 1).contiguous().to(eng.hdt)


def nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu().numpy()


def values_of_pieces(pieces, K, layout, scale):
    """The fp32 values a matrix of hi/lo operand pieces stands for: (hi + lo) / scale."""
    if layout == hip.PIECES_PAIRS:
        b = pieces.reshape(pieces.shape[0], K // 32, 2, 32)
        hi, lo = b[:, :, 0].reshape(-1, K), b[:, :, 1].reshape(-1, K)
    else:
        hi, lo = pieces[:, :K], pieces[:, K:2 * K]
    return (hi.double() + lo.double()) / scale


@pytest.mark.parametrize("mode", ["half", "fp32"])
def test_encoder_taps_against_reference_goldens(golden_dir, engines, mode):
    """dinov2.py:297-321 with the taps of video_depth.py:48-51: four normalised token maps without the class token."""
    eng = engines[mode]
    g = np.load(os.path.join(golden_dir, "g1_encoder.npz"))
    x = torch.from_numpy(np.random.RandomState(11).standard_normal((2, 3, 70, 98)).astype(np.float32)).cuda()
    with torch.cuda.device(0):
        taps = eng.encoder(x, 2, 70, 98)
    for i, t in enumerate(taps):
        if eng.taps_split:                                # fp32 mode: the tap norms write the operand pieces of the head's first product
            t = values_of_pieces(t, 384, eng.layout, eng.act_scale)
        err = rel(t.float().reshape(2, 35, 384).cpu().numpy(), g[f"tap{i}"])
        print(f"{mode} tap{i}: {err:.3e}")
        assert err <= (HALF_ENCODER if mode == "half" else FP32_MODE), i


@pytest.mark.parametrize("mode", ["half", "fp32"])
@pytest.mark.parametrize("idx,C,B,T,seed,key", [(0, 192, 1, 32, 132, "mm0_T32"), (0, 192, 1, 5, 105, "mm0_T5"), (2, 64, 1, 32, 152, "mm2_T32"),
                                                (2, 64, 1, 5, 125, "mm2_T5"), (1, 384, 1, 32, 142, "mm1_T32"), (1, 384, 1, 5, 115, "mm1_T5"),
                                                (3, 64, 2, 4, 199, "mm3_B2_T4")])
def test_temporal_modules_against_reference_goldens(golden_dir, engines, mode, idx, C, B, T, seed, key):
    """motion_module.py:102-126 (GroupNorm, proj_in, two temporal self-attentions with the position table, GEGLU feed-forward,
    proj_out, residual) for each of the four modules' widths, full and short windows, and a batch of two."""
    eng = engines[mode]
    g = np.load(os.path.join(golden_dir, "g2_temporal.npz"))
    inp = np.random.RandomState(seed).standard_normal((B * T, C, 3, 4)).astype(np.float32)
    with torch.cuda.device(0):
        out = eng.motion(idx, nhwc(eng, inp), B, T)
    err = rel(nchw(out), g[key])
    print(f"{mode} {key}: {err:.3e}")
    assert err <= (HALF_HEAD if mode == "half" else FP32_MODE)


@pytest.mark.parametrize("mode", ["half", "fp32"])
def test_fusion_blocks_against_reference_goldens(golden_dir, engines, mode):
    """util/blocks.py:135-162: two inputs with a given output size, one input (refinenet4), and the x2 default."""
    eng = engines[mode]
    g = np.load(os.path.join(golden_dir, "g3_head_blocks.npz"))
    rs = np.random.RandomState(31)
    a = rs.standard_normal((2, 64, 5, 7)).astype(np.float32)
    b = rs.standard_normal((2, 64, 5, 7)).astype(np.float32)
    with torch.cuda.device(0):
        cases = (("fusion3", eng.fusion(3, nhwc(eng, a), nhwc(eng, b), 10, 14)),
                 ("fusion4", eng.fusion(4, nhwc(eng, a), None, 9, 13)),
                 ("fusion1", eng.fusion(1, nhwc(eng, a), nhwc(eng, b), 10, 14)))
    for key, out in cases:
        err = rel(nchw(out), g[key])
        print(f"{mode} {key}: {err:.3e}")
        assert err <= (HALF_HEAD if mode == "half" else FP32_MODE), key
