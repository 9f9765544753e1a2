"""The CPU oracle against the golden vectors captured from the unmodified reference (oracle/make_goldens.py)."""
import json
import os

import numpy as np
import pytest
import torch

import vda_oracle as O
from video_depth_anything import spec

TOL = 2e-5          # fp32 restatement vs fp32 reference, relative to the tensor's max magnitude


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def sd():
    return {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**spec.MODEL_CONFIGS["vits"], seed=1).items()}


def frames_for(n, h, w, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        r = (xx * 255.0 / max(w - 1, 1) + 3 * i) % 256
        g = (yy * 255.0 / max(h - 1, 1) + 5 * i) % 256
        b = 255 - r
        f = np.stack([r, g, b], -1) + rs.randint(-20, 21, (h, w, 3))
        out[i] = np.clip(f, 0, 255).astype(np.uint8)
    return out


@pytest.mark.parametrize("enc", ["vits", "vitl"])
def test_state_dict_inventory_matches_reference(golden_dir, enc):
    want = json.load(open(os.path.join(golden_dir, f"keys_{enc}.json")))
    got = spec.state_dict_spec(**spec.MODEL_CONFIGS[enc])
    assert list(got.keys()) == list(want.keys())
    for k, (shape, _) in got.items():
        assert list(shape) == want[k], k
    assert len(got) == {"vits": 351, "vitl": 519}[enc]


def test_encoder_taps_and_pos_tables(golden_dir, sd):
    g = np.load(os.path.join(golden_dir, "g1_encoder.npz"))
    x = torch.from_numpy(np.random.RandomState(11).standard_normal((2, 3, 70, 98)).astype(np.float32))
    taps = O.encoder_taps(sd, x, "vits")
    for i in range(4):
        assert rel(taps[i].numpy(), g[f"tap{i}"]) < TOL
    for hp, wp in ((5, 7), (37, 66), (74, 74), (37, 37)):
        pe = O.pos_embed_for(sd["pretrained.pos_embed"], hp, wp).numpy()
        assert rel(pe[0, :, :8], g[f"pos_{hp}x{wp}"]) < 1e-6


def test_temporal_modules(golden_dir, sd):
    g = np.load(os.path.join(golden_dir, "g2_temporal.npz"))
    for idx, C in ((0, 192), (2, 64), (1, 384)):
        for T in (32, 5):
            rs = np.random.RandomState(100 + idx * 10 + T)
            inp = torch.from_numpy(rs.standard_normal((T, C, 3, 4)).astype(np.float32))
            assert rel(O.temporal_module(sd, idx, inp, T).numpy(), g[f"mm{idx}_T{T}"]) < TOL
    inp = torch.from_numpy(np.random.RandomState(199).standard_normal((8, 64, 3, 4)).astype(np.float32))
    assert rel(O.temporal_module(sd, 3, inp, 4).numpy(), g["mm3_B2_T4"]) < TOL


def test_head_blocks(golden_dir, sd):
    g = np.load(os.path.join(golden_dir, "g3_head_blocks.npz"))
    rs = np.random.RandomState(31)
    a = torch.from_numpy(rs.standard_normal((2, 64, 5, 7)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal((2, 64, 5, 7)).astype(np.float32))
    assert rel(O.fusion_block(sd, 3, a, b, size=(10, 14)).numpy(), g["fusion3"]) < TOL
    assert rel(O.fusion_block(sd, 4, a, None, size=(9, 13)).numpy(), g["fusion4"]) < TOL
    assert rel(O.fusion_block(sd, 1, a, b, size=None).numpy(), g["fusion1"]) < TOL


@pytest.mark.parametrize("name,shape,seed", [("T32", (1, 32, 3, 70, 98), 41), ("T4", (1, 4, 3, 70, 98), 42),
                                             ("B2T3", (2, 3, 3, 56, 42), 43), ("T32_sq", (1, 32, 3, 42, 42), 44)])
def test_full_forward(golden_dir, sd, name, shape, seed):
    g = np.load(os.path.join(golden_dir, "g4_forward.npz"))
    x = torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))
    out = O.forward(sd, x, "vits").numpy()
    assert out.shape == g[name].shape
    assert rel(out, g[name]) < TOL


@pytest.mark.parametrize("n,bs", [(1, 32), (31, 32), (32, 32), (33, 32), (33, 16), (60, 32), (60, 16), (77, 32)])
def test_whole_video(golden_dir, sd, n, bs):
    g = np.load(os.path.join(golden_dir, "g5_video.npz"))
    frames = frames_for(n, 28, 42, 500 + n)
    out = O.infer_video_depth(sd, list(frames), "vits", batch_size=bs)
    assert out.shape == (n, 28, 42) and out.dtype == np.float32
    assert rel(out, g[f"N{n}_bs{bs}"]) < TOL


def test_host_tables(golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_host.npz"))
    for h0, w0, s, h, w in g["sizes"]:
        assert O.network_size(int(h0), int(w0), int(s)) == (int(h), int(w))
    s, t = O.scale_shift(np.concatenate(list(g["ls_pred"])), np.concatenate(list(g["ls_target"])))
    assert abs(s - g["ls_out"][0]) < 1e-6 and abs(t - g["ls_out"][1]) < 1e-6
    out = np.stack(O.blend(list(g["blend_pre"]), list(g["blend_post"])))
    assert rel(out, g["blend_out"]) < 1e-7


def test_vitl_width_blocks(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_vitl_width.npz"))
    big = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(
        **spec.MODEL_CONFIGS["vitl"], seed=2, only=["pretrained.blocks.3.", "head.scratch.refinenet2."]).items()}
    tok = torch.from_numpy(np.random.RandomState(71).standard_normal((2, 40, 1024)).astype(np.float32))
    assert rel(O.encoder_block(big, 3, tok, 16).numpy(), g["block"]) < TOL
    rs = np.random.RandomState(72)
    a = torch.from_numpy(rs.standard_normal((1, 256, 6, 5)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal((1, 256, 6, 5)).astype(np.float32))
    assert rel(O.fusion_block(big, 2, a, b, size=(12, 10)).numpy(), g["fusion"]) < TOL


def test_cubic_resize_restatement_properties():
    """The OpenCV-style bicubic restatement has no fixture to be pinned by; check what the arithmetic guarantees: identity
    at equal size, constants preserved (the four weights sum to one), and exact 2x decimation phase (taps at +-0.5)."""
    rs = np.random.RandomState(3)
    img = rs.rand(12, 17, 3).astype(np.float32)
    assert np.array_equal(O.cv_resize_cubic(img, 12, 17), img)
    const = np.full((9, 11, 3), 0.37, np.float32)
    assert np.abs(O.cv_resize_cubic(const, 20, 7) - 0.37).max() < 1e-6
    ramp = np.tile(np.arange(16, dtype=np.float32)[None, :, None], (4, 1, 1))
    half = O.cv_resize_cubic(ramp, 4, 8)                  # linear ramps are reproduced exactly away from the border
    assert np.abs(half[:, 1:-1, 0] - (np.arange(8)[1:-1] * 2 + 0.5)).max() < 1e-5
    frame = rs.randint(0, 256, (10, 14, 3), dtype=np.uint8)
    assert np.array_equal(O.preprocess_frame(frame, 10, 14), O.normalize_frame(frame))


def test_clamped_profile_forward_and_video(golden_dir):
    """Depth maps with exact zeros (mixed-sign last 1x1 convolution): forward goldens G4b and whole-video goldens G5b."""
    sdc = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**spec.MODEL_CONFIGS["vits"], seed=1, profile="clamped").items()}
    g = np.load(os.path.join(golden_dir, "g4b_forward_clamped.npz"))
    for name, shape, seed in (("T4", (1, 4, 3, 70, 98), 42), ("B2T3", (2, 3, 3, 56, 42), 43)):
        x = torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))
        out = O.forward(sdc, x, "vits").numpy()
        assert rel(out, g[name]) < TOL and np.mean(g[name] == 0) >= 0.2
        assert np.mean((out == 0) != (g[name] == 0)) < 1e-3
    g = np.load(os.path.join(golden_dir, "g5b_video_clamped.npz"))
    for n, bs in ((33, 32), (33, 16)):
        frames = np.random.RandomState(700 + n).randint(0, 256, (n, 28, 42, 3)).astype(np.uint8)
        out = O.infer_video_depth(sdc, list(frames), "vits", batch_size=bs)
        assert rel(out, g[f"N{n}_bs{bs}"]) < TOL and np.mean(out == 0) >= 0.1


def lidar_cases():
    rs = np.random.RandomState(81)
    n, h, w = 40, 12, 16
    depths = (rs.rand(n, h, w).astype(np.float32) * 20 + 0.5)
    depths[7, 5, 9] = 0.0
    return {"global": (depths, {"global": (None, None, 12.5)}),
            "sparse": (depths, {3: (9, 5, 8.0), 7: (9, 5, 30.0), 20: (100, 100, 15.0), 33: (0, 11, 4.25)}),
            "single": (depths, {12: (4, 4, 7.0)}), "none": (depths, {}),
            "mixed": (depths, {"global": (2, 3, 10.0), 5: (15, 11, 40.0), 6: (1, 1, 2.0)})}


def test_lidar_rescaling_oracle_and_host_logic(golden_dir):
    """Metric post-processing (metric_depth/run2.py:42-67): the oracle against the reference's outputs, and the package's own
    host arithmetic (reference pixels, scale factors, temporal propagation without scipy) against the oracle."""
    from video_depth_anything import metric_post as MP
    g = np.load(os.path.join(golden_dir, "g8_lidar.npz"))
    for name, (depths, refs) in lidar_cases().items():
        scaled, factors = O.apply_lidar_scaling(list(depths), refs)
        assert rel(np.stack(scaled), g[name]) < 1e-7
        assert np.array_equal(np.asarray(factors), g[name + "_factors"])
        n, h, w = depths.shape
        pix = MP.reference_pixels(refs, n, h, w)
        pred = np.array([depths[i, r[1], r[0]] if r else 0.0 for i, r in enumerate(pix)], dtype=np.float64)
        mine = MP._scale_factors(pred, pix)
        assert np.abs(mine - g[name + "_factors"]).max() <= 1e-12 * np.abs(g[name + "_factors"]).max()


def test_depth_visualisation_oracle_matches_reference_frames(golden_dir):
    """utils/dc_utils.py:72-82: what the reference hands to its video writer, bit for bit; the package's inferno byte table."""
    g = np.load(os.path.join(golden_dir, "g9_visualisation.npz"))
    table = np.load(os.path.join(os.path.dirname(spec.__file__), "inferno_u8.npy"))
    assert np.array_equal(O.colorize_depth(g["depths"], table), g["inferno"])
    assert np.array_equal(O.colorize_depth(g["depths"], None), g["gray"])


def colour_tables():
    return np.load(os.path.join(os.path.dirname(spec.__file__), "colormaps_u8.npz"))


def test_colour_variant_oracle_matches_reference_frames(golden_dir):
    """The restatements of run1.py:28-60, run1dist.py:46-91,173-177 and run4.py:29-42 against the bytes / maps the
    reference's own functions returned (G10), through the shipped colour tables."""
    g = np.load(os.path.join(golden_dir, "g10_colour_variants.npz"))
    t = colour_tables()
    assert all(t[k].shape == (259, 3) and t[k].dtype == np.uint8 for k in t.files)
    for name in ("wide", "near"):
        d = g[name]
        assert np.array_equal(O.apply_traffic_zones(d, t["Reds"], t["viridis"], 10.0, 30.0), g[name + "_zones"])
        assert np.array_equal(O.apply_traffic_zones(d, t["Reds"], t["viridis"]), g[name + "_zones_default"])
        vmin, vmax = np.nanmin(d), np.nanmax(d)
        for cmap in ("turbo", "plasma"):
            assert np.array_equal(O.apply_contrast_stretching(d, t[cmap]), g[f"{name}_stretch_{cmap}"])
            assert np.array_equal(O.apply_distance_capping(d, vmin, vmax, 25.0, t[cmap]), g[f"{name}_cap_{cmap}"])
            assert np.array_equal(O.apply_colormap(d, vmin, vmax, t[cmap]), g[f"{name}_plain_{cmap}"])
        assert np.array_equal(O.focus_range(np.nan_to_num(d, nan=3.0), 0.0, 50.0), g[name + "_focus"])
    assert (g["wide_zones"][3, 4] != 0).any() and tuple(g["wide_zones"][5, 9]) == (0, 0, 0)      # boundary pixel coloured, NaN black


def test_equalisation_and_warp_restatements_have_the_documented_properties():
    """cv2.equalizeHist and cv2.remap cannot be imported here: their restatements are checked against what the algorithms
    guarantee (parity with OpenCV itself is unpinned)."""
    rs = np.random.RandomState(5)
    g = (rs.rand(40, 50) ** 3 * 200).astype(np.uint8)
    e = O.equalize_hist(g)
    assert e.min() == 0 and e.max() == 255
    order = np.argsort(g.reshape(-1), kind="stable")
    assert (np.diff(e.reshape(-1)[order].astype(int)) >= 0).all()              # monotone in the input value
    assert np.array_equal(O.equalize_hist(np.full((4, 4), 7, np.uint8)), np.full((4, 4), 7, np.uint8))
    prev = rs.rand(12, 17).astype(np.float32) * 30
    zero = np.zeros((12, 17, 2), np.float32)
    assert np.array_equal(O.warp_depth(prev, zero), prev)
    shift = zero.copy()
    shift[..., 0] = 1.0                                                        # sample one pixel to the right
    w = O.warp_depth(prev, shift)
    assert np.array_equal(w[:, :-1], prev[:, 1:]) and np.array_equal(w[:, -1], prev[:, -1])      # mirrored border repeats the edge
    half = zero.copy()
    half[..., 1] = 0.5
    w = O.warp_depth(prev, half)
    assert np.allclose(w[:-1], 0.5 * (prev[:-1] + prev[1:]), rtol=1e-6)
    scaled = [prev, prev * 1.1, prev * 0.9]
    ref = O.temporal_refine(scaled, [zero, zero])
    assert np.allclose(ref[2], 0.7 * scaled[2] + 0.3 * (0.7 * scaled[1] + 0.3 * scaled[0]), rtol=1e-6)
