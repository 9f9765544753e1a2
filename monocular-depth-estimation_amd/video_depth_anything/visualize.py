"""Depth video -> 8-bit visualisation on the device (reference: utils/dc_utils.py:72-82, the ``is_depths`` branch of
``save_video``): global min / max normalisation to 0..255 (float32 arithmetic, truncating cast) and the inferno colour
table, or plain grayscale.  The table in ``inferno_u8.npy`` is matplotlib's 256-entry inferno list times 255, truncated
to uint8, exactly as the reference indexes it (written by oracle/make_goldens.py).

Also the per-frame colour variants of the reference's other drivers, same names and arguments: ``apply_traffic_zones``
(run1.py:28-60), ``apply_histogram_equalization``, ``apply_contrast_stretching``, ``apply_distance_capping``
(run1dist.py:27-91), ``create_range_mask`` / ``enhance_range`` (run4.py:29-42).  Each takes a float32 ``[H, W]`` map as a numpy
array or a device tensor and returns what the reference returns (a numpy array), computed by one or two launches; scalar
parameters are evaluated with the very numpy expressions the reference uses, so their values and float32 / float64 types are
whatever the installed numpy gives the reference.  Percentiles are taken with numpy on a host copy of the map (selection, not
arithmetic).  ``colormaps_u8.npz`` holds matplotlib's tables for the colour maps the drivers name (259 rows: the 256 entries
times 255, truncated, then the under / over / bad colours)."""
import os

import numpy as np
import torch

from . import hip

_LUT = {}


def inferno_table(device):
    t = _LUT.get(device)
    if t is None:
        table = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "inferno_u8.npy"))
        assert table.shape == (256, 3) and table.dtype == np.uint8
        t = _LUT[device] = torch.from_numpy(table).to(device).contiguous()
    return t


def colorize_depth(depths, grayscale=False, chunk=64):
    """``depths``: float32 ``[N, H, W]`` numpy array or device tensor.  Returns uint8 ``[N, H, W, 3]`` (``[N, H, W]`` with
    ``grayscale``) as a numpy array; the minimum and maximum are taken over the whole video.  A numpy video is streamed
    through the device in chunks of frames, once per pass."""
    host = isinstance(depths, np.ndarray)
    dev = torch.device(f"cuda:{torch.cuda.current_device()}") if host else depths.device
    if dev.type != "cuda":
        raise RuntimeError("colorize_depth runs on the device: no CPU path")
    n, h, w = depths.shape

    def block(c0):
        b = depths[c0:c0 + chunk]
        return torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(dev) if host else b.float().contiguous()

    with torch.cuda.device(dev):
        lut = None if grayscale else inferno_table(dev)
        mm = torch.empty(hip.MINMAX_FLOATS, dtype=torch.float32, device=dev)
        lohi = torch.tensor([float("inf"), float("-inf")], device=dev)
        for c0 in range(0, n, chunk):                    # pass 1: the video's range
            b = block(c0)
            hip.minmax_f32(b, mm, b.numel())
            lohi = torch.stack([torch.minimum(lohi[0], mm[0]), torch.maximum(lohi[1], mm[1])])
        out = np.empty((n, h, w) if grayscale else (n, h, w, 3), dtype=np.uint8)
        for c0 in range(0, n, chunk):                    # pass 2: quantise and look up
            b = block(c0)
            o = torch.empty(out[c0:c0 + chunk].shape, dtype=torch.uint8, device=dev)
            hip.colorize_u8(b, lohi, lut, o, b.numel())
            out[c0:c0 + chunk] = o.cpu().numpy()
    return out


# ---- colour variants of run1.py / run1dist.py / run4.py ---------------------------------------------------------------------------
_CMAPS = {}


def colormap_table(name, device):
    """matplotlib's table for ``name`` as uint8 [259, 3] on the device (see the module docstring)."""
    key = (name, device)
    t = _CMAPS.get(key)
    if t is None:
        tables = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "colormaps_u8.npz"))
        if name not in tables.files:
            raise ValueError(f"colour map {name!r} is not shipped (have: {', '.join(sorted(tables.files))})")
        t = _CMAPS[key] = torch.from_numpy(tables[name]).to(device).contiguous()
        assert tuple(t.shape) == (hip.CMAP_ROWS, 3) and t.dtype == torch.uint8
    return t


def _on_device(depth):
    if isinstance(depth, np.ndarray):
        dev = torch.device(f"cuda:{torch.cuda.current_device()}")
        return torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)).to(dev), dev
    if depth.device.type != "cuda":
        raise RuntimeError("the colour variants run on the device: no CPU path")
    return depth.float().contiguous(), depth.device


def _host(depth):
    return depth if isinstance(depth, np.ndarray) else depth.detach().cpu().numpy()


def _is_f64(*values):
    """Whether numpy evaluates float32-map arithmetic with these scalars in float64 (a numpy.float64 scalar promotes the map;
    Python floats and numpy.float32 scalars do not)."""
    return (np.zeros(1, np.float32) + sum(values)).dtype == np.float64


def apply_traffic_zones(depth, critical_max=10, traffic_max=30):
    """run1.py:28-60: Reds up to ``critical_max``, viridis up to ``traffic_max``, dark grey beyond."""
    d, dev = _on_device(depth)
    with torch.cuda.device(dev):
        out = torch.empty(tuple(d.shape) + (3,), dtype=torch.uint8, device=dev)
        hip.colorize_zones_u8(d, colormap_table("Reds", dev), colormap_table("viridis", dev), out, d.numel(),
                              np.float32(critical_max), np.float32(traffic_max), np.float32(traffic_max - critical_max))
        return out.cpu().numpy()


def _colormapped(d, dev, colormap, sub, div, **kw):
    out = torch.empty(tuple(d.shape) + (3,), dtype=torch.uint8, device=dev)
    hip.colormap_u8(d, colormap_table(colormap, dev), out, d.numel(), sub, div, **kw)
    return out.cpu().numpy()


def apply_colormap(depth, vmin, vmax, colormap="turbo"):
    """run1dist.py:173-177: ``cmap((depth - vmin) / (vmax - vmin + 1e-8))`` as bytes."""
    d, dev = _on_device(depth)
    div = vmax - vmin + 1e-8
    with torch.cuda.device(dev):
        return _colormapped(d, dev, colormap, vmin, div, f64=_is_f64(vmin, div))


def apply_contrast_stretching(depth, percentile_low=2, percentile_high=98, colormap="turbo"):
    """run1dist.py:46-65: clip to the two percentiles of the map, normalise, colour."""
    flat = _host(depth).flatten()
    flat = flat[~np.isnan(flat)]
    lo, hi = np.percentile(flat, percentile_low), np.percentile(flat, percentile_high)
    d, dev = _on_device(depth)
    with torch.cuda.device(dev):
        return _colormapped(d, dev, colormap, lo, hi - lo, clip_lo=lo, clip_hi=hi, f64=_is_f64(lo, hi))


def apply_distance_capping(depth, vmin, vmax, cap, colormap="turbo"):
    """run1dist.py:67-91: colours vary between ``vmin`` and ``cap``; everything from ``cap`` on takes the last colour."""
    d, dev = _on_device(depth)
    div = cap - vmin + 1e-8
    with torch.cuda.device(dev):
        return _colormapped(d, dev, colormap, vmin, div, clip_lo=vmin, clip_hi=cap, cap=cap, f64=_is_f64(vmin, div))


def equalization_table(hist):
    """cv2.equalizeHist's byte table from a 256-bin histogram (OpenCV's rule restated; OpenCV is not importable here, so this
    step's parity is unpinned): the first occupied bin maps to 0, bin i to round(255 * (cdf(i) - h0) / (total - h0))."""
    hist = np.asarray(hist, dtype=np.int64)
    total = int(hist.sum())
    first = int(np.argmax(hist > 0))
    lut = np.zeros(256, np.uint8)
    if hist[first] == total:
        lut[:] = first
        return lut
    scale = np.float32(255.0) / np.float32(total - hist[first])
    acc = 0
    for i in range(first + 1, 256):
        acc += int(hist[i])
        lut[i] = min(max(int(np.rint(np.float32(acc) * scale)), 0), 255)
    return lut


def apply_histogram_equalization(depth, vmin, vmax, colormap="turbo"):
    """run1dist.py:27-44: quantise to bytes between ``vmin`` and ``vmax``, equalise the byte histogram, colour."""
    d, dev = _on_device(depth)
    n = d.numel()
    with torch.cuda.device(dev):
        g = torch.empty(n, dtype=torch.uint8, device=dev)
        hip.quantize_u8(d, g, n, np.float32(vmin), np.float32(vmax - vmin))
        hist = torch.zeros(256, dtype=torch.int32, device=dev)
        hip.histogram_u8(g, hist, n)
        remap = torch.from_numpy(equalization_table(hist.cpu().numpy())).to(dev)
        out = torch.empty(tuple(d.shape) + (3,), dtype=torch.uint8, device=dev)
        hip.colormap_bytes_u8(g, remap, colormap_table(colormap, dev), out, n)
        return out.cpu().numpy()


def focus_range(depth, min_range=0.0, max_range=50.0):
    """run4.py:105-111: ``depth * create_range_mask(depth, ...)`` followed by ``enhance_range`` on the product.  Returns the
    float32 map the reference stores back into its depth array."""
    d, dev = _on_device(depth)
    with torch.cuda.device(dev):
        masked = torch.empty_like(d)
        hip.range_focus_f32(d, masked, d.numel(), far=np.float32(max_range), far_weight=np.float32(0.1))
        host = masked.cpu().numpy()
        in_range = host[(host >= min_range) & (host <= max_range)]
        if in_range.size == 0:
            return host
        p2, p98 = np.percentile(in_range, [2, 98])
        div = p98 - p2 + 1e-8
        out = torch.empty_like(d)
        hip.range_focus_f32(masked, out, d.numel(), stretch=True, sub=p2, div=div, clip_lo=0.0, clip_hi=1.0, f64=_is_f64(p2, div))
        return out.cpu().numpy()
