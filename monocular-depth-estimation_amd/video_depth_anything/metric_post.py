"""Metric post-processing on the device: LiDAR reference rescaling of a depth video.

Reference: metric_depth/run2.py:42-67 (``apply_scaling_and_refinement`` without ``temporal_refine``; the same code inline in
metric_depth/run.py:233-264 and run3.py).  Per frame with a reference ``(x, y, range)`` the scale is range / predicted
depth at that pixel (skipped where the prediction is not positive); frames without one take the linear interpolation or
extrapolation over the frame index of the factors that exist (a single factor applies to every frame); every frame is
multiplied by its factor.  The per-frame samples and the multiply run as two launches over device-resident maps
(``infer_video_depth_device``); the N scalars in between are host arithmetic in float64, as in the reference.

The optional temporal refinement of the reference (run2.py:22-40, 68-77) is Farneback optical flow, a remap of the previous
refined map along the flow and a 0.7 / 0.3 blend.  ``temporal_refine`` does the remap and the blend on the device from flow
fields the caller supplies; the flow estimator itself is an OpenCV routine and is not part of this library, and with OpenCV
unavailable the remap's parity with ``cv2.remap`` is unpinned (the blend is exact).
"""
import numpy as np
import torch

from . import hip


def reference_pixels(lidar_refs, n, h, w):
    """Per frame ``(x, y, range)`` after the reference's defaults and clamping (run2.py:45-53), or None."""
    out = []
    for i in range(n):
        ref = lidar_refs.get(i, lidar_refs.get("global", None))
        if not ref:
            out.append(None)
            continue
        x, y, rng = ref
        if x is None or y is None:
            x, y = w // 2, h // 2
        x, y = min(int(x), w - 1), min(int(y), h - 1)
        out.append((x % w, y % h, float(rng)))          # a negative index wraps, as numpy indexing does
    return out


def propagate_scale_factors(scale_factors):
    """Fill the frames whose factor is still exactly 1.0 (run2.py:56-66; scipy's linear interp1d with
    fill_value="extrapolate" restated: slope of the bracketing pair, of the first / last pair outside the range)."""
    s = np.asarray(scale_factors, dtype=np.float64).copy()
    valid = np.where(s != 1.0)[0]
    if len(valid) >= 2:
        xs, ys = valid.astype(np.float64), s[valid]
        for i in np.where(s == 1.0)[0]:
            hi = int(np.clip(np.searchsorted(xs, i), 1, len(xs) - 1))
            lo = hi - 1
            slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo])
            s[i] = slope * (i - xs[lo]) + ys[lo]
    elif len(valid) == 1:
        s[:] = s[valid[0]]
    return s


def apply_lidar_scaling(depths, lidar_refs):
    """``depths``: fp32 ``[N, H, W]`` device tensor (scaled in place and returned) or numpy array (a scaled copy is returned,
    staged through the device in chunks).  ``lidar_refs``: ``{frame index | "global": (x, y, range)}`` as the reference's
    drivers build it from ``--lidar_ref`` (run2.py:142-151).  Returns ``(scaled depths, float64 scale factors [N])``."""
    if isinstance(depths, np.ndarray):
        dev = torch.device(f"cuda:{torch.cuda.current_device()}")
        out = np.empty(depths.shape, dtype=np.float32)
        n, h, w = depths.shape
        refs = reference_pixels(lidar_refs, n, h, w)
        pred = np.array([depths[i, r[1], r[0]] if r else 0.0 for i, r in enumerate(refs)], dtype=np.float64)
        scales = _scale_factors(pred, refs)
        sc = torch.from_numpy(scales.astype(np.float32)).to(dev)
        for c0 in range(0, n, 64):
            blk = torch.from_numpy(np.ascontiguousarray(depths[c0:c0 + 64], dtype=np.float32)).to(dev)
            hip.scale_frames(blk, sc[c0:c0 + 64], blk, blk.shape[0], h * w)
            out[c0:c0 + 64] = blk.cpu().numpy()
        return out, scales
    if not depths.is_cuda or depths.dtype != torch.float32 or depths.dim() != 3 or not depths.is_contiguous():
        raise ValueError("expected a contiguous fp32 [N, H, W] device tensor or a numpy array")
    n, h, w = depths.shape
    refs = reference_pixels(lidar_refs, n, h, w)
    with torch.cuda.device(depths.device):
        xy = torch.tensor([[r[0], r[1]] if r else [0, 0] for r in refs], dtype=torch.int32).to(depths.device)
        pred = torch.empty(n, dtype=torch.float32, device=depths.device)
        hip.sample_points(depths, xy, pred, n, h, w)
        scales = _scale_factors(pred.cpu().numpy().astype(np.float64), refs)
        hip.scale_frames(depths, torch.from_numpy(scales.astype(np.float32)).to(depths.device), depths, n, h * w)
    return depths, scales


def _scale_factors(pred, refs):
    s = np.ones(len(refs), dtype=np.float64)
    for i, r in enumerate(refs):
        if r and pred[i] > 0:
            # python float / numpy float32 scalar: the reference's division runs in float32 (numpy >= 2 promotion rules)
            s[i] = float(np.float32(r[2]) / np.float32(pred[i]))
    return propagate_scale_factors(s)


def temporal_refine(scaled_depths, flows, prev_depth=None, w_cur=0.7, w_prev=0.3):
    """The refinement chain of run2.py:68-77 on device-resident maps: ``refined[0] = scaled[0]`` and
    ``refined[i] = w_cur * scaled[i] + w_prev * warp(refined[i - 1], flows[i - 1])``, where ``flows[i - 1]`` (fp32 ``[H, W, 2]``)
    is the optical flow from frame i - 1 to frame i and ``warp`` samples like ``cv2.remap(INTER_LINEAR, BORDER_REFLECT)``.
    With ``prev_depth`` (the last refined map of the previous chunk) frame 1 blends with it unwarped, as the reference does.
    ``scaled_depths``: fp32 ``[N, H, W]`` device tensor or numpy array; ``flows``: ``[N - 1, H, W, 2]`` likewise (``flows[0]`` is
    not read when ``prev_depth`` is given).  Returns the refined maps in the type of ``scaled_depths``."""
    host = isinstance(scaled_depths, np.ndarray)
    dev = torch.device(f"cuda:{torch.cuda.current_device()}") if host else scaled_depths.device

    def on_dev(t):
        return torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(dev) if isinstance(t, np.ndarray) else t.float().contiguous()

    with torch.cuda.device(dev):
        scaled = on_dev(scaled_depths)
        n, h, w = scaled.shape
        refined = torch.empty_like(scaled)
        refined[0] = scaled[0]
        zero_flow = None
        for i in range(1, n):
            if i == 1 and prev_depth is not None:
                prev = on_dev(prev_depth)
                if zero_flow is None:
                    zero_flow = torch.zeros((h, w, 2), dtype=torch.float32, device=dev)
                flow = zero_flow                          # the reference blends the previous chunk's map as it is
            else:
                prev, flow = refined[i - 1], on_dev(flows[i - 1])
            hip.warp_blend_f32(prev, scaled[i], flow, refined[i], h, w, w_cur, w_prev)
        return refined.cpu().numpy() if host else refined
