"""Depth video -> 8-bit visualisation on the device (reference: utils/dc_utils.py:72-82, the ``is_depths`` branch of
``save_video``): global min / max normalisation to 0..255 (float32 arithmetic, truncating cast) and the inferno colour
table, or plain grayscale.  The table in ``inferno_u8.npy`` is matplotlib's 256-entry inferno list times 255, truncated
to uint8, exactly as the reference indexes it (written by oracle/make_goldens.py)."""
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
