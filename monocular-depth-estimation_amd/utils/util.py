"""Host-side (numpy) forms of the stitch arithmetic, kept importable under the reference's module path ``utils.util``
for drivers that call them directly (reference: utils/util.py:16-74).  ``infer_video_depth`` itself does this work on
the device (csrc/stitch.hip)."""
import numpy as np


def compute_scale_and_shift(prediction, target, mask, scale_only=False):
    """Least-squares ``target ~ scale * prediction + shift`` over ``mask`` in float32; (1, 0) for a singular system."""
    p = np.asarray(prediction, dtype=np.float32)
    t = np.asarray(target, dtype=np.float32)
    m = np.asarray(mask, dtype=np.float32)
    a00, a01, a11 = np.sum(m * p * p), np.sum(m * p), np.sum(m)
    b0 = np.sum(m * p * t)
    if scale_only:
        return b0 / (a00 + 1e-6), 0
    b1 = np.sum(m * t)
    det = a00 * a11 - a01 * a01
    if det == 0:
        return 1, 0
    return (a11 * b0 - a01 * b1) / det, (-a01 * b0 + a00 * b1) / det


def get_interpolate_frames(frame_list_pre, frame_list_post):
    """Cross-fade two equally long frame lists; the weight of the second rises linearly from 0 to 1."""
    n = len(frame_list_pre)
    assert n == len(frame_list_post)
    step = 1.0 / (n - 1)
    w = [0.0] + [i * step for i in range(1, n - 1)] + [1.0]
    return [a * (1 - wi) + b * wi for a, b, wi in zip(frame_list_pre, frame_list_post, w)]
