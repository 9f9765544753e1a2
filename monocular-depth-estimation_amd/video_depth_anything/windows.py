"""Host-side window logic of ``infer_video_depth``: network size rule, window composition with key-frame
carry-over, rank partition of windows, and the affine stitch driven on the device.

Reference: video_depth_anything/video_depth.py:30-33 (constants), 68-72 (aspect rule), 97-103 (padding), 112-132
(window assembly), 166-209 (stitch); util/transform.py:51-107 (size rule); utils/util.py:40-74 (least squares, blend).
"""
import math

import torch

from . import hip

# infer settings of the reference (video_depth.py:29-33)
INFER_LEN = 32
OVERLAP = 10
KEYFRAMES = [0, 12, 24, 25, 26, 27, 28, 29, 30, 31]
INTERP_LEN = 8


def network_size(frame_h, frame_w, input_size=518, clamp_420=False):
    """(H, W) of the network input: keep aspect, shorter side >= input_size, both multiples of 14; for aspect ratios
    above 1.78 the target is first reduced (video_depth.py:68-72).  ``clamp_420`` reproduces video_depth.py:75-78."""
    ratio = max(frame_h, frame_w) / min(frame_h, frame_w)
    if ratio > 1.78:
        input_size = int(input_size * 1.777 / ratio)
        input_size = round(input_size / 14) * 14
    if clamp_420:
        input_size = min(input_size, 420)
        input_size = round(input_size / 14) * 14
    scale = max(input_size / frame_w, input_size / frame_h)

    def snap(v):
        y = int(round_half_even(v / 14) * 14)
        if y < input_size:
            y = int(math.ceil(v / 14) * 14)
        return y

    return snap(scale * frame_h), snap(scale * frame_w)


def round_half_even(x):
    """numpy.round semantics (round half to even), which the reference's size rule uses (transform.py:52)."""
    return round(x)


def window_plan(n_frames, batch_size):
    """Composition of every window.  Returns (plan, L, O) where plan[w] is a list of L slots, each either
    ("new", video frame index) or ("prev", slot of the previous window's input)."""
    L = min(INFER_LEN, batch_size)
    O = min(OVERLAP, L // 2)
    step = L - O
    kf = [k for k in KEYFRAMES if k < L][:O]
    plan = []
    for w, f0 in enumerate(range(0, n_frames, step)):
        slots = [("new", min(f0 + i, n_frames - 1)) for i in range(L)]     # the tail is padded with the last frame
        if w > 0:
            for i, k in enumerate(kf):
                slots[i] = ("prev", k)
        plan.append(slots)
    return plan, L, O


def resolve_plan(plan):
    """Replace every ("prev", k) by the video frame it ultimately refers to, so a window can be built without the
    previous window (needed when windows are sharded over ranks).  Returns a list of lists of frame indices."""
    out = []
    for w, slots in enumerate(plan):
        cur = []
        for kind, k in slots:
            cur.append(k if kind == "new" else out[w - 1][k])
        out.append(cur)
    return out


def partition(n_windows, world, rank):
    """Contiguous block of windows [lo, hi) for ``rank``: ceil(n/world) windows per rank."""
    per = -(-n_windows // world)
    lo = min(rank * per, n_windows)
    return lo, min(lo + per, n_windows)


class Stitcher:
    """Sequential affine alignment + cross-fade of full windows, all arithmetic on the device.

    ``push(depth)`` takes one window's 32 raw maps ``[32, H0, W0]`` fp32 (in window order) and returns the list of maps that
    became final.  Mirrors video_depth.py:169-205: window 0 passes through and provides the two reference key frames
    (slots 0 and 12); a later window k is fitted ``s*d + t`` on its slots 0, 1 against the references, its slots 2..9 are
    cross-faded onto the previous 8 aligned maps, slots 10..31 are appended, and the second reference becomes its
    aligned slot 12.
    """

    def __init__(self, device):
        self.dev = device
        self.tail = None          # last 8 aligned maps (still open for blending)
        self.ref0 = None
        self.ref1 = None
        self.sums = torch.zeros(hip.LSQ_SUMS_DOUBLES, dtype=torch.float64, device=device)   # 5 totals + partial-sum scratch
        self.ss = torch.zeros(2, dtype=torch.float32, device=device)
        self.started = False

    def _affine(self, x, pre=None, w_pre=0.0, w_post=1.0):
        out = torch.empty_like(x)
        hip.affine_blend(x, pre, out, self.ss, float(w_pre), float(w_post), x.numel())
        return out

    def fit(self, d0, d1):
        """Least-squares (scale, shift) of [d0, d1] onto [ref0, ref1]; stays on the device in ``self.ss``."""
        self.sums.zero_()
        hip.scale_shift_sums(d0, self.ref0, self.sums, d0.numel())
        hip.scale_shift_sums(d1, self.ref1, self.sums, d1.numel())
        hip.solve_scale_shift(self.sums, self.ss)

    def push(self, depth):
        assert depth.shape[0] == INFER_LEN
        final = []
        if not self.started:
            self.started = True
            self.ref0, self.ref1 = depth[0].clone(), depth[12].clone()
            final = [depth[i] for i in range(INFER_LEN - INTERP_LEN)]
            self.tail = [depth[i].clone() for i in range(INFER_LEN - INTERP_LEN, INFER_LEN)]
            return final
        self.fit(depth[0].contiguous(), depth[1].contiguous())
        n = INTERP_LEN
        for i in range(n):       # weights of utils/util.py:67-73: 0, 1/7, ..., 6/7, 1 on the new window's side
            w_post = 0.0 if i == 0 else (1.0 if i == n - 1 else i * (1.0 / (n - 1)))
            final.append(self._affine(depth[2 + i].contiguous(), self.tail[i], 1.0 - w_post, w_post))
        for i in range(OVERLAP, INFER_LEN - INTERP_LEN):
            final.append(self._affine(depth[i].contiguous()))
        self.tail = [self._affine(depth[i].contiguous()) for i in range(INFER_LEN - INTERP_LEN, INFER_LEN)]
        self.ref1 = self._affine(depth[12].contiguous())
        return final

    def flush(self):
        t, self.tail = self.tail or [], None
        return t
