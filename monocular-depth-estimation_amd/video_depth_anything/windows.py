"""Host-side window logic of ``infer_video_depth``: network size rule, window composition with key-frame
carry-over, rank partition of windows, and the affine alignment driven on the device.

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


BLOCK_MAX = 8      # consecutive windows a rank takes per round (see schedule)


def schedule(n_windows, world, block_max=BLOCK_MAX):
    """Which rank runs which window, round by round: ``rounds[r][rank]`` is a (possibly empty) run of consecutive windows.

    A rank takes up to ``block_max`` CONSECUTIVE windows per round, because consecutive windows share ten of their 32
    frames and the second of them can take those frames' encoder outputs from the first.  The block of a round is no
    longer than an even split of what is left, so a short video still occupies every rank: with ``b = min(block_max,
    ceil(left / world))`` rank j takes windows ``first + j*b .. first + (j+1)*b - 1`` of the round.  All windows of a round
    finish together, and the sequential alignment follows round by round.  ``block_max = 1`` is plain round-robin."""
    rounds, first = [], 0
    while first < n_windows:
        left = n_windows - first
        b = max(1, min(block_max, -(-left // world)))
        rounds.append([list(range(min(first + j * b, n_windows), min(first + (j + 1) * b, n_windows))) for j in range(world)])
        first += min(left, world * b)
    return rounds


class Aligner:
    """Sequential affine alignment of full windows, all arithmetic on the device (mirrors video_depth.py:169-205).

    ``advance(w, d0, d1, d12)`` takes the raw maps of slots 0, 1 and 12 of window w, in window order: window 0 provides the
    two reference maps (its slots 0 and 12); a later window is fitted ``s*d + t`` on its slots 0, 1 against the references
    and the second reference becomes its aligned slot 12.  Until the next ``advance``, ``apply`` aligns a raw map of that
    window and ``blend`` cross-fades one of its slots 2..9 onto the previous window's aligned tail.  The fit never leaves
    the device, so the host runs ahead of the GPU.
    """

    def __init__(self, device):
        self.dev = device
        self.ref0 = None
        self.ref1 = None
        self.sums = torch.zeros(hip.LSQ_SUMS_DOUBLES, dtype=torch.float64, device=device)   # 5 totals + partial-sum scratch
        self.ss = torch.zeros(2, dtype=torch.float32, device=device)
        self.window = -1

    def advance(self, w, d0, d1, d12):
        assert w == self.window + 1, "windows must be advanced in order"
        self.window = w
        if w == 0:
            self.ref0, self.ref1 = d0.clone(), d12.clone()
            return
        self.sums.zero_()
        hip.scale_shift_sums(d0.contiguous(), self.ref0, self.sums, d0.numel())
        hip.scale_shift_sums(d1.contiguous(), self.ref1, self.sums, d1.numel())
        hip.solve_scale_shift(self.sums, self.ss)
        self.ref1 = self.apply(d12, w)

    def apply(self, x, w, pre=None, w_pre=0.0, w_post=1.0, out=None):
        """max(s*x + t, 0) with the current window's fit (window 0 is the reference frame of the video: unchanged).  ``x`` may
        be one map or a contiguous run of maps (one launch); ``out``: a contiguous tensor of x's size to write into."""
        assert w == self.window
        if w == 0 and pre is None:
            if out is not None:
                out.copy_(x)
                return out
            return x
        if out is None:
            out = torch.empty_like(x)
        hip.affine_blend(x.contiguous(), pre, out, self.ss, float(w_pre), float(w_post), x.numel())
        return out

    def blend(self, x, pre, i, out=None):
        """Cross-fade of the current window's slot 2 + i (aligned) onto ``pre``, the previous window's aligned slot 24 + i,
        with the weights of utils/util.py:67-73: 0, 1/7, ..., 6/7, 1 on the new window's side."""
        n = INTERP_LEN
        w_post = 0.0 if i == 0 else (1.0 if i == n - 1 else i * (1.0 / (n - 1)))
        return self.apply(x, self.window, pre, 1.0 - w_post, w_post, out=out)
