"""Whole-video inference around ``VideoDepthAnything.forward`` (reference: video_depth.py:67-209).

Differences from the reference's data flow (results are the same function of the frames):
  * frames are uploaded once as uint8 and preprocessed on the device; a frame that appears in two windows is
    preprocessed once and cached, instead of a 103 MB fp32 host->device copy per window;
  * the full-resolution upsample, the least-squares alignment and the cross-fade run on the device; only final maps are
    copied back;
  * with ``torch.distributed`` initialised (one process per GPU, backend "nccl" = RCCL) the windows are split into
    contiguous blocks per rank.  A window depends only on INPUT frames, so ranks compute independently; afterwards the
    ranks all-gather, per window, the three maps the alignment chain needs (slots 0, 1, 12) plus each rank's last eight
    raw maps for the cross-fade at rank boundaries, replay the (cheap) sequential fit, align their own frames and gather
    the result.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import hip
from .windows import INFER_LEN, INTERP_LEN, OVERLAP, Stitcher, network_size, partition, resolve_plan, window_plan


def _dist_state():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


class FrameSource:
    """Preprocessed network-size frames on the device, computed on demand from the uint8 video and cached.  The frames a
    window is missing are staged through one pinned host buffer, uploaded with one asynchronous copy and preprocessed by
    one kernel launch."""

    def __init__(self, frames, H, W, device, keep=64):
        self.frames, self.H, self.W, self.dev, self.keep = frames, H, W, device, keep
        self.cache = {}

    def window(self, ids):
        """fp32 [len(ids), 3, H, W] for the given video frame indices (repeats allowed)."""
        missing = sorted({i for i in ids if i not in self.cache})
        if missing:
            h0, w0 = self.frames[missing[0]].shape[:2]
            stage = torch.empty((len(missing), h0, w0, 3), dtype=torch.uint8)
            if self.dev.type == "cuda":
                stage = stage.pin_memory()
            host = stage.numpy()
            for j, i in enumerate(missing):
                f = self.frames[i]
                host[j] = f if f.dtype == np.uint8 else np.clip(f, 0, 255).astype(np.uint8)
            u8 = stage.to(self.dev, non_blocking=True)
            t = torch.empty((len(missing), 3, self.H, self.W), dtype=torch.float32, device=self.dev)
            hip.preprocess_u8(u8, t, len(missing), h0, w0, self.H, self.W)
            for j, i in enumerate(missing):
                self.cache[i] = t[j]
        out = torch.stack([self.cache[i] for i in ids], 0)
        if len(self.cache) > self.keep:
            for k in sorted(self.cache)[: len(self.cache) - self.keep]:
                del self.cache[k]
        return out


class HostSink:
    """Final depth maps device -> host: each batch is copied into pinned memory on a side stream while the next window is
    computed; ``result()`` waits for the copies and assembles the [n, h0, w0] array."""

    def __init__(self, n, h0, w0, device):
        self.out = np.empty((n, h0, w0), dtype=np.float32)
        self.n, self.done, self.pending = n, 0, []
        # host tensors (the CPU doubles the host-logic tests run this module with) are copied synchronously
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None

    def push(self, maps):
        k = min(len(maps), self.n - self.done)
        if k <= 0:
            return
        block = torch.stack(list(maps[:k]), 0)
        if self.stream is None:
            self.out[self.done:self.done + k] = block.numpy()
            self.done += k
            return
        ready = torch.cuda.Event()
        ready.record()
        host = torch.empty(block.shape, dtype=torch.float32).pin_memory()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            host.copy_(block, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.stream)
        block.record_stream(self.stream)
        self.pending.append((copied, host, self.done, k))
        self.done += k

    def result(self):
        for copied, host, start, k in self.pending:
            copied.synchronize()
            self.out[start:start + k] = host.numpy()
        self.pending = []
        return self.out


def run_video(model, frames, target_fps, input_size=518, device="cuda", batch_size=16, forward_fn=None):
    n = len(frames)
    if n == 0:
        raise ValueError("no frames")
    h0, w0 = frames[0].shape[:2]
    dev = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}") if isinstance(device, str) else device
    clamp = bool(getattr(model, "clamp_input_size_on_gpu", False)) and str(device).startswith("cuda")
    H, W = network_size(h0, w0, input_size, clamp_420=clamp)
    fwd = forward_fn or model.forward
    plan, L, O = window_plan(n, batch_size)
    frame_ids = resolve_plan(plan)
    n_win = len(plan)
    world, rank = _dist_state()
    lo, hi = partition(n_win, world, rank)
    src = FrameSource(frames, H, W, dev)

    def raw_window(w):
        """Raw full-resolution depth [L, h0, w0] of window w."""
        x = src.window(frame_ids[w]).unsqueeze(0)
        d = fwd(x).float().reshape(L, H, W).contiguous()
        if (H, W) == (h0, w0):
            return d
        out = torch.empty((L, h0, w0), dtype=torch.float32, device=dev)
        hip.bilinear_f32(d, out, L, H, W, h0, w0, relu=False)
        return out

    stitched = L == INFER_LEN and batch_size >= INFER_LEN

    # ---- single process -------------------------------------------------------------------------------------------------
    if world == 1:
        sink = HostSink(n, h0, w0, dev)
        if stitched:
            st = Stitcher(dev)
            for w in range(n_win):
                sink.push(st.push(raw_window(w)))
            sink.push(st.flush())
        else:
            for w in range(n_win):      # reference quirk: overlapping windows concatenated unaligned (video_depth.py:166-168)
                d = raw_window(w)
                sink.push([d[i] for i in range(L)])
        return sink.result(), target_fps

    # ---- one process per GPU ------------------------------------------------------------------------------------------------
    mine = [raw_window(w) for w in range(lo, hi)]
    per = -(-n_win // world)
    if not stitched:
        local = torch.zeros((per * L, h0, w0), dtype=torch.float32, device=dev)
        for j, d in enumerate(mine):
            local[j * L:(j + 1) * L] = d
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        full = torch.cat(gathered, 0)[: n_win * L]
        return full[:n].cpu().numpy(), target_fps

    # key maps of every window (slots 0, 1, 12) and every rank's last 8 raw maps
    keys = torch.zeros((per, 3, h0, w0), dtype=torch.float32, device=dev)
    for j, d in enumerate(mine):
        keys[j, 0], keys[j, 1], keys[j, 2] = d[0], d[1], d[12]
    tail = mine[-1][INFER_LEN - INTERP_LEN:].contiguous() if mine else torch.zeros((INTERP_LEN, h0, w0), dtype=torch.float32, device=dev)
    all_keys = [torch.empty_like(keys) for _ in range(world)]
    all_tail = [torch.empty_like(tail) for _ in range(world)]
    dist.all_gather(all_keys, keys)
    dist.all_gather(all_tail, tail)
    all_keys = torch.cat(all_keys, 0)[:n_win]

    # replay the sequential fit over all windows (every rank, redundantly: K small reductions)
    st = Stitcher(dev)
    st.started = True
    st.ref0, st.ref1 = all_keys[0, 0].contiguous(), all_keys[0, 2].contiguous()
    scales = [None] * n_win                 # device tensors [scale, shift] per window
    for w in range(1, n_win):
        st.fit(all_keys[w, 0].contiguous(), all_keys[w, 1].contiguous())
        scales[w] = st.ss.clone()
        st.ref1 = st._affine(all_keys[w, 2].contiguous())

    ident = torch.tensor([1.0, 0.0], dtype=torch.float32, device=dev)

    def aligned(x, w, pre=None, w_pre=0.0, w_post=1.0):
        out = torch.empty_like(x)
        hip.affine_blend(x.contiguous(), pre, out, scales[w] if scales[w] is not None else ident, float(w_pre), float(w_post), x.numel())
        return out

    # frames this rank owns: window w contributes video frames 22w+2 .. 22w+23 (window 0: 0 .. 23); the very last window
    # also owns its slots 24..31
    pieces = []
    for j, d in enumerate(mine):
        w = lo + j
        if w == 0:
            pieces += [d[i] for i in range(INFER_LEN - INTERP_LEN)]
        else:
            if j > 0:
                prev_tail, pw = mine[j - 1][INFER_LEN - INTERP_LEN:], w - 1
            else:
                prev_tail, pw = all_tail[rank - 1], w - 1
            for i in range(INTERP_LEN):
                w_post = 0.0 if i == 0 else (1.0 if i == INTERP_LEN - 1 else i * (1.0 / (INTERP_LEN - 1)))
                pre = aligned(prev_tail[i], pw) if pw > 0 else prev_tail[i].contiguous()
                pieces.append(aligned(d[2 + i], w, pre, 1.0 - w_post, w_post))
            pieces += [aligned(d[i], w) for i in range(OVERLAP, INFER_LEN - INTERP_LEN)]
        if w == n_win - 1:
            pieces += [aligned(d[i], w) if w > 0 else d[i] for i in range(INFER_LEN - INTERP_LEN, INFER_LEN)]
    step = INFER_LEN - OVERLAP
    cap = per * step + INFER_LEN
    local = torch.zeros((cap, h0, w0), dtype=torch.float32, device=dev)
    cnt = torch.tensor([len(pieces)], dtype=torch.int64, device=dev)
    if pieces:
        local[: len(pieces)] = torch.stack(pieces, 0)
    gathered = [torch.empty_like(local) for _ in range(world)]
    counts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(gathered, local)
    dist.all_gather(counts, cnt)
    full = torch.cat([g[: int(c.item())] for g, c in zip(gathered, counts)], 0)
    return full[:n].cpu().numpy(), target_fps
