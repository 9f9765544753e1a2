"""Whole-video inference around ``VideoDepthAnything.forward`` (reference: video_depth.py:67-209).

Differences from the reference's data flow (results are the same function of the frames):
  * frames are uploaded once as uint8 and preprocessed on the device; a frame that appears in several windows is
    preprocessed once and cached, instead of a 103 MB fp32 host->device copy per window;
  * a window takes the encoder outputs of the frames it shares with the window before it (ten of 32) from that window
    instead of recomputing them (``model.reuse_encoder_features``);
  * the full-resolution upsample, the least-squares alignment and the cross-fade run on the device; only final maps are
    copied back, from a side stream, into a small ring of pinned buffers, while the next window is computed;
  * with ``torch.distributed`` initialised (one process per GPU, backend "nccl" = RCCL) the windows are dealt in rounds:
    per round a rank takes a block of up to eight consecutive windows (windows.schedule; consecutive, so that the carry-
    over above works on every rank).  A window depends only on INPUT frames, so the ranks compute a round independently.
    Per round each rank then contributes the eleven raw maps of each of its windows that other ranks need -- slots 0, 1,
    12 (the alignment chain, video_depth.py:180-201) and slots 24..31 (the cross-fade partner of the next window, :189)
    -- to ONE all-gather, issued asynchronously so that it travels while the next round is computed.  Every rank replays
    the round's cheap sequential fits, aligns the frames of its own windows and streams them to its host.  No final frame crosses the links unless the caller wants every rank to return the whole video
    (``shard_output="replicate"``, the default, for drop-in callers; "local" keeps a rank's frames on that rank).
"""
import queue
import threading
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from . import hip
from .windows import BLOCK_MAX, INFER_LEN, INTERP_LEN, OVERLAP, Aligner, network_size, resolve_plan, schedule, window_plan

KEY_SLOTS = (0, 1, 12)                      # raw maps of a window that the alignment chain reads
PACK = len(KEY_SLOTS) + INTERP_LEN          # maps a rank contributes per round: the key maps and the 8-map tail


def _dist_state():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


class FrameSource:
    """Preprocessed network-size frames on the device, computed on demand from the uint8 video and cached (least recently
    used frames are dropped first, so the key frames every window re-reads stay resident).  The frames a window is missing
    are staged through a pinned host buffer, uploaded with one asynchronous copy and preprocessed by one kernel launch.
    ``prefetch(ids)`` does that for a later window on a worker thread and an upload stream of its own, while the device
    computes the current window: the staging copy then costs the enqueuing thread nothing and the upload no time on the
    compute stream (together 1.2 of a ViT-S window's 9.8 ms)."""

    def __init__(self, frames, H, W, device, keep=96):
        self.frames, self.H, self.W, self.dev, self.keep = frames, H, W, device, keep
        self.cache = OrderedDict()
        self.stages = [None, None]           # pinned staging buffers, used in turn
        self.uploaded = [None, None]         # per buffer, event: its last upload has been read
        self.turn = 0
        self.up = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self.job = None                      # (thread, result holder) of the prefetch in flight

    def _load(self, missing):
        """Stage, upload and preprocess the frames `missing` on the current stream; returns their fp32 tensor."""
        h0, w0 = self.frames[missing[0]].shape[:2]
        b = self.turn
        self.turn ^= 1
        if self.stages[b] is None or self.stages[b].shape[0] < len(missing):
            self.stages[b] = torch.empty((max(len(missing), INFER_LEN), h0, w0, 3), dtype=torch.uint8)
            if self.dev.type == "cuda":
                self.stages[b] = self.stages[b].pin_memory()
        elif self.uploaded[b] is not None:
            self.uploaded[b].synchronize()                  # the previous upload from this buffer has been read
        host = self.stages[b].numpy()
        for j, i in enumerate(missing):
            f = self.frames[i]
            host[j] = f if f.dtype == np.uint8 else np.clip(f, 0, 255).astype(np.uint8)
        u8 = self.stages[b][: len(missing)].to(self.dev, non_blocking=True)
        if self.dev.type == "cuda":
            self.uploaded[b] = torch.cuda.Event()
            self.uploaded[b].record()
        t = torch.empty((len(missing), 3, self.H, self.W), dtype=torch.float32, device=self.dev)
        hip.preprocess_u8(u8, t, len(missing), h0, w0, self.H, self.W)
        return t

    def prefetch(self, ids):
        """Start loading those of the frames `ids` that are not cached; ``window`` picks the result up."""
        if self.up is None:
            return
        self._collect()
        missing = sorted({i for i in ids if i not in self.cache})
        if not missing:
            return
        box = {}

        def job():
            try:
                with torch.cuda.device(self.dev), torch.cuda.stream(self.up):      # (both are per thread)
                    box["t"] = self._load(missing)
                    box["done"] = torch.cuda.Event()
                    box["done"].record(self.up)
            except BaseException as e:
                box["error"] = e

        th = threading.Thread(target=job, name="vda-frame-prefetch", daemon=True)
        th.start()
        self.job = (th, box, missing)

    def _collect(self):
        """Take over what the prefetch in flight loaded: the current stream waits for it, the frames enter the cache."""
        if self.job is None:
            return
        th, box, missing = self.job
        self.job = None
        th.join()
        if "error" in box:
            raise box["error"]
        cur = torch.cuda.current_stream()
        cur.wait_event(box["done"])
        box["t"].record_stream(cur)           # allocated on the upload stream, read on this one from now on
        for j, i in enumerate(missing):
            self.cache[i] = box["t"][j]

    def close(self):
        if self.job is not None:
            self.job[0].join()
            self.job = None

    def window(self, ids):
        """fp32 [len(ids), 3, H, W] for the given video frame indices (repeats allowed)."""
        self._collect()
        missing = sorted({i for i in ids if i not in self.cache})
        if missing:
            t = self._load(missing)
            for j, i in enumerate(missing):
                self.cache[i] = t[j]
        for i in ids:
            self.cache.move_to_end(i)
        out = torch.stack([self.cache[i] for i in ids], 0)
        while len(self.cache) > self.keep:
            self.cache.popitem(last=False)
        return out


class HostSink:
    """Final depth maps device -> host.  ``push(start, maps)`` copies a run of consecutive video frames into one of a few
    pinned staging buffers on a side stream while the next window is computed; a worker thread waits for each copy and moves
    it into the [n, h0, w0] result (waiting for an event and copying an array both release the interpreter lock, so the
    thread that enqueues the next window's kernels is not held up: at ViT-S rates that copy was a fifth of the wall time).
    Pinned memory stays at ``ring`` windows whatever the video length: ``push`` blocks while all of them are in flight."""

    def __init__(self, n, h0, w0, device, ring=3):
        self.out = np.zeros((n, h0, w0), dtype=np.float32)      # untouched pages cost nothing (frames of other ranks)
        self.have = np.zeros(n, dtype=bool)
        self.n, self.ring, self.made = n, ring, 0
        # host tensors (the CPU doubles the host-logic tests run this module with) are copied synchronously
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self.free, self.todo, self.worker, self.error = queue.SimpleQueue(), queue.SimpleQueue(), None, None

    def _work(self):
        while True:
            item = self.todo.get()
            if item is None:
                return
            copied, host, start, k = item
            try:
                if self.error is None:
                    copied.synchronize()
                    self.out[start:start + k] = host[:k].numpy()
            except BaseException as e:                            # kept for result(); the buffers keep circulating
                self.error = e
            self.free.put(host)

    def push(self, start, maps):
        k = min(len(maps), self.n - start)
        if k <= 0:
            return
        block = maps[:k] if torch.is_tensor(maps) else torch.stack(list(maps[:k]), 0)      # a contiguous run of maps, or a list
        self.have[start:start + k] = True
        if self.stream is None:
            self.out[start:start + k] = block.numpy()
            return
        if self.worker is None:
            self.worker = threading.Thread(target=self._work, name="vda-host-sink", daemon=True)
            self.worker.start()
        if self.free.empty() and self.made < self.ring:
            host = torch.empty((INFER_LEN,) + tuple(block.shape[1:]), dtype=torch.float32).pin_memory()
            self.made += 1
        else:
            host = self.free.get()                                # the oldest copy in flight, once the worker is done with it
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            host[:k].copy_(block, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.stream)
        block.record_stream(self.stream)
        self.todo.put((copied, host, start, k))

    def close(self):
        """Ends the worker (after the copies already handed to it)."""
        if self.worker is not None:
            self.todo.put(None)
            self.worker.join()
            self.worker = None

    def result(self):
        self.close()
        if self.error is not None:
            raise self.error
        return self.out


class DeviceSink:
    """The HostSink interface for callers that keep the depth video in HBM (metric post-processing on the device)."""

    def __init__(self, n, h0, w0, device):
        self.out = torch.zeros((n, h0, w0), dtype=torch.float32, device=device)
        self.have = np.zeros(n, dtype=bool)
        self.n = n

    def push(self, start, maps):
        k = min(len(maps), self.n - start)
        if k > 0:
            self.out[start:start + k] = maps[:k] if torch.is_tensor(maps) else torch.stack(list(maps[:k]), 0)
            self.have[start:start + k] = True

    def close(self):
        pass

    def result(self):
        return self.out


def owned_runs(n_frames, n_win, stitched, L):
    """Per window, the (first video frame, first slot, count) of the output frames it finalises.  Stitched: window 0 gives
    frames 0..23, window w slots 2..23 -> frames 22w+2.., the last window also its slots 24..31 (video_depth.py:176-205);
    unstitched: window w's L slots land at output positions w*L.. (the reference concatenates, :166-168, 209)."""
    runs = []
    for w in range(n_win):
        if not stitched:
            start, slot, cnt = w * L, 0, L
        else:
            step = INFER_LEN - OVERLAP
            slot = 0 if w == 0 else OVERLAP - INTERP_LEN
            start = w * step + slot
            cnt = (INFER_LEN if w == n_win - 1 else INFER_LEN - INTERP_LEN) - slot
        cnt = max(0, min(cnt, n_frames - start))
        runs.append((start, slot, cnt))
    return runs


def run_video(model, frames, target_fps, input_size=518, device="cuda", batch_size=16, forward_fn=None, shard_output=None,
              keep_on_device=False):
    """``infer_video_depth``.  ``keep_on_device``: return the depth video as a device tensor instead of a numpy array."""
    n = len(frames)
    if n == 0:
        raise ValueError("no frames")
    h0, w0 = frames[0].shape[:2]
    if isinstance(device, str):
        if device == "cuda":
            p = next(model.parameters(), None) if model is not None else None
            device = p.device if p is not None and p.is_cuda else f"cuda:{torch.cuda.current_device()}"
        device = torch.device(device)
    dev = device
    if dev.type == "cuda":          # every launch below takes the current stream of the CURRENT device: make it `dev`
        with torch.cuda.device(dev):
            return _run_video(model, frames, target_fps, input_size, dev, batch_size, forward_fn, shard_output, n, h0, w0, keep_on_device)
    return _run_video(model, frames, target_fps, input_size, dev, batch_size, forward_fn, shard_output, n, h0, w0, keep_on_device)


def _run_video(model, frames, target_fps, input_size, dev, batch_size, forward_fn, shard_output, n, h0, w0, keep_on_device):
    clamp = bool(getattr(model, "clamp_input_size_on_gpu", False)) and dev.type == "cuda"
    H, W = network_size(h0, w0, input_size, clamp_420=clamp)
    # a window starts with frames of the window before it: a model that can carry their encoder outputs over is told
    # which video frame each slot holds
    keyed = forward_fn is None and bool(getattr(model, "reuse_encoder_features", False))
    fwd = forward_fn or model.forward
    if keyed:
        model.reset_frame_cache()
    shard_output = shard_output or getattr(model, "shard_output", "replicate")
    if shard_output not in ("replicate", "local"):
        raise ValueError(f"shard_output {shard_output!r}: expected 'replicate' or 'local'")
    plan, L, O = window_plan(n, batch_size)
    frame_ids = resolve_plan(plan)
    n_win = len(plan)
    world, rank = _dist_state()
    src = FrameSource(frames, H, W, dev)
    stitched = L == INFER_LEN and batch_size >= INFER_LEN
    runs = owned_runs(n, n_win, stitched, L)
    if keep_on_device and world > 1 and shard_output != "local":
        raise ValueError("keep_on_device with several ranks needs shard_output='local' (a rank keeps its own frames)")
    sink = DeviceSink(n, h0, w0, dev) if keep_on_device else HostSink(n, h0, w0, dev)
    try:
        return _run_windows(model, sink, fwd, keyed, src, frame_ids, runs, stitched, shard_output, world, rank, dev, n, n_win,
                            L, H, W, h0, w0, target_fps, keep_on_device)
    finally:
        sink.close()             # (an exception on the way must not leave the sink's worker thread waiting)
        src.close()


def _run_windows(model, sink, fwd, keyed, src, frame_ids, runs, stitched, shard_output, world, rank, dev, n, n_win,
                 L, H, W, h0, w0, target_fps, keep_on_device):
    # several ranks: a rank takes a block of consecutive windows per round (they share frames, see schedule); one process
    # finishes window by window; unaligned windows need no rounds at all
    if not stitched:
        rounds = schedule(n_win, world, n_win)
    else:
        rounds = schedule(n_win, world, int(getattr(model, "shard_block", BLOCK_MAX)) if world > 1 else 1)
    owner_of = {w: j for rd in rounds for j, ws in enumerate(rd) for w in ws}
    stats = dict(world=world, windows=n_win, rounds=len(rounds), gathered_bytes=0,
                 gathered_bytes_per_window_per_rank=PACK * h0 * w0 * 4 if world > 1 else 0)

    mine_in_order = [w for rd in rounds for w in rd[rank]]
    after = dict(zip(mine_in_order, mine_in_order[1:]))          # the window this rank computes next

    def raw_window(w):
        """Raw full-resolution depth [L, h0, w0] of window w."""
        x = src.window(frame_ids[w]).unsqueeze(0)
        d = (fwd(x, frame_keys=frame_ids[w]) if keyed else fwd(x)).float().reshape(L, H, W).contiguous()
        if w in after:                         # its frames are staged and uploaded while the device works on this one
            src.prefetch(frame_ids[after[w]])
        if (H, W) == (h0, w0):
            return d
        out = torch.empty((L, h0, w0), dtype=torch.float32, device=dev)
        hip.bilinear_f32(d, out, L, H, W, h0, w0, relu=False)
        return out

    if not stitched:        # reference quirk: overlapping windows concatenated unaligned; nothing to exchange
        for w in rounds[0][rank]:
            d = raw_window(w)
            start, slot, cnt = runs[w]
            sink.push(start, [d[slot + i] for i in range(cnt)])
    else:
        al = Aligner(dev)
        prev_tail = None                  # aligned slots 24..31 of the window before the next one this rank finishes
        pending = None
        for rd in range(len(rounds) + 1):
            cur = None
            if rd < len(rounds):
                mine = rounds[rd][rank]
                block = max(len(ws) for ws in rounds[rd])
                ds = [raw_window(w) for w in mine]
                if world > 1:
                    pack = torch.zeros((block * PACK, h0, w0), dtype=torch.float32, device=dev)
                    for b, d in enumerate(ds):
                        for j, s in enumerate(KEY_SLOTS):
                            pack[b * PACK + j] = d[s]
                        pack[b * PACK + len(KEY_SLOTS):(b + 1) * PACK] = d[INFER_LEN - INTERP_LEN:]
                    # the overlap maps only; asynchronous: it travels while the next round's windows are computed
                    packs = torch.empty((world, block * PACK, h0, w0), dtype=torch.float32, device=dev)
                    stats["gathered_bytes"] += packs.numel() * 4
                    if dist.get_backend() == "nccl":
                        work = dist.all_gather_into_tensor(packs, pack, async_op=True)
                    else:       # other backends (gloo in tests and single-GPU rehearsals): blocking, through host memory
                        work = None
                        parts = [torch.empty((block * PACK, h0, w0), dtype=torch.float32) for _ in range(world)]
                        dist.all_gather(parts, pack.cpu())
                        packs.copy_(torch.stack(parts, 0))
                else:
                    packs, work = None, None
                cur = (rd, ds, packs, work)
            if pending is not None:
                prd, pds, ppacks, pwork = pending
                if pwork is not None:
                    pwork.wait()
                for j in range(world):
                    for b, w in enumerate(rounds[prd][j]):
                        own = j == rank
                        if world > 1:
                            keys = ppacks[j, b * PACK:b * PACK + len(KEY_SLOTS)]
                            tail = ppacks[j, b * PACK + len(KEY_SLOTS):(b + 1) * PACK]
                        else:
                            keys, tail = [pds[b][s] for s in KEY_SLOTS], pds[b][INFER_LEN - INTERP_LEN:]
                        al.advance(w, keys[0], keys[1], keys[2])
                        if own:
                            pd = pds[b]
                            start, slot, cnt = runs[w]
                            if w == 0:                             # the video's reference window: unchanged
                                sink.push(start, pd[slot:slot + cnt])
                            elif cnt > 0:
                                final = torch.empty((cnt, h0, w0), dtype=torch.float32, device=dev)
                                i = 0
                                while i < cnt and slot + i < OVERLAP:      # slots 2..9: cross-fade onto the previous window's tail
                                    q = slot + i - (OVERLAP - INTERP_LEN)
                                    al.blend(pd[slot + i], prev_tail[q], q, out=final[i])
                                    i += 1
                                if i < cnt:                        # every other slot: scale and shift, one launch for the run
                                    al.apply(pd[slot + i:slot + cnt], w, out=final[i:])
                                sink.push(start, final)
                        # the aligned tail of window w is read by the owner of window w + 1
                        if w + 1 < n_win and owner_of[w + 1] == rank:
                            prev_tail = al.apply(tail if torch.is_tensor(tail) else torch.stack(list(tail), 0), w)
            pending = cur

    if keyed:
        model.reset_frame_cache()
    out = sink.result()
    owner = np.full(n, -1, dtype=np.int64)
    for w, (start, slot, cnt) in enumerate(runs):
        owner[start:start + cnt] = owner_of[w]
    if model is not None:
        model.last_owned_frames = owner == rank
        model.last_shard_stats = stats
    if world > 1 and shard_output == "replicate" and not keep_on_device:
        _replicate(out, owner, world, rank, dev)
    return out, target_fps


def _replicate(out, owner, world, rank, dev, chunk=32):
    """Every rank receives the frames the other ranks own (drop-in callers expect the whole video from every rank): one
    broadcast per owner and chunk of frames, staged through the device for the RCCL backend."""
    via_device = dist.get_backend() == "nccl"
    for root in range(world):
        ids = np.nonzero(owner == root)[0]
        for c0 in range(0, len(ids), chunk):
            sel = ids[c0:c0 + chunk]
            buf = torch.from_numpy(out[sel]) if rank == root else torch.empty((len(sel),) + out.shape[1:], dtype=torch.float32)
            if via_device:
                buf = buf.to(dev)
            dist.broadcast(buf, src=root)
            if rank != root:
                out[sel] = buf.cpu().numpy()
