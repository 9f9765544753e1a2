"""Drop-in ``VideoDepthAnything`` for MI355X: same constructor, ``state_dict`` keys, ``forward`` and
``infer_video_depth`` as the reference class (reference: video_depth_anything/video_depth.py:35-209 and
metric_depth/video_depth_anything/video_depth.py:67-270), computed by hand-written gfx950 kernels.

There is no CPU or eager-PyTorch path: ``forward`` raises if the module or the input is not on a ROCm device or the
kernel library is not built.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .engine import Engine
from .packing import ENCODER_DTYPE, PRECISIONS, PackedWeights
from .spec import ENCODERS, state_dict_spec
from .windows import INFER_LEN, KEYFRAMES, OVERLAP

__all__ = ["VideoDepthAnything", "INFER_LEN", "OVERLAP", "KEYFRAMES"]


def _install(root, dotted, tensor, buffer):
    """Register ``tensor`` under the dotted name, creating plain container modules on the way."""
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        nxt = getattr(mod, name, None) if name in mod._modules else None
        if nxt is None:
            nxt = nn.Module()
            mod.add_module(name, nxt)
        mod = nxt
    if buffer:
        mod.register_buffer(parts[-1], tensor, persistent=True)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class VideoDepthAnything(nn.Module):
    def __init__(self, encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024], use_bn=False, use_clstoken=False,
                 num_frames=32, pe="ape"):
        super().__init__()
        if encoder not in ENCODERS:
            raise ValueError(f"encoder {encoder!r} is not supported (vits, vitl)")
        if use_bn or use_clstoken or pe != "ape":
            raise NotImplementedError("only use_bn=False, use_clstoken=False, pe='ape' (the published checkpoints) are built")
        assert num_frames > 0
        self.encoder = encoder
        self.features = features
        self.out_channels = list(out_channels)
        self.num_frames = num_frames
        self.intermediate_layer_idx = {"vits": [2, 5, 8, 11], "vitl": [4, 11, 17, 23]}
        # The reference clamps input_size to 420 when device == 'cuda' (video_depth.py:75-78), which on ROCm would
        # make 518-pixel inference impossible; its CPU path and the metric copy do not clamp.  Off unless asked for.
        self.clamp_input_size_on_gpu = False
        # 16-bit operand type of the DPT head and temporal modules (the encoder is always bf16).  fp16 has the same MFMA
        # rate as bf16 and 3 more mantissa bits; with bf16 here the depth error is 1.6e-2 max-relative (above the 1e-2
        # bound), with fp16 it is set by the bf16 encoder alone.  The reference's own GPU default is fp16 autocast.
        self.head_dtype = torch.float16
        # 16-bit operand type of the encoder's products and stored activations (the residual stream, the norms' statistics,
        # softmax and all accumulators are fp32 either way).  VDA_ENCODER_DTYPE=fp16|bf16 sets the default.
        self.encoder_dtype = ENCODER_DTYPE
        # "half": 16-bit activations (bf16 encoder, head_dtype head).  "fp32": fp32 activations and three-term IEEE-half
        # hi/lo products on the same matrix cores (about 2^-21 per product) -- what ``infer_video_depth(fp32=True)``
        # selects, the counterpart of the reference running without autocast (video_depth.py:136).
        self.precision = "half"
        # fp32 mode: layout of the operand pieces (packing.PIECES_TRIPLE / PIECES_PAIRS); None = the library's default (pairs)
        self.pieces_layout = None
        gen = torch.Generator().manual_seed(0)
        for key, (shape, kind) in state_dict_spec(encoder, features, self.out_channels, num_frames).items():
            if kind in ("nw",):
                t = torch.ones(shape)
            elif kind in ("b", "nb", "b_out0", "b_out2"):
                t = torch.zeros(shape)
            elif kind == "ls":
                t = torch.ones(shape)
            elif kind == "pe":
                from .spec import sinusoid_table
                t = torch.from_numpy(sinusoid_table(shape[1], shape[2]))
            else:
                t = torch.empty(shape).normal_(0.0, 0.02, generator=gen)
            _install(self, key, t, buffer=(kind == "pe"))
        # frame groups of the encoder that run on concurrent HIP streams (see Engine.encoder)
        self.encoder_streams = int(os.environ.get("VDA_ENCODER_STREAMS", "2"))
        # 2: independent parts of the DPT head run side by side on two streams (Engine.head); 1: everything on the caller's
        self.head_streams = int(os.environ.get("VDA_HEAD_STREAMS", "2"))
        # multi-rank infer_video_depth: "replicate" = every rank returns the whole video, "local" = a rank returns only the
        # frames of its own windows (zeros elsewhere; ``last_owned_frames`` is the mask) and no final frame is exchanged
        self.shard_output = "replicate"
        # multi-rank infer_video_depth: consecutive windows a rank takes per round (windows.schedule); 1 = round-robin
        self.shard_block = 8
        # infer_video_depth: consecutive windows share ten of their 32 frames (video_depth.py:157-164) and the encoder treats
        # frames one by one, so a window takes the encoder outputs of those frames from the window before it on this
        # process instead of recomputing them (the temporal head still sees all 32).  False recomputes, like the reference.
        self.reuse_encoder_features = True
        # 16-bit mode: fold the encoder blocks' LayerNorms (block.py:84,87) into the qkv / fc1 products.  None = the engine's
        # choice (on at ViT-S width, off at ViT-L width, or what VDA_FOLD_LAYERNORM said when the engine was built)
        self.fold_layernorm = None
        self._engines = {}            # precision -> (weights key, Engine)
        self._flat = None             # the parameter and buffer tensors, listed once (see _weights_key)

    # ---- weights -> kernel layouts, rebuilt whenever the parameters change or move -----------------------------------
    def _apply(self, fn, *a, **kw):               # .to() / .cuda() / .float() replace buffers and parameter storage
        self._flat = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._flat = None
        return super().load_state_dict(*a, **kw)

    def _weights_key(self):
        """Identity of the current weights: (address, in-place version) of every tensor.  The tensor list itself is cached
        (walking the module tree costs more than a ViT-S window's launch overhead can spare) and rebuilt after every call
        that can replace tensor objects."""
        if self._flat is None:
            self._flat = list(self.state_dict(keep_vars=True).values())
        p = self.pretrained.cls_token
        return (p.device, self.head_dtype, self.encoder_dtype, self.pieces_layout, tuple((t.data_ptr(), t._version) for t in self._flat))

    def _ensure_engine(self):
        dev = self.pretrained.cls_token.device
        if dev.type != "cuda":
            raise RuntimeError("VideoDepthAnything (MI355X build) has no CPU path: call .to('cuda') on a ROCm device first")
        hip.lib()
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision {self.precision!r}: expected 'half' or 'fp32' (or a measurement form of packing.PRECISIONS)")
        key = self._weights_key()
        cached = self._engines.get(self.precision)
        if cached is None or cached[0] != key:
            packed = PackedWeights(self.state_dict(), self.encoder, self.features, self.out_channels, dev, self.head_dtype,
                                   precision=self.precision, layout=self.pieces_layout, encoder_dtype=self.encoder_dtype)
            cached = (key, Engine(packed))
            self._engines[self.precision] = cached
        return cached[1]

    def reset_frame_cache(self):
        """Drop the encoder outputs kept by keyed forward calls (start and end of a video)."""
        for _, eng in self._engines.values():
            eng.kept = None

    @torch.no_grad()
    def forward(self, x, frame_keys=None):
        """``x[B, T, 3, H, W]`` (ImageNet-normalised RGB, H and W multiples of 14, T <= num_frames) -> ``[B, T, H, W]`` fp32.

        ``frame_keys`` (no counterpart in the reference; B = 1): one hashable identity per frame.  The leading frames whose
        keys were also given to the previous keyed call take their encoder outputs from that call; the caller vouches that
        equal keys mean equal pixels and calls ``reset_frame_cache()`` when the keys start to mean something else."""
        if x.dim() != 5 or x.shape[2] != 3:
            raise ValueError(f"expected input [B, T, 3, H, W], got {tuple(x.shape)}")
        B, T, _, H, W = x.shape
        assert H % 14 == 0, f"Input image height {H} is not a multiple of patch height 14"
        assert W % 14 == 0, f"Input image width {W} is not a multiple of patch width: 14"
        if T > self.num_frames:
            raise ValueError(f"T={T} exceeds the temporal position table ({self.num_frames} frames)")
        if not x.is_cuda:
            raise RuntimeError("forward() needs a device tensor: the MI355X build has no CPU path")
        eng = self._ensure_engine()
        eng.encoder_streams = self.encoder_streams
        eng.head_streams = self.head_streams
        eng.fold_layernorm = eng.fold_default if self.fold_layernorm is None else bool(self.fold_layernorm)
        with torch.cuda.device(x.device):
            return eng.forward(x.float().contiguous(), frame_keys)

    @torch.no_grad()
    def infer_video_depth(self, frames, target_fps, input_size=518, device="cuda", fp32=False, batch_size=16,
                          use_cpu_offload=False):
        """Depth for every frame of a video (reference: video_depth.py:67-209).

        ``frames``: uint8 RGB ``[N, H0, W0, 3]`` array or a list of ``[H0, W0, 3]`` frames.  Returns
        ``(float32 [N, H0, W0], target_fps)``.  Windows of ``min(32, batch_size)`` frames with key-frame carry-over;
        with ``batch_size >= 32`` consecutive windows are affinely aligned and cross-faded, otherwise they are
        concatenated unaligned exactly like the reference does.  ``fp32=True`` runs the windows in the fp32 storage mode
        (see ``precision``); ``use_cpu_offload`` is accepted for signature compatibility (weights stay in HBM).

        Unlike the reference's relative-depth copy, ``device='cuda'`` does NOT clamp ``input_size`` to 420 (video_depth.py:75-78;
        the reference's CPU path and its metric copy do not clamp either): set ``clamp_input_size_on_gpu = True`` for that.
        With ``torch.distributed`` initialised the windows are dealt over the ranks (see ``shard_output``).
        """
        return self._infer_video(frames, target_fps, input_size, device, fp32, batch_size, False)

    @torch.no_grad()
    def infer_video_depth_device(self, frames, target_fps, input_size=518, device="cuda", fp32=False, batch_size=16):
        """``infer_video_depth`` that leaves the result in HBM: returns ``(fp32 device tensor [N, H0, W0], target_fps)``,
        for post-processing on the device (``metric_post.apply_lidar_scaling``).  No counterpart in the reference."""
        return self._infer_video(frames, target_fps, input_size, device, fp32, batch_size, True)

    def _infer_video(self, frames, target_fps, input_size, device, fp32, batch_size, keep_on_device):
        from .pipeline import run_video
        saved = self.precision
        if fp32:
            self.precision = "fp32"
        try:
            return run_video(self, frames, target_fps, input_size=input_size, device=device, batch_size=batch_size,
                             keep_on_device=keep_on_device)
        finally:
            self.precision = saved
