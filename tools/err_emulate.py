"""Diagnostic (CPU): emulate bf16 operand/output rounding in chosen stages of the fp32 oracle head to find which
stages dominate the final depth error.   python tools/err_emulate.py"""
import os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vda_oracle as O
from video_depth_anything import spec
import torch.nn.functional as RF

enc = "vits"; H, W, T = 70, 98, 32
cfg = spec.MODEL_CONFIGS[enc]
sd = {k: torch.from_numpy(v) for k, v in spec.synth_state_dict(**cfg, seed=1).items()}
key_of = {id(v): k for k, v in sd.items()}
x = torch.from_numpy(np.random.RandomState(41).standard_normal((1, T, 3, H, W)).astype(np.float32))
hp, wp = H // 14, W // 14
r = lambda t: t.bfloat16().float()
EXEMPT = []          # substrings of weight keys whose op stays fp32
OUT_ROUND = True
RW = True; RA = True
def lowp(w):
    k = key_of.get(id(w), "?")
    return not any(e in k for e in EXEMPT)
class Proxy(types.SimpleNamespace):
    def __getattr__(self, n): return getattr(RF, n)
P = Proxy()
def conv2d(x, w, b=None, **kw):
    if lowp(w):
        o = RF.conv2d(r(x) if RA else x, r(w) if RW else w, b, **kw); return r(o) if OUT_ROUND else o
    return RF.conv2d(x, w, b, **kw)
def convT(x, w, b=None, **kw):
    if lowp(w):
        o = RF.conv_transpose2d(r(x) if RA else x, r(w) if RW else w, b, **kw); return r(o) if OUT_ROUND else o
    return RF.conv_transpose2d(x, w, b, **kw)
def linear(x, w, b=None):
    if lowp(w):
        o = RF.linear(r(x) if RA else x, r(w) if RW else w, b); return r(o) if OUT_ROUND else o
    return RF.linear(x, w, b)
P.conv2d, P.conv_transpose2d, P.linear = conv2d, convT, linear
def fin(d): return RF.relu(RF.interpolate(d, size=(H, W), mode="bilinear", align_corners=True)).squeeze(1).numpy()
with torch.no_grad():
    taps = O.encoder_taps(sd, x.flatten(0, 1), enc)
    want = fin(O.head_forward(sd, taps, hp, wp, T))
    O.F = P
    def run(name, exempt, out_round=True):
        global EXEMPT, OUT_ROUND
        EXEMPT, OUT_ROUND = exempt, out_round
        got = fin(O.head_forward(sd, [r(t) for t in taps], hp, wp, T))
        print(f"{name:58s} max_rel {O.max_rel_err(got, want):.3e} abs_rel {O.abs_rel(got, want):.3e}")
    run("everything bf16", [])
    RW=False; run("activations only rounded (weights exact), out fp32", [], False)
    RW=True; RA=False; run("weights only rounded (activations exact), out fp32", [], False)
    RA=True
    run("everything bf16, outputs kept fp32", [], False)
    run("oc2 exact", ["output_conv2"])
    run("oc1+oc2 exact", ["output_conv"])
    run("refinenet1 + oc exact", ["output_conv", "refinenet1"])
    run("refinenet1,2 + oc exact", ["output_conv", "refinenet1", "refinenet2"])
    run("all refinenets + oc exact", ["output_conv", "refinenet"])
    run("motion modules exact", ["motion_modules"])
    run("only motion modules bf16", ["output_conv", "refinenet", "projects", "resize_layers", "_rn"])
    run("only projects/resize/rn bf16", ["output_conv", "refinenet", "motion_modules"])
    run("only oc bf16", ["refinenet", "motion_modules", "projects", "resize_layers", "_rn"])
    run("only refinenet1 bf16", ["output_conv", "refinenet2", "refinenet3", "refinenet4", "motion_modules", "projects", "resize_layers", "_rn"])
    run("nothing bf16 (taps rounded only)", ["e", "o"])
