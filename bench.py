"""Throughput of the per-window forward pass (frames per second, 32-frame windows at 518 px) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step is one 32-frame window of VDA-Large at 518x518 through ``VideoDepthAnything.forward`` with the input already in
HBM.  Windows depend only on input frames, so with N > 1 every rank runs its own windows (weak scaling, no data-path
collective); the reported value is all ranks' frames divided by the slowest rank's time.

The JSON line also carries
  roofline      : the dominant kernel (the 256x256-tile bf16 MFMA GEMM in the instantiation of the proj / fc2 products) timed
                  per launch with HIP events on the stream it is launched on: algorithmic FLOP per launch / average launch
                  time vs the dense bf16 peak.  In the timed steps the encoder's two frame groups (and the head's independent
                  branches) run on two streams and their launches overlap, so these times come from a second pass of the same K steps with the groups on
                  one stream (``--encoder-streams 1`` runs the whole bench that way, as the committed rocprof summary does);
  fp32_mode     : the same window in the fp32 storage mode (``precision = "fp32"``, what ``infer_video_depth(fp32=True)``
                  runs): the mode that meets the parity bounds also where the final ReLU clamps pixels to 0; N = 1 only;
  vits          : the same measurement for VDA-Small (the metric names both models), N = 1 only;
  end_to_end    : the whole-video path (`infer_video_depth`) from host frames to host depth, PCIe included (N = 1 only);
  sharded_video : N > 1: one video through `infer_video_depth` with its windows dealt over the ranks (RCCL all-gather of
                  the overlap maps only, every rank keeps its own frames);
  cpu_baseline  : the CPU oracle (fp32 torch restatement of the reference) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "monocular-depth-estimation_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0
# reference model under FlopCounterMode, GFLOP per model-frame (2*MAC), T = 32
GFLOP_PER_FRAME = {("vitl", 518, 518): 1404.67, ("vits", 518, 518): 121.28, ("vitl", 518, 924): 2761.52,
                   ("vits", 518, 924): 264.43, ("vitl", 1036, 1036): 7822.31, ("vits", 1036, 1036): 898.78,
                   ("vitl", 518, 686): 1938.96, ("vits", 518, 686): 175.41}
# (kernel, epilogue form, operand type) of the proj / fc2 products, the fp32 residual update; with the encoder's LayerNorms
# folded into qkv / fc1 (VDA_FOLD_LAYERNORM=1; off by default at ViT-L width) the form that also emits the rounded rows
DOMINANT = ("gemm256_kernel", "f32_res32", "bf16")
DOMINANT_FOLDED = ("gemm256_kernel", "f32_res32_emit", "bf16")
TRAFFIC_PROFILE = "profiles/r3_vitl518_hbm_traffic.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--encoder", default="vitl", choices=["vits", "vitl"])
    ap.add_argument("--height", type=int, default=518)
    ap.add_argument("--width", type=int, default=518)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--precision", default="half", choices=["half", "fp32"],
                    help="half: 16-bit activations (the headline); fp32: fp32 activations, three-term fp16 hi/lo products")
    ap.add_argument("--encoder-streams", type=int, default=None,
                    help="frame groups of the encoder that run on concurrent HIP streams (library default: 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-vits", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true")
    ap.add_argument("--no-sharded-video", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single device: every rank uses cuda:0 and the gloo backend")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    from video_depth_anything import hip, spec
    from video_depth_anything.video_depth import VideoDepthAnything

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build(encoder):
        model = VideoDepthAnything(**spec.MODEL_CONFIGS[encoder])     # random-init weights of the named architecture
        model = model.to(dev).eval()
        model.precision = args.precision
        if args.encoder_streams is not None:
            model.encoder_streams = args.encoder_streams
        return model

    def timed_steps(model, x, steps, warmup):
        """Seconds for `steps` windows after `warmup` untimed ones (barrier + device synchronize on both sides, slowest rank)."""
        for _ in range(warmup):
            model(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(x)
        barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    def kernel_pass(model, x, steps):
        """The same steps with the encoder's frame groups on ONE stream and every plain GEMM launch bracketed by HIP events."""
        saved = model.encoder_streams
        model.encoder_streams = 1
        try:
            model(x)
            torch.cuda.synchronize()
            timing = hip.enable_gemm_timing()
            for _ in range(steps):
                model(x)
            torch.cuda.synchronize()
        finally:
            hip.disable_gemm_timing()
            model.encoder_streams = saved
        return timing

    def roofline_of(timing, tag, label):
        launches, flops, ms = timing.summary(tag)
        if not launches:
            return None
        ach = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None, "kernel": label, "launches": launches,
                "avg_launch_us": round(1e3 * ms / launches, 2), "avg_gflop_per_launch": round(flops / launches / 1e9, 3),
                "measured_in": "a second pass of the same steps with the encoder's frame groups on one stream (in the timed "
                               "steps they overlap on two streams); HIP events on the launching stream"}

    T, H, W = args.frames, args.height, args.width
    model = build(args.encoder)
    x = torch.from_numpy(np.random.RandomState(rank).standard_normal((1, T, 3, H, W)).astype(np.float32)).to(dev)
    dt = timed_steps(model, x, args.steps, args.warmup)
    headline = (args.encoder, H, W, T, args.precision) == ("vitl", 518, 518, 32, "half")

    out = None
    enc16 = "f16" if model.encoder_dtype == torch.float16 else "bf16"       # VDA_ENCODER_DTYPE=fp16 selects IEEE-half operands
    if rank == 0:
        fps = world * args.steps * T / dt
        out = {
            "metric": "frames/sec at 518px (VDA-L, 32-frame windows)" if headline
            else f"frames/sec ({args.encoder}, {T}-frame windows at {H}x{W}, precision {args.precision})",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": enc16 if args.precision == "half" else "f32 (3 x fp16 hi/lo MFMA passes)", "data": "synthetic",
            "config": {"workload": f"Video-Depth-Anything-{'Large (ViT-L)' if args.encoder == 'vitl' else 'Small (ViT-S)'} "
                                   f"forward, 1x{T}x3x{H}x{W} window resident in HBM, random-init weights",
                       "windows_per_step_per_gpu": 1, "parallelism": f"window-parallel x{world}",
                       "encoder_streams": model.encoder_streams,
                       "encoder_layernorms": "folded into qkv / fc1" if model._ensure_engine().folds_layernorms() else "separate pass",
                       "operands": f"encoder {enc16} MFMA, head fp16 MFMA, fp32 accumulate" if args.precision == "half"
                       else "fp32 activations; every product hi*hi + lo*hi + hi*lo on fp16 MFMA, fp32 accumulate",
                       "parity": "16-bit mode: max-rel <= 1e-2 / AbsRel <= 0.5 % on depth maps without clamped pixels, 4-8e-3 of "
                                 "the map maximum absolute where pixels are clamped to 0; --precision fp32 meets 1e-2 / 0.5 % "
                                 "on both (tests/test_model_gpu.py)"},
        }
        gf = GFLOP_PER_FRAME.get((args.encoder, H, W))
        if gf:
            out["model_tflops"] = round(fps / world * gf / 1e3, 2)
            out["model_mfma_frac"] = round(fps / world * gf / 1e3 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)
    if not args.no_kernel_timing and rank == 0 and args.precision == "half":
        timing = kernel_pass(model, x, args.steps)
        # the single kernel with the largest share of the window: the 256x256-tile bf16 GEMM instantiated for an fp32 output
        # with an fp32 residual (the proj and fc2 products, 48 launches per ViT-L window)
        folded = timing.summary(DOMINANT_FOLDED)[0] > timing.summary(DOMINANT)[0]
        if folded:
            roof = roofline_of(timing, DOMINANT_FOLDED, "gemm256_kernel<LINEAR, EPI_SPEC_F32_RES32_EMIT, bf16> (encoder proj and fc2)")
            pmc_name = "gemm256_kernelILi0ELb0ELi14EDF16b"
        else:
            roof = roofline_of(timing, DOMINANT, "gemm256_kernel<LINEAR, EPI_SPEC_F32_RES32, bf16> (encoder proj and fc2)")
            pmc_name = "gemm256_kernelILi0ELb0ELi3EDF16b"
        if roof is None:      # ViT-S: the dominant tag is whichever plain GEMM kernel took the most time
            tags = sorted(timing.tags(), key=lambda t: -timing.summary(t)[2])
            if tags:
                roof = roofline_of(timing, tags[0], " / ".join(tags[0]))
        if roof is not None:
            fam = {}
            for tag in timing.tags():
                n_, f_, ms_ = timing.summary(tag)
                fam[" / ".join(tag)] = {"launches": n_, "avg_launch_us": round(1e3 * ms_ / n_, 2),
                                        "tflops": round(f_ / (ms_ * 1e-3) / 1e12, 2)}
            n_, f_, ms_ = timing.summary(None)
            fam["all plain linear GEMMs"] = {"launches": n_, "tflops": round(f_ / (ms_ * 1e-3) / 1e12, 2), "ms_per_step": round(ms_ / args.steps, 3)}
            roof["other_instantiations"] = fam
            if headline:
                roof.update(pmc_traffic(pmc_name))
            out["roofline"] = roof
    if rank == 0 and world == 1 and not args.no_fp32_mode and args.precision == "half":
        try:
            out["fp32_mode"] = fp32_mode_line(model, x, timed_steps, args)
        except Exception as e:      # noqa: BLE001  (an auxiliary block: reported in the line, not swallowed)
            out["fp32_mode"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if rank == 0 and world == 1 and not args.no_fp32_mode and args.precision == "half" and enc16 == "bf16":
        try:
            out["fp16_encoder"] = fp16_encoder_line(model, x, timed_steps, args)
        except Exception as e:      # noqa: BLE001
            out["fp16_encoder"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    del x
    if rank == 0 and world == 1 and not args.no_vits and args.encoder == "vitl":
        out["vits"] = vits_line(build, timed_steps, kernel_pass, roofline_of, args, dev)
    if world == 1 and not args.no_end_to_end:
        try:
            out["end_to_end"] = end_to_end(model, H, W)
        except Exception as e:      # noqa: BLE001
            out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if world > 1 and not args.no_sharded_video:
        try:        # an auxiliary block: a failure in it must not cost the run its headline line
            sv = sharded_video(model, H, W, world, rank, dev, barrier)
        except Exception as e:      # noqa: BLE001  (reported in the line, not swallowed)
            sv = {"error": f"{type(e).__name__}: {e}"[:400]}
        if rank == 0:
            out["sharded_video"] = sv
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.encoder, H, W, T)
        if args.encoder == "vitl" and not args.no_vits:
            out["vits"]["cpu_baseline"] = cpu_baseline("vits", 518, 518, 32)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def fp16_encoder_line(model, x, timed_steps, args):
    """The headline mode with IEEE-half instead of bf16 encoder operands (``model.encoder_dtype = torch.float16``): the same
    kernels and matrix-core rate, three more significand bits, a third to a quarter of the 16-bit mode's error."""
    saved = model.encoder_dtype
    model.encoder_dtype = torch.float16
    try:
        dt = timed_steps(model, x, args.steps, args.warmup)
    finally:
        model.encoder_dtype = saved
    fps = args.steps * x.shape[1] / dt
    return {"metric": "frames/sec, 16-bit mode with fp16 encoder operands", "value": round(fps, 2), "unit": "frames/s",
            "ms_per_step": round(1e3 * dt / args.steps, 3), "steps": args.steps, "warmup": args.warmup, "dtype": "f16",
            "parity": "max-rel 2-3e-3 on smooth depth maps (bf16 operands: 4-9e-3); AbsRel 0.4 % / floored max-rel above the bound "
                      "where pixels are clamped to 0 (tests/test_model_gpu.py, profiles/r3_precision_table_fp16_encoder.txt)"}


def fp32_mode_line(model, x, timed_steps, args):
    """The same model and window in the fp32 storage mode: fp32 activations, every product hi*hi + lo*hi + hi*lo of
    IEEE-half pieces on the 16-bit matrix cores (three times the matrix work of the headline mode)."""
    steps = max(3, min(args.steps, 6))
    model.precision = "fp32"
    try:
        dt = timed_steps(model, x, steps, 2)
    finally:
        model.precision = args.precision
        model._engines.pop("fp32", None)           # frees the fp32 mode's packed weights and workspaces
        torch.cuda.empty_cache()
    T = x.shape[1]
    fps = steps * T / dt
    line = {"metric": "frames/sec, fp32 storage mode (infer_video_depth(fp32=True))", "value": round(fps, 2), "unit": "frames/s",
            "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": 2,
            "dtype": "f32 activations; products as 3 fp16 hi/lo MFMA passes, fp32 accumulate",
            "parity": "max-rel <= 1e-2 / AbsRel <= 0.5 % on smooth and on clamped depth maps (tests/test_model_gpu.py)"}
    gf = GFLOP_PER_FRAME.get((args.encoder, x.shape[3], x.shape[4]))
    if gf:
        line["model_tflops"] = round(fps * gf / 1e3, 2)
        line["matrix_tflops_issued"] = round(3 * fps * gf / 1e3, 2)
    return line


def vits_line(build, timed_steps, kernel_pass, roofline_of, args, dev):
    """VDA-Small, the other model the metric names: one 32-frame window at 518x518 per step, same measurement."""
    model = build("vits")
    x = torch.from_numpy(np.random.RandomState(7).standard_normal((1, 32, 3, 518, 518)).astype(np.float32)).to(dev)
    steps = max(args.steps, 20)
    dt = timed_steps(model, x, steps, args.warmup)
    fps = steps * 32 / dt
    gf = GFLOP_PER_FRAME[("vits", 518, 518)]
    line = {"metric": "frames/sec at 518px (VDA-S, 32-frame windows)", "value": round(fps, 2), "unit": "frames/s",
            "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "model_tflops": round(fps * gf / 1e3, 2),
            "model_mfma_frac": round(fps * gf / 1e3 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    if not args.no_kernel_timing and args.precision == "half":
        timing = kernel_pass(model, x, steps)
        tags = sorted(timing.tags(), key=lambda t: -timing.summary(t)[2])
        if tags:
            line["roofline"] = roofline_of(timing, tags[0], " / ".join(tags[0]) + " (the plain GEMM instantiation with the largest share)")
            n_, f_, ms_ = timing.summary(None)
            line["roofline"]["all_plain_linear_gemms"] = {"launches": n_, "tflops": round(f_ / (ms_ * 1e-3) / 1e12, 2),
                                                          "ms_per_step": round(ms_ / steps, 3)}
    return line


def pmc_traffic(kernel_substring):
    """HBM-side bytes per launch of the dominant kernel on the flagship workload, from the committed counter passes
    (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` over this same command, one pass each; FETCH_SIZE doubled for
    gfx950's 64-byte tally of 128-byte requests): counters cannot be collected from inside the timed process."""
    for rel in (TRAFFIC_PROFILE, "profiles/r1_step4_vitl518_hbm_traffic.json"):
        try:
            rows = json.load(open(os.path.join(ROOT, rel)))
            row = next(r for r in rows if kernel_substring in r["kernel"])
        except (OSError, StopIteration, ValueError):
            continue
        return {"traffic": round(row["hbm_bytes_per_launch"]), "traffic_unit": "bytes per launch (fetch x2 + write)",
                "traffic_source": rel}
    return {}


def end_to_end(model, H, W, n=98):
    """``infer_video_depth`` on a synthetic uint8 video held in host memory (frames already at network size): upload,
    device preprocessing, all windows, alignment, cross-fade and the copy of every depth map back to the host.  Not part
    of `value`: it is the PCIe-inclusive rate of the whole video path, in VIDEO frames (a window advances 22 frames)."""
    frames = np.random.RandomState(1).randint(0, 256, (n, H, W, 3), dtype=np.uint8)
    from video_depth_anything.windows import window_plan
    windows = len(window_plan(n, 32)[0])
    model.infer_video_depth(frames[:33], 30, input_size=min(H, W), device="cuda", batch_size=32)      # warm-up
    res = {}
    # second leg: every window recomputes the encoder outputs of the ten frames it shares with the window before it, as
    # the reference does (reuse_encoder_features = False)
    for name, reuse in (("", True), ("_recomputing_shared_frames", False)):
        model.reuse_encoder_features = reuse
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        depth, _ = model.infer_video_depth(frames, 30, input_size=min(H, W), device="cuda", batch_size=32)
        dt = time.perf_counter() - t0
        assert depth.shape == (n, H, W)
        res["video_frames_per_s" + name] = round(n / dt, 2)
        res["seconds" + name] = round(dt, 3)
    model.reuse_encoder_features = True
    res.update({"model_frames_per_s_recomputing_shared_frames": round(32 * windows / res["seconds_recomputing_shared_frames"], 2),
                "frames": n, "windows": windows,
                "path": "host uint8 frames -> infer_video_depth(batch_size=32) -> host fp32 depth (PCIe both ways included); by "
                        "default a window takes the encoder outputs of the 10 frames it shares with the window before it from "
                        "that window"})
    return res


def sharded_video(model, H, W, world, rank, dev, barrier, windows_per_rank=12):
    """One video through ``infer_video_depth`` with its windows dealt over the ranks in rounds of up to eight consecutive
    windows per rank: per round each rank contributes 11 raw maps per window (slots 0, 1, 12 and 24..31) to one RCCL
    all-gather, replays the alignment chain, aligns its own frames and copies them to its own host memory
    (``shard_output='local'``: no final frame is exchanged)."""
    n = 22 * world * windows_per_rank               # exactly world * windows_per_rank windows
    base = np.random.RandomState(5).randint(0, 256, (64, H, W, 3), dtype=np.uint8)
    frames = [base[i % 64] for i in range(n)]       # every rank holds the same video (a list of frames, 64 distinct ones)
    model.shard_output = "local"
    model.infer_video_depth(frames[:22 * world + 10], 30, input_size=min(H, W), device=str(dev), batch_size=32)      # warm-up
    barrier()
    t0 = time.perf_counter()
    depth, _ = model.infer_video_depth(frames, 30, input_size=min(H, W), device=str(dev), batch_size=32)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    owned = torch.tensor([int(model.last_owned_frames.sum())], dtype=torch.int64, device=dev)
    if dist.get_backend() == "nccl":
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(owned)
    else:
        tmax, owned = tmax.cpu(), owned.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(owned)
    dt = float(tmax.item())
    st = model.last_shard_stats
    return {"video_frames_per_s": round(n / dt, 2), "model_frames_per_s": round(32 * st["windows"] / dt, 2), "frames": n,
            "windows": st["windows"], "rounds": st["rounds"], "seconds": round(dt, 3), "ranks_seen": dist.get_world_size(),
            "backend": dist.get_backend(), "frames_owned_over_all_ranks": int(owned.item()),
            "all_gathered_bytes": st["gathered_bytes"],
            "all_gathered_bytes_per_window_per_rank": st["gathered_bytes_per_window_per_rank"],
            "path": "host uint8 frames on every rank -> infer_video_depth(batch_size=32, shard_output='local') -> each rank's own "
                    "frames in its host memory; a rank runs blocks of up to 8 consecutive windows (the later ones take the "
                    "encoder outputs of the 10 shared frames from the one before); collectives: one all-gather of 11 fp32 "
                    "maps per window and rank per round"}


def cpu_info():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, len(cores)


def cpu_baseline(encoder, H, W, T=32):
    """The CPU oracle on the same workload: one 4-frame warm-up window, then three timed full windows (T frames each),
    median reported.  A 32-frame ViT-L window takes about 40 s on 16 threads, so this leg is most of the bench's wall time;
    a progress line goes to stderr after every window."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vda_oracle as O
    from video_depth_anything import spec
    cfg = spec.MODEL_CONFIGS[encoder]
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    sd = {}
    g = torch.Generator().manual_seed(0)
    for k, (shape, kind) in spec.state_dict_spec(**cfg).items():       # throughput does not depend on the values: cheap random fill
        sd[k] = torch.randn(shape, generator=g) * 0.02
    rs = np.random.RandomState(0)
    O.forward(sd, torch.from_numpy(rs.standard_normal((1, 4, 3, H, W)).astype(np.float32)), encoder)      # warm-up
    print(f"cpu_baseline {encoder}: warm-up window done", file=sys.stderr, flush=True)
    times = []
    for i in range(3):
        x = torch.from_numpy(rs.standard_normal((1, T, 3, H, W)).astype(np.float32))
        t0 = time.perf_counter()
        O.forward(sd, x, encoder)
        times.append(time.perf_counter() - t0)
        print(f"cpu_baseline {encoder}: window {i + 1}/3 took {times[-1]:.1f} s", file=sys.stderr, flush=True)
    med = statistics.median(times)
    model, phys = cpu_info()
    return {"value": round(T / med, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "cpu_model": model, "physical_cores_in_machine": phys, "threads_used": threads,
            "sample": f"CPU oracle (fp32 torch restatement of the reference), 1 warm-up window of 4 frames + 3 timed windows of "
                      f"{T} frames at {H}x{W}; median {med:.1f} s per window ({', '.join(f'{t:.1f}' for t in times)})"}


if __name__ == "__main__":
    main()
