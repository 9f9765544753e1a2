"""Throughput of the per-window forward pass (frames per second, 32-frame windows at 518 px) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step is one 32-frame window of VDA-Large at 518x518 through ``VideoDepthAnything.forward`` with the input already in
HBM.  Windows depend only on input frames, so with N > 1 every rank runs its own windows (weak scaling, no data-path
collective); the reported value is all ranks' frames divided by the slowest rank's time.

The JSON line also carries
  roofline     : the dominant kernel (the 256x256-tile bf16 MFMA GEMM, in the instantiation of the proj / fc2 products) timed per launch with HIP events on
                 its own stream inside the timed region: algorithmic FLOP per launch / average launch time vs the dense bf16 peak;
  end_to_end   : the whole-video path (`infer_video_depth`) from host frames to host depth, PCIe included (N = 1 only);
  cpu_baseline : the CPU oracle (fp32 torch restatement of the reference) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
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
                   ("vits", 518, 924): 264.43, ("vitl", 1036, 1036): 7822.31, ("vits", 1036, 1036): 898.78}


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
                    help="half: 16-bit activations (the headline); fp32: fp32 activations, three-term bf16 hi/lo products")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
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

    cfg = spec.MODEL_CONFIGS[args.encoder]
    model = VideoDepthAnything(**cfg)             # random-init weights of the named architecture (no checkpoints offline)
    model = model.to(dev).eval()
    model.precision = args.precision
    T, H, W = args.frames, args.height, args.width
    x = torch.from_numpy(np.random.RandomState(rank).standard_normal((1, T, 3, H, W)).astype(np.float32)).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model(x)
    barrier()
    timing = None
    if not args.no_kernel_timing and rank == 0:
        timing = hip.enable_gemm_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model(x)
    barrier()
    dt = time.perf_counter() - t0
    if timing is not None:
        hip.disable_gemm_timing()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        fps = world * args.steps * T / dt
        out = {
            "metric": "frames/sec at 518px (VDA-L, 32-frame windows)" if (args.encoder, H, W, T, args.precision) == ("vitl", 518, 518, 32, "half")
            else f"frames/sec ({args.encoder}, {T}-frame windows at {H}x{W}, precision {args.precision})",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "half" else "f32 (3 x bf16 hi/lo MFMA passes)", "data": "synthetic",
            "config": {"workload": f"Video-Depth-Anything-{'Large (ViT-L)' if args.encoder == 'vitl' else 'Small (ViT-S)'} "
                                   f"forward, 1x{T}x3x{H}x{W} window resident in HBM, random-init weights",
                       "windows_per_step_per_gpu": 1, "parallelism": f"window-parallel x{world}",
                       "operands": "encoder bf16 MFMA, head fp16 MFMA, fp32 accumulate" if args.precision == "half"
                       else "fp32 activations; every product hi*hi + lo*hi + hi*lo on bf16 MFMA, fp32 accumulate"},
        }
        gf = GFLOP_PER_FRAME.get((args.encoder, H, W))
        if gf:
            out["model_tflops"] = round(fps / world * gf / 1e3, 2)
            out["model_mfma_frac"] = round(fps / world * gf / 1e3 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)
        if timing is not None:
            # the single kernel with the largest share of the window: the 256x256-tile bf16 GEMM instantiated for an fp32
            # output with an fp32 residual (the proj and fc2 products, 48 launches per window)
            launches, flops, ms = timing.summary("f32_res32")
            if launches:
                ach = flops / (ms * 1e-3) / 1e12
                out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                                   "kernel": "gemm256_kernel<LINEAR, EPI_SPEC_F32_RES32, bf16> (encoder proj and fc2)",
                                   "launches": launches, "avg_launch_us": round(1e3 * ms / launches, 2),
                                   "avg_gflop_per_launch": round(flops / launches / 1e9, 3)}
                fam = {}
                for form, label in (("own", "qkv"), ("own_gelu", "fc1"), (None, "all four encoder linears")):
                    n_, f_, ms_ = timing.summary(form)
                    if n_:
                        fam[label] = {"launches": n_, "avg_launch_us": round(1e3 * ms_ / n_, 2), "tflops": round(f_ / (ms_ * 1e-3) / 1e12, 2)}
                out["roofline"]["other_instantiations"] = fam
                if (args.encoder, H, W, T, args.precision) == ("vitl", 518, 518, 32, "half"):
                    out["roofline"].update(pmc_traffic("gemm256_kernelILi0ELb0ELi3EDF16b"))
        if not args.no_end_to_end and world == 1:
            out["end_to_end"] = end_to_end(model, H, W)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.encoder, H, W)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel_substring):
    """HBM-side bytes per launch of the dominant kernel on the flagship workload, from the committed counter passes
    (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` over this same command, one pass each; FETCH_SIZE doubled for
    gfx950's 64-byte tally of 128-byte requests): counters cannot be collected from inside the timed process."""
    path = os.path.join(ROOT, "profiles", "r1_step4_vitl518_hbm_traffic.json")
    try:
        rows = json.load(open(path))
        row = next(r for r in rows if kernel_substring in r["kernel"])
    except (OSError, StopIteration, ValueError):
        return {}
    return {"traffic": round(row["hbm_bytes_per_launch"]), "traffic_unit": "bytes per launch (fetch x2 + write)",
            "traffic_source": "profiles/r1_step4_vitl518_hbm_traffic.json"}


def end_to_end(model, H, W, n=98):
    """``infer_video_depth`` on a synthetic uint8 video held in host memory (frames already at network size): upload,
    device preprocessing, all windows, alignment, cross-fade and the copy of every depth map back to the host.  Not part
    of `value`: it is the PCIe-inclusive rate of the whole video path, in VIDEO frames (a window advances 22 frames)."""
    frames = np.random.RandomState(1).randint(0, 256, (n, H, W, 3), dtype=np.uint8)
    model.infer_video_depth(frames[:33], 30, input_size=min(H, W), device="cuda", batch_size=32)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    depth, _ = model.infer_video_depth(frames, 30, input_size=min(H, W), device="cuda", batch_size=32)
    dt = time.perf_counter() - t0
    assert depth.shape == (n, H, W)
    from video_depth_anything.windows import window_plan
    windows = len(window_plan(n, 32)[0])
    return {"video_frames_per_s": round(n / dt, 2), "model_frames_per_s": round(32 * windows / dt, 2), "frames": n,
            "windows": windows, "seconds": round(dt, 3),
            "path": "host uint8 frames -> infer_video_depth(batch_size=32) -> host fp32 depth (PCIe both ways included)"}


def cpu_baseline(encoder, H, W):
    """The CPU oracle on a bounded sample: one short window (per-frame cost of the encoder and head does not depend on the
    window length; only the negligible temporal attention does), sized for roughly 15-25 s on 16 host threads."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vda_oracle as O
    from video_depth_anything import spec
    cfg = spec.MODEL_CONFIGS[encoder]
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    sd = {}
    shapes = spec.state_dict_spec(**cfg)
    g = torch.Generator().manual_seed(0)
    for k, (shape, kind) in shapes.items():       # throughput does not depend on the values: cheap random fill
        sd[k] = torch.randn(shape, generator=g) * 0.02
    frames = 12 if encoder == "vitl" else 32
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, frames, 3, H, W)).astype(np.float32))
    t0 = time.perf_counter()
    O.forward(sd, x, encoder)
    dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"CPU oracle (fp32 torch restatement), one {frames}-frame window at {H}x{W}, single run, {dt:.1f} s"}


if __name__ == "__main__":
    main()
