"""The allocation-free GroupNorm entry point (vda_groupnorm_static: partial sums in a device array of the library itself)
against vda_groupnorm with a caller-provided workspace: identical bits, on two calls in a row, the second one smaller."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from video_depth_anything import hip  # noqa: E402

DEV = "cuda"


def run_both(frames, hw, C, dt, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn((frames, hw, C), generator=g) * 2 + 0.3).to(dt).to(DEV)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV)
    work = torch.empty((frames * hip.GROUPNORM_MAX_SPLITS * 32 * 2,), dtype=torch.float64, device=DEV)
    want = torch.full_like(x, float("nan"))
    hip.groupnorm(x, want, w, b, work, frames, hw, C, 32, 1e-6)
    got = torch.full_like(x, float("nan"))
    # the first call on a device may load the code object and looks the scratch array's address up; from then on the free
    # memory the DRIVER reports (not torch's allocator, which would not see a hipMalloc inside the library) must not move
    hip.groupnorm_static(x, torch.empty_like(x), w, b, frames, hw, C, 32, 1e-6)
    torch.cuda.synchronize()
    before, before_torch = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
    hip.groupnorm_static(x, got, w, b, frames, hw, C, 32, 1e-6)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == before and torch.cuda.memory_allocated() == before_torch
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want)


def test_groupnorm_static_matches_workspace_variant_twice():
    run_both(32, 37 * 37, 1024, torch.float16, 1)        # 32 frames x 64 splits of partial sums
    run_both(3, 19 * 19, 256, torch.bfloat16, 2)          # smaller: fewer frames and splits; stale sums beyond them are never read
    run_both(64, 5 * 7, 64, torch.float32, 3)             # the maximum frame count


def test_groupnorm_static_under_graph_capture():
    """The callers it is for: captured into a HIP graph (no allocation is possible there) and replayed on new input.  The first
    call on a device happens outside the capture (it resolves the scratch array's address, which may load the code object)."""
    g = torch.Generator().manual_seed(7)
    frames, hw, C = 4, 37 * 37, 256
    x = (torch.randn((frames, hw, C), generator=g) * 2).half().to(DEV)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV)
    out = torch.zeros_like(x)
    hip.groupnorm_static(x, out, w, b, frames, hw, C, 32, 1e-6)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        hip.groupnorm_static(x, out, w, b, frames, hw, C, 32, 1e-6)
    x.copy_((torch.randn((frames, hw, C), generator=g) * 3 + 1).half())
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    work = torch.empty((frames * hip.GROUPNORM_MAX_SPLITS * 32 * 2,), dtype=torch.float64, device=DEV)
    want = torch.empty_like(x)
    hip.groupnorm(x, want, w, b, work, frames, hw, C, 32, 1e-6)
    assert torch.equal(out, want)


def test_groupnorm_static_rejects_more_frames_than_its_scratch_holds():
    x = torch.zeros((65, 4, 64), dtype=torch.float16, device=DEV)
    w = torch.ones(64, device=DEV)
    with pytest.raises(RuntimeError, match="-2"):
        hip.groupnorm_static(x, torch.empty_like(x), w, w, 65, 4, 64, 32, 1e-6)
