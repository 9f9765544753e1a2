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
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    hip.groupnorm_static(x, got, w, b, frames, hw, C, 32, 1e-6)
    assert torch.cuda.memory_allocated() == before
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want)


def test_groupnorm_static_matches_workspace_variant_twice():
    run_both(32, 37 * 37, 1024, torch.float16, 1)        # 32 frames x 64 splits of partial sums
    run_both(3, 19 * 19, 256, torch.bfloat16, 2)          # smaller: fewer frames and splits; stale sums beyond them are never read
    run_both(64, 5 * 7, 64, torch.float32, 3)             # the maximum frame count


def test_groupnorm_static_rejects_more_frames_than_its_scratch_holds():
    x = torch.zeros((65, 4, 64), dtype=torch.float16, device=DEV)
    w = torch.ones(64, device=DEV)
    with pytest.raises(RuntimeError, match="-2"):
        hip.groupnorm_static(x, torch.empty_like(x), w, w, 65, 4, 64, 32, 1e-6)
