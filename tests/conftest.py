import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _poison_uninitialised_tensors():
    """NVSR_POISON_EMPTY=1 (a diagnostic run of the suite, round 5): every floating-point tensor that torch.empty / empty_like / new_empty hands out
    is filled with NaN, so a kernel that reads memory nobody wrote -- or leaves part of an output it promises unwritten -- is loud.  In an ordinary
    run the caching allocator hands such a kernel the previous iteration's block: stale values of the right shape and magnitude, which is how the
    refine workload's once-per-process weight packing stayed invisible for most of a round."""
    import torch

    def wrap(fn):
        def poisoned(*a, **k):
            t = fn(*a, **k)
            if t.is_floating_point() and t.numel():
                t.fill_(float("nan"))
            return t
        return poisoned

    torch.empty = wrap(torch.empty)
    torch.empty_like = wrap(torch.empty_like)
    torch.Tensor.new_empty = wrap(torch.Tensor.new_empty)


if os.environ.get("NVSR_POISON_EMPTY") == "1":
    _poison_uninitialised_tensors()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def hip():
    """The HIP product path (C-ABI library through the host-side mirror).  Fails loudly if the .so is missing."""
    import torch

    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    import nvsr_amd

    nvsr_amd.capi.lib()  # raises if the extension is not built
    return nvsr_amd


def sample_pdf_tolerance(bins, weights, u, eps=2e-6, floor=3e-6, w_noise=0.0):
    """Per-sample tolerance for inverse-CDF sampling (nerf_helpers.py:668-702).  A sample is
    b0 + (u - c0) / (c1 - c0) * (b1 - b0): an fp32 cdf of ~64 terms carries up to ~64*ulp(1)/2 = eps of absolute rounding noise (order of the
    cumsum), which the division by a small bin mass (c1 - c0) amplifies.  tol = floor + 2*eps*(b1-b0)/denom,
    or the whole bin width where the bin mass sits on the reference's 1e-5 threshold."""
    w = weights.astype(np.float64) + 1e-5
    pdf = w / w.sum(-1, keepdims=True)
    # w_noise: absolute noise of the input weights themselves (when they come out of an fp32 MLP + compositing);
    # the pdf normalisation divides it by the ray's total weight
    eps = eps + w_noise / w.sum(-1)
    cdf = np.concatenate([np.zeros_like(pdf[:, :1]), np.cumsum(pdf, -1)], -1)
    tol = np.empty(u.shape, np.float64)
    for i in range(u.shape[0]):
        idx = np.searchsorted(cdf[i], u[i].astype(np.float64), side="right")
        tol_i = np.zeros(u.shape[1])
        for shift in (-1, 0, 1):          # a rounding-level change of u vs cdf may move the sample to a neighbour bin
            ii = np.clip(idx + shift, 0, cdf.shape[1])
            below = np.maximum(ii - 1, 0)
            above = np.minimum(ii, cdf.shape[1] - 1)
            denom = cdf[i, above] - cdf[i, below]
            width = np.abs(bins[i, above] - bins[i, below])
            # `denom < 1e-5 -> 1` (nerf_helpers.py:698) is a discontinuity: an empty bin of an opaque ray has mass
            # 1e-5/(1+63e-5), i.e. within fp32 noise of the threshold, and may land anywhere in its bin
            knife = np.abs(denom - 1e-5) < 4 * eps[i]
            denom = np.where(denom < 1e-5, 1.0, denom)
            tol_i = np.maximum(tol_i, np.where(knife, width, 2 * eps[i] * width / denom))
            # u sitting on a cdf knot (always the case for u = 1.0 in det mode): searchsorted may pick either side, and if
            # a bin next to the knot is below the 1e-5 threshold the sample collapses to that bin's start instead
            ui = u[i].astype(np.float64)
            on_knot = (np.abs(ui - cdf[i, below]) < 4 * eps[i]) | (np.abs(ui - cdf[i, above]) < 4 * eps[i])
            lo_n = np.maximum(below - 1, 0)
            hi_n = np.minimum(above + 1, cdf.shape[1] - 1)
            span = np.abs(bins[i, hi_n] - bins[i, lo_n])
            tol_i = np.maximum(tol_i, np.where(on_knot, span, 0.0))
        tol[i] = floor + tol_i
    return tol
