import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def gemm_knobs():
    """gemm_knobs(STDIFF_GEMM_TILE="3", ...): set kernel-selection knobs for this test.  The library reads its
    environment once per process, so the fixture calls stdiff_reload_env after setting and after restoring them."""
    from stdiff_amd import _lib
    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            saved.setdefault(k, os.environ.get(k))
            os.environ[k] = str(v)
        _lib.reload_env()
    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    _lib.reload_env()


@pytest.fixture
def progress(capsys):
    """progress("...") for tests that run for minutes: the line goes to the real stdout even under capture, and is
    appended to gpurun_out/progress.log -- the GPU box's harness ends a command that writes nothing for three minutes."""
    import time
    t0 = time.time()

    def say(msg):
        line = f"[{time.time() - t0:6.1f} s] {msg}"
        with capsys.disabled():
            print(line, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "progress.log"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass
    return say


def elementwise(a, b, rtol):
    """Element-wise view of a tolerance next to the max-norm figure the tests assert (north_star states its tolerances
    as rtol): for got `a` and reference `b`, the fraction of elements with |a - b| <= rtol * |b| + atol for
      atol = rtol * max|b|   ("full": what the max-norm criterion  max|a-b| <= rtol * max|b|  implies for every element),
      atol = 0.1 * rtol * max|b|   ("tenth": ten times stricter on the elements near zero),
      atol = 0   ("pure": numpy's rtol alone; elements near zero cannot meet it in any finite-precision path),
    and the median of |a - b| / |b|."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    d, scale = np.abs(a - b), float(np.abs(b).max())
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = d / np.abs(b)
    return {"full": float((d <= rtol * np.abs(b) + rtol * scale).mean()),
            "tenth": float((d <= rtol * np.abs(b) + 0.1 * rtol * scale).mean()),
            "pure": float((d <= rtol * np.abs(b)).mean()),
            "median_rel": float(np.nanmedian(rel)), "max_over_scale": float(d.max() / max(scale, 1e-300))}


def elementwise_line(tag, a, b, rtol):
    e = elementwise(a, b, rtol)
    return e, (f"{tag}: rtol {rtol:g}: elements within rtol|b| + atol: {100 * e['full']:.3f} % (atol = rtol x scale), "
               f"{100 * e['tenth']:.3f} % (atol = 0.1 rtol x scale), {100 * e['pure']:.2f} % (atol = 0); median |a-b|/|b| "
               f"{e['median_rel']:.2e}; max|a-b| / scale {e['max_over_scale']:.2e}")
