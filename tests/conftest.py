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
