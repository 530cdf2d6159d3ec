import os
import sys

import pytest

# The CPU oracle's OpenMP regions are tiny at test sizes; on a 256-thread host a 256-wide team makes every region cost
# milliseconds (fork/join + per-thread tile allocation), which turned a 20 s test into > 10 minutes.  Must be set before
# libgomp loads (i.e. before tests/oracle_lib.py dlopens the oracle).
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

# the suite drives the test-support entry points (virtual communicator, debug options, kernel variants): the testing flavour of the
# library (cup3d_amd/capi.py); tests/test_gpu_release_flavour.py runs the release build in subprocesses
os.environ.setdefault("CUP3D_HIP_FLAVOUR", "testing")

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CUP3D_TEST_TORCH_FIRST"):   # bisect of round 5's exit-time abort (scripts/gpu_round6.sh exitsubset): torch's ROCm libraries before ours
    import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout; ignored when the plugin is absent)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- a GPU session that is cut off from outside must still say what happened (round 4's one failing whole-suite run left two lines of
# dots): with CUP3D_LIVE_LOG=<file> every test's start and outcome is appended to <file> AS IT HAPPENS (line-buffered, fsync'ed), and a
# failure's full traceback -- pytest's own report only prints them when the session ends -- goes there at once.
_LIVE = os.environ.get("CUP3D_LIVE_LOG")


def _live(text):
    if not _LIVE:
        return
    import time
    with open(_LIVE, "a") as f:
        f.write(time.strftime("%H:%M:%S ") + text + "\n")
        f.flush()
        os.fsync(f.fileno())


def pytest_runtest_logstart(nodeid, location):
    _live(f"START {nodeid}")


def pytest_runtest_logreport(report):
    if report.when == "call" or report.outcome != "passed":
        _live(f"{report.outcome.upper():7s} {report.nodeid} [{report.when}] {getattr(report, 'duration', 0.0):.1f} s")
    if report.failed:
        _live("TRACEBACK of " + report.nodeid + "\n" + report.longreprtext + "\n" + "\n".join(f"--- captured {k} ---\n{v}" for k, v in report.sections))
