import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The suite checks THIS tree's libzkgpu.so: a stale ZKGPU_LIB (the A/B switch of zkvm_amd/_lib.py) would silently put
    # another build under test.
    if os.environ.get("ZKGPU_LIB") and os.environ.get("ZKGPU_ALLOW_LIB_OVERRIDE") != "1":
        raise pytest.UsageError("ZKGPU_LIB is set: unset it, or set ZKGPU_ALLOW_LIB_OVERRIDE=1 to test that library on purpose")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liboracle.so through ctypes); built on first use."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session", autouse=True)
def heartbeat():
    """Some GPU tests wait minutes for the CPU oracle (a 2^20-gate proof on the host cores) without printing anything.
    A line every 20 seconds in gpurun_out/pytest_heartbeat.log shows a watching harness that the run is alive."""
    import threading
    import time
    out_dir = os.path.join(ROOT, "gpurun_out")
    stop = threading.Event()
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "pytest_heartbeat.log")
        open(path, "a").close()
    except OSError:
        yield
        return

    def beat():
        while not stop.wait(20.0):
            try:
                with open(path, "a") as f:
                    f.write("%d %s\n" % (time.time(), os.environ.get("PYTEST_CURRENT_TEST", "")))
            except OSError:
                return

    t = threading.Thread(target=beat, daemon=True)
    t.start()
    yield
    stop.set()
