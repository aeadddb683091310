"""The C boundary without Python in the process: tests/c_client/client.c is built by __graft_entry__.build() against
include/zkgpu.h and libzkgpu.so alone and run here as a child process."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client_runs_on_the_gpu():
    exe = os.path.join(ROOT, "tests", "c_client", "client")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c client ok" in out.stdout


def test_rccl_exchange_library_with_one_rank():
    """libzkgpu_rccl.so (include/zkgpu_rccl.h): the partial-sum all-gather behind a shared commitment and both
    collectives on device buffers, through RCCL with a communicator of one GPU."""
    exe = os.path.join(ROOT, "tests", "c_client", "rccl_client")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rccl client ok" in out.stdout
