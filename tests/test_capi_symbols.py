"""CPU-side checks of the C boundary: the library loads and exports every symbol include/zkgpu.h declares
(no compute calls: those need a GPU), and the host-only entry points agree with the oracle."""
import os
import re

import numpy as np
import pytest

import model as M
from conftest import GOLDEN, ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "zkgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from zkvm_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libzkgpu.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), name
    # the Python binding covers the same set
    assert sorted(_lib.SIGNATURES) == names


def test_rccl_library_exports_every_declared_symbol():
    """libzkgpu_rccl.so, the optional RCCL implementation of the exchange callbacks (include/zkgpu_rccl.h)."""
    import ctypes
    path = os.path.join(ROOT, "zkvm_amd", "libzkgpu_rccl.so")
    assert os.path.exists(path), "build libzkgpu_rccl.so first (__graft_entry__.build())"
    text = open(os.path.join(ROOT, "include", "zkgpu_rccl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(zkgpu_rccl_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 7
    lib = ctypes.CDLL(path)
    for name in names:
        assert hasattr(lib, name), name
    # without a GPU the identifier cannot be created: the call must return an error, not crash
    import torch
    if not torch.cuda.is_available():
        buf = (ctypes.c_uint8 * 128)()
        lib.zkgpu_rccl_init.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        out = ctypes.c_void_p()
        assert lib.zkgpu_rccl_init(ctypes.byref(out), buf, 3, 2, 0) != 0   # rank outside the world


def test_init_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import zkvm_amd
    with pytest.raises(zkvm_amd.ZkGpuError) as e:
        zkvm_amd.Context(0)
    assert e.value.code == -3


def test_domain_new_matches_oracle(orc):
    import zkvm_amd
    for num in (0, 1, 2, 3, 5, 1 << 10, (1 << 20) - 6, 1 << 23, 1 << 27, (1 << 31)):
        d = zkvm_amd.EvaluationDomain(num)
        o = orc.domain_new(num)
        assert d.size == o["size"] and d.log_size_of_group == o["log_size_of_group"]
        for name in ("size_as_field_element", "size_inv", "group_gen", "group_gen_inv", "generator_inv"):
            assert np.array_equal(getattr(d, name), o[name]), (num, name)
    with pytest.raises(zkvm_amd.InvalidEvalDomainSize):
        zkvm_amd.EvaluationDomain((1 << 31) + 1)


def test_host_compress_and_sum(orc):
    import zkvm_amd
    data = open(os.path.join(GOLDEN, "g1_compressed_valid_test_vectors.dat"), "rb").read()
    xy, inf = orc.g1_multiples_of_generator(0, 40)
    for k in range(40):
        assert zkvm_amd.g1_compress(xy[k], inf[k]) == data[48 * k:48 * k + 48]
    g = orc.g1_generator()
    pts = np.stack([orc.g1_mul(g, M.fr_to_mont(k)) for k in (3, 10, 0, 25)])
    sxy, sinf = zkvm_amd.g1_sum_to_affine(pts)
    assert sinf == 0 and zkvm_amd.g1_compress(sxy, sinf) == data[48 * 38:48 * 39]
    sxy, sinf = zkvm_amd.g1_sum_to_affine(np.zeros((0, 18), np.uint64))
    assert sinf == 1 and zkvm_amd.g1_compress(sxy, sinf) == data[:48]


def test_rust_bindings_declare_every_entry_point():
    """rust/zkgpu-sys mirrors include/zkgpu.h one to one (it cannot be compiled here, so at least keep it complete)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "zkgpu.h")).read()
    rust = open(os.path.join(root, "rust", "zkgpu-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"\b(zkgpu_[a-z0-9_]+)\s*\(", header))
    bound = set(re.findall(r"pub fn (zkgpu_[a-z0-9_]+)", rust))
    assert declared == bound


def test_header_is_valid_c_and_cxx(tmp_path):
    """include/zkgpu.h compiles as C99 and as C++17 on its own (what a cgo / FFI binding generator consumes)."""
    import os
    import subprocess
    from conftest import ROOT
    inc = os.path.join(ROOT, "include")
    c_src = tmp_path / "use.c"
    c_src.write_text('#include "zkgpu.h"\nint main(void) { (void)&zkgpu_domain_new; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(c_src)])
    cxx_src = tmp_path / "use.cpp"
    cxx_src.write_text('#include "zkgpu.h"\nint main() { (void)&zkgpu_domain_new; return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(cxx_src)])
