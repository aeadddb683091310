"""ctypes binding of zkvm_amd/libzkgpu.so (the C boundary declared in include/zkgpu.h).

The library is hand-written HIP for gfx950; there is no CPU fallback.  Loading succeeds without a GPU (so
that the symbols can be checked), but zkgpu_init fails with ZKGPU_ERR_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKGPU_LIB names another build of the library, for A/B measurements of kernel variants only: the override is announced
# on stderr when the library is loaded, bench.py records the path it measured, and the test suite and the smoke entry
# refuse to run under it.
DEFAULT_LIB_PATH = os.path.join(_HERE, "libzkgpu.so")
LIB_PATH = os.environ.get("ZKGPU_LIB") or DEFAULT_LIB_PATH
LIB_OVERRIDDEN = os.path.abspath(LIB_PATH) != os.path.abspath(DEFAULT_LIB_PATH)

ERRORS = {
    -1: "ZKGPU_ERR_INVALID_ARG", -2: "ZKGPU_ERR_HIP", -3: "ZKGPU_ERR_NO_DEVICE", -4: "ZKGPU_ERR_DOMAIN_SIZE",
    -5: "ZKGPU_ERR_DEGREE_ZERO", -6: "ZKGPU_ERR_DEGREE_LARGE", -7: "ZKGPU_ERR_NOT_LOADED", -8: "ZKGPU_ERR_OOM",
    -9: "ZKGPU_ERR_QUOTIENT_SHAPE", -10: "ZKGPU_ERR_INTERNAL",
}


class ZkGpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "ZKGPU_ERR"), code, message))
        self.code = code


class Domain(C.Structure):
    _fields_ = [("size", C.c_uint64), ("log_size_of_group", C.c_uint32), ("size_as_field_element", C.c_uint64 * 4),
                ("size_inv", C.c_uint64 * 4), ("group_gen", C.c_uint64 * 4), ("group_gen_inv", C.c_uint64 * 4),
                ("generator_inv", C.c_uint64 * 4)]


def build(force=False):
    """Compile libzkgpu.so for gfx950 with the Makefile in csrc/ (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"])
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); mirrors include/zkgpu.h one to one
_P = C.c_void_p
SIGNATURES = {
    "zkgpu_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "zkgpu_destroy": (None, [_P]),
    "zkgpu_last_error": (C.c_char_p, [_P]),
    "zkgpu_sync": (C.c_int, [_P]),
    "zkgpu_device_bytes": (C.c_size_t, [_P]),
    "zkgpu_timer_start": (C.c_int, [_P]),
    "zkgpu_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "zkgpu_domain_new": (C.c_int, [C.c_uint64, C.POINTER(Domain)]),
    "zkgpu_ntt": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.c_int]),
    "zkgpu_ntt_dev": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_uint32, C.c_int]),
    "zkgpu_ntt_dev_noalloc": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_uint32, C.c_int]),
    "zkgpu_srs_load": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "zkgpu_srs_load_dev": (C.c_int, [_P, _P, C.c_size_t]),
    "zkgpu_msm": (C.c_int, [_P, _P, C.c_size_t, _P, _P]),
    "zkgpu_msm_dev": (C.c_int, [_P, _P, C.c_size_t, _P, _P]),
    "zkgpu_msm_partial_dev": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P]),
    "zkgpu_g1_sum_to_affine": (C.c_int, [_P, C.c_size_t, _P, _P]),
    "zkgpu_msm_kernels_dev": (C.c_int, [_P, _P, C.c_size_t]),
    "zkgpu_commit": (C.c_int, [_P, _P, C.c_size_t, _P, _P]),
    "zkgpu_commit_many": (C.c_int, [_P, C.c_uint32, _P, _P, _P, _P]),
    "zkgpu_g1_compress": (C.c_int, [_P, C.c_uint8, _P]),
    "zkgpu_ntt_set_params": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "zkgpu_srs_setup": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "zkgpu_srs_export": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "zkgpu_srs_load_raw": (C.c_int, [_P, _P, C.c_size_t]),
    "zkgpu_srs_export_raw": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zkgpu_msm_get_params": (C.c_int, [_P, _P, _P]),
    "zkgpu_msm_profile_dev": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "zkgpu_msm_set_params": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "zkgpu_field_mul_selfcheck": (C.c_int, [C.c_int, _P, _P, C.c_size_t, _P]),
    "zkgpu_field_inv_selfcheck": (C.c_int, [C.c_int, _P, C.c_size_t, _P]),
    "zkgpu_g1_bucket_selfcheck": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P]),
    "zkgpu_g1_bucket_selfcheck_tree": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P]),
    "zkgpu_prover_set_sharding": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, _P, _P]),
    "zkgpu_set_collective": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "zkgpu_ntt_slice_dev": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32]),
    "zkgpu_intt_sharded_dev": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_int]),
    "zkgpu_prover_key_load": (C.c_int, [_P, C.c_uint32, C.c_uint64, _P, C.c_char_p, C.c_size_t, _P, C.c_size_t]),
    "zkgpu_prover_load_bytes": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "zkgpu_prover_to_bytes": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "zkgpu_prover_key_commitments": (C.c_int, [_P, _P, _P]),
    "zkgpu_prover_key_poly": (C.c_int, [_P, C.c_int, _P]),
    "zkgpu_prover_key_eval_len": (C.c_size_t, [_P]),
    "zkgpu_prover_key_eval8n": (C.c_int, [_P, C.c_int, _P]),
    "zkgpu_g1_decompress": (C.c_int, [_P, _P, _P]),
    "zkgpu_proof_to_bytes": (C.c_int, [_P, _P, _P, _P]),
    "zkgpu_proof_from_bytes": (C.c_int, [_P, _P, _P, _P]),
    "zkgpu_prover_force_full_coset": (C.c_int, [_P, C.c_int]),
    "zkgpu_prover_last_proof_coset": (C.c_int, [_P]),
    "zkgpu_quotient": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(C.c_size_t)]),
    "zkgpu_quotient_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "zkgpu_perm_z": (C.c_int, [_P, C.c_uint32, _P, _P, _P, _P, _P]),
    "zkgpu_poly_eval": (C.c_int, [_P, _P, C.c_size_t, _P, _P]),
    "zkgpu_poly_ruffini": (C.c_int, [_P, _P, C.c_size_t, _P, _P]),
    "zkgpu_linearization": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zkgpu_aggregate_witness": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P, C.POINTER(C.c_size_t)]),
    "zkgpu_prove": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P, _P, _P, _P]),
    "zkgpu_ctx_clone": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "zkgpu_device_bytes_shared": (C.c_size_t, [_P]),
    "zkgpu_witness_prefetch": (C.c_int, [_P, _P]),
    "zkgpu_host_alloc": (C.c_void_p, [C.c_size_t]),
    "zkgpu_host_free": (None, [_P]),
    "zkgpu_merkle_batch_circuit_size": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "zkgpu_merkle_batch_circuit": (C.c_int, [C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, _P, C.c_uint64, _P, _P, _P, _P, _P,
                                             C.c_uint32]),
    "zkgpu_merkle_batch_witness": (C.c_int, [C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, _P, C.c_uint64, _P, _P, C.c_uint32]),
    "zkgpu_poseidon_merkle4": (C.c_int, [_P, C.c_size_t, _P, _P, _P, _P]),
    "zkgpu_transcript_challenge": (C.c_int, [C.c_char_p, C.c_char_p, _P, C.c_size_t, C.c_char_p, _P, C.c_size_t]),
}


def lib():
    """The loaded library.  Raises if libzkgpu.so has not been built: the product never degrades to a CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("zkvm_amd/libzkgpu.so is missing: build it with __graft_entry__.build() or "
                               "`make -C zkvm_amd/csrc` (hipcc, gfx950)")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; a process must not run two HIP runtimes (the second
        # one finds no GPU).  Importing torch first makes the loader resolve this library's libamdhip64.so.7
        # dependency to the copy torch has already mapped, so device pointers are shared between the two.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        if LIB_OVERRIDDEN:
            import sys
            print("zkvm_amd: ZKGPU_LIB is set, using %s instead of %s" % (LIB_PATH, DEFAULT_LIB_PATH), file=sys.stderr, flush=True)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if not LIB_OVERRIDDEN:
                    raise
                continue   # an older build under comparison lacks the newer entry points
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
