"""Host-side mirror of the reference's interface for the accelerated path, over the C boundary.

Names and behaviour follow the `plonk` crate: EvaluationDomain (plonk/src/fft/domain.rs), CommitKey
(plonk/src/commitment_scheme/kzg10/key.rs).  Field elements are numpy uint64 arrays of shape (n, 4)
(Fr, Montgomery limbs) and (n, 12) (affine G1 points) on the host, or torch int64 CUDA tensors of the same
shapes when the data is already resident in HBM.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ZkGpuError

FFT, IFFT, COSET_FFT, COSET_IFFT = 0, 1, 2, 3


class InvalidEvalDomainSize(ValueError):
    """plonk Error::InvalidEvalDomainSize"""


class PolynomialDegreeIsZero(ValueError):
    """plonk Error::PolynomialDegreeIsZero"""


class PolynomialDegreeTooLarge(ValueError):
    """plonk Error::PolynomialDegreeTooLarge"""


_ERR_CLASSES = {-4: InvalidEvalDomainSize, -5: PolynomialDegreeIsZero, -6: PolynomialDegreeTooLarge}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _np_u64(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, width)


class Context:
    """One GPU, one HIP stream, resident twiddle tables and SRS (zkgpu_ctx)."""

    def __init__(self, device=0):
        self._lib = _lib.lib()
        h = C.c_void_p()
        rc = self._lib.zkgpu_init(C.byref(h), int(device))
        if rc != 0:
            raise ZkGpuError(rc, self._lib.zkgpu_last_error(None).decode())
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zkgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.zkgpu_last_error(self._h).decode()
            cls = _ERR_CLASSES.get(rc)
            if cls is not None:
                raise cls(msg)
            raise ZkGpuError(rc, msg)

    def sync(self):
        self._check(self._lib.zkgpu_sync(self._h))

    def device_bytes(self):
        return int(self._lib.zkgpu_device_bytes(self._h))

    def timer_start(self):
        self._check(self._lib.zkgpu_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        self._check(self._lib.zkgpu_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    # ---- NTT --------------------------------------------------------------------------------------------
    def ntt(self, data, log_n, mode, out=None):
        """Transform `data` (numpy (len, 4) on the host, or a torch CUDA int64 tensor) over the 2^log_n domain.
        Host input returns a new numpy array; device input writes `out` (default: a new tensor) and returns it."""
        n = 1 << log_n
        if _is_torch(data):
            import torch
            assert data.is_cuda and data.dtype == torch.int64 and data.is_contiguous()
            len_in = data.numel() // 4
            if out is None:
                out = torch.empty((n, 4), dtype=torch.int64, device=data.device)
            assert out.numel() == 4 * n and out.is_contiguous()
            self._check(self._lib.zkgpu_ntt_dev(self._h, data.data_ptr(), len_in, out.data_ptr(), log_n, mode))
            return out
        a = _np_u64(data, 4)
        len_in = a.shape[0]
        buf = np.zeros((n, 4), dtype=np.uint64)
        m = min(len_in, n)
        buf[:m] = a[:m]
        self._check(self._lib.zkgpu_ntt(self._h, buf.ctypes.data, len_in if len_in <= n else n, log_n, mode))
        return buf

    # ---- SRS / MSM --------------------------------------------------------------------------------------
    def srs_load(self, xy, infinity=None):
        if _is_torch(xy):
            self._check(self._lib.zkgpu_srs_load_dev(self._h, xy.data_ptr(), xy.numel() // 12))
            return
        pts = _np_u64(xy, 12)
        inf = None
        if infinity is not None:
            inf = np.ascontiguousarray(infinity, dtype=np.uint8)
        self._check(self._lib.zkgpu_srs_load(self._h, pts.ctypes.data, inf.ctypes.data if inf is not None else None,
                                             pts.shape[0]))

    def msm(self, scalars):
        """-> (xy limbs (12,), infinity flag): affine sum_i scalars[i] * SRS[i]."""
        out = np.zeros(12, dtype=np.uint64)
        inf = C.c_uint8(0)
        if _is_torch(scalars):
            self._check(self._lib.zkgpu_msm_dev(self._h, scalars.data_ptr(), scalars.numel() // 4, out.ctypes.data,
                                                C.addressof(inf)))
        else:
            s = _np_u64(scalars, 4)
            self._check(self._lib.zkgpu_msm(self._h, s.ctypes.data, s.shape[0], out.ctypes.data, C.addressof(inf)))
        return out, int(inf.value)

    def msm_partial(self, scalars_dev, base=0):
        """Projective partial sum over the point range [base, base + len) -> (18,) limbs."""
        out = np.zeros(18, dtype=np.uint64)
        self._check(self._lib.zkgpu_msm_partial_dev(self._h, scalars_dev.data_ptr(), scalars_dev.numel() // 4, base,
                                                    out.ctypes.data))
        return out

    def msm_kernels(self, scalars_dev):
        self._check(self._lib.zkgpu_msm_kernels_dev(self._h, scalars_dev.data_ptr(), scalars_dev.numel() // 4))

    def commit(self, coeffs):
        c = _np_u64(coeffs, 4)
        out = np.zeros(12, dtype=np.uint64)
        inf = C.c_uint8(0)
        self._check(self._lib.zkgpu_commit(self._h, c.ctypes.data, c.shape[0], out.ctypes.data, C.addressof(inf)))
        return out, int(inf.value)


def g1_sum_to_affine(xyz):
    pts = _np_u64(xyz, 18)
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    rc = _lib.lib().zkgpu_g1_sum_to_affine(pts.ctypes.data, pts.shape[0], out.ctypes.data, C.addressof(inf))
    if rc != 0:
        raise ZkGpuError(rc, "g1_sum_to_affine")
    return out, int(inf.value)


def g1_compress(xy, infinity=0):
    p = _np_u64(xy, 12)
    out = np.zeros(48, dtype=np.uint8)
    rc = _lib.lib().zkgpu_g1_compress(p.ctypes.data, int(infinity), out.ctypes.data)
    if rc != 0:
        raise ZkGpuError(rc, "g1_compress")
    return bytes(out)


class EvaluationDomain:
    """EvaluationDomain::new(num_coeffs) and its four transforms (plonk/src/fft/domain.rs:106-196)."""

    def __init__(self, num_coeffs, ctx=None):
        d = _lib.Domain()
        rc = _lib.lib().zkgpu_domain_new(int(num_coeffs), C.byref(d))
        if rc == -4:
            raise InvalidEvalDomainSize("log2(size) must be below 32")
        if rc != 0:
            raise ZkGpuError(rc, "domain_new")
        self.size = int(d.size)
        self.log_size_of_group = int(d.log_size_of_group)
        for name in ("size_as_field_element", "size_inv", "group_gen", "group_gen_inv", "generator_inv"):
            setattr(self, name, np.array(list(getattr(d, name)), dtype=np.uint64))
        self.ctx = ctx

    def fft(self, coeffs, out=None):
        return self.ctx.ntt(coeffs, self.log_size_of_group, FFT, out)

    def ifft(self, evals, out=None):
        return self.ctx.ntt(evals, self.log_size_of_group, IFFT, out)

    def coset_fft(self, coeffs, out=None):
        return self.ctx.ntt(coeffs, self.log_size_of_group, COSET_FFT, out)

    def coset_ifft(self, evals, out=None):
        return self.ctx.ntt(evals, self.log_size_of_group, COSET_IFFT, out)


class CommitKey:
    """CommitKey { powers_of_g } with commit() (plonk/src/commitment_scheme/kzg10/key.rs:32-35,108-129)."""

    def __init__(self, ctx, powers_of_g, infinity=None):
        self.ctx = ctx
        self.len = (powers_of_g.numel() if _is_torch(powers_of_g) else np.asarray(powers_of_g).size) // 12
        ctx.srs_load(powers_of_g, infinity)

    def max_degree(self):
        return self.len - 1

    def commit(self, polynomial):
        """-> (xy, infinity).  Raises PolynomialDegreeIsZero / PolynomialDegreeTooLarge like the reference."""
        return self.ctx.commit(polynomial)
