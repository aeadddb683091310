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
# phases of zkgpu_msm_profile_dev, in order
MSM_PHASE_NAMES = ("digits", "range_partition", "count_pass", "offsets", "scatter_pass", "schedule", "accumulate",
                   "oversized", "reduce")


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

    def clone(self):
        """zkgpu_ctx_clone: another context on the same GPU (own stream and buffers) that shares this one's SRS table
        and prover key by reference; Prover.on(clone) gives the prover object for it."""
        h = C.c_void_p()
        self._check(self._lib.zkgpu_ctx_clone(self._h, C.byref(h)))
        other = Context.__new__(Context)
        other._lib, other._h, other.device = self._lib, h, self.device
        return other

    def device_bytes_shared(self):
        return int(self._lib.zkgpu_device_bytes_shared(self._h))

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

    def force_full_coset(self, on=True):
        """Keys loaded from now on evaluate the quotient over the reference's 8n coset whatever their gates
        (zkgpu_prover_force_full_coset)."""
        self._check(self._lib.zkgpu_prover_force_full_coset(self._h, 1 if on else 0))

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

    def ntt_noalloc(self, data, log_n, mode, out):
        """zkgpu_ntt_dev_noalloc: the transform on device tensors without any allocation inside the library (its
        inter-pass buffer is a static device array, 2^20 elements at most; the tables must be resident)."""
        n = 1 << log_n
        assert data.is_cuda and data.is_contiguous() and out.is_cuda and out.is_contiguous() and out.numel() == 4 * n
        self._check(self._lib.zkgpu_ntt_dev_noalloc(self._h, data.data_ptr(), data.numel() // 4, out.data_ptr(), log_n, mode))
        return out

    def ntt_slice(self, data, log_n, mode, rank, world, out=None):
        """zkgpu_ntt_slice_dev: elements rank + world * k of fft / coset_fft of `data` (a CUDA tensor) over the 2^log_n
        domain, with one local transform of 2^log_n / world points."""
        import torch
        m = (1 << log_n) // world
        if out is None:
            out = torch.empty((m, 4), dtype=torch.int64, device=data.device)
        assert data.is_cuda and data.is_contiguous() and out.is_contiguous() and out.numel() == 4 * m
        self._check(self._lib.zkgpu_ntt_slice_dev(self._h, data.data_ptr(), data.numel() // 4, out.data_ptr(), log_n, mode,
                                                  rank, world))
        return out

    def intt_sharded(self, slice_dev, log_n, mode, out=None):
        """zkgpu_intt_sharded_dev: ifft / coset_ifft of evaluations held as slices by the ranks registered with
        sharding.enable_collectives -> all 2^log_n coefficients.  Collective: every rank calls it."""
        import torch
        n = 1 << log_n
        if out is None:
            out = torch.empty((n, 4), dtype=torch.int64, device=slice_dev.device)
        assert slice_dev.is_cuda and slice_dev.is_contiguous() and out.is_contiguous() and out.numel() == 4 * n
        self._check(self._lib.zkgpu_intt_sharded_dev(self._h, slice_dev.data_ptr(), out.data_ptr(), log_n, mode))
        return out

    def ntt_set_params(self, tile_log=0, elems_per_thread_log=0):
        """Tuning knobs of the transform (0 = default)."""
        self._check(self._lib.zkgpu_ntt_set_params(self._h, int(tile_log), int(elems_per_thread_log)))

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

    def srs_setup(self, tau, g_xy, n_points):
        """PublicParameters::setup on the device: SRS[i] = tau^i * g."""
        t, g = _np_u64(tau, 4), _np_u64(g_xy, 12)
        self._check(self._lib.zkgpu_srs_setup(self._h, t.ctypes.data, g.ctypes.data, int(n_points)))

    def srs_export(self, n):
        xy = np.zeros((n, 12), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        self._check(self._lib.zkgpu_srs_export(self._h, xy.ctypes.data if n else None, inf.ctypes.data if n else None, n))
        return xy, inf

    def srs_load_raw(self, data):
        """CommitKey::from_slice_unchecked: u64 count + 97 bytes per point."""
        buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self._check(self._lib.zkgpu_srs_load_raw(self._h, buf.ctypes.data, buf.size))

    def srs_export_raw(self):
        """CommitKey::to_raw_var_bytes of the resident SRS."""
        need = C.c_size_t(0)
        self._check(self._lib.zkgpu_srs_export_raw(self._h, None, 0, C.byref(need)))
        out = np.zeros(need.value, dtype=np.uint8)
        self._check(self._lib.zkgpu_srs_export_raw(self._h, out.ctypes.data, out.size, C.byref(need)))
        return bytes(out)

    def msm_set_params(self, window_bits=0, segment_bits=0):
        """Tuning knobs (0 = automatic); window_bits applies to the next srs_load."""
        self._check(self._lib.zkgpu_msm_set_params(self._h, int(window_bits), int(segment_bits)))

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

    def msm_get_params(self):
        """-> (window bits, table levels = digits per scalar) of the loaded SRS."""
        c, k = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.zkgpu_msm_get_params(self._h, C.byref(c), C.byref(k)))
        return int(c.value), int(k.value)

    def msm_profile(self, scalars_dev):
        """Phase times of one device MSM (HIP events on the library's stream), milliseconds."""
        ms = (C.c_float * len(MSM_PHASE_NAMES))()
        self._check(self._lib.zkgpu_msm_profile_dev(self._h, scalars_dev.data_ptr(), scalars_dev.numel() // 4, ms))
        out = dict(zip(MSM_PHASE_NAMES, [float(x) for x in ms]))
        out["sort"] = sum(out[k] for k in MSM_PHASE_NAMES[:6])     # everything before the accumulate launch
        return out

    def quotient_kernel_ms(self):
        """Device milliseconds of the quotient kernel in the most recent quotient / proof on this context."""
        ms = C.c_float(0)
        self._check(self._lib.zkgpu_quotient_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def commit(self, coeffs):
        c = _np_u64(coeffs, 4)
        out = np.zeros(12, dtype=np.uint64)
        inf = C.c_uint8(0)
        self._check(self._lib.zkgpu_commit(self._h, c.ctypes.data, c.shape[0], out.ctypes.data, C.addressof(inf)))
        return out, int(inf.value)


def _commit_many(self, polys):
    """Up to four KZG commitments in one batched MSM -> [(xy limbs, infinity flag), ...] (zkgpu_commit_many)."""
    arrs = [_np_u64(p, 4) for p in polys]
    m = len(arrs)
    ptrs = (C.c_void_p * m)(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * m)(*[a.shape[0] for a in arrs])
    out = np.zeros((m, 12), dtype=np.uint64)
    inf = np.zeros(m, dtype=np.uint8)
    self._check(self._lib.zkgpu_commit_many(self._h, m, ptrs, lens, out.ctypes.data, inf.ctypes.data))
    return [(out[j], int(inf[j])) for j in range(m)]


Context.commit_many = _commit_many


def g1_sum_to_affine(xyz):
    pts = _np_u64(xyz, 18)
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    rc = _lib.lib().zkgpu_g1_sum_to_affine(pts.ctypes.data, pts.shape[0], out.ctypes.data, C.addressof(inf))
    if rc != 0:
        raise ZkGpuError(rc, "g1_sum_to_affine")
    return out, int(inf.value)


def field_mul_selfcheck(field, a, b):
    """Host-side run of the kernels' Montgomery product: field 0 = Fr (n, 4), 1 = Fq (n, 6)."""
    width = 4 if field == 0 else 6
    x, y = _np_u64(a, width), _np_u64(b, width)
    out = np.zeros_like(x)
    rc = _lib.lib().zkgpu_field_mul_selfcheck(int(field), x.ctypes.data, y.ctypes.data, x.shape[0], out.ctypes.data)
    if rc != 0:
        raise ZkGpuError(rc, "field_mul_selfcheck")
    return out


def field_inv_selfcheck(field, a):
    """Host-side run of the library's field inversion (Montgomery form in and out): field 0 = Fr (n, 4), 1 = Fq (n, 6)."""
    width = 4 if field == 0 else 6
    x = _np_u64(a, width)
    out = np.zeros_like(x)
    rc = _lib.lib().zkgpu_field_inv_selfcheck(int(field), x.ctypes.data, x.shape[0], out.ctypes.data)
    if rc != 0:
        raise ZkGpuError(rc, "field_inv_selfcheck")
    return out


def g1_bucket_selfcheck(xy, negate=None, infinity=None, tree=False):
    """Host-side run of the MSM kernels' bucket arithmetic -> (projective sum (18,), bound violations); `tree` sums
    accumulators pairwise as the bucket reduction does instead of adding the points one by one."""
    pts = _np_u64(xy, 12)
    n = pts.shape[0]
    neg = np.ascontiguousarray(negate if negate is not None else np.zeros(n), dtype=np.uint8)
    inf = np.ascontiguousarray(infinity if infinity is not None else np.zeros(n), dtype=np.uint8)
    out = np.zeros(18, dtype=np.uint64)
    viol = C.c_uint64(0)
    fn = _lib.lib().zkgpu_g1_bucket_selfcheck_tree if tree else _lib.lib().zkgpu_g1_bucket_selfcheck
    rc = fn(pts.ctypes.data if n else None, inf.ctypes.data if n else None,
                                              neg.ctypes.data if n else None, n, out.ctypes.data, C.addressof(viol))
    if rc != 0:
        raise ZkGpuError(rc, "g1_bucket_selfcheck")
    return out, int(viol.value)


def g1_compress(xy, infinity=0):
    p = _np_u64(xy, 12)
    out = np.zeros(48, dtype=np.uint8)
    rc = _lib.lib().zkgpu_g1_compress(p.ctypes.data, int(infinity), out.ctypes.data)
    if rc != 0:
        raise ZkGpuError(rc, "g1_compress")
    return bytes(out)


def g1_decompress(data):
    """G1Affine::from_bytes (coset-bls12_381/src/g1/coset.rs:73-140) -> (xy limbs (12,), infinity flag); ValueError
    where the reference returns Err."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    if buf.shape[0] != 48:
        raise ValueError("a compressed G1 point has 48 bytes")
    xy = np.zeros(12, dtype=np.uint64)
    inf = np.zeros(1, dtype=np.uint8)
    rc = _lib.lib().zkgpu_g1_decompress(buf.ctypes.data, xy.ctypes.data, inf.ctypes.data)
    if rc != 0:
        raise ValueError("not a compressed point of the G1 subgroup")
    return xy, int(inf[0])


PROOF_BYTES = 1008


def proof_to_bytes(proof):
    """Proof::to_bytes (plonk/src/proof_system/proof.rs:95-123) of what Prover.prove returns -> 1008 bytes."""
    comms = _np_u64(proof["comms"], 12)
    inf = np.ascontiguousarray(proof["comms_inf"], dtype=np.uint8)
    evals = _np_u64(proof["evals"], 4)
    assert comms.shape[0] == 11 and inf.shape[0] == 11 and evals.shape[0] == 15
    out = np.zeros(PROOF_BYTES, dtype=np.uint8)
    rc = _lib.lib().zkgpu_proof_to_bytes(comms.ctypes.data, inf.ctypes.data, evals.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ZkGpuError(rc, "proof_to_bytes")
    return bytes(out)


def proof_from_bytes(data):
    """Proof::from_bytes (proof.rs:125-157) -> dict(comms, comms_inf, evals); ValueError where the reference returns Err."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    if buf.shape[0] != PROOF_BYTES:
        raise ValueError("a proof has %d bytes" % PROOF_BYTES)
    comms = np.zeros((11, 12), dtype=np.uint64)
    inf = np.zeros(11, dtype=np.uint8)
    evals = np.zeros((15, 4), dtype=np.uint64)
    rc = _lib.lib().zkgpu_proof_from_bytes(buf.ctypes.data, comms.ctypes.data, inf.ctypes.data, evals.ctypes.data)
    if rc != 0:
        raise ValueError("malformed proof bytes")
    return dict(comms=comms, comms_inf=inf, evals=evals)


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


N_SELECTORS = 15
SELECTOR_NAMES = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
                  "q_variable_group_add", "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"]
COMMITMENT_NAMES = ["a_comm", "b_comm", "c_comm", "d_comm", "z_comm", "t_low_comm", "t_mid_comm", "t_high_comm",
                    "t_fourth_comm", "w_z_chall_comm", "w_z_chall_w_comm"]
EVALUATION_NAMES = ["a_eval", "b_eval", "c_eval", "d_eval", "a_w_eval", "b_w_eval", "d_w_eval", "q_arith_eval", "q_c_eval",
                    "q_l_eval", "q_r_eval", "s_sigma_1_eval", "s_sigma_2_eval", "s_sigma_3_eval", "z_eval"]
# Wall-clock laps of one prove() at the points where the transcript needs the device.  The coset transforms of the
# wires and of z are launched behind their commitments, so their time falls into the first two laps, and "quotient" is
# the public-input term, the quotient kernel and the inverse coset transform.
PHASE_NAMES = ["wire_commits", "permutation", "quotient", "quotient_commits", "evaluations", "linearization", "openings",
               "total"]


class QuotientTooShort(ValueError):
    """The quotient has fewer than 3n + 1 coefficients: the reference's prove() panics on its slicing here."""


_ERR_CLASSES[-9] = QuotientTooShort


class Prover:
    """Compiler::preprocess + Prover::prove of the reference (plonk/src/compiler.rs:78-390,
    plonk/src/compiler/prover.rs:210-509) with the ProverKey resident in HBM.  The SRS must be loaded into `ctx`
    first (CommitKey / Context.srs_load)."""

    def __init__(self, ctx, n, constraints, selector_evals, label=b"zkgpu", pi_indexes=()):
        sel = _np_u64(selector_evals, 4)
        assert sel.shape[0] == N_SELECTORS * n
        idx = np.ascontiguousarray(list(pi_indexes), dtype=np.uint64)
        self.ctx, self.n, self.constraints = ctx, int(n), int(constraints)
        ctx._check(ctx._lib.zkgpu_prover_key_load(ctx._h, int(n), int(constraints), sel.ctypes.data, bytes(label),
                                                  len(label), idx.ctypes.data if idx.size else None, idx.size))

    @classmethod
    def try_from_bytes(cls, ctx, data, pi_indexes=()):
        """Prover::try_from_bytes (plonk/src/compiler/prover.rs:137-207): the serialized prover's keys go straight into
        device memory; its commit key replaces the SRS loaded in `ctx`."""
        buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data,
                                   dtype=np.uint8)
        idx = np.ascontiguousarray(list(pi_indexes), dtype=np.uint64)
        n = C.c_uint32(0)
        ctx._check(ctx._lib.zkgpu_prover_load_bytes(ctx._h, buf.ctypes.data if buf.size else None, buf.size,
                                                    idx.ctypes.data if idx.size else None, idx.size, C.byref(n)))
        self = cls.__new__(cls)
        self.ctx, self.n = ctx, int(n.value)
        self.constraints = int.from_bytes(bytes(buf[40:48]), "big")
        return self

    def on(self, ctx):
        """This prover on a clone of its context (Context.clone): the same resident key, another stream."""
        other = Prover.__new__(Prover)
        other.ctx, other.n, other.constraints = ctx, self.n, self.constraints
        return other

    def to_bytes(self):
        """Prover::to_bytes (plonk/src/compiler/prover.rs:85-134) as a numpy uint8 array."""
        size = C.c_size_t(0)
        self.ctx._check(self.ctx._lib.zkgpu_prover_to_bytes(self.ctx._h, None, 0, C.byref(size)))
        out = np.empty(size.value, dtype=np.uint8)
        self.ctx._check(self.ctx._lib.zkgpu_prover_to_bytes(self.ctx._h, out.ctypes.data, out.size, C.byref(size)))
        return out

    def selector_commitments(self):
        xy = np.zeros((N_SELECTORS, 12), dtype=np.uint64)
        inf = np.zeros(N_SELECTORS, dtype=np.uint8)
        self.ctx._check(self.ctx._lib.zkgpu_prover_key_commitments(self.ctx._h, xy.ctypes.data, inf.ctypes.data))
        return xy, inf

    def key_poly(self, which):
        out = np.zeros((self.n, 4), dtype=np.uint64)
        self.ctx._check(self.ctx._lib.zkgpu_prover_key_poly(self.ctx._h, which, out.ctypes.data))
        return out

    def key_eval8n_len(self):
        """Elements of one key vector over the 8n coset held by this context: 8n, or 8n / world under
        sharding.enable_collectives (the slice i = rank + world * k)."""
        return int(self.ctx._lib.zkgpu_prover_key_eval_len(self.ctx._h))

    def key_eval8n(self, which):
        out = np.zeros((self.key_eval8n_len(), 4), dtype=np.uint64)
        self.ctx._check(self.ctx._lib.zkgpu_prover_key_eval8n(self.ctx._h, which, out.ctypes.data))
        return out

    def quotient(self, z, a, b, c, d, pi, challenges7):
        """build_quotient_polynomial -> coefficient array with leading zeros dropped."""
        polys = [_np_u64(p, 4) for p in (z, a, b, c, d, pi)]
        lens = (C.c_size_t * 6)(*[p.shape[0] for p in polys])
        packed = np.ascontiguousarray(np.concatenate(polys, axis=0))
        ch = _np_u64(challenges7, 4)
        out = np.zeros((8 * self.n, 4), dtype=np.uint64)
        t_len = C.c_size_t(0)
        self.ctx._check(self.ctx._lib.zkgpu_quotient(self.ctx._h, packed.ctypes.data, lens, ch.ctypes.data,
                                                     out.ctypes.data, C.byref(t_len)))
        return out[:t_len.value].copy()

    def linearization(self, z, t_low, t_mid, t_high, t_fourth, challenges7, z_challenge, evals15, public_inputs):
        """build_linearization_polynomial -> coefficient array with leading zeros dropped."""
        polys = [_np_u64(p, 4) for p in (z, t_low, t_mid, t_high, t_fourth)]
        lens = (C.c_size_t * 5)(*[p.shape[0] for p in polys])
        packed = np.ascontiguousarray(np.concatenate(polys, axis=0))
        pis = _np_u64(public_inputs, 4)
        cap = max([self.n] + [p.shape[0] for p in polys])
        out = np.zeros((cap, 4), dtype=np.uint64)
        r_len = C.c_size_t(0)
        self.ctx._check(self.ctx._lib.zkgpu_linearization(
            self.ctx._h, packed.ctypes.data, lens, _np_u64(challenges7, 4).ctypes.data, _np_u64(z_challenge, 4).ctypes.data,
            _np_u64(evals15, 4).ctypes.data, pis.ctypes.data if pis.size else None, pis.shape[0], out.ctypes.data, cap,
            C.byref(r_len)))
        return out[:r_len.value].copy()

    def prefetch(self, wires):
        """zkgpu_witness_prefetch: start the upload of a coming proof's witness (a numpy array, ideally over pinned
        memory from pinned_fr); the prove() call that is given the same array takes the uploaded copy."""
        w = _np_u64(wires, 4)
        assert w.shape[0] == 4 * self.n
        self.ctx._check(self.ctx._lib.zkgpu_witness_prefetch(self.ctx._h, w.ctypes.data))

    def prove(self, wires, public_inputs, blinders14, timings=False):
        if _is_torch(wires):
            assert wires.is_cuda and wires.is_contiguous() and wires.numel() == 16 * self.n
            w_ptr = wires.data_ptr()
        else:
            w = _np_u64(wires, 4)
            assert w.shape[0] == 4 * self.n
            w_ptr = w.ctypes.data
        pi = _np_u64(public_inputs, 4)
        bl = _np_u64(blinders14, 4)
        assert bl.shape[0] == 14
        comms = np.zeros((11, 12), dtype=np.uint64)
        inf = np.zeros(11, dtype=np.uint8)
        evals = np.zeros((15, 4), dtype=np.uint64)
        chal = np.zeros((10, 4), dtype=np.uint64)
        tm = (C.c_float * 8)()
        self.ctx._check(self.ctx._lib.zkgpu_prove(self.ctx._h, w_ptr, pi.ctypes.data if pi.size else None,
                                                  pi.shape[0], bl.ctypes.data, comms.ctypes.data, inf.ctypes.data,
                                                  evals.ctypes.data, chal.ctypes.data, tm if timings else None))
        out = dict(comms=comms, comms_inf=inf, evals=evals, challenges=chal,
                   coset=int(self.ctx._lib.zkgpu_prover_last_proof_coset(self.ctx._h)))
        if timings:
            out["timings_ms"] = dict(zip(PHASE_NAMES, [float(x) for x in tm]))
        return out


class PinnedFr:
    """`count` Fr elements of pinned host memory (zkgpu_host_alloc) as a numpy (count, 4) uint64 array in `.array`;
    uploads from it run on the copy engine beside the device's work.  Freed by close() or with the object."""

    def __init__(self, count):
        self._lib = _lib.lib()
        self._ptr = self._lib.zkgpu_host_alloc(int(count) * 32)
        if not self._ptr:
            raise MemoryError("zkgpu_host_alloc(%d bytes) failed" % (int(count) * 32))
        buf = (C.c_uint64 * (int(count) * 4)).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=np.uint64).reshape(int(count), 4)

    def close(self):
        if getattr(self, "_ptr", None):
            self.array = None
            self._lib.zkgpu_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def perm_z(ctx, wires, sigma_evals, beta, gamma):
    w = _np_u64(wires, 4)
    s = _np_u64(sigma_evals, 4)
    n = w.shape[0] // 4
    out = np.zeros((n, 4), dtype=np.uint64)
    ctx._check(ctx._lib.zkgpu_perm_z(ctx._h, n, w.ctypes.data, s.ctypes.data, _np_u64(beta, 4).ctypes.data,
                                     _np_u64(gamma, 4).ctypes.data, out.ctypes.data))
    return out


def poly_eval(ctx, coeffs, x):
    c = _np_u64(coeffs, 4)
    out = np.zeros(4, dtype=np.uint64)
    ctx._check(ctx._lib.zkgpu_poly_eval(ctx._h, c.ctypes.data if c.size else None, c.shape[0], _np_u64(x, 4).ctypes.data,
                                        out.ctypes.data))
    return out


def poly_ruffini(ctx, coeffs, root):
    c = _np_u64(coeffs, 4)
    out = np.zeros((max(c.shape[0] - 1, 0), 4), dtype=np.uint64)
    ctx._check(ctx._lib.zkgpu_poly_ruffini(ctx._h, c.ctypes.data if c.size else None, c.shape[0],
                                           _np_u64(root, 4).ctypes.data, out.ctypes.data if out.size else None))
    return out


def aggregate_witness(ctx, polys, point, v):
    """CommitKey::compute_aggregate_witness -> coefficient array with leading zeros dropped."""
    arrs = [_np_u64(p, 4) for p in polys]
    lens = (C.c_size_t * len(arrs))(*[a.shape[0] for a in arrs])
    packed = np.ascontiguousarray(np.concatenate(arrs, axis=0)) if arrs else np.zeros((0, 4), np.uint64)
    cap = max([a.shape[0] for a in arrs] + [1])
    out = np.zeros((cap, 4), dtype=np.uint64)
    n_out = C.c_size_t(0)
    ctx._check(ctx._lib.zkgpu_aggregate_witness(ctx._h, packed.ctypes.data, lens, len(arrs), _np_u64(point, 4).ctypes.data,
                                                _np_u64(v, 4).ctypes.data, out.ctypes.data, C.byref(n_out)))
    return out[:n_out.value].copy()


def transcript_challenge(label, msg_label, msg, ch_label, n):
    out = np.zeros(n, dtype=np.uint8)
    m = np.frombuffer(bytes(msg), dtype=np.uint8).copy()
    rc = _lib.lib().zkgpu_transcript_challenge(bytes(label), bytes(msg_label), m.ctypes.data if m.size else None, m.size,
                                               bytes(ch_label), out.ctypes.data, n)
    if rc != 0:
        raise ZkGpuError(rc, "transcript_challenge")
    return bytes(out)
