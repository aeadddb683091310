"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/liboracle.so (the CPU restatement of the
reference's algorithms).  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package.

All field elements travel as numpy uint64 arrays of little-endian limbs in Montgomery form:
Fr (..., 4), Fq (..., 6), affine points (..., 12) = x|y with a separate uint8 infinity flag,
projective points (..., 18) = x|y|z.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

N_SEL = 15
SEL_NAMES = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
             "q_variable_group_add", "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"]
EVAL_NAMES = ["a_eval", "b_eval", "c_eval", "d_eval", "a_w_eval", "b_w_eval", "d_w_eval", "q_arith_eval", "q_c_eval",
              "q_l_eval", "q_r_eval", "s_sigma_1_eval", "s_sigma_2_eval", "s_sigma_3_eval", "z_eval"]
COMM_NAMES = ["a_comm", "b_comm", "c_comm", "d_comm", "z_comm", "t_low_comm", "t_mid_comm", "t_high_comm",
              "t_fourth_comm", "w_z_chall_comm", "w_z_chall_w_comm"]


def build(force=False):
    """Compile liboracle.so with the Makefile next to this file (g++ only)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_msm.restype = C.c_double
        _lib.orc_fft.restype = C.c_double
        _lib.orc_msm_window_size.restype = C.c_size_t
        _lib.orc_poly_ruffini.restype = C.c_size_t
        _lib.orc_aggregate_witness.restype = C.c_size_t
        _lib.orc_prover_new.restype = C.c_void_p
        _lib.orc_prover_poly.restype = C.c_size_t
        _lib.orc_prover_eval8n.restype = C.c_size_t
        _lib.orc_quotient.restype = C.c_size_t
        _lib.orc_linearization.restype = C.c_size_t
        _lib.orc_prover_to_bytes.restype = C.c_size_t
    return _lib


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- Fr ----------------------------------------------------------------------------------------------
_FR_OPS = {"add": 0, "sub": 1, "mul": 2, "neg": 3, "square": 4, "invert": 5, "reduce": 6, "from_raw": 7}


def fr_op(op, a, b=None):
    a = _u64(a).reshape(-1, 4)
    out = np.empty_like(a)
    bb = None
    if b is not None:
        bb = _u64(np.broadcast_to(_u64(b).reshape(-1, 4), a.shape))
    lib().orc_fr_op(_FR_OPS[op], _p(a), _p(bb) if bb is not None else None, _p(out), C.c_size_t(a.shape[0]))
    return out


def fr_pow(a, by4):
    a = _u64(a).reshape(4)
    by = _u64(by4).reshape(4)
    out = np.empty(4, np.uint64)
    lib().orc_fr_pow(_p(a), _p(by), _p(out))
    return out


def fr_to_bytes(a):
    a = _u64(a).reshape(4)
    out = np.empty(32, np.uint8)
    lib().orc_fr_to_bytes(_p(a), _p(out))
    return bytes(out)


def fr_from_bytes(b):
    buf = np.frombuffer(bytes(b), np.uint8).copy()
    out = np.empty(4, np.uint64)
    ok = lib().orc_fr_from_bytes(_p(buf), _p(out))
    return out, bool(ok)


def fr_from_bytes_wide(b):
    buf = np.frombuffer(bytes(b), np.uint8).copy()
    out = np.empty(4, np.uint64)
    lib().orc_fr_from_bytes_wide(_p(buf), _p(out))
    return out


def fr_from_u512(words8):
    w = _u64(words8).reshape(8)
    out = np.empty(4, np.uint64)
    lib().orc_fr_from_u512(_p(w), _p(out))
    return out


def fr_divn(a, n):
    a = _u64(a).reshape(4).copy()
    lib().orc_fr_divn(_p(a), C.c_uint32(n))
    return a


def batch_inversion(v):
    v = _u64(v).reshape(-1, 4).copy()
    lib().orc_batch_inversion(_p(v), C.c_size_t(v.shape[0]))
    return v


# ---- Fq ----------------------------------------------------------------------------------------------
_FQ_OPS = {"add": 0, "sub": 1, "mul": 2, "neg": 3, "square": 4, "invert": 5, "sqrt": 6}


def fq_op(op, a, b=None):
    a = _u64(a).reshape(-1, 6)
    out = np.empty_like(a)
    bb = None
    if b is not None:
        bb = _u64(np.broadcast_to(_u64(b).reshape(-1, 6), a.shape))
    lib().orc_fq_op(_FQ_OPS[op], _p(a), _p(bb) if bb is not None else None, _p(out), C.c_size_t(a.shape[0]))
    return out


def fq_to_bytes(a):
    a = _u64(a).reshape(6)
    out = np.empty(48, np.uint8)
    lib().orc_fq_to_bytes(_p(a), _p(out))
    return bytes(out)


def fq_from_bytes(b):
    buf = np.frombuffer(bytes(b), np.uint8).copy()
    out = np.empty(6, np.uint64)
    ok = lib().orc_fq_from_bytes(_p(buf), _p(out))
    return out, bool(ok)


def fq_lex_largest(a):
    return bool(lib().orc_fq_lex_largest(_p(_u64(a).reshape(6))))


# ---- G1 ----------------------------------------------------------------------------------------------
def g1_generator():
    out = np.empty(18, np.uint64)
    lib().orc_g1_generator(_p(out))
    return out


def g1_identity():
    out = np.empty(18, np.uint64)
    lib().orc_g1_identity(_p(out))
    return out


def g1_add(a, b):
    out = np.empty(18, np.uint64)
    lib().orc_g1_add(_p(_u64(a)), _p(_u64(b)), _p(out))
    return out


def g1_double(a):
    out = np.empty(18, np.uint64)
    lib().orc_g1_double(_p(_u64(a)), _p(out))
    return out


def g1_add_mixed(a, bxy, binf=0):
    out = np.empty(18, np.uint64)
    lib().orc_g1_add_mixed(_p(_u64(a)), _p(_u64(bxy)), C.c_uint8(binf), _p(out))
    return out


def g1_mul(a, scalar):
    out = np.empty(18, np.uint64)
    lib().orc_g1_mul(_p(_u64(a)), _p(_u64(scalar)), _p(out))
    return out


def g1_to_affine(a):
    xy = np.empty(12, np.uint64)
    inf = C.c_uint8(0)
    lib().orc_g1_to_affine(_p(_u64(a)), _p(xy), C.byref(inf))
    return xy, inf.value


def g1_is_on_curve(a):
    return bool(lib().orc_g1_is_on_curve(_p(_u64(a))))


def g1_compress(xy, inf=0):
    out = np.empty(48, np.uint8)
    lib().orc_g1_compress(_p(_u64(xy)), C.c_uint8(inf), _p(out))
    return bytes(out)


def g1_multiples_of_generator(start, n):
    xy = np.empty((n, 12), np.uint64)
    inf = np.empty(n, np.uint8)
    lib().orc_g1_multiples_of_generator(C.c_uint64(start), C.c_size_t(n), _p(xy), _p(inf))
    return xy, inf


def srs_powers(g_proj, tau, n):
    xy = np.empty((n, 12), np.uint64)
    inf = np.empty(n, np.uint8)
    lib().orc_srs_powers(_p(_u64(g_proj)), _p(_u64(tau)), C.c_size_t(n), _p(xy), _p(inf))
    return xy, inf


# ---- MSM / FFT -----------------------------------------------------------------------------------------
def msm(points_xy, scalars, inf=None, threads=1, with_time=False):
    """msm_variable_base(points, scalars) -> projective (18 limbs)."""
    pts = _u64(points_xy).reshape(-1, 12)
    sc = _u64(scalars).reshape(-1, 4)
    out = np.empty(18, np.uint64)
    infp = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, np.uint8)
        infp = _p(inf)
    secs = lib().orc_msm(_p(pts), infp, _p(sc), C.c_size_t(sc.shape[0]), C.c_size_t(pts.shape[0]), C.c_uint(threads),
                         _p(out))
    return (out, secs) if with_time else out


def msm_window_size(n):
    return int(lib().orc_msm_window_size(C.c_size_t(n)))


_FFT_MODES = {"fft": 0, "ifft": 1, "coset_fft": 2, "coset_ifft": 3}


def fft(mode, data, domain_size=None, with_time=False):
    """EvaluationDomain::new(domain_size).{fft,ifft,coset_fft,coset_ifft}(data)."""
    d = _u64(data).reshape(-1, 4)
    if domain_size is None:
        domain_size = d.shape[0]
    size = 1
    while size < domain_size:
        size *= 2
    out = np.empty((size, 4), np.uint64)
    secs = lib().orc_fft(_FFT_MODES[mode], _p(d), C.c_size_t(d.shape[0]), C.c_size_t(domain_size), _p(out))
    if secs < 0:
        raise ValueError("InvalidEvalDomainSize")
    return (out, secs) if with_time else out


def domain_new(num_coeffs):
    size = C.c_uint64(0)
    out5 = np.empty((5, 4), np.uint64)
    log_size = lib().orc_domain_new(C.c_size_t(num_coeffs), C.byref(size), _p(out5))
    if log_size < 0:
        raise ValueError("InvalidEvalDomainSize")
    return dict(size=size.value, log_size_of_group=log_size, size_as_field_element=out5[0], size_inv=out5[1],
                group_gen=out5[2], group_gen_inv=out5[3], generator_inv=out5[4])


def vanishing_poly_over_coset(domain_size, poly_degree):
    out = np.empty((domain_size, 4), np.uint64)
    lib().orc_vanishing_poly_over_coset(C.c_size_t(domain_size), C.c_uint64(poly_degree), _p(out))
    return out


def lagrange_coefficients(domain_size, tau):
    out = np.empty((domain_size, 4), np.uint64)
    lib().orc_lagrange_coefficients(C.c_size_t(domain_size), _p(_u64(tau)), _p(out))
    return out


# ---- polynomial / protocol ------------------------------------------------------------------------------
def poly_evaluate(coeffs, x):
    c = _u64(coeffs).reshape(-1, 4)
    out = np.empty(4, np.uint64)
    lib().orc_poly_evaluate(_p(c), C.c_size_t(c.shape[0]), _p(_u64(x)), _p(out))
    return out


def poly_ruffini(coeffs, root):
    c = _u64(coeffs).reshape(-1, 4)
    out = np.empty((max(c.shape[0], 1), 4), np.uint64)
    n = lib().orc_poly_ruffini(_p(c), C.c_size_t(c.shape[0]), _p(_u64(root)), _p(out))
    return out[:n].copy()


def _pack(polys):
    polys = [_u64(p).reshape(-1, 4) for p in polys]
    lens = (C.c_size_t * len(polys))(*[p.shape[0] for p in polys])
    packed = np.concatenate(polys, axis=0) if polys else np.empty((0, 4), np.uint64)
    return _u64(packed), lens, max([p.shape[0] for p in polys] + [1])


def aggregate_witness(polys, point, v):
    packed, lens, mx = _pack(polys)
    out = np.empty((mx, 4), np.uint64)
    n = lib().orc_aggregate_witness(_p(packed), lens, C.c_size_t(len(polys)), _p(_u64(point)), _p(_u64(v)), _p(out))
    return out[:n].copy()


def permutation_vec(wires, sigma_polys, beta, gamma):
    w = _u64(wires).reshape(4, -1, 4)
    n = w.shape[1]
    packed, lens, _ = _pack(sigma_polys)
    out = np.empty((n, 4), np.uint64)
    lib().orc_permutation_vec(C.c_size_t(n), _p(w), _p(packed), lens, _p(_u64(beta)), _p(_u64(gamma)), _p(out))
    return out


def sigma_to_lagrange(sigma_map):
    m = _u64(sigma_map).reshape(-1)
    out = np.empty((m.shape[0], 4), np.uint64)
    lib().orc_sigma_to_lagrange(C.c_size_t(m.shape[0]), _p(m), _p(out))
    return out


class Prover:
    """Compiler::preprocess + Prover of the reference, restated on the CPU."""

    def __init__(self, n, constraints, selector_evals, srs_xy, label=b"zkgpu", pi_indexes=(), threads=1):
        sel = _u64(selector_evals).reshape(N_SEL, n, 4)
        srs = _u64(srs_xy).reshape(-1, 12)
        idx = _u64(np.asarray(list(pi_indexes), dtype=np.uint64))
        rc = C.c_int(0)
        self.n, self.constraints, self.threads = n, constraints, threads
        self._h = C.c_void_p(lib().orc_prover_new(C.c_size_t(n), C.c_size_t(constraints), _p(sel), _p(srs),
                                                  C.c_size_t(srs.shape[0]), C.c_char_p(bytes(label)), _p(idx),
                                                  C.c_size_t(idx.shape[0]), C.c_uint(threads), C.byref(rc)))
        self.preprocess_rc = rc.value

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_prover_free(self._h)
            self._h = None

    def poly(self, which):
        out = np.empty((self.n, 4), np.uint64)
        k = lib().orc_prover_poly(self._h, C.c_int(which), _p(out))
        return out[:k].copy()

    def eval8n(self, which):
        out = np.empty((8 * self.n, 4), np.uint64)
        lib().orc_prover_eval8n(self._h, C.c_int(which), _p(out))
        return out

    def selector_comms(self):
        xy = np.empty((N_SEL, 12), np.uint64)
        inf = np.empty(N_SEL, np.uint8)
        lib().orc_prover_selector_comms(self._h, _p(xy), _p(inf))
        return xy, inf

    def quotient(self, z, a, b, c, d, pi, challenges7, threads=None, with_time=False):
        packed, lens, _ = _pack([z, a, b, c, d, pi])
        out = np.empty((8 * self.n, 4), np.uint64)
        secs = C.c_double(0)
        k = lib().orc_quotient(self._h, _p(packed), lens, _p(_u64(challenges7).reshape(7, 4)),
                               C.c_uint(threads or self.threads), _p(out), C.byref(secs))
        return (out[:k].copy(), secs.value) if with_time else out[:k].copy()

    def linearization(self, z, t_low, t_mid, t_high, t_fourth, challenges7, z_challenge, evals15, pub_inputs):
        packed, lens, mx = _pack([z, t_low, t_mid, t_high, t_fourth])
        pi = _u64(pub_inputs).reshape(-1, 4)
        out = np.empty((mx + self.n + 8, 4), np.uint64)
        k = lib().orc_linearization(self._h, _p(packed), lens, _p(_u64(challenges7).reshape(7, 4)),
                                    _p(_u64(z_challenge)), _p(_u64(evals15).reshape(15, 4)), _p(pi),
                                    C.c_size_t(pi.shape[0]), _p(out))
        return out[:k].copy()

    def set_vk_n(self, vk_n):
        """VerifierKey.n of a prover rebuilt from bytes (widget.rs:102-106); a compiled prover has n = constraints."""
        lib().orc_prover_set_vk_n(self._h, C.c_uint64(vk_n))

    def to_bytes(self):
        """Prover::to_bytes (compiler/prover.rs:109-134) as a numpy uint8 array"""
        size = lib().orc_prover_to_bytes(self._h, None, C.c_size_t(0))
        out = np.empty(size, np.uint8)
        lib().orc_prover_to_bytes(self._h, _p(out), C.c_size_t(size))
        return out

    def prove(self, wires, pub_inputs, blinders14, threads=None):
        w = _u64(wires).reshape(4, self.n, 4)
        pi = _u64(pub_inputs).reshape(-1, 4)
        bl = _u64(blinders14).reshape(14, 4)
        comms = np.empty((11, 12), np.uint64)
        inf = np.empty(11, np.uint8)
        evals = np.empty((15, 4), np.uint64)
        chal = np.empty((10, 4), np.uint64)
        secs = C.c_double(0)
        rc = lib().orc_prove(self._h, _p(w), _p(pi), C.c_size_t(pi.shape[0]), _p(bl), C.c_uint(threads or self.threads),
                             _p(comms), _p(inf), _p(evals), _p(chal), C.byref(secs))
        return dict(rc=rc, comms=comms, comms_inf=inf, evals=evals, challenges=chal, secs=secs.value)


def verifier_challenges(label, sel_xy, sel_inf, constraints, pub_inputs, comms_xy, comms_inf, evals15):
    """verifier.rs:187-217 + proof.rs:186-255 -> (11, 4): beta gamma alpha range logic fixed var z v v_w u"""
    pi = _u64(pub_inputs).reshape(-1, 4)
    out = np.empty((11, 4), np.uint64)
    lib().orc_verifier_challenges(C.c_char_p(bytes(label)), C.c_size_t(len(label)), _p(_u64(sel_xy).reshape(N_SEL, 12)),
                                  _p(np.ascontiguousarray(sel_inf, dtype=np.uint8)), C.c_uint64(constraints), _p(pi),
                                  C.c_size_t(pi.shape[0]), _p(_u64(comms_xy).reshape(11, 12)),
                                  _p(np.ascontiguousarray(comms_inf, dtype=np.uint8)), _p(_u64(evals15).reshape(15, 4)),
                                  _p(out))
    return out


def strobe_conformance():
    out = np.empty(64, np.uint8)
    lib().orc_strobe_conformance(_p(out))
    return bytes(out[:32]), bytes(out[32:])


def merlin_simple(label, msg_label, msg, ch_label, n):
    out = np.empty(n, np.uint8)
    m = np.frombuffer(bytes(msg), np.uint8).copy()
    lib().orc_merlin_simple(C.c_char_p(label), C.c_char_p(msg_label), _p(m), C.c_size_t(len(msg)), C.c_char_p(ch_label),
                            _p(out), C.c_size_t(n))
    return bytes(out)


def keccak_f1600(state25):
    s = _u64(state25).reshape(25).copy()
    lib().orc_keccak_f1600(_p(s))
    return s
