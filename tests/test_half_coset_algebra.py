"""The algebra behind the prover's quotient over 4n points, checked on the CPU against the oracle's quotient (which
follows the reference over the 8n coset, plonk/src/proof_system/quotient_poly.rs:15-100).

The quotient t has at most 4n + 7 coefficients (the permutation term of the numerator reaches degree 5n + 6, a widget's
term 5n + 3, the rest less).  The library
(zkvm_amd/csrc/prover.hip: quotient_half_coset) evaluates the numerator on the 4n points g w_4n^k, which fixes t modulo
X^4n - g^4n, and takes the seven top coefficients from the top coefficients of the round polynomials:
    t_(4n+i) = [alpha (z prod_w (w + beta k_w X + gamma) - z(wX) prod_w (w + beta sigma_w + gamma))]_(5n+i),  i < 7.
(a key with widget gates adds the top coefficients of its widget terms, which the GPU tests compare through whole
proofs).  Both statements are verified here with Python integers on the oracle's own t."""
import numpy as np
import pytest

import model as M

R = M.R_MOD


def _ints(a):
    return M.fr_vec_from_mont(np.ascontiguousarray(a).reshape(-1, 4))


def _mul_top(a, b, keep):
    """Top `keep` coefficients of a * b, both given by their top coefficients in descending order."""
    out = [0] * keep
    for i, x in enumerate(a[:keep]):
        for j, y in enumerate(b[:keep - i]):
            out[i + j] = (out[i + j] + x * y) % R
    return out


@pytest.mark.parametrize("log_n,seed", [(4, 1), (6, 2), (8, 3)])
def test_top_coefficients_and_folding(orc, log_n, seed):
    from zkvm_amd import circuits
    n = 1 << log_n
    c = circuits.arithmetic_circuit(n, seed=seed)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    srs, _ = orc.g1_multiples_of_generator(1, n + 7)
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"algebra", c["pi_indexes"], threads=2)
    assert op.preprocess_rc == 0
    ch = M.random_fr_mont(7, seed=40 + seed)
    alpha, beta, gamma = _ints(ch[:3])
    # blinded round polynomials of the lengths prove() produces (prover.rs:64-83)
    bl = M.random_fr_mont(11, seed=50 + seed)
    polys = []
    for k in range(4):
        p = np.zeros((n + 2, 4), np.uint64)
        p[:n] = orc.fft("ifft", c["wires"][k])
        v = _ints(p)
        b = _ints(bl[2 * k:2 * k + 2])
        for i in range(2):
            v[i] = (v[i] - b[i]) % R
            v[n + i] = b[i]
        polys.append(M.fr_vec_to_mont(v))
    sigma_polys = [op.poly(11 + k) for k in range(4)]
    zvec = orc.permutation_vec(c["wires"], sigma_polys, ch[1], ch[2])
    zv = _ints(orc.fft("ifft", zvec)) + [0, 0, 0]
    b = _ints(bl[8:11])
    for i in range(3):
        zv[i] = (zv[i] - b[i]) % R
        zv[n + i] = b[i]
    z = M.fr_vec_to_mont(zv)
    dense_pi = np.zeros((n, 4), np.uint64)
    dense_pi[0] = c["public_inputs"][0]
    pi = orc.fft("ifft", dense_pi)
    t = _ints(op.quotient(z, *polys, pi, ch))
    assert len(t) == 4 * n + 7           # the circuit is satisfied: exact division, full length with random blinders

    # the seven top coefficients from those of the factors
    omega = M.domain_root(n)
    top = lambda v, deg: [v[deg - j] if 0 <= deg - j < len(v) else 0 for j in range(7)]
    ident = top(zv, n + 2)
    copy = [x * pow(omega, (n + 2 - j) % n, R) % R for j, x in enumerate(ident)]       # z(wX): z_j w^j, w^n = 1
    for k in range(4):
        w = top(_ints(polys[k]), n + 1)
        s = top(_ints(sigma_polys[k]), n + 1)
        ident = _mul_top(ident, w, 7)                       # beta k_w X + gamma (k_w = 1, 7, 13, 17) lie at degrees 0, 1
        copy = _mul_top(copy, [(x + beta * y) % R for x, y in zip(w, s)], 7)
    want_top = [alpha * (a - b) % R for a, b in zip(ident, copy)]                      # degrees 5n+6 ... 5n of N
    assert [t[4 * n + 6 - j] for j in range(7)] == want_top

    # t modulo X^4n - g^4n is what the 4n coset sees: its evaluations there equal those of the folded polynomial
    g4n = pow(7, 4 * n, R)
    folded = [(t[i] + (g4n * t[4 * n + i] if i < 7 else 0)) % R for i in range(4 * n)]
    ev_t = orc.fft("coset_fft", M.fr_vec_to_mont(t), 8 * n)[::2]                       # every second point of the 8n coset
    ev_f = orc.fft("coset_fft", M.fr_vec_to_mont(folded), 4 * n)
    assert np.array_equal(ev_t, ev_f)
    # and the way back: inverse transform over 4n points, then the two patches
    r = _ints(orc.fft("coset_ifft", ev_f, 4 * n))
    rebuilt = [(r[i] - (g4n * t[4 * n + i] if i < 7 else 0)) % R for i in range(4 * n)] + t[4 * n:]
    assert rebuilt == t


@pytest.mark.parametrize("which", ["range_logic", "curve"])
def test_quotient_of_a_widget_key_has_4n_plus_7_coefficients(orc, which):
    """The degree bound that lets keys with widget gates take the 4n route too: the oracle's quotient (over 8n points)
    of a satisfied circuit with range and logic gates, and of one with fixed-base and curve-addition gates."""
    import test_composer as T
    c = (T.widget_circuit(0xbeef, 0x1234, 16) if which == "range_logic" else T.curve_circuit()).tables()
    n = c["n"]
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    assert any(c["selectors"][7 + w].any() for w in range(4))
    srs, _ = orc.g1_multiples_of_generator(1, n + 7)
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"algebra", c["pi_indexes"], threads=4)
    assert op.preprocess_rc == 0
    ch = M.random_fr_mont(7, seed=77)
    bl = M.random_fr_mont(11, seed=78)
    polys = []
    for k in range(4):
        v = _ints(orc.fft("ifft", c["wires"][k])) + [0, 0]
        b = _ints(bl[2 * k:2 * k + 2])
        for i in range(2):
            v[i] = (v[i] - b[i]) % R
            v[n + i] = b[i]
        polys.append(M.fr_vec_to_mont(v))
    zvec = orc.permutation_vec(c["wires"], [op.poly(11 + k) for k in range(4)], ch[1], ch[2])
    zv = _ints(orc.fft("ifft", zvec)) + [0, 0, 0]
    b = _ints(bl[8:11])
    for i in range(3):
        zv[i] = (zv[i] - b[i]) % R
        zv[n + i] = b[i]
    dense_pi = np.zeros((n, 4), np.uint64)
    for j, i in enumerate(c["pi_indexes"]):
        dense_pi[i] = c["public_inputs"][j]
    t = _ints(op.quotient(M.fr_vec_to_mont(zv), *polys, orc.fft("ifft", dense_pi), ch))
    assert len(t) == 4 * n + 7
    g4n = pow(7, 4 * n, R)
    folded = [(t[i] + (g4n * t[4 * n + i] if i < 7 else 0)) % R for i in range(4 * n)]
    assert np.array_equal(orc.fft("coset_fft", M.fr_vec_to_mont(t), 8 * n)[::2], orc.fft("coset_fft", M.fr_vec_to_mont(folded), 4 * n))
