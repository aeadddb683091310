"""Pins the CPU oracle's G1 arithmetic, MSM, FFT and transcript:
 * generator / 2G limbs                      coset-bls12_381/src/g1.rs:218-238, 1246-1272
 * k*G for k = 0..999, compressed            coset-bls12_381/src/tests/g1_compressed_valid_test_vectors.dat
                                             (recipe tests/mod.rs:4-37), committed under tests/golden/
 * msm_variable_base_test / pippenger_test   coset-bls12_381/src/coset/multiscalar_mul.rs:240-278
 * FFT: the reference holds no known-answer vector; pinned to an O(n^2) big-integer DFT instead
"""
import os

import numpy as np
import pytest

import model as M
from conftest import GOLDEN

U = np.uint64


def eq(a, b):
    return np.array_equal(np.asarray(a, U).reshape(-1), np.asarray(b, U).reshape(-1))


def test_generator_matches_model(orc):
    g = orc.g1_generator()
    assert M.projective_to_affine_int(g) == (M.GEN_X, M.GEN_Y)
    assert (M.GEN_Y * M.GEN_Y - M.GEN_X ** 3 - 4) % M.P_MOD == 0
    assert orc.g1_is_on_curve(g) and orc.g1_is_on_curve(orc.g1_identity())


def test_doubling_kat(orc):
    two_g, inf = orc.g1_to_affine(orc.g1_double(orc.g1_generator()))
    want = np.array([0x53e978ce58a9ba3c, 0x3ea0583c4f3d65f9, 0x4d20bb47f0012960, 0xa54c664ae5b2b5d9, 0x26b552a39d7eb21f,
                     0x0008895d26e68785, 0x70110b3298293940, 0xda33c5393f1f6afc, 0xb86edfd16a5aa785, 0xaec6d1c9e7b1c895,
                     0x25cfc2b522d11720, 0x06361c83f8d09b15], dtype=U)
    assert inf == 0 and eq(two_g, want)
    ident = orc.g1_double(orc.g1_identity())
    assert orc.g1_to_affine(ident)[1] == 1


def test_compressed_vectors_k_times_g(orc):
    """e = identity; 1000 times: compare compress(affine(e)) with the vector, e += G (tests/mod.rs:4-37)."""
    data = open(os.path.join(GOLDEN, "g1_compressed_valid_test_vectors.dat"), "rb").read()
    assert len(data) == 48000
    e = orc.g1_identity()
    g = orc.g1_generator()
    gxy, _ = orc.g1_to_affine(g)
    for k in range(1000):
        xy, inf = orc.g1_to_affine(e)
        assert orc.g1_compress(xy, inf) == data[48 * k:48 * k + 48], k
        # alternate full and mixed addition so both formulas are pinned
        e = orc.g1_add(e, g) if k % 2 == 0 else orc.g1_add_mixed(e, gxy, 0)
    # the same vectors through scalar multiplication and through the model's compression
    for k in (0, 1, 2, 3, 17, 255, 256, 999):
        p = orc.g1_mul(g, M.fr_to_mont(k))
        xy, inf = orc.g1_to_affine(p)
        assert orc.g1_compress(xy, inf) == data[48 * k:48 * k + 48]
        assert M.compress(M.ec_mul((M.GEN_X, M.GEN_Y), k)) == data[48 * k:48 * k + 48]


def test_add_edge_cases(orc):
    g = orc.g1_generator()
    ident = orc.g1_identity()
    gxy, _ = orc.g1_to_affine(g)
    neg_g = g.copy()
    neg_g[6:12] = orc.fq_op("neg", g[6:12])[0]
    assert orc.g1_to_affine(orc.g1_add(g, neg_g))[1] == 1                       # P + (-P) = O
    assert eq(orc.g1_to_affine(orc.g1_add(g, g))[0], orc.g1_to_affine(orc.g1_double(g))[0])   # complete formula doubles
    assert eq(orc.g1_to_affine(orc.g1_add_mixed(g, gxy, 0))[0], orc.g1_to_affine(orc.g1_double(g))[0])
    assert eq(orc.g1_to_affine(orc.g1_add(ident, g))[0], gxy) and eq(orc.g1_to_affine(orc.g1_add(g, ident))[0], gxy)
    assert eq(orc.g1_to_affine(orc.g1_add_mixed(ident, gxy, 0))[0], gxy)
    assert eq(orc.g1_add_mixed(g, gxy, 1), g)                                    # adding the affine identity


def test_multiples_helper(orc):
    xy, inf = orc.g1_multiples_of_generator(0, 6)
    assert inf.tolist() == [1, 0, 0, 0, 0, 0]
    for k in range(1, 6):
        assert M.affine_from_limbs(xy[k]) == M.ec_mul((M.GEN_X, M.GEN_Y), k)


# ---- MSM ---------------------------------------------------------------------------------------------------
def test_msm_variable_base_test(orc):
    """multiscalar_mul.rs:271-278: one point, scalar 100."""
    gxy, _ = orc.g1_to_affine(orc.g1_generator())
    got = orc.msm(gxy.reshape(1, 12), M.fr_to_mont(100).reshape(1, 4))
    data = open(os.path.join(GOLDEN, "g1_compressed_valid_test_vectors.dat"), "rb").read()
    xy, inf = orc.g1_to_affine(got)
    assert orc.g1_compress(xy, inf) == data[4800:4848]


def test_msm_all_bases_g_small_sums(orc):
    """Any MSM over bases = G whose scalar sum is < 1000 is pinned by the k*G vectors."""
    data = open(os.path.join(GOLDEN, "g1_compressed_valid_test_vectors.dat"), "rb").read()
    gxy, _ = orc.g1_to_affine(orc.g1_generator())
    for n, scalars in ((5, [0, 1, 1, 3, 7]), (40, list(range(40))), (33, [1] * 33), (64, [0] * 63 + [999])):
        pts = np.tile(gxy, (n, 1))
        got = orc.msm(pts, M.fr_vec_to_mont(scalars))
        xy, inf = orc.g1_to_affine(got)
        k = sum(scalars)
        assert orc.g1_compress(xy, inf) == data[48 * k:48 * k + 48]


def test_msm_pippenger_recipe(orc):
    """multiscalar_mul.rs:240-269 recipe applied to msm_variable_base: points (1+i)G, scalars a0 + i*step,
    compared with the naive sum for sizes 512, 256, ..., 1."""
    n = 512
    a0 = pow(2128506, -1, M.R_MOD)
    step = pow(4443282, -1, M.R_MOD)
    pts, inf = orc.g1_multiples_of_generator(1, n)
    assert not inf.any()
    scalars = [(a0 + i * step) % M.R_MOD for i in range(n)]
    sc = M.fr_vec_to_mont(scalars)
    size = n
    while size > 0:
        got = M.projective_to_affine_int(orc.msm(pts[:size], sc[:size]))
        k = sum((1 + i) * scalars[i] for i in range(size)) % M.R_MOD      # everything is a multiple of G
        assert got == M.ec_mul((M.GEN_X, M.GEN_Y), k), size
        size //= 2


def test_msm_window_rule(orc):
    assert orc.msm_window_size(31) == 3 and orc.msm_window_size(32) == 6
    assert orc.msm_window_size(1 << 10) == 9 and orc.msm_window_size(1 << 20) == 16 and orc.msm_window_size(1 << 24) == 19


def test_msm_threads_and_truncation(orc):
    pts, _ = orc.g1_multiples_of_generator(1, 100)
    sc = M.random_fr_mont(80, 3)
    sc[5] = 0
    sc[6] = M.fr_to_mont(1)
    a = orc.msm(pts, sc, threads=1)
    b = orc.msm(pts, sc, threads=4)
    assert eq(a, b)
    want = M.ec_mul((M.GEN_X, M.GEN_Y), sum((1 + i) * M.fr_from_mont(sc[i]) for i in range(80)) % M.R_MOD)
    assert M.projective_to_affine_int(a) == want
    # scalars longer than points: the zip truncates to the points
    sc2 = M.random_fr_mont(120, 4)
    c = orc.msm(pts, sc2)
    want = M.ec_mul((M.GEN_X, M.GEN_Y), sum((1 + i) * M.fr_from_mont(sc2[i]) for i in range(100)) % M.R_MOD)
    assert M.projective_to_affine_int(c) == want


# ---- FFT ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 4, 8, 64, 256])
def test_fft_vs_naive_dft(orc, n):
    vals = M.random_fr_ints(n, 100 + n)
    mont = M.fr_vec_to_mont(vals)
    assert M.fr_vec_from_mont(orc.fft("fft", mont)) == M.naive_dft(vals, n)
    assert M.fr_vec_from_mont(orc.fft("ifft", mont)) == M.naive_dft(vals, n, inverse=True)
    assert M.fr_vec_from_mont(orc.fft("coset_fft", mont)) == M.naive_dft(vals, n, coset=True)
    assert M.fr_vec_from_mont(orc.fft("coset_ifft", mont)) == M.naive_dft(vals, n, inverse=True, coset=True)


def test_fft_padding_and_truncation(orc):
    vals = M.random_fr_ints(11, 7)
    mont = M.fr_vec_to_mont(vals)
    # shorter input is zero-padded; the coset powers are applied to the un-padded coefficients
    got = M.fr_vec_from_mont(orc.fft("coset_fft", mont, domain_size=32))
    assert got == [M.poly_eval(vals, 7 * pow(M.domain_root(32), i, M.R_MOD) % M.R_MOD) for i in range(32)]
    # longer input is truncated to the domain (domain.rs:146)
    got = M.fr_vec_from_mont(orc.fft("fft", mont, domain_size=8))
    assert got == M.naive_dft(vals[:8], 8)
    # round trips
    x = M.random_fr_mont(1 << 10, 9)
    assert eq(orc.fft("ifft", orc.fft("fft", x)), x)
    assert eq(orc.fft("coset_ifft", orc.fft("coset_fft", x)), x)


def test_domain_fields(orc):
    for k in range(0, 12):
        d = orc.domain_new(1 << k)
        assert d["size"] == 1 << k and d["log_size_of_group"] == k
        w = M.fr_from_mont(d["group_gen"])
        assert w == M.domain_root(1 << k)
        assert M.fr_from_mont(d["group_gen_inv"]) == pow(w, -1, M.R_MOD)
        assert M.fr_from_mont(d["size_inv"]) == pow(1 << k, -1, M.R_MOD)
        assert M.fr_from_mont(d["generator_inv"]) == pow(7, -1, M.R_MOD)
    assert orc.domain_new(5)["size"] == 8 and orc.domain_new(0)["size"] == 1
    with pytest.raises(ValueError):
        orc.domain_new(1 << 32)


def test_vanishing_and_lagrange(orc):
    n = 16
    vh = M.fr_vec_from_mont(orc.vanishing_poly_over_coset(8 * n, n))
    w = M.domain_root(8 * n)
    assert vh == [(pow(7 * pow(w, i, M.R_MOD), n, M.R_MOD) - 1) % M.R_MOD for i in range(8 * n)]
    assert len(set(vh)) == 8
    tau = M.random_fr_ints(1, 77)[0]
    lag = M.fr_vec_from_mont(orc.lagrange_coefficients(n, M.fr_to_mont(tau)))
    wn = M.domain_root(n)
    zh = (pow(tau, n, M.R_MOD) - 1) % M.R_MOD
    for i in range(n):
        wi = pow(wn, i, M.R_MOD)
        assert lag[i] == zh * wi % M.R_MOD * pow(n * (tau - wi), -1, M.R_MOD) % M.R_MOD
    # tau on the domain: indicator vector
    lag = M.fr_vec_from_mont(orc.lagrange_coefficients(n, M.fr_to_mont(pow(wn, 5, M.R_MOD))))
    assert lag == [1 if i == 5 else 0 for i in range(n)]


# ---- transcript (merlin 3.0.0 is not under /root/reference; strings recalled from merlin's own tests) -------
def test_keccak_f1600_zero_state(orc):
    # first lanes of Keccak-f[1600] applied to the all-zero state (public Keccak team test vector)
    s = orc.keccak_f1600(np.zeros(25, U))
    assert int(s[0]) == 0xF1258F7940E1DDE7 and int(s[1]) == 0x84D5CCF933C0478A


def test_strobe_conformance(orc):
    prf1, prf2 = orc.strobe_conformance()
    assert prf1.hex() == "b48e645ca17c667fd5206ba57a6a228d72d8e1903814d3f17f622996d7cfefb0"
    assert prf2.hex() == "07e45cce8078cee259e3e375bb85d75610e2d1e1201c5f645045a194edd49ff8"


def test_merlin_equivalence_simple(orc):
    got = orc.merlin_simple(b"test protocol", b"some label", b"some data", b"challenge", 32)
    assert got.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
