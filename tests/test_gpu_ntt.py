"""GPU parity: zkgpu_ntt / zkgpu_ntt_dev against the CPU oracle's EvaluationDomain, bit for bit on raw limbs."""
import numpy as np
import pytest

import model as M

pytestmark = pytest.mark.gpu

MODES = ["fft", "ifft", "coset_fft", "coset_ifft"]


@pytest.fixture(scope="module")
def ctx():
    import zkvm_amd
    c = zkvm_amd.Context(0)
    yield c
    c.close()


def edge_vectors(n):
    p_minus_1 = M.fr_to_mont(M.R_MOD - 1)
    one = M.fr_to_mont(1)
    delta = np.zeros((n, 4), np.uint64)
    delta[0] = one
    raw_max = M.int_to_limbs(M.R_MOD - 1, 4)          # the largest canonical limb pattern: sums grow fastest
    alt = np.tile(raw_max, (n, 1))
    alt[1::2] = 0                                      # max / 0 alternating: differences stay at the maximum
    return {"zero": np.zeros((n, 4), np.uint64), "delta": delta, "ones": np.tile(one, (n, 1)),
            "r_minus_1": np.tile(p_minus_1, (n, 1)), "raw_max": np.tile(raw_max, (n, 1)), "max_zero": alt}


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 12, 13, 14, 16])
def test_ntt_matches_oracle(ctx, orc, log_n):
    n = 1 << log_n
    x = M.random_fr_mont(n, 1000 + log_n)
    for mode_id, mode in enumerate(MODES):
        got = ctx.ntt(x, log_n, mode_id)
        want = orc.fft(mode, x, n)
        assert np.array_equal(got, want), (mode, log_n)


@pytest.mark.parametrize("log_n", [4, 9, 13, 18])
def test_ntt_edge_vectors(ctx, orc, log_n):
    n = 1 << log_n
    for name, x in edge_vectors(n).items():
        for mode_id, mode in enumerate(MODES):
            assert np.array_equal(ctx.ntt(x, log_n, mode_id), orc.fft(mode, x, n)), (name, mode)


def test_ntt_padding_and_truncation(ctx, orc):
    # shorter input: zero-padded, coset powers over the un-padded coefficients (domain.rs:146,182-185)
    x = M.random_fr_mont(1000 + 3, 5)
    for mode_id, mode in enumerate(MODES):
        assert np.array_equal(ctx.ntt(x, 13, mode_id), orc.fft(mode, x, 1 << 13)), mode
    # longer input: truncated to the domain
    x = M.random_fr_mont(40, 6)
    for mode_id, mode in enumerate(MODES):
        assert np.array_equal(ctx.ntt(x, 5, mode_id), orc.fft(mode, x, 32)), mode
    # empty input
    assert np.array_equal(ctx.ntt(np.zeros((0, 4), np.uint64), 6, 0), np.zeros((64, 4), np.uint64))


def test_ntt_small_vs_bigint_model(ctx):
    n = 64
    vals = M.random_fr_ints(n, 42)
    x = M.fr_vec_to_mont(vals)
    assert M.fr_vec_from_mont(ctx.ntt(x, 6, 0)) == M.naive_dft(vals, n)
    assert M.fr_vec_from_mont(ctx.ntt(x, 6, 3)) == M.naive_dft(vals, n, inverse=True, coset=True)


def test_ntt_device_pointers_full_size(ctx, orc):
    """2^20 points resident in HBM (BASELINE config 1): oracle parity, round trips and linearity."""
    import torch
    log_n = 20
    n = 1 << log_n
    x = M.random_fr_mont(n, 1)
    xd = torch.from_numpy(x.view(np.int64)).cuda()
    yd = ctx.ntt(xd, log_n, 0)
    ctx.sync()
    y = yd.cpu().numpy().view(np.uint64)
    assert np.array_equal(y, orc.fft("fft", x, n))
    # in place inverse round trip
    ctx.ntt(yd, log_n, 1, out=yd)
    ctx.sync()
    assert np.array_equal(yd.cpu().numpy().view(np.uint64), x)
    # coset round trip
    zd = ctx.ntt(xd, log_n, 2)
    ctx.ntt(zd, log_n, 3, out=zd)
    ctx.sync()
    assert np.array_equal(zd.cpu().numpy().view(np.uint64), x)


def test_coset_fft_8n_of_short_polynomial(ctx, orc):
    """coset FFT of an (n+3)-coefficient polynomial zero-padded to 8n (the quotient's use), n = 2^14 here so the
    serial oracle finishes in seconds; the 2^23 case is covered by the round trip below."""
    n = 1 << 14
    x = M.random_fr_mont(n + 3, 2)
    got = ctx.ntt(x, 17, 2)
    assert np.array_equal(got, orc.fft("coset_fft", x, 8 * n))


def test_ntt_2_23_round_trip_and_spot_values(ctx):
    import torch
    log_n = 23
    n = 1 << log_n
    coeffs = M.random_fr_mont((1 << 20) + 3, 2)
    xd = torch.from_numpy(coeffs.view(np.int64)).cuda()
    ev = ctx.ntt(xd, log_n, 2)              # coset_fft of the zero-padded polynomial
    back = ctx.ntt(ev, log_n, 3)            # coset_ifft
    ctx.sync()
    b = back.cpu().numpy().view(np.uint64)
    assert np.array_equal(b[:coeffs.shape[0]], coeffs)
    assert not b[coeffs.shape[0]:].any()
    # spot-check evaluations against Horner in big integers: f(g * w^i)
    ints = M.fr_vec_from_mont(coeffs[:64])     # a 64-coefficient polynomial is enough for a direct check
    xs = torch.from_numpy(coeffs[:64].copy().view(np.int64)).cuda()
    ev2 = ctx.ntt(xs, log_n, 2)
    ctx.sync()
    e = ev2.cpu().numpy().view(np.uint64)
    w = M.domain_root(n)
    for i in (0, 1, 2, 12345, n // 2, n - 1):
        assert M.fr_from_mont(e[i]) == M.poly_eval(ints, 7 * pow(w, i, M.R_MOD) % M.R_MOD)


def test_domain_size_error(ctx):
    import zkvm_amd
    with pytest.raises(zkvm_amd.InvalidEvalDomainSize):
        zkvm_amd.EvaluationDomain(1 << 32)
    d = zkvm_amd.EvaluationDomain(5, ctx)
    assert d.size == 8 and d.log_size_of_group == 3


@pytest.mark.parametrize("tile_log,ept_log", [(6, 1), (7, 3), (9, 2), (10, 1), (12, 1), (12, 2)])
def test_ntt_tile_sizes(ctx, orc, tile_log, ept_log):
    """Every tile size cuts the network into different passes; all of them give the oracle's result."""
    try:
        ctx.ntt_set_params(tile_log, ept_log)
        for log_n in (5, 13, 16):
            n = 1 << log_n
            x = M.random_fr_mont(n - 3, seed=tile_log * 100 + log_n)
            for mode, name in ((0, "fft"), (1, "ifft"), (2, "coset_fft"), (3, "coset_ifft")):
                assert np.array_equal(ctx.ntt(x, log_n, mode), orc.fft(name, x, n)), (name, log_n)
            # zero-padded inputs: the first pass skips the butterflies of its empty rows
            for short in (1, 3, n // 8 + 3, n // 4, n // 2 + 1):
                xs = x[:short]
                for mode, name in ((0, "fft"), (2, "coset_fft"), (3, "coset_ifft")):
                    assert np.array_equal(ctx.ntt(xs, log_n, mode), orc.fft(name, xs, n)), (name, log_n, short)
    finally:
        ctx.ntt_set_params(0, 0)


def test_ntt_fuzz_against_oracle():
    """Random sizes, input lengths, modes and tile geometries (tools/fuzz_ntt.py)."""
    import os
    import runpy
    import sys
    from conftest import ROOT
    argv = sys.argv
    sys.argv = ["fuzz_ntt.py", "150", "5"]
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "fuzz_ntt.py"), run_name="__main__")
    finally:
        sys.argv = argv
