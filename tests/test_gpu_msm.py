"""GPU parity: zkgpu_msm / zkgpu_commit against the CPU oracle's msm_variable_base + affine conversion."""
import os

import numpy as np
import pytest

import model as M
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import zkvm_amd
    c = zkvm_amd.Context(0)
    yield c
    c.close()


def oracle_affine(orc, pts, sc, inf=None):
    return orc.g1_to_affine(orc.msm(pts, sc, inf=inf, threads=8))


def check(ctx, orc, pts, sc, inf=None):
    ctx.srs_load(pts, inf)
    got_xy, got_inf = ctx.msm(sc)
    want_xy, want_inf = oracle_affine(orc, pts, sc, inf)
    assert got_inf == want_inf
    assert np.array_equal(got_xy, want_xy)


def test_msm_k_times_g_vectors(ctx, orc):
    """All bases = G: the result is (sum of scalars) * G, pinned by the reference's k*G vectors."""
    import zkvm_amd
    data = open(os.path.join(GOLDEN, "g1_compressed_valid_test_vectors.dat"), "rb").read()
    gxy, _ = orc.g1_to_affine(orc.g1_generator())
    for n, scalars in ((1, [100]), (5, [0, 1, 1, 3, 7]), (40, list(range(40))), (33, [1] * 33), (64, [0] * 63 + [999])):
        ctx.srs_load(np.tile(gxy, (n, 1)))
        xy, inf = ctx.msm(M.fr_vec_to_mont(scalars))
        k = sum(scalars)
        assert zkvm_amd.g1_compress(xy, inf) == data[48 * k:48 * k + 48], (n, k)
    # sum == 0 -> identity, encoded as the reference's affine identity
    ctx.srs_load(np.tile(gxy, (3, 1)))
    xy, inf = ctx.msm(M.fr_vec_to_mont([0, 0, 0]))
    assert inf == 1 and zkvm_amd.g1_compress(xy, inf) == data[:48]
    xy, inf = ctx.msm(M.fr_vec_to_mont([5, M.R_MOD - 5, 0]))
    assert inf == 1


@pytest.mark.parametrize("n", [1, 2, 31, 32, 100, 1000, 5000])
def test_msm_random_small(ctx, orc, n):
    pts, _ = orc.g1_multiples_of_generator(1, n)
    sc = M.random_fr_mont(n, 50 + n)
    check(ctx, orc, pts, sc)


def test_msm_edge_scalars(ctx, orc):
    n = 2048
    pts, _ = orc.g1_multiples_of_generator(3, n)
    r_minus_1 = M.fr_to_mont(M.R_MOD - 1)
    check(ctx, orc, pts, np.tile(r_minus_1, (n, 1)))                      # all r-1: one giant bucket per window
    check(ctx, orc, pts, np.tile(M.fr_to_mont(1), (n, 1)))                # all ones
    check(ctx, orc, pts, np.tile(M.random_fr_mont(1, 9), (n, 1)))         # all equal
    check(ctx, orc, pts, np.zeros((n, 4), np.uint64))                     # all zero -> identity
    # witness-like: 30% zeros, 20% ones, 20% < 2^16, 30% uniform
    rng = np.random.Generator(np.random.PCG64(4))
    sc = M.random_fr_mont(n, 4)
    kind = rng.random(n)
    small = M.fr_vec_to_mont([int(v) for v in rng.integers(0, 1 << 16, size=n)])
    sc[kind < 0.3] = 0
    sc[(kind >= 0.3) & (kind < 0.5)] = M.fr_to_mont(1)
    mask = (kind >= 0.5) & (kind < 0.7)
    sc[mask] = small[mask]
    check(ctx, orc, pts, sc)


def test_msm_infinity_points_and_truncation(ctx, orc):
    pts, inf = orc.g1_multiples_of_generator(0, 64)     # element 0 is the point at infinity
    sc = M.random_fr_mont(64, 8)
    check(ctx, orc, pts, sc, inf)
    # more scalars than points: truncated like the reference's zip
    pts, _ = orc.g1_multiples_of_generator(1, 50)
    sc = M.random_fr_mont(80, 12)
    ctx.srs_load(pts)
    got_xy, got_inf = ctx.msm(sc)
    want_xy, want_inf = oracle_affine(orc, pts, sc[:50])
    assert got_inf == want_inf and np.array_equal(got_xy, want_xy)
    # fewer scalars than points
    got_xy, _ = ctx.msm(sc[:20])
    want_xy, _ = oracle_affine(orc, pts[:20], sc[:20])
    assert np.array_equal(got_xy, want_xy)


def test_commit_degree_checks(ctx, orc):
    import zkvm_amd
    pts, _ = orc.g1_multiples_of_generator(1, 16)
    ck = zkvm_amd.CommitKey(ctx, pts)
    assert ck.max_degree() == 15
    with pytest.raises(zkvm_amd.PolynomialDegreeIsZero):
        ck.commit(np.zeros((4, 4), np.uint64))
    with pytest.raises(zkvm_amd.PolynomialDegreeIsZero):
        ck.commit(M.fr_vec_to_mont([5, 0, 0]))
    with pytest.raises(zkvm_amd.PolynomialDegreeTooLarge):
        ck.commit(M.random_fr_mont(17, 1))
    poly = M.random_fr_mont(16, 2)
    xy, inf = ck.commit(poly)
    want_xy, want_inf = oracle_affine(orc, pts, poly)
    assert inf == want_inf and np.array_equal(xy, want_xy)
    # leading zeros beyond the key length are fine: degree is what counts
    padded = np.concatenate([poly, np.zeros((8, 4), np.uint64)])
    xy2, _ = ck.commit(padded)
    assert np.array_equal(xy2, xy)


@pytest.mark.parametrize("n", [300, 5000, 1 << 15])
def test_commit_many_ragged_batches(ctx, orc, n):
    """zkgpu_commit_many: up to four polynomials of very different lengths in one batched pass (shared sort,
    accumulate and reduction launches) give the commitments of four separate zkgpu_commit calls = the oracle's."""
    import zkvm_amd
    pts, _ = orc.g1_multiples_of_generator(5, n)
    ctx.srs_load(pts)
    lens = [n, 2, n // 3 + 1, n - 1]
    polys = [M.random_fr_mont(k, 900 + k) for k in lens]
    polys[3][n // 2:] = 0                                   # leading zeros: the degree is what counts
    polys[2][::2] = M.fr_to_mont(M.R_MOD - 1)               # half the scalars equal: deep buckets
    for m in (4, 3, 2, 1):
        got = ctx.commit_many(polys[:m])
        assert len(got) == m
        for j in range(m):
            want_xy, want_inf = oracle_affine(orc, pts[:lens[j]], polys[j])
            assert got[j][1] == want_inf and np.array_equal(got[j][0], want_xy), (m, j)
    one_xy, _ = ctx.commit(polys[2])
    assert np.array_equal(ctx.commit_many([polys[2]])[0][0], one_xy)
    # the first failing degree check is the error returned, before any work (prover.rs:262-265)
    with pytest.raises(zkvm_amd.PolynomialDegreeIsZero):
        ctx.commit_many([polys[0], M.fr_vec_to_mont([7, 0, 0]), M.random_fr_mont(n + 5, 1)])
    with pytest.raises(zkvm_amd.PolynomialDegreeTooLarge):
        ctx.commit_many([polys[0], M.random_fr_mont(n + 5, 1), M.fr_vec_to_mont([7])])
    with pytest.raises(zkvm_amd.ZkGpuError):
        ctx.commit_many([polys[0]] * 5)


def test_window_width_by_table_size(orc):
    """The measured choices of the window model: 13-bit windows up to an SRS for 2^13 gates, 16 bits up to 2^17
    points, 20 above (and never more: 22 bits at 2^24 points was 20 % slower)."""
    import zkvm_amd
    from zkvm_amd import circuits
    c = zkvm_amd.Context(0)
    try:
        for n, want in ((1 << 10, 13), ((1 << 13) + 7, 13), ((1 << 14) + 7, 16), ((1 << 17) + 7, 16), ((1 << 18) + 7, 20),
                        ((1 << 20) + 7, 20)):
            c.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n)
            bits, levels = c.msm_get_params()
            assert (bits, levels) == (want, 255 // want + 1), n
    finally:
        c.close()


def test_msm_2_16_device_scalars_and_sharding(ctx, orc):
    """65536 distinct points (tau-powers SRS), scalars resident in HBM; the two half-range partial sums add up to
    the full result (the point-range sharding used across GPUs)."""
    import torch
    import zkvm_amd
    n = 1 << 16
    tau = M.fr_to_mont(0x1234567890abcdef1234567890abcdef % M.R_MOD)
    pts, _ = orc.srs_powers(orc.g1_generator(), tau, 256)
    # a 2^16 SRS by scalar multiplication would take minutes on the CPU: use consecutive multiples instead
    pts, _ = orc.g1_multiples_of_generator(7, n)
    sc = M.random_fr_mont(n, 3)
    ctx.srs_load(pts)
    sd = torch.from_numpy(sc.view(np.int64)).cuda()
    got_xy, got_inf = ctx.msm(sd)
    want_xy, want_inf = oracle_affine(orc, pts, sc)
    assert got_inf == want_inf and np.array_equal(got_xy, want_xy)
    half = n // 2
    p0 = ctx.msm_partial(sd[:half].contiguous(), 0)
    p1 = ctx.msm_partial(sd[half:].contiguous(), half)
    xy, inf = zkvm_amd.g1_sum_to_affine(np.stack([p0, p1]))
    assert inf == want_inf and np.array_equal(xy, want_xy)


def test_partial_range_arguments_cannot_wrap(ctx, orc):
    """A huge base must be refused as an argument error, not wrap around the range check and index past the table."""
    import ctypes as C
    import torch
    import zkvm_amd
    pts, _ = orc.g1_multiples_of_generator(1, 64)
    ctx.srs_load(pts)
    sd = torch.from_numpy(M.random_fr_mont(64, 5).view(np.int64)).cuda()
    out = np.zeros(18, dtype=np.uint64)
    size_max = C.c_size_t(-1).value
    for n, base in ((2, size_max - 1), (64, size_max - 63), (1, 64), (65, 0), (2, 63)):
        rc = ctx._lib.zkgpu_msm_partial_dev(ctx._h, sd.data_ptr(), n, base, out.ctypes.data)
        assert rc == -1, (n, base)
    assert ctx._lib.zkgpu_msm_partial_dev(ctx._h, sd.data_ptr(), 1, 63, out.ctypes.data) == 0


def test_msm_repeated_scalars_default_window(ctx, orc):
    """Runs of equal scalars give buckets from a few entries to thousands on every level: below the oversize
    threshold they stay with one thread, above it they are split into workgroup tasks (one and several chunks)."""
    n = 1 << 14
    pts, _ = orc.g1_multiples_of_generator(5, n)
    sc = M.random_fr_mont(n, 77)
    pos = 0
    for run in (3, 40, 63, 64, 65, 66, 100, 255, 256, 257, 700, 4096, 4500):
        sc[pos:pos + run] = sc[pos]
        pos += run
    assert pos < n
    check(ctx, orc, pts, sc)


@pytest.mark.parametrize("window_bits,segment_bits", [(4, 1), (5, 3), (9, 2), (13, 4), (16, 3), (18, 6), (21, 5), (22, 4)])
def test_msm_window_and_segment_widths(ctx, orc, window_bits, segment_bits):
    """The table of window multiples and the bucket reduction are exact for every window width (1 to 64 levels,
    8 to 2^20 buckets) and segment width; partial ranges index the table with their base."""
    n = 700
    pts, _ = orc.g1_multiples_of_generator(2, n)
    sc = M.random_fr_mont(n, 900 + window_bits)
    sc[0] = 0
    sc[1] = M.fr_to_mont(1)
    sc[2] = M.fr_to_mont(M.R_MOD - 1)
    sc[3:40] = sc[3]                      # one oversized bucket per level
    try:
        ctx.msm_set_params(window_bits, segment_bits)
        check(ctx, orc, pts, sc)
        import torch
        import zkvm_amd
        sd = torch.from_numpy(sc[100:300].view(np.int64)).cuda()
        part = ctx.msm_partial(sd, 100)
        xy, inf = zkvm_amd.g1_sum_to_affine(part.reshape(1, 18))
        want_xy, want_inf = oracle_affine(orc, pts[100:300], sc[100:300])
        assert inf == want_inf and np.array_equal(xy, want_xy)
    finally:
        ctx.msm_set_params(0, 0)


def test_msm_2_22_properties(ctx, orc):
    """At the benchmark's size the oracle is too slow, so the sum is checked through properties of the group:
    range shards add up to the whole, and the sum over s plus the sum over (r - s) is the identity."""
    import torch
    import zkvm_amd
    from zkvm_amd import circuits
    n = 1 << 22
    base, _ = orc.g1_multiples_of_generator(1, 1 << 12)
    pts = torch.from_numpy(base.view(np.int64)).cuda().repeat(n >> 12, 1).contiguous()
    ctx.srs_load(pts)
    del pts
    sc = M.random_fr_mont(n, seed=2222)
    sc[:7] = 0
    sd = torch.from_numpy(sc.view(np.int64)).cuda()
    whole = ctx.msm_partial(sd, 0)
    xy, inf = zkvm_amd.g1_sum_to_affine(whole.reshape(1, 18))
    assert inf == 0
    # three unequal shards
    cuts = [0, 1_000_003, 3 * (1 << 20) + 17, n]
    parts = np.stack([ctx.msm_partial(sd[cuts[k]:cuts[k + 1]].contiguous(), cuts[k]) for k in range(3)])
    xy2, inf2 = zkvm_amd.g1_sum_to_affine(parts)
    assert inf2 == 0 and np.array_equal(xy, xy2)
    # negated scalars: limb-wise r - s on the Montgomery residues (negation commutes with the Montgomery map)
    r_limbs = np.array([0xffffffff00000001, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48], dtype=np.uint64)
    neg = np.zeros_like(sc)
    borrow = np.zeros(n, dtype=np.uint64)
    for k in range(4):
        a = np.full(n, r_limbs[k], dtype=np.uint64)
        t = a - sc[:, k]
        b1 = (a < sc[:, k]).astype(np.uint64)
        t2 = t - borrow
        b2 = (t < borrow).astype(np.uint64)
        neg[:, k] = t2
        borrow = b1 | b2
    neg[(sc == 0).all(axis=1)] = 0
    other = ctx.msm_partial(torch.from_numpy(neg.view(np.int64)).cuda(), 0)
    _, inf3 = zkvm_amd.g1_sum_to_affine(np.stack([whole, other]))
    assert inf3 == 1


def test_short_top_window_goes_through_the_task_path(ctx, orc):
    """255 = 14 * 18 + 3: with an 18-bit window the top digits of all scalars fall on 8 buckets.  Those must be cut into
    workgroup tasks; left to one thread each they are chains of n/8 dependent additions (a hundred times slower)."""
    import time
    import torch
    n = 1 << 16
    pts, _ = orc.g1_multiples_of_generator(9, n)
    sc = M.random_fr_mont(n, 55)
    try:
        ctx.msm_set_params(18, 0)
        ctx.srs_load(pts)
        sd = torch.from_numpy(sc.view(np.int64)).cuda()
        got_xy, got_inf = ctx.msm(sd)
        t0 = time.perf_counter()
        ctx.msm(sd)
        elapsed = time.perf_counter() - t0
    finally:
        ctx.msm_set_params(0, 0)
    want_xy, want_inf = oracle_affine(orc, pts, sc)
    assert got_inf == want_inf and np.array_equal(got_xy, want_xy)
    assert elapsed < 0.03, elapsed


def test_kzg_opening_checked_by_pairing(ctx, orc):
    """The reference's KZG tests (kzg10/key.rs:341-459) check commit + open with the pairing:
    e(C - v G, h) = e(W, [tau] h - z h) for W = commit((p - v) / (X - z)).  Commitment, evaluation and the Ruffini
    quotient come from the GPU, the pairing from the oracle."""
    import plonk_check
    from oracle import pairing as Pg
    from zkvm_amd import api
    n = 200
    tau = M.random_fr_mont(1, seed=321)[0]
    tau_int = M.fr_from_mont(tau)
    srs, _ = orc.srs_powers(orc.g1_generator(), tau, n)
    ctx.srs_load(srs)
    poly = M.random_fr_mont(n, 11)
    z = M.random_fr_mont(1, 12)[0]
    c_xy, c_inf = ctx.commit(poly)
    v = api.poly_eval(ctx, poly, z)
    w_poly = api.poly_ruffini(ctx, poly, z)          # (p - p(z)) / (X - z): the remainder is dropped
    w_xy, w_inf = ctx.commit(w_poly)
    g = M.affine_from_limbs(srs[0])
    C, W = M.affine_from_limbs(c_xy, c_inf), M.affine_from_limbs(w_xy, w_inf)
    v_int, z_int = M.fr_from_mont(v), M.fr_from_mont(z)
    vg = M.ec_mul(g, v_int)
    lhs = M.ec_add(C, None if vg is None else (vg[0], (-vg[1]) % M.P_MOD))
    h, tau_h = plonk_check.opening_key_g2(tau_int)
    rhs_g2 = Pg.g2_add(tau_h, Pg.g2_neg(Pg.g2_mul(h, z_int)))
    neg_w = (W[0], (-W[1]) % M.P_MOD)
    assert Pg.pairing_product_is_one([(lhs, h), (neg_w, rhs_g2)])
    wrong = M.ec_add(lhs, g)
    assert not Pg.pairing_product_is_one([(wrong, h), (neg_w, rhs_g2)])


def test_msm_fuzz_against_oracle():
    """Random small point sets with repeated and negated points, points at infinity, colliding scalars and random
    window / segment widths (tools/fuzz_msm.py): the exceptional cases of the bucket arithmetic, found by chance."""
    import os
    import runpy
    import sys
    from conftest import ROOT
    argv = sys.argv
    sys.argv = ["fuzz_msm.py", "150", "7"]
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "fuzz_msm.py"), run_name="__main__")
    finally:
        sys.argv = argv
