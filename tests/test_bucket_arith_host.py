"""Host-side check (no GPU) of the arithmetic inside the MSM kernels: 28-bit unsaturated-limb Fq and extended
Jacobian mixed additions, against the oracle's G1 arithmetic, including the exceptional cases of the incomplete
formulas and the operand bounds the lazy reduction relies on."""
import numpy as np
import pytest

import model as M


def _want(orc, pts, neg, inf):
    acc = orc.g1_identity()
    for i in range(len(pts)):
        if inf is not None and inf[i]:
            continue
        p = np.concatenate([pts[i], M.fq_to_mont(1)])
        if neg is not None and neg[i]:
            p[6:12] = orc.fq_op("neg", p[6:12])
        acc = orc.g1_add(acc, p)
    return orc.g1_to_affine(acc)


def _check(orc, pts, neg=None, inf=None):
    from zkvm_amd import api
    want = _want(orc, pts, neg, inf)
    for tree in (False, True):       # point-by-point chain, then pairwise accumulator tree
        out, viol = api.g1_bucket_selfcheck(pts, neg, inf, tree=tree)
        assert viol == 0
        got = api.g1_sum_to_affine(out.reshape(1, 18))
        assert got[1] == int(want[1])
        if not got[1]:
            assert np.array_equal(got[0], want[0])


def test_random_chains(orc):
    pts, _ = orc.g1_multiples_of_generator(3, 400)
    rng = np.random.Generator(np.random.PCG64(1))
    for n in (1, 2, 3, 17, 400):
        idx = rng.integers(0, 400, size=n)
        neg = rng.integers(0, 2, size=n).astype(np.uint8)
        _check(orc, pts[idx], neg)


def test_exceptional_cases(orc):
    pts, _ = orc.g1_multiples_of_generator(1, 8)
    g, g2, g3 = pts[0], pts[1], pts[2]
    z = np.zeros(12, np.uint64)
    cases = [
        ([g, g], [0, 0]),                 # P + P -> doubling path
        ([g, g], [1, 1]),                 # (-P) + (-P)
        ([g, g], [0, 1]),                 # P - P -> identity
        ([g, g, g2], [0, 1, 0]),          # identity, then restart
        ([g, g, g, g], [0, 0, 0, 0]),     # 2P + P (ordinary), 3P + P
        ([g, g, g2, g2], [0, 0, 1, 0]),   # 2P - 2P by value: accumulator equals the negated table point
        ([g, g, g2], [0, 0, 0]),          # 2P + 2P by value -> doubling path with a non-trivial accumulator
        ([g2, g, g], [0, 1, 1]),          # back to identity through two steps
        ([g3, g, g2], [0, 1, 1]),         # 3P - P - 2P
        ([g] * 33, [0] * 33),
        ([g] * 64, [0] * 64),             # tree: every level adds two equal accumulators (doubling path)
        ([g, g2, g, g2], [0, 0, 1, 1]),   # tree: (P + 2P) + (-P - 2P) cancels at the top
        ([g, g, g2, g3], [0, 0, 0, 1]),   # tree: 2P (by doubling) + (2P - 3P)
        ([g] * 6, [0, 1, 0, 1, 1, 0]),
    ]
    for plist, neg in cases:
        _check(orc, np.stack(plist), np.array(neg, np.uint8))
    _check(orc, np.stack([g, z, g2]), None, np.array([0, 1, 0], np.uint8))   # infinity entries are skipped
    _check(orc, np.stack([z, z]), None, np.array([1, 1], np.uint8))
    _check(orc, np.zeros((0, 12), np.uint64))


def test_limb_edge_values(orc):
    """Points whose coordinates sit at the edges of the field do not exist on demand; stress the limb paths instead
    with long alternating chains that keep the accumulator's magnitude at its maximum."""
    pts, _ = orc.g1_multiples_of_generator(1000, 64)
    rng = np.random.Generator(np.random.PCG64(7))
    idx = rng.integers(0, 64, size=3000)
    neg = rng.integers(0, 2, size=3000).astype(np.uint8)
    _check(orc, pts[idx], neg)
