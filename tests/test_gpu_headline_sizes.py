"""GPU parity at the sizes the benchmark is quoted on, bit for bit against the CPU oracle on raw limbs: a 2^20-point
commitment over a real tau-power SRS (uniform and witness-like scalars), the 2^23-point coset transforms of the quotient,
one whole 2^20-gate proof, a 2^17-row proof of a key with every widget, and - where the oracle would need minutes - the 2^24-point commitment through properties of
the group.  The oracle legs run for tens of seconds each (the reference's FFT is serial, its MSM window-parallel)."""
import numpy as np
import pytest

import model as M

pytestmark = pytest.mark.gpu

LOG_N = 20
N = 1 << LOG_N


@pytest.fixture(scope="module")
def ctx():
    import zkvm_amd
    c = zkvm_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def srs20(ctx, orc):
    """SRS of a 2^20-gate circuit (2^20 + 7 powers of tau times G), generated on the GPU, exported, and pinned
    independently of the library: the first powers by the oracle's PublicParameters::setup restatement, scattered ones
    by big-integer scalar multiplication."""
    tau = M.random_fr_mont(1, seed=2020)[0]
    g = orc.g1_generator()
    gxy, _ = orc.g1_to_affine(g)
    ctx.srs_setup(tau, gxy, N + 7)
    xy, inf = ctx.srs_export(N + 7)
    assert not inf.any()
    want_xy, want_inf = orc.srs_powers(g, tau, 48)
    assert np.array_equal(xy[:48], want_xy) and not want_inf.any()
    tau_int = M.fr_from_mont(tau)
    g_aff = M.affine_from_limbs(gxy)
    for i in (1 << 10, (1 << 19) + 12345, N - 1, N + 6):
        assert M.affine_from_limbs(xy[i]) == M.ec_mul(g_aff, pow(tau_int, i, M.R_MOD)), i
    return dict(tau=tau, gxy=gxy, xy=xy)


def witness_like(n, seed):
    """30 % zeros, 20 % ones, 20 % below 2^16, 30 % uniform: the shape of a wire column (values in Montgomery form,
    so the digits the bucket method sees are those of 0, 1 and the small integers themselves)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sc = M.random_fr_mont(n, seed)
    kind = rng.random(n)
    sc[kind < 0.3] = 0
    sc[(kind >= 0.3) & (kind < 0.5)] = M.fr_to_mont(1)
    mask = (kind >= 0.5) & (kind < 0.7)
    sc[mask] = M.fr_vec_to_mont([int(v) for v in rng.integers(0, 1 << 16, size=int(mask.sum()))])
    return sc


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["uniform", "witness_like"])
def test_msm_2_20_on_tau_power_srs(ctx, orc, srs20, kind):
    """multiscalar_mul.rs:143-220 + commitment.rs:34-38 at 2^20 points: every limb of the affine sum."""
    import torch
    if kind == "uniform":
        sc = M.random_fr_mont(N, 3)
    else:
        sc = witness_like(N, 4)
    sd = torch.from_numpy(sc.view(np.int64)).cuda()
    got_xy, got_inf = ctx.msm(sd)
    want_xy, want_inf = orc.g1_to_affine(orc.msm(srs20["xy"][:N], sc, threads=16))
    assert got_inf == int(want_inf) and np.array_equal(got_xy, want_xy)
    # the host-pointer entry point and the commitment of the same coefficients agree
    xy2, inf2 = ctx.commit(sc)
    assert inf2 == got_inf and np.array_equal(xy2, got_xy)


@pytest.mark.timeout(600)
def test_coset_transforms_2_23_match_oracle(ctx, orc):
    """fft/domain.rs:176-196 on the 8n domain of a 2^20-gate circuit: coset_fft of an (n+3)-coefficient polynomial and
    coset_ifft of a full vector, all 2^23 outputs."""
    import torch
    n8 = 8 * N
    coeffs = M.random_fr_mont(N + 3, 2)
    ev = ctx.ntt(torch.from_numpy(coeffs.view(np.int64)).cuda(), LOG_N + 3, 2)
    ctx.sync()
    want = orc.fft("coset_fft", coeffs, n8)
    assert np.array_equal(ev.cpu().numpy().view(np.uint64), want)
    del ev
    full = M.random_fr_mont(n8, 23)
    back = ctx.ntt(torch.from_numpy(full.view(np.int64)).cuda(), LOG_N + 3, 3)
    ctx.sync()
    assert np.array_equal(back.cpu().numpy().view(np.uint64), orc.fft("coset_ifft", full, n8))


@pytest.mark.timeout(900)
def test_prove_2_20_equals_oracle_proof(ctx, orc, srs20):
    """compiler/prover.rs:210-509 at the headline size: the 11 commitments, 15 evaluations and 10 challenges of the GPU
    proof against the oracle's on the same circuit, SRS and blinders."""
    import torch
    from zkvm_amd import api, circuits
    c = circuits.arithmetic_circuit(N, seed=20)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    # srs20 left the context holding this SRS
    gp = api.Prover(ctx, N, c["constraints"], c["selectors"].reshape(-1, 4), b"headline", c["pi_indexes"])
    bl = circuits.random_blinders(20)
    wires = torch.from_numpy(c["wires"].reshape(-1, 4).view(np.int64)).cuda()
    got = gp.prove(wires, c["public_inputs"], bl)
    del wires
    op = orc.Prover(N, c["constraints"], c["selectors"], srs20["xy"], b"headline", c["pi_indexes"], threads=16)
    assert op.preprocess_rc == 0
    sxy, sinf = gp.selector_commitments()
    wxy, winf = op.selector_comms()
    assert np.array_equal(sinf, winf) and np.array_equal(sxy, wxy)
    c["selectors"] = None
    want = op.prove(c["wires"], c["public_inputs"], bl)
    assert want["rc"] == 0
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(got[key], want[key]), key


@pytest.mark.timeout(900)
def test_prove_2_17_rows_with_every_widget_equals_oracle_proof(orc, srs20):
    """The same for a key in which every widget has gates (range, logic, fixed-base multiplication, curve addition:
    widget/*/proverkey.rs), 2^17 rows: the proof over the 4n coset equals the oracle's, which follows the reference
    over 8n points."""
    import torch
    import zkvm_amd
    from zkvm_amd import api, circuits
    n = 1 << 17
    c = circuits.widget_circuit(n)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    own = zkvm_amd.Context(0)      # the module's context keeps its 2^20-point SRS
    try:
        own.srs_load(srs20["xy"][:n + 7])
        gp = api.Prover(own, n, c["constraints"], c["selectors"].reshape(-1, 4), b"headline-widgets", c["pi_indexes"])
        bl = circuits.random_blinders(17)
        wires = torch.from_numpy(c["wires"].reshape(-1, 4).view(np.int64)).cuda()
        got = gp.prove(wires, c["public_inputs"], bl)
        assert got["coset"] == 4
    finally:
        own.close()
    op = orc.Prover(n, c["constraints"], c["selectors"], srs20["xy"][:n + 7], b"headline-widgets", c["pi_indexes"], threads=16)
    assert op.preprocess_rc == 0
    want = op.prove(c["wires"], c["public_inputs"], bl)
    assert want["rc"] == 0
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(got[key], want[key]), key


def negate_mont(sc):
    """limb-wise r - s on Montgomery residues (negation commutes with the Montgomery map); zero stays zero"""
    n = sc.shape[0]
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
    return neg


@pytest.mark.timeout(600)
def test_msm_2_24_distinct_points_properties(orc):
    """The largest commitment the reference's configurations name: 2^24 distinct tau-powers.  The oracle needs minutes
    there, so the sum is checked through the group: unequal range shards add up to the whole, a 2^16-point slice of it
    equals the oracle's sum over the exported points, and the sums over s and r - s cancel."""
    import torch
    import zkvm_amd
    from zkvm_amd import circuits
    n = 1 << 24
    c = zkvm_amd.Context(0)
    try:
        c.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n)
        sc = M.random_fr_mont(n, seed=2424)
        sc[:5] = 0
        sd = torch.from_numpy(sc.view(np.int64)).cuda()
        whole = c.msm_partial(sd, 0)
        xy, inf = zkvm_amd.g1_sum_to_affine(whole.reshape(1, 18))
        assert inf == 0
        cuts = [0, 3_000_017, (1 << 23) + 5, 13_000_000, n]
        parts = np.stack([c.msm_partial(sd[cuts[k]:cuts[k + 1]].contiguous(), cuts[k]) for k in range(4)])
        xy2, inf2 = zkvm_amd.g1_sum_to_affine(parts)
        assert inf2 == 0 and np.array_equal(xy, xy2)
        # one slice against the oracle, on the points as exported
        lo, m = cuts[2], 1 << 16
        pts, pinf = c.srs_export(lo + m)
        assert not pinf.any()
        slice_part = c.msm_partial(sd[lo:lo + m].contiguous(), lo)
        sxy, sinf = zkvm_amd.g1_sum_to_affine(slice_part.reshape(1, 18))
        wxy, winf = orc.g1_to_affine(orc.msm(pts[lo:], sc[lo:lo + m], threads=16))
        assert sinf == int(winf) and np.array_equal(sxy, wxy)
        del pts
        other = c.msm_partial(torch.from_numpy(negate_mont(sc).view(np.int64)).cuda(), 0)
        _, inf3 = zkvm_amd.g1_sum_to_affine(np.stack([whole, other]))
        assert inf3 == 1
    finally:
        c.close()


@pytest.mark.timeout(600)
def test_msm_2_22_as_eight_range_shards(orc):
    """The multi-GPU commitment's shape at size: eight contexts, each generating and holding ONLY its own range
    SRS[g n / 8, (g + 1) n / 8) of 2^22 + 7 tau-powers (as sharding.srs_setup_range does per rank) and summing its slice of
    the scalars; the eight 144-byte partial sums added in rank order equal the commitment of one context holding all
    points, and one shard equals the oracle's sum over the same points."""
    import torch
    import zkvm_amd
    from zkvm_amd import circuits, sharding
    n, world = (1 << 22) + 7, 8
    tau, g = circuits.seeded_tau(5), circuits.g1_generator_limbs()
    sc = witness_like(n, 2222)
    sc[1::3] = M.random_fr_mont(len(sc[1::3]), 2223)
    whole_ctx = zkvm_amd.Context(0)
    try:
        whole_ctx.srs_setup(tau, g, n)
        want_xy, want_inf = whole_ctx.msm(torch.from_numpy(sc.view(np.int64)).cuda())
        pts, _ = whole_ctx.srs_export(n)
    finally:
        whole_ctx.close()
    parts = []
    for rank in range(world):
        base, count = sharding.shard_range(n, rank, world)
        c = zkvm_amd.Context(0)
        try:
            sharding.srs_setup_range(c, tau, g, base, count)
            xy, inf = c.srs_export(count)
            assert not inf.any() and np.array_equal(xy, pts[base:base + count]), rank
            part = c.msm_partial(torch.from_numpy(sc[base:base + count].view(np.int64)).cuda(), 0)
            parts.append(part)
            if rank == 5:
                sxy, sinf = zkvm_amd.g1_sum_to_affine(part.reshape(1, 18))
                m = 1 << 17        # the oracle's leg: the shard's first 2^17 points
                wxy, winf = orc.g1_to_affine(orc.msm(xy[:m], sc[base:base + m], threads=16))
                head = c.msm_partial(torch.from_numpy(sc[base:base + m].view(np.int64)).cuda(), 0)
                hxy, hinf = zkvm_amd.g1_sum_to_affine(head.reshape(1, 18))
                assert hinf == int(winf) and np.array_equal(hxy, wxy)
        finally:
            c.close()
    got_xy, got_inf = sharding.combine(np.stack(parts))
    assert got_inf == want_inf == 0 and np.array_equal(got_xy, want_xy)


@pytest.mark.timeout(900)
def test_merkle_batch_circuit_2_24_gates_proof_accepted(orc):
    """The reference's largest configuration with its own gate mix: 519 openings of a height-32 Poseidon-Merkle tree in
    one circuit (16.76 M gates, n = 2^24, 8n = 2^27-point transforms, 519 public roots), generated by the library's
    native front-end, proved on one GPU and accepted by the restated verifier with its pairing; a tampered proof is
    refused.  The CPU oracle would need about half an hour for this proof."""
    import torch
    import plonk_check
    import zkvm_amd
    from zkvm_amd import api, circuits, poseidon as P
    height, k = 32, 519
    tree = P.Tree(height, native=True)
    rng = np.random.default_rng(24)
    spots = sorted({int(p) for p in rng.integers(0, 1 << 62, k + 8)})[:k]      # 4^32 = 2^64 leaf positions
    for p in spots:
        tree.insert(p, P.hash_to_scalar(p.to_bytes(8, "little")))
    openings = [tree.opening(p) for p in spots]
    leaves = [tree.leaves[p] for p in spots]
    assert openings[7].verify(leaves[7]) and not openings[7].verify(leaves[8])
    circ = P.batch_opening_circuit(openings, leaves, threads=16)
    n = circ["n"]
    assert n == 1 << 24 and circ["constraints"] == 4 + k * 32289 and len(circ["pi_indexes"]) == k
    ctx = zkvm_amd.Context(0)
    try:
        tau = M.random_fr_mont(1, seed=4040)[0]
        gxy, _ = orc.g1_to_affine(orc.g1_generator())
        ctx.srs_setup(tau, gxy, n + 7)
        circuits.fill_sigma_rows(circ, ctx)
        gp = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"merkle-batch", circ["pi_indexes"])
        sxy, sinf = gp.selector_commitments()
        circ["selectors"] = None
        wires = torch.from_numpy(circ["wires"].reshape(-1, 4).view(np.int64)).cuda()
        proof = gp.prove(wires, circ["public_inputs"], circuits.random_blinders(24))
        del wires
        tau_int = M.fr_from_mont(tau)
        h, tau_h = plonk_check.opening_key_g2(tau_int)
        assert plonk_check.verify(n, circ["constraints"], b"merkle-batch", proof, sxy, sinf, circ["public_inputs"],
                                  circ["pi_indexes"], gxy, h, tau_h, orc)
        bad = dict(proof)
        bad["evals"] = proof["evals"].copy()
        bad["evals"][3, 0] ^= np.uint64(1)
        assert not plonk_check.verify(n, circ["constraints"], b"merkle-batch", bad, sxy, sinf, circ["public_inputs"],
                                      circ["pi_indexes"], gxy, h, tau_h, orc)
    finally:
        ctx.close()
