"""The quotient over the 4n coset.  The reference evaluates the quotient's numerator on 8n points
(plonk/src/proof_system/quotient_poly.rs:28,46-51); the library takes every second one of them and restores the seven
top coefficients (keys with widget gates: tests/test_composer.py).  Proofs must equal the oracle's (which follows
the reference over 8n points) bit for bit on either route, and a witness that violates a gate or a copy constraint must
end in the reference's error."""
import numpy as np
import pytest

import model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import zkvm_amd
    c = zkvm_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def srs(orc):
    tau = M.random_fr_mont(1, seed=91)[0]
    xy, inf = orc.srs_powers(orc.g1_generator(), tau, (1 << 13) + 7)
    assert not inf.any()
    return xy


def _circuit(orc, n, seed):
    from zkvm_amd import circuits
    c = circuits.arithmetic_circuit(n, seed=seed)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    return c


def _same(got, want):
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(got[key], want[key]), key


@pytest.mark.parametrize("log_n,expect", [(3, 8), (4, 4), (5, 4), (9, 4), (13, 4)])
def test_route_and_parity(ctx, orc, srs, log_n, expect):
    """n < 16 keeps the 8n coset; from n = 16 on the proof runs over 4n points; forced back to 8n it is the same proof."""
    from zkvm_amd import api, circuits
    n = 1 << log_n
    c = _circuit(orc, n, seed=50 + log_n)
    ctx.srs_load(srs[:n + 7])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], b"half", c["pi_indexes"], threads=8)
    assert op.preprocess_rc == 0
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"half", c["pi_indexes"])
    for s in (1, 2):   # twice: the buffers of the first proof are reused
        bl = circuits.random_blinders(s)
        want = op.prove(c["wires"], c["public_inputs"], bl)
        assert want["rc"] == 0
        got = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
        assert got["coset"] == expect
        _same(got, want)
    ctx.force_full_coset(True)
    try:
        gp8 = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"half", c["pi_indexes"])
        got = gp8.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
        assert got["coset"] == 8
        _same(got, want)
    finally:
        ctx.force_full_coset(False)


def test_key_vectors_over_8n_are_built_on_demand(ctx, orc, srs):
    """A key that proves over the 4n coset still hands out the reference's 8n vectors (inspection, to_bytes, the
    piecewise quotient)."""
    from zkvm_amd import api
    n = 64
    c = _circuit(orc, n, seed=71)
    ctx.srs_load(srs[:n + 7])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], b"half", c["pi_indexes"], threads=4)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"half", c["pi_indexes"])
    for k in (0, 6, 11, 14, 15):
        assert np.array_equal(gp.key_eval8n(k), op.eval8n(k)), k


def test_violated_gate_takes_the_8n_route_and_fails_like_the_reference(ctx, orc, srs):
    from zkvm_amd import api, circuits
    n = 64
    c = _circuit(orc, n, seed=72)
    ctx.srs_load(srs[:n + 7])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], b"half", c["pi_indexes"], threads=4)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"half", c["pi_indexes"])
    bl = circuits.random_blinders(4)
    w = c["wires"].copy()
    w[1, 9, 0] ^= np.uint64(1)
    assert op.prove(w, c["public_inputs"], bl)["rc"] == 2
    with pytest.raises(api.PolynomialDegreeTooLarge):
        gp.prove(w.reshape(-1, 4), c["public_inputs"], bl)
    assert ctx._lib.zkgpu_prover_last_proof_coset(ctx._h) in (0, 8)
    # a wrong public input violates gate 0 only through the PI term
    pi = c["public_inputs"].copy()
    pi[0, 0] ^= np.uint64(2)
    assert op.prove(c["wires"], pi, bl)["rc"] == 2
    with pytest.raises(api.PolynomialDegreeTooLarge):
        gp.prove(c["wires"].reshape(-1, 4), pi, bl)
    # and the key still proves a good witness afterwards, over the 4n coset
    want = op.prove(c["wires"], c["public_inputs"], bl)
    got = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
    assert got["coset"] == 4
    _same(got, want)


def test_broken_copy_constraint_takes_the_8n_route(ctx, orc, srs):
    """Every gate holds but a copied value differs: the permutation accumulator does not return to 1, the numerator
    does not vanish on the domain, and the reference ends in the degree error of a quotient commitment."""
    from zkvm_amd import api, circuits
    n = 128
    c = _circuit(orc, n, seed=73)
    sel = c["selectors"]
    # an addition gate (a + b + d - c = 0) whose left wire copies the previous output
    row = None
    for i in range(2, c["constraints"]):
        if c["sigma_map"][0, i] != i and M.fr_vec_from_mont(sel[4][i:i + 1])[0] == 1:
            row = i
            break
    assert row is not None
    w_int = [M.fr_vec_from_mont(c["wires"][k][row:row + 1])[0] for k in range(4)]
    w = c["wires"].copy()
    w[0, row] = M.fr_vec_to_mont([(w_int[0] + 5) % M.R_MOD])[0]
    w[3, row] = M.fr_vec_to_mont([(w_int[3] - 5) % M.R_MOD])[0]
    ctx.srs_load(srs[:n + 7])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], b"half", c["pi_indexes"], threads=4)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"half", c["pi_indexes"])
    bl = circuits.random_blinders(6)
    want = op.prove(w, c["public_inputs"], bl)
    assert want["rc"] == 2
    with pytest.raises(api.PolynomialDegreeTooLarge):
        gp.prove(w.reshape(-1, 4), c["public_inputs"], bl)
    got = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
    assert got["coset"] == 4
    _same(got, op.prove(c["wires"], c["public_inputs"], bl))


def test_serialized_key_proves_over_the_4n_coset(ctx, orc, srs):
    """A key loaded from the reference's byte format takes the even entries of its stored 8n vectors."""
    from zkvm_amd import api, circuits
    n = 256
    c = _circuit(orc, n, seed=74)
    ctx.srs_load(srs[:n + 7])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], b"half", c["pi_indexes"], threads=4)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"half", c["pi_indexes"])
    blob = gp.to_bytes()
    assert np.array_equal(blob, op.to_bytes())
    gp2 = api.Prover.try_from_bytes(ctx, blob, c["pi_indexes"])
    bl = circuits.random_blinders(7)
    want = op.prove(c["wires"], c["public_inputs"], bl)
    got = gp2.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
    assert got["coset"] == 4
    _same(got, want)
    assert np.array_equal(gp2.to_bytes(), blob)
