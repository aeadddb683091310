"""The Composer mirror (zkvm_amd/composer.py) end to end, the way the reference tests its circuits
(coset-poseidon/src/hades/permutation/gadget.rs:199-277: compile, prove, verify, and a must-fail case): tables from
the composer, proof from the CPU oracle, acceptance by the verifier restatement with its pairing."""
import numpy as np
import pytest

import model as M

R = M.R_MOD


def cubic_circuit(x, claimed=None, bit=1):
    """x^3 + x + 5 = y with y public, a boolean and a select on the side."""
    from zkvm_amd.composer import Composer, Constraint
    c = Composer.initialized()
    y = (x ** 3 + x + 5) % R if claimed is None else claimed
    wx = c.append_witness(x)
    x2 = c.gate_mul(Constraint().mult(1).a(wx).b(wx))
    x3 = c.gate_mul(Constraint().mult(1).a(x2).b(wx))
    s = c.gate_add(Constraint().left(1).right(1).constant(5).a(x3).b(wx))
    wy = c.append_public(y)
    c.assert_equal(s, wy)
    wb = c.append_witness(bit)
    c.component_boolean(wb)
    sel = c.component_select(wb, wx, x2)
    c.assert_equal(sel, wx if bit == 1 else x2)
    one = c.component_select_one(wb, wx)
    c.assert_equal_constant(c.component_select_zero(wb, one), (x if bit else 0))
    k = c.append_constant(77)
    c.assert_equal(k, c.append_constant(77))
    return c


def _oracle_prover(orc, circ, label=b"composer"):
    circ["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(circ["sigma_map"][k]) for k in range(4)])
    tau = M.random_fr_mont(1, seed=5151)[0]
    srs, _ = orc.srs_powers(orc.g1_generator(), tau, circ["n"] + 7)
    op = orc.Prover(circ["n"], circ["constraints"], circ["selectors"], srs, label, circ["pi_indexes"])
    assert op.preprocess_rc == 0
    return op, srs, M.fr_from_mont(tau)


def test_initialized_composer_matches_the_reference_layout():
    """composer.rs:139-195: witnesses 0 and 1, two constant gates, two dummy gates that are satisfied"""
    from zkvm_amd.composer import Composer
    c = Composer.initialized()
    assert c.constraints() == 4 and c.witnesses[:2] == [0, 1]
    for coeff, w in c.gates:
        a, b, o, d = (c.witnesses[i] for i in w)
        assert (coeff["q_m"] * a * b + coeff["q_l"] * a + coeff["q_r"] * b + coeff["q_o"] * o + coeff["q_f"] * d
                + coeff["q_c"]) % R == 0
    t = c.tables()
    assert t["n"] == 4 and t["pi_indexes"] == []
    # copy constraints: every witness used twice sits on one cycle
    assert sorted(t["sigma_map"].reshape(-1).tolist()) == list(range(16))


@pytest.mark.parametrize("bit", [0, 1])
def test_circuit_proves_and_verifies(orc, bit):
    from zkvm_amd import circuits
    import plonk_check
    comp = cubic_circuit(3, bit=bit)
    circ = comp.tables()
    assert circ["public_inputs"].shape[0] == 1 and M.fr_from_mont(circ["public_inputs"][0]) == 35
    # every gate holds and every copy cycle carries one value
    flat = np.concatenate([circ["wires"][k] for k in range(4)])
    sm = circ["sigma_map"].reshape(-1)
    assert np.array_equal(flat, flat[sm])
    op, srs, tau = _oracle_prover(orc, circ)
    proof = op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(1))
    assert proof["rc"] == 0
    sxy, sinf = op.selector_comms()
    h, tau_h = plonk_check.opening_key_g2(tau)
    args = (sxy, sinf, circ["public_inputs"], circ["pi_indexes"], srs[0], h, tau_h, orc)
    assert plonk_check.verify(circ["n"], circ["constraints"], b"composer", proof, *args)
    wrong = M.fr_vec_to_mont([36])
    assert not plonk_check.verify(circ["n"], circ["constraints"], b"composer", proof, sxy, sinf, wrong, circ["pi_indexes"],
                                  srs[0], h, tau_h, orc)


def test_false_statement_cannot_be_proved(orc):
    """The must-fail case: with a wrong claimed output the quotient is no polynomial and the commitment of its
    fourth part fails (PolynomialDegreeTooLarge), as in the reference's hades gadget test."""
    from zkvm_amd import circuits
    circ = cubic_circuit(3, claimed=36).tables()
    op, _, _ = _oracle_prover(orc, circ)
    assert op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(2))["rc"] == 2


@pytest.mark.gpu
def test_gpu_proof_of_a_composed_circuit(orc):
    import zkvm_amd
    from zkvm_amd import api, circuits
    circ = cubic_circuit(12345, bit=1).tables()
    op, srs, _ = _oracle_prover(orc, circ)
    ctx = zkvm_amd.Context(0)
    try:
        ctx.srs_load(srs)
        gp = api.Prover(ctx, circ["n"], circ["constraints"], circ["selectors"].reshape(-1, 4), b"composer", circ["pi_indexes"])
        bl = circuits.random_blinders(3)
        want = op.prove(circ["wires"], circ["public_inputs"], bl)
        got = gp.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
        assert want["rc"] == 0
        for key in ("challenges", "evals", "comms_inf", "comms"):
            assert np.array_equal(got[key], want[key]), key
    finally:
        ctx.close()


def widget_circuit(value=0xbeef, other=0x1234, bits=16, fail=False):
    """Range and logic components: value < 2^bits, and / xor of two witnesses exposed as public inputs."""
    from zkvm_amd.composer import Composer
    c = Composer.initialized()
    a = c.append_witness(value)
    b = c.append_witness(other)
    c.component_range(a, bits // 2 if not fail else (bits // 2 - 4))      # fail: claims 8 bits fewer
    mask = (1 << bits) - 1
    w_and = c.append_logic_and(a, b, bits // 2)
    w_xor = c.append_logic_xor(a, b, bits // 2)
    assert c[w_and] == (value & other & mask) and c[w_xor] == ((value ^ other) & mask)
    c.assert_equal(w_and, c.append_public(c[w_and]))
    c.assert_equal(w_xor, c.append_public(c[w_xor]))
    return c


@pytest.mark.parametrize("bits", [16, 10, 64])
def test_range_and_logic_widgets_prove_and_verify(orc, bits):
    """Circuits with q_range and q_logic rows: the oracle's range/logic quotient and linearization terms and the
    verifier's matching commitments terms (widget/range|logic/verifierkey.rs) agree, i.e. the proof is accepted."""
    from zkvm_amd import circuits
    import plonk_check
    value = 0xbeefcafe12345678 & ((1 << bits) - 1)
    circ = widget_circuit(value, 0x0f0f33335555aaaa, bits).tables()
    assert M.fr_vec_from_mont(circ["selectors"][7]).count(1) >= 1 and any(M.fr_vec_from_mont(circ["selectors"][8]))
    op, srs, tau = _oracle_prover(orc, circ, b"widgets")
    proof = op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(bits))
    assert proof["rc"] == 0
    sxy, sinf = op.selector_comms()
    assert not sinf[7] and not sinf[8]
    h, tau_h = plonk_check.opening_key_g2(tau)
    assert plonk_check.verify(circ["n"], circ["constraints"], b"widgets", proof, sxy, sinf, circ["public_inputs"],
                              circ["pi_indexes"], srs[0], h, tau_h, orc)
    bad = dict(proof)
    bad["evals"] = proof["evals"].copy()
    bad["evals"][6] = M.fr_to_mont((M.fr_from_mont(bad["evals"][6]) + 1) % R)     # d_w_eval feeds the range term
    assert not plonk_check.verify(circ["n"], circ["constraints"], b"widgets", bad, sxy, sinf, circ["public_inputs"],
                                  circ["pi_indexes"], srs[0], h, tau_h, orc)


def test_out_of_range_value_cannot_be_proved(orc):
    from zkvm_amd import circuits
    circ = widget_circuit(0xbeef, 0x1234, 16, fail=True).tables()
    op, _, _ = _oracle_prover(orc, circ, b"widgets")
    assert op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(4))["rc"] == 2


@pytest.mark.gpu
def test_gpu_proof_with_range_and_logic_widgets(orc):
    """The quotient kernel's widget variant and the host-side linearization scalars on a satisfied circuit."""
    import zkvm_amd
    from zkvm_amd import api, circuits
    circ = widget_circuit(0xbeefcafe, 0x0f0f3333, 32).tables()
    op, srs, _ = _oracle_prover(orc, circ, b"widgets")
    ctx = zkvm_amd.Context(0)
    try:
        ctx.srs_load(srs)
        gp = api.Prover(ctx, circ["n"], circ["constraints"], circ["selectors"].reshape(-1, 4), b"widgets", circ["pi_indexes"])
        bl = circuits.random_blinders(5)
        want = op.prove(circ["wires"], circ["public_inputs"], bl)
        got = gp.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
        assert want["rc"] == 0
        for key in ("challenges", "evals", "comms_inf", "comms"):
            assert np.array_equal(got[key], want[key]), key
    finally:
        ctx.close()


def curve_circuit(scalar=0x1234567890abcdef1234567, wrong=False):
    """[scalar] G by the fixed-base component, plus G + [scalar] G by the curve-addition gate, both exposed."""
    from zkvm_amd import composer as C
    c = C.Composer.initialized()
    k = c.append_witness(scalar)
    p = c.component_mul_generator(k)
    want = C.jubjub_mul(C.JUBJUB_GENERATOR, scalar)
    assert (c[p[0]], c[p[1]]) == want
    g = c.append_point(C.JUBJUB_GENERATOR)
    s = c.component_add_point(g, p)
    total = C.jubjub_add(C.JUBJUB_GENERATOR, want)
    assert (c[s[0]], c[s[1]]) == total == C.jubjub_mul(C.JUBJUB_GENERATOR, scalar + 1)
    if wrong:
        total = C.jubjub_add(total, C.JUBJUB_GENERATOR)
    c.assert_equal_public_point(s, total)
    return c


def test_jubjub_helpers():
    from zkvm_amd import composer as C
    g = C.JUBJUB_GENERATOR
    on_curve = lambda p: (-p[0] * p[0] + p[1] * p[1] - 1 - C.EDWARDS_D * p[0] * p[0] * p[1] * p[1]) % R == 0
    assert on_curve(g) and on_curve(C.jubjub_mul(g, 12345))
    assert C.jubjub_mul(g, C.JUBJUB_SCALAR_MODULUS) == C.JUBJUB_IDENTITY          # prime-order subgroup
    for k in (1, 2, 3, 0xdeadbeef, C.JUBJUB_SCALAR_MODULUS - 1):
        naf = C.windowed_naf_2(k)
        assert sum(d << i for i, d in enumerate(naf)) == k and set(naf) <= {-1, 0, 1}
        assert all(not (naf[i] and naf[i + 1]) for i in range(255))                # non-adjacent form


def test_curve_widgets_prove_and_verify(orc):
    """Fixed-base scalar multiplication and curve addition gates: accepted by the verifier's ecc terms
    (widget/ecc/**/verifierkey.rs); a wrong public point cannot be proved."""
    from zkvm_amd import circuits
    import plonk_check
    circ = curve_circuit().tables()
    assert circ["n"] == 512
    op, srs, tau = _oracle_prover(orc, circ, b"curve")
    proof = op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(8))
    assert proof["rc"] == 0
    sxy, sinf = op.selector_comms()
    assert not sinf[9] and not sinf[10]
    h, tau_h = plonk_check.opening_key_g2(tau)
    assert plonk_check.verify(circ["n"], circ["constraints"], b"curve", proof, sxy, sinf, circ["public_inputs"],
                              circ["pi_indexes"], srs[0], h, tau_h, orc)
    bad = curve_circuit(wrong=True).tables()
    op2, _, _ = _oracle_prover(orc, bad, b"curve")
    assert op2.prove(bad["wires"], bad["public_inputs"], circuits.random_blinders(8))["rc"] == 2


@pytest.mark.gpu
def test_gpu_proof_with_curve_widgets(orc):
    import zkvm_amd
    from zkvm_amd import api, circuits
    circ = curve_circuit(0xabcdef0123456789).tables()
    op, srs, _ = _oracle_prover(orc, circ, b"curve")
    ctx = zkvm_amd.Context(0)
    try:
        ctx.srs_load(srs)
        gp = api.Prover(ctx, circ["n"], circ["constraints"], circ["selectors"].reshape(-1, 4), b"curve", circ["pi_indexes"])
        bl = circuits.random_blinders(9)
        want = op.prove(circ["wires"], circ["public_inputs"], bl)
        got = gp.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
        assert want["rc"] == 0
        for key in ("challenges", "evals", "comms_inf", "comms"):
            assert np.array_equal(got[key], want[key]), key
    finally:
        ctx.close()


def _violate_one_widget(circ, w):
    """A witness that breaks gates of widget w and nothing else: a variable all of whose cells lie in rows without an
    arithmetic gate and without another widget's selector (in the row and in the row before it, which reads it as its
    next row), changed in every one of its cells: every copy constraint and every arithmetic gate still hold."""
    n = circ["n"]
    sel = circ["selectors"]
    sigma = circ["sigma_map"].reshape(-1)                      # cell col * n + row -> the next cell of its cycle
    arith = sel[6].any(axis=1)
    other = np.zeros(n, dtype=bool)
    for k in range(4):
        if k != w:
            other |= sel[7 + k].any(axis=1)
    own = np.nonzero(sel[7 + w].any(axis=1))[0]
    for row in np.unique(np.concatenate([own, (own + 1) % n])):
        for col in range(4):
            cells = [col * n + int(row)]
            while int(sigma[cells[-1]]) != cells[0]:
                cells.append(int(sigma[cells[-1]]))
            rows = [c % n for c in cells]
            if any(arith[r] or other[r] or other[r - 1] for r in rows):
                continue
            bad = circ["wires"].copy()
            value = M.fr_from_mont(bad[col, row])
            for c in cells:
                bad[c // n, c % n] = M.fr_to_mont((value + 1) % R)
            flat = bad.reshape(4 * n, 4)
            assert np.array_equal(flat, flat[sigma])           # the copy constraints hold
            return bad
    raise AssertionError("no variable of widget %d stands alone" % w)


@pytest.mark.gpu
@pytest.mark.parametrize("subset", [(0, 1, 2, 3), (0,), (1,), (2,), (3,), (1, 3)])
def test_gpu_proof_with_every_widget_in_one_key(orc, subset):
    """circuits.widget_circuit: range, logic, fixed-base and curve-addition gates in one key of 2^12 rows (one launch per
    widget whose selector is not identically zero, summed per point, then the arithmetic and permutation terms); for
    the subsets the other widgets' selector rows are zeroed, which leaves their gates unconstrained but satisfied.
    Proof = the oracle's over the 4n coset and, forced, over the reference's 8n coset; accepted by the verifier; a
    violated range gate ends in the oracle's error; so does a witness that violates gates of one widget only (every
    arithmetic gate and copy constraint intact: only the row-wise widget check can send it to the 8n coset)."""
    import plonk_check
    import zkvm_amd
    from zkvm_amd import api, circuits
    circ = circuits.widget_circuit(1 << 12, repeats=2)
    for w in range(4):
        if w not in subset:
            circ["selectors"][7 + w] = 0
    op, srs, tau = _oracle_prover(orc, circ, b"all-widgets")
    ctx = zkvm_amd.Context(0)
    try:
        ctx.srs_load(srs)
        gp = api.Prover(ctx, circ["n"], circ["constraints"], circ["selectors"].reshape(-1, 4), b"all-widgets", circ["pi_indexes"])
        bl = circuits.random_blinders(11)
        want = op.prove(circ["wires"], circ["public_inputs"], bl)
        got = gp.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
        assert want["rc"] == 0 and got["coset"] == 4
        for key in ("challenges", "evals", "comms_inf", "comms"):
            assert np.array_equal(got[key], want[key]), key
        if subset == (0, 1, 2, 3):
            sxy, sinf = gp.selector_commitments()
            gxy, _ = orc.g1_to_affine(orc.g1_generator())
            h, tau_h = plonk_check.opening_key_g2(tau)
            assert plonk_check.verify(circ["n"], circ["constraints"], b"all-widgets", got, sxy, sinf, circ["public_inputs"],
                                      circ["pi_indexes"], gxy, h, tau_h, orc)
        if 0 in subset:
            row = int(np.nonzero(circ["selectors"][7].any(axis=1))[0][3])
            bad = circ["wires"].copy()
            bad[1, row, 0] ^= np.uint64(4)
            assert op.prove(bad, circ["public_inputs"], bl)["rc"] == 2
            with pytest.raises(api.PolynomialDegreeTooLarge):
                gp.prove(bad.reshape(-1, 4), circ["public_inputs"], bl)
        for w in subset:
            bad = _violate_one_widget(circ, w)
            assert op.prove(bad, circ["public_inputs"], bl)["rc"] == 2, w
            with pytest.raises(api.PolynomialDegreeTooLarge):
                gp.prove(bad.reshape(-1, 4), circ["public_inputs"], bl)
            assert ctx._lib.zkgpu_prover_last_proof_coset(ctx._h) in (0, 8)
        again = gp.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)     # and the key still proves over 4n
        assert again["coset"] == 4 and np.array_equal(again["comms"], want["comms"])
        if subset == (0, 1, 2, 3):
            # a clone of the context (zkgpu_ctx_clone) proves with the same resident key, widget selectors included
            twin = ctx.clone()
            try:
                cloned = gp.on(twin).prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
                assert cloned["coset"] == 4
                for key in ("challenges", "evals", "comms_inf", "comms"):
                    assert np.array_equal(cloned[key], want[key]), key
                with pytest.raises(api.PolynomialDegreeTooLarge):
                    gp.on(twin).prove(_violate_one_widget(circ, 2).reshape(-1, 4), circ["public_inputs"], bl)
            finally:
                twin.close()
        if subset in ((0, 1, 2, 3), (1, 3)):
            # the reference's byte format (its 8n vectors are built for it on demand) and a key restored from it
            blob = gp.to_bytes()
            assert np.array_equal(blob, op.to_bytes())
            restored = api.Prover.try_from_bytes(ctx, blob, circ["pi_indexes"]).prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
            assert restored["coset"] == 4
            for key in ("challenges", "evals", "comms_inf", "comms"):
                assert np.array_equal(restored[key], want[key]), key
            ctx.srs_load(srs)                                                          # the blob carried its own commit key
        ctx.force_full_coset(True)
        try:
            gp8 = api.Prover(ctx, circ["n"], circ["constraints"], circ["selectors"].reshape(-1, 4), b"all-widgets", circ["pi_indexes"])
            got8 = gp8.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
            assert got8["coset"] == 8
            for key in ("challenges", "evals", "comms_inf", "comms"):
                assert np.array_equal(got8[key], want[key]), key
        finally:
            ctx.force_full_coset(False)
    finally:
        ctx.close()
