"""The circuits the reference proves, built by the host-side mirror (zkvm_amd/poseidon.py) and run the way the
reference tests them: known-answer hashes (coset-poseidon/src/hades.rs:73-141), the Hades gadget's compile / prove /
verify / must-fail cases (coset-poseidon/src/hades/permutation/gadget.rs:182-277), openings of the Poseidon-Merkle
tree (coset-merkle/src/opening.rs:254-284) and the opening circuit (merkle-plonk/src/lib.rs:117-131)."""
import numpy as np
import pytest

import model as M

R = M.R_MOD

# coset-poseidon/src/hades.rs:73-84 (little-endian hex) and :113-141 (big-endian hex of the digests)
KAT_INPUTS = [
    "bb67ed265bf1db490ded2e1ede55c0d14c55521509dc73f9c354e98ab76c9625",
    "7e74220084d75e10c89e9435d47bb5b8075991b2e29be3b84421dac3b1ee6007",
    "5ce5481a4d78cca03498f72761da1b9f1d2aa8fb300be39f0e4fe2534f9d4308",
    "b1e710e3c4a8c35154b0ce4e4f4af6f498ebd79f8e7cdf3150372c7501be250b",
    "33c9e2025f86b5d82149f1ab8e20a168fc3d99d09b48cbce0286db8752cc3306",
    "e98206bfdce791e4e5144079b997d4fc25006194b35655f0e48490b26e24ea35",
    "86d2a95cc552de8d5bb20bd4a407fee5ffdc314e93dfe6b2dc792bc71fd8cc2d",
    "4edd8307ce28a8c70963d20a7bc28df1e1720bbbc93878a18bd07fad7d51fa15",
    "eabc7a296704a68aa01f95adc85f6dd758b175745336d8fc795a17984024b21e",
    "cfc108673c93df305e31c283b9c767b7097ae4e174a223e0c24b15a67b701a3a",
]
KAT_DIGESTS = {
    3: 0x26abf2d0476f154e69bf19740092fe36265680c294462b8e759ad73a99567dd5,
    4: 0x1cc40219c7ec92919d6db7a41cd41953333a2ed544606daca182e4eaa6c7db2d,
    5: 0x707c98a0e9a6e4832ac33ee08811bce122017a58dbbbf66a2f6fcdc69d45462d,
    6: 0x26905a794d3d2fb0c3ed2276abc696c27a5bfdea7f106e596cbeedd86891c461,
    8: 0x1b98a2c5f1fe54d21b5ce9bf0dcc99ea8784a64f3c544fa06d3f73569741006e,
    10: 0x211b7ea21c9afca93dabdfbda8b2d5275b2dd802fed87bb431e98557c61667d2,
}
LABEL = b"hades_gadget_tester"         # gadget.rs:206


def gates_hold(c):
    """every arithmetic gate of a composer is satisfied by its witnesses (public inputs included)"""
    for index, (k, w) in enumerate(c.gates):
        a, b, o, d = (c.witnesses[i] for i in w)
        pi = c.public_inputs.get(index, 0)
        if k["q_arith"] and (k["q_m"] * a * b + k["q_l"] * a + k["q_r"] * b + k["q_o"] * o + k["q_f"] * d
                             + k["q_c"] + pi) % R:
            return False
    return True


def seeded_state(seed):
    return [M.fr_from_mont(x) for x in M.random_fr_mont(5, seed=seed)]


def test_known_answer_hashes():
    from zkvm_amd import poseidon as P

    class ZeroTag(P.ScalarPermutation):            # hades.rs:52-65: the test's own Safe with tag = 0
        def tag(self, data):
            return 0

    values = [int.from_bytes(bytes.fromhex(h), "little") for h in KAT_INPUTS]
    for count, want in KAT_DIGESTS.items():
        sponge = P.Sponge(ZeroTag(), [(P.ABSORB, count), (P.ABSORB, 1), (P.SQUEEZE, 1)], 0)
        sponge.absorb(count, values[:count])
        sponge.absorb(1, [1])
        sponge.squeeze(1)
        assert sponge.finish()[0] == want, count


def test_constant_tables():
    """hades/round_constants.rs:41-50: canonical, none zero; the MDS matrix likewise"""
    from zkvm_amd import poseidon as P
    assert len(P.ROUND_CONSTANTS) == 68 and all(len(r) == 5 for r in P.ROUND_CONSTANTS)
    assert all(0 < v < R for row in P.ROUND_CONSTANTS + P.MDS_MATRIX for v in row)


def test_permutation_is_deterministic():
    """hades/permutation/scalar.rs:88-100"""
    from zkvm_amd import poseidon as P
    x = P.ScalarPermutation().permute([17] * 5)
    assert x == P.ScalarPermutation().permute([17] * 5) != P.ScalarPermutation().permute([19] * 5)


def test_sponge_rejects_calls_off_the_pattern():
    from zkvm_amd import poseidon as P
    with pytest.raises(P.IOPatternViolation):
        P.Sponge(P.ScalarPermutation(), [(P.SQUEEZE, 1)], 0)
    with pytest.raises(P.IOPatternViolation):
        P.digest(P.Domain.MERKLE4, [1, 2, 3])                       # hash.rs:56-58
    with pytest.raises(P.IOPatternViolation):
        P.digest(P.Domain.MERKLE2, [1, 2, 3])
    s = P.Sponge(P.ScalarPermutation(), [(P.ABSORB, 2), (P.SQUEEZE, 1)], 0)
    with pytest.raises(P.IOPatternViolation):
        s.absorb(3, [1, 2, 3])
    s = P.Sponge(P.ScalarPermutation(), [(P.ABSORB, 2), (P.SQUEEZE, 1)], 0)
    s.absorb(2, [1, 2])
    with pytest.raises(P.IOPatternViolation):
        s.finish()
    # neighbouring calls of one kind are one call to the tag; the domain separator tells hashes apart
    assert P.tag_input([(P.ABSORB, 1), (P.ABSORB, 3), (P.SQUEEZE, 1)], 15) == P.tag_input([(P.ABSORB, 4), (P.SQUEEZE, 1)], 15)
    assert P.tag_input([(P.ABSORB, 4), (P.SQUEEZE, 1)], 15) == bytes.fromhex("80000004" "00000001" "000000000000000f")
    assert P.digest(P.Domain.MERKLE4, [1, 2, 3, 4]) != P.digest(P.Domain.OTHER, [1, 2, 3, 4])
    assert len(P.digest(P.Domain.OTHER, [1, 2, 3, 4, 5, 6], output_len=3)) == 3
    assert len(P.digest(P.Domain.MERKLE2, [1, 2], output_len=3)) == 1          # hash.rs:86-90


def test_gadget_follows_the_scalar_permutation():
    """The gates reproduce the values: same outputs, every gate satisfied, and the gate count of the layout
    (5 for the first round's constants, 10 per MDS layer, 3 per S-box)."""
    from zkvm_amd import poseidon as P
    state = seeded_state(0xbeef)
    want = P.ScalarPermutation().permute(state)
    c = P.permutation_circuit(state, want)
    assert gates_hold(c)
    assert c.constraints() == 4 + 5 + 68 * 10 + (8 * 5 + 60) * 3 + 5 and c.size() == 1024
    c2 = P.Composer.initialized()
    out = P.GadgetPermutation(c2).permute([c2.append_witness(v) for v in state])
    assert [c2[w] for w in out] == want
    # the hash gadget and the hash agree, whatever the domain
    for domain, values in ((P.Domain.MERKLE4, [5, 6, 7, 8]), (P.Domain.MERKLE2, [9, 10]), (P.Domain.OTHER, list(range(1, 10)))):
        c3 = P.Composer.initialized()
        got = P.digest_gadget(c3, domain, [c3.append_witness(v) for v in values])
        assert [c3[w] for w in got] == P.digest(domain, values) and gates_hold(c3)


def test_tree_openings():
    """coset-merkle/src/opening.rs:254-284 with Poseidon items: each opening verifies its own leaf only"""
    from zkvm_amd import poseidon as P
    tree = P.Tree(3)
    assert tree.capacity() == 64 and tree.root() == 0 and tree.opening(5) is None
    leaves = {pos: P.digest(P.Domain.OTHER, [pos + 1])[0] for pos in (0, 5, 6, 21, 63)}
    for pos, leaf in leaves.items():
        tree.insert(pos, leaf)
    for pos, leaf in leaves.items():
        o = tree.opening(pos)
        assert o.root == tree.root() and len(o.branch) == 3 and all(len(level) == 4 for level in o.branch)
        assert o.positions == [pos // 16, pos // 4 % 4, pos % 4]
        assert o.verify(leaf) and not o.verify(leaf + 1)
    # leaves 5 and 6 share their parent; an absent sibling counts as zero
    assert tree.opening(5).branch[2] == [0, leaves[5], leaves[6], 0]
    with pytest.raises(IndexError):
        tree.insert(64, 1)


def merkle_circuit(height, position, seed=7, wrong_leaf=False):
    from zkvm_amd import poseidon as P
    tree = P.Tree(height)
    rng = np.random.default_rng(seed)
    spots = {position} | {int(p) for p in rng.integers(0, tree.capacity(), 6)}
    for p in spots:
        tree.insert(p, P.digest(P.Domain.OTHER, [p, seed])[0])
    leaf = tree.leaves[position]
    return P.opening_circuit(tree.opening(position), leaf + (1 if wrong_leaf else 0)), tree


def _prover(orc, circ, label):
    circ["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(circ["sigma_map"][k]) for k in range(4)])
    tau = M.random_fr_mont(1, seed=0xbeef)[0]
    srs, _ = orc.srs_powers(orc.g1_generator(), tau, circ["n"] + 7)
    op = orc.Prover(circ["n"], circ["constraints"], circ["selectors"], srs, label, circ["pi_indexes"])
    assert op.preprocess_rc == 0
    return op, srs, M.fr_from_mont(tau)


def _accepted(orc, circ, op, srs, tau, label, proof):
    import plonk_check
    sxy, sinf = op.selector_comms()
    h, tau_h = plonk_check.opening_key_g2(tau)
    return plonk_check.verify(circ["n"], circ["constraints"], label, proof, sxy, sinf, circ["public_inputs"],
                              circ["pi_indexes"], srs[0], h, tau_h, orc)


@pytest.mark.parametrize("case", ["preimage", "preimage_constant"])
def test_hades_gadget_proves_and_verifies(orc, case):
    """gadget.rs:211-251"""
    from zkvm_amd import circuits, poseidon as P
    state = seeded_state(0xbeef) if case == "preimage" else [5000] * 5
    circ = P.permutation_circuit(state, P.ScalarPermutation().permute(state)).tables()
    op, srs, tau = _prover(orc, circ, LABEL)
    proof = op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(0xbeef))
    assert proof["rc"] == 0 and _accepted(orc, circ, op, srs, tau, LABEL, proof)


def test_hades_gadget_wrong_preimage_fails(orc):
    """gadget.rs:253-277: input (0, 31, 0, 0, 0) against the image of (31, 31, 31, 31, 31)"""
    from zkvm_amd import circuits, poseidon as P
    c = P.permutation_circuit([0, 31, 0, 0, 0], P.ScalarPermutation().permute([31] * 5))
    assert not gates_hold(c)
    circ = c.tables()
    op, _, _ = _prover(orc, circ, LABEL)
    assert op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(0xbeef))["rc"] == 2


def test_opening_circuit_proves_and_verifies(orc):
    """merkle-plonk/src/lib.rs:117-131 on a tree of height 3: the root is the one public input"""
    from zkvm_amd import circuits
    c, tree = merkle_circuit(3, 37)
    assert gates_hold(c) and c.public_input_values() == [tree.root()]
    circ = c.tables()
    assert circ["n"] == 4096
    op, srs, tau = _prover(orc, circ, b"merkle")
    proof = op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(11))
    assert proof["rc"] == 0 and _accepted(orc, circ, op, srs, tau, b"merkle", proof)
    bad, _ = merkle_circuit(3, 37, wrong_leaf=True)
    assert not gates_hold(bad)
    circ = bad.tables()
    op, _, _ = _prover(orc, circ, b"merkle")
    assert op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(11))["rc"] == 2


def _gpu_prove(circ, srs, label, blinders):
    import zkvm_amd
    from zkvm_amd import api
    ctx = zkvm_amd.Context(0)
    try:
        ctx.srs_load(srs)
        gp = api.Prover(ctx, circ["n"], circ["constraints"], circ["selectors"].reshape(-1, 4), label, circ["pi_indexes"])
        return gp.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], blinders)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["preimage", "preimage_constant", "preimage_fails"])
def test_gpu_hades_gadget(orc, case):
    """The reference's three gadget tests with the GPU prover: same proof as the CPU prover bit for bit, accepted
    by the verifier; the wrong preimage is refused with the same error."""
    from zkvm_amd import circuits, poseidon as P
    if case == "preimage_fails":
        state, image = [0, 31, 0, 0, 0], P.ScalarPermutation().permute([31] * 5)
    else:
        state = seeded_state(0xbeef) if case == "preimage" else [5000] * 5
        image = P.ScalarPermutation().permute(state)
    circ = P.permutation_circuit(state, image).tables()
    op, srs, tau = _prover(orc, circ, LABEL)
    bl = circuits.random_blinders(0xbeef)
    want = op.prove(circ["wires"], circ["public_inputs"], bl)
    if case == "preimage_fails":
        from zkvm_amd import api
        assert want["rc"] == 2
        with pytest.raises(api.PolynomialDegreeTooLarge):      # the fourth quotient part does not fit the SRS
            _gpu_prove(circ, srs, LABEL, bl)
        return
    got = _gpu_prove(circ, srs, LABEL, bl)
    assert want["rc"] == 0
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(got[key], want[key]), key
    assert _accepted(orc, circ, op, srs, tau, LABEL, got)


@pytest.mark.gpu
def test_gpu_merkle_opening_height_17(orc):
    """One opening of a tree of 4^17 = 2^34 leaves: 17 Poseidon hashes, about 17 500 gates, a 2^15-row proof."""
    from zkvm_amd import circuits
    c, tree = merkle_circuit(17, 0x2_4680_1357, seed=3)
    assert gates_hold(c)
    circ = c.tables()
    assert circ["n"] == 1 << 15 and M.fr_from_mont(circ["public_inputs"][0]) == tree.root()
    op, srs, tau = _prover(orc, circ, b"merkle")
    bl = circuits.random_blinders(17)
    want = op.prove(circ["wires"], circ["public_inputs"], bl)
    got = _gpu_prove(circ, srs, b"merkle", bl)
    assert want["rc"] == 0
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(got[key], want[key]), key
    assert _accepted(orc, circ, op, srs, tau, b"merkle", got)


# ---- the native generator (host code of the library: no GPU needed) -------------------------------------------------
def _tree_with_openings(height, count, seed):
    from zkvm_amd import poseidon as P
    tree = P.Tree(height)
    rng = np.random.default_rng(seed)
    spots = sorted({int(p) for p in rng.integers(0, tree.capacity(), count + 3)})[:count]
    for p in spots:
        tree.insert(p, P.digest(P.Domain.OTHER, [p, seed])[0])
    return tree, spots


def test_native_merkle4_digest_equals_the_mirror():
    from zkvm_amd import poseidon as P
    groups = [seeded_state(s)[:4] for s in range(5)] + [[0, 0, 0, 0], [R - 1] * 4]
    assert P.merkle4_native(groups) == [P.digest(P.Domain.MERKLE4, g)[0] for g in groups]


@pytest.mark.parametrize("height", [1, 3, 8])
def test_native_circuit_of_one_opening_equals_the_composer(height):
    """k = 1: gate for gate the tables of the reference's OpeningCircuit as the Python Composer builds them -
    selectors, the copy-constraint permutation, the wire columns and the public input."""
    from zkvm_amd import poseidon as P
    tree, spots = _tree_with_openings(height, 2, seed=height)
    opening, leaf = tree.opening(spots[1]), tree.leaves[spots[1]]
    want = P.opening_circuit(opening, leaf).tables()
    got = P.batch_opening_circuit([opening], [leaf], threads=2)
    assert got["n"] == want["n"] and got["constraints"] == want["constraints"] and got["pi_indexes"] == want["pi_indexes"]
    for key in ("selectors", "sigma_map", "wires", "public_inputs"):
        assert np.array_equal(got[key], want[key]), key
    # the per-proof half alone (what Prover::prove regenerates, prover.rs:214-238), into arrays used before
    wires, pi = P.batch_opening_witness([tree.opening(spots[0])], [tree.leaves[spots[0]]], threads=1)
    assert not np.array_equal(wires, want["wires"])
    again = P.batch_opening_witness([opening], [leaf], threads=1, out=(wires, pi))
    assert again[0] is wires and np.array_equal(wires, want["wires"]) and np.array_equal(pi, want["public_inputs"])


def test_native_batch_circuit_equals_the_composer_on_several_openings(orc):
    """k openings on one Composer (the loop of merkle-plonk/src/lib.rs:275-332 folded into one circuit): the native
    generator's blocks against the gate-by-gate mirror, a false leaf breaks exactly its own block, and the CPU prover's
    proof of the batch circuit is accepted."""
    from zkvm_amd import circuits, poseidon as P
    from zkvm_amd.composer import Composer, Constraint
    tree, spots = _tree_with_openings(2, 5, seed=21)
    openings = [tree.opening(p) for p in spots]
    leaves = [tree.leaves[p] for p in spots]
    c = Composer.initialized()
    for o, leaf in zip(openings, leaves):
        root = P.opening_gadget(c, o, c.append_witness(leaf))
        c.append_gate(Constraint().left(-1).a(root).public(o.root))
    assert gates_hold(c)
    want = c.tables()
    for threads in (1, 3, 16):
        got = P.batch_opening_circuit(openings, leaves, threads=threads)
        assert got["n"] == want["n"] and got["constraints"] == want["constraints"] and got["pi_indexes"] == want["pi_indexes"]
        for key in ("selectors", "sigma_map", "wires", "public_inputs"):
            assert np.array_equal(got[key], want[key]), (key, threads)
        wires, pi = P.batch_opening_witness(openings, leaves, threads=threads)
        assert np.array_equal(wires, want["wires"]) and np.array_equal(pi, want["public_inputs"]), threads
    assert [M.fr_from_mont(v) for v in got["public_inputs"]] == [tree.root()] * 5
    op, srs, tau = _prover(orc, got, b"merkle-batch")
    proof = op.prove(got["wires"], got["public_inputs"], circuits.random_blinders(5))
    assert proof["rc"] == 0 and _accepted(orc, got, op, srs, tau, b"merkle-batch", proof)
    bad = P.batch_opening_circuit(openings, leaves[:2] + [leaves[2] + 1] + leaves[3:])
    block = (want["constraints"] - 4) // 5
    diff = np.nonzero((bad["wires"] != want["wires"]).any(axis=(0, 2)))[0]
    assert diff.size and diff.min() >= 4 + 2 * block and diff.max() < 4 + 3 * block
    op, _, _ = _prover(orc, bad, b"merkle-batch")
    assert op.prove(bad["wires"], bad["public_inputs"], circuits.random_blinders(5))["rc"] == 2


def test_native_circuit_argument_checks():
    import ctypes as C
    from zkvm_amd import _lib
    lib = _lib.lib()
    gates, n = C.c_uint64(0), C.c_uint64(0)
    assert lib.zkgpu_merkle_batch_circuit_size(0, 1, C.byref(gates), C.byref(n)) == -1
    assert lib.zkgpu_merkle_batch_circuit_size(32, 0, C.byref(gates), C.byref(n)) == -1
    assert lib.zkgpu_merkle_batch_circuit_size(32, 1, C.byref(gates), C.byref(n)) == 0
    assert gates.value == 32293 and n.value == 1 << 15          # one height-32 opening: the reference's workload
    assert lib.zkgpu_merkle_batch_circuit_size(32, 519, C.byref(gates), C.byref(n)) == 0
    assert n.value == 1 << 24 and gates.value > (1 << 24) - 32289
    assert lib.zkgpu_merkle_batch_circuit_size(32, 1 << 20, C.byref(gates), C.byref(n)) == -4
