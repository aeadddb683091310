"""GPU parity of the PLONK prover path through the C boundary: permutation accumulator, polynomial evaluation and
Ruffini division, key preprocessing, quotient polynomial and the whole of Prover::prove, bit for bit against the
CPU oracle on the same seeded circuits (arithmetic gates, copy constraints, one public input)."""
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
    tau = M.random_fr_mont(1, seed=77)[0]
    g = orc.g1_generator()
    xy, inf = orc.srs_powers(g, tau, (1 << 12) + 7)
    assert not inf.any()
    return xy


def _circuit(orc, n, seed=5, constraints=None):
    from zkvm_amd import circuits
    c = circuits.arithmetic_circuit(n, constraints=constraints, seed=seed)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    return c


def test_fill_sigma_rows_matches_oracle(ctx, orc):
    from zkvm_amd import circuits
    c = circuits.arithmetic_circuit(64, seed=3)
    want = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    circuits.fill_sigma_rows(c, ctx)
    assert np.array_equal(c["selectors"][11:15], want)


@pytest.mark.parametrize("log_n", [2, 5, 10, 13, 15])
def test_perm_z(ctx, orc, log_n):
    from zkvm_amd import api
    n = 1 << log_n
    wires = M.random_fr_mont(4 * n, seed=log_n).reshape(4, n, 4)
    sig = M.random_fr_mont(4 * n, seed=100 + log_n).reshape(4, n, 4)
    beta, gamma = M.random_fr_mont(2, seed=200 + log_n)
    sigma_polys = [orc.fft("ifft", sig[k]) for k in range(4)]
    want = orc.permutation_vec(wires, sigma_polys, beta, gamma)
    got = api.perm_z(ctx, wires, sig, beta, gamma)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("length", [0, 1, 2, 63, 64, 65, 1000, 4096, (1 << 14) + 3, (1 << 17) + 11])
def test_poly_eval(ctx, orc, length):
    from zkvm_amd import api
    c = M.random_fr_mont(length, seed=length) if length else np.zeros((0, 4), np.uint64)
    x = M.random_fr_mont(1, seed=9)[0]
    assert np.array_equal(api.poly_eval(ctx, c, x), orc.poly_evaluate(c, x))
    zero = np.zeros(4, np.uint64)
    assert np.array_equal(api.poly_eval(ctx, c, zero), orc.poly_evaluate(c, zero))


@pytest.mark.parametrize("length", [1, 2, 3, 100, 1024, (1 << 13) + 5, (1 << 16) + 9])
def test_poly_ruffini(ctx, orc, length):
    from zkvm_amd import api
    c = M.random_fr_mont(length, seed=length + 1)
    root = M.random_fr_mont(1, seed=10)[0]
    want = orc.poly_ruffini(c, root)
    got = api.poly_ruffini(ctx, c, root)
    assert got.shape[0] == length - 1
    assert np.array_equal(got[:want.shape[0]], want)
    assert not got[want.shape[0]:].any()
    zero = np.zeros(4, np.uint64)
    want = orc.poly_ruffini(c, zero)
    got = api.poly_ruffini(ctx, c, zero)
    assert np.array_equal(got[:want.shape[0]], want) and not got[want.shape[0]:].any()


def _provers(ctx, orc, srs, c, label=b"parity"):
    from zkvm_amd import api
    n = c["n"]
    ctx.srs_load(srs[:n + 7])
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), label, c["pi_indexes"])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], label, c["pi_indexes"], threads=4)
    assert op.preprocess_rc == 0
    return gp, op


@pytest.mark.parametrize("log_n", [4, 10])
def test_key_preprocess(ctx, orc, srs, log_n):
    n = 1 << log_n
    c = _circuit(orc, n)
    gp, op = _provers(ctx, orc, srs, c)
    for k in range(15):
        want = op.poly(k)
        got = gp.key_poly(k)
        assert np.array_equal(got[:want.shape[0]], want), k
        assert not got[want.shape[0]:].any(), k
    for k in range(16):
        assert np.array_equal(gp.key_eval8n(k), op.eval8n(k)), k
    # member 16: the first Lagrange polynomial on the 8n coset (compute_first_lagrange_poly_scaled with scale 1,
    # quotient_poly.rs:222-231)
    e0 = np.zeros((n, 4), np.uint64)
    e0[0] = M.fr_to_mont(1)
    l1 = orc.fft("coset_fft", orc.fft("ifft", e0), 8 * n)
    assert np.array_equal(gp.key_eval8n(16), l1)
    gxy, ginf = gp.selector_commitments()
    oxy, oinf = op.selector_comms()
    assert np.array_equal(ginf, oinf)
    for k in range(15):
        if not oinf[k]:
            assert np.array_equal(gxy[k], oxy[k]), k


@pytest.mark.parametrize("log_n", [4, 10])
def test_quotient(ctx, orc, srs, log_n):
    n = 1 << log_n
    c = _circuit(orc, n, seed=8)
    gp, op = _provers(ctx, orc, srs, c)
    ch = M.random_fr_mont(7, seed=31)
    wires = c["wires"]
    # blinded wire polynomials (n + 2 coefficients) exercise the padded coset FFTs
    polys = []
    bl = M.random_fr_mont(8, seed=32)
    for k in range(4):
        p = np.zeros((n + 2, 4), np.uint64)
        p[:n] = orc.fft("ifft", wires[k])
        polys.append(p)
    sigma_polys = [op.poly(11 + k) for k in range(4)]
    zvec = orc.permutation_vec(wires, sigma_polys, ch[1], ch[2])
    z = orc.fft("ifft", zvec)
    dense_pi = np.zeros((n, 4), np.uint64)
    dense_pi[0] = c["public_inputs"][0]
    pi = orc.fft("ifft", dense_pi)
    want = op.quotient(z, *polys, pi, ch)
    got = gp.quotient(z, *polys, pi, ch)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert want.shape[0] <= 4 * n   # the circuit is satisfied: the division is exact (degree 4n - 5)


@pytest.mark.parametrize("log_n", [3, 6, 11])
def test_linearization(ctx, orc, srs, log_n):
    """build_linearization_polynomial (linearization_poly.rs:146-234) on its own: random round polynomials, quotient
    parts of ragged lengths, evaluations and challenges against the oracle's restatement, coefficient for coefficient."""
    n = 1 << log_n
    c = _circuit(orc, n, seed=60 + log_n)
    gp, op = _provers(ctx, orc, srs, c)
    for trial, t4_len in enumerate((n + 6, 3, 0)):
        z = M.random_fr_mont(n + 3, seed=61 + trial)
        parts = [M.random_fr_mont(k, seed=62 + 10 * trial + j) for j, k in enumerate((n + 1, n + 1, n + 1, t4_len))]
        ch = M.random_fr_mont(7, seed=63 + trial)
        z_ch = M.random_fr_mont(1, seed=64 + trial)[0]
        evals = M.random_fr_mont(15, seed=65 + trial)
        want = op.linearization(z, *parts, ch, z_ch, evals, c["public_inputs"])
        got = gp.linearization(z, *parts, ch, z_ch, evals, c["public_inputs"])
        assert got.shape == want.shape and np.array_equal(got, want), (log_n, trial)
    # a z polynomial with leading zeros: its degree picks the domain of L_1(z) (permutation/proverkey.rs:156)
    z = M.random_fr_mont(n + 3, seed=70)
    z[n // 2:] = 0
    want = op.linearization(z, *parts, ch, z_ch, evals, c["public_inputs"])
    got = gp.linearization(z, *parts, ch, z_ch, evals, c["public_inputs"])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("lens", [[5], [1], [40, 7, 64, 3], [1 << 13, (1 << 13) + 3, 100], [300] * 8])
def test_aggregate_witness(ctx, orc, lens):
    """CommitKey::compute_aggregate_witness (kzg10/key.rs:131-146): powers of v over ragged polynomials, then the
    division by (X - z) with the remainder dropped."""
    from zkvm_amd import api
    polys = [M.random_fr_mont(k, seed=80 + i + k) for i, k in enumerate(lens)]
    point, v = M.random_fr_mont(2, seed=81 + len(lens))
    want = orc.aggregate_witness(polys, point, v)
    got = api.aggregate_witness(ctx, polys, point, v)
    assert got.shape == want.shape and np.array_equal(got, want)
    # the point zero: the quotient is a shift
    zero = np.zeros(4, np.uint64)
    assert np.array_equal(api.aggregate_witness(ctx, polys, zero, v), orc.aggregate_witness(polys, zero, v))


@pytest.mark.parametrize("log_n,seed", [(3, 1), (4, 2), (7, 3), (10, 4), (12, 5)])
def test_prove(ctx, orc, srs, log_n, seed):
    from zkvm_amd import api, circuits
    n = 1 << log_n
    c = _circuit(orc, n, seed=seed)
    gp, op = _provers(ctx, orc, srs, c)
    bl = circuits.random_blinders(seed)
    want = op.prove(c["wires"], c["public_inputs"], bl)
    assert want["rc"] == 0
    got = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl, timings=True)
    assert np.array_equal(got["challenges"], want["challenges"])
    assert np.array_equal(got["evals"], want["evals"])
    assert np.array_equal(got["comms_inf"], want["comms_inf"])
    for k in range(11):
        assert np.array_equal(got["comms"][k], want["comms"][k]), api.COMMITMENT_NAMES[k]
    assert got["timings_ms"]["total"] > 0
    # and as the 1008 bytes the reference writes to a proof file (proof_system/proof.rs:95-123)
    assert api.proof_to_bytes(got) == api.proof_to_bytes(want)


def test_prove_twice_same_key(ctx, orc, srs):
    """The per-proof buffers are reused: a second proof with other blinders on the same key stays exact."""
    from zkvm_amd import circuits
    c = _circuit(orc, 256, seed=12)
    gp, op = _provers(ctx, orc, srs, c)
    for s in (1, 2):
        bl = circuits.random_blinders(s)
        want = op.prove(c["wires"], c["public_inputs"], bl)
        got = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
        assert np.array_equal(got["comms"], want["comms"]) and np.array_equal(got["evals"], want["evals"])


def test_prefetched_witnesses(ctx, orc, srs):
    """zkgpu_witness_prefetch: the loop of the reference's service (a fresh witness per proof,
    merkle-plonk/src/lib.rs:275-332) with witness i + 1 uploaded beside proof i from two pinned host buffers; every
    proof equals the oracle's for its own witness and public input, also when an upload is never consumed."""
    from zkvm_amd import api, circuits
    n = 1 << 10
    c = _circuit(orc, n, seed=31)
    gp, op = _provers(ctx, orc, srs, c)
    # gate 0 reads q_l * a + PI = 0 and its left wire is tied to nothing else: a witness per value of a[0]
    witnesses = []
    for k in range(4):
        w = c["wires"].copy()
        pub = c["public_inputs"].copy()
        if k:
            v = int(circuits.limbs_to_ints(M.random_fr_mont(1, 900 + k))[0])
            w[0, 0] = circuits.ints_to_limbs([v])[0]
            pub = circuits.ints_to_limbs([(circuits.R_MOD - v) % circuits.R_MOD])
        witnesses.append((w, pub, circuits.random_blinders(50 + k)))
    want = [op.prove(w, pub, bl) for w, pub, bl in witnesses]
    assert all(p["rc"] == 0 for p in want) and not np.array_equal(want[0]["comms"][0], want[1]["comms"][0])
    pins = [api.PinnedFr(4 * n) for _ in range(2)]
    try:
        pins[0].array[:] = witnesses[0][0].reshape(-1, 4)
        gp.prefetch(pins[0].array)
        for i, (w, pub, bl) in enumerate(witnesses):
            if i + 1 < len(witnesses):
                pins[(i + 1) & 1].array[:] = witnesses[i + 1][0].reshape(-1, 4)   # proof i - 1, which read it, has returned
                gp.prefetch(pins[(i + 1) & 1].array)
            got = gp.prove(pins[i & 1].array, pub, bl)
            for key in ("comms", "comms_inf", "evals", "challenges"):
                assert np.array_equal(got[key], want[i][key]), (i, key)
        # an upload nobody consumes, then a proof from another (pageable) array and one from a device tensor
        gp.prefetch(pins[0].array)
        w, pub, bl = witnesses[2]
        got = gp.prove(np.ascontiguousarray(w.reshape(-1, 4)), pub, bl)
        assert np.array_equal(got["comms"], want[2]["comms"]) and np.array_equal(got["evals"], want[2]["evals"])
        import torch
        wd = torch.from_numpy(w.reshape(-1, 4).view(np.int64)).cuda()
        got = gp.prove(wd, pub, bl)
        assert np.array_equal(got["comms"], want[2]["comms"]) and np.array_equal(got["evals"], want[2]["evals"])
        ctx.sync()
    finally:
        for p in pins:
            p.close()


def test_two_host_threads_share_one_context(ctx, orc, srs):
    """Prover::prove takes &self (compiler/prover.rs:21,210): several host threads may prove with one prover.  The
    library serialises calls on a context, so concurrent proofs and transforms on one context stay exact."""
    import threading
    from zkvm_amd import circuits
    c = _circuit(orc, 512, seed=14)
    gp, op = _provers(ctx, orc, srs, c)
    blinders = [circuits.random_blinders(s) for s in range(4)]
    wants = [op.prove(c["wires"], c["public_inputs"], bl) for bl in blinders]
    x = M.random_fr_mont(1 << 12, 15)
    want_fft = orc.fft("coset_fft", x, 1 << 12)
    results, errors = {}, []

    def prove_worker(t):
        try:
            for rep in range(3):
                for k in (t, t + 2):
                    results[(t, rep, k)] = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], blinders[k])
        except Exception as e:   # surfaced below: an exception in a thread would otherwise be lost
            errors.append(e)

    def ntt_worker():
        try:
            for rep in range(20):
                results[("ntt", rep)] = ctx.ntt(x, 12, 2)
        except Exception as e:
            errors.append(e)

    threads = [threading.Thread(target=prove_worker, args=(t,)) for t in (0, 1)] + [threading.Thread(target=ntt_worker)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (key, got) in results.items():
        if key[0] == "ntt":
            assert np.array_equal(got, want_fft)
        else:
            want = wants[key[2]]
            for f in ("challenges", "evals", "comms_inf", "comms"):
                assert np.array_equal(got[f], want[f]), (key, f)


def test_cloned_contexts_share_table_and_key(orc, srs):
    """zkgpu_ctx_clone = #[derive(Clone)] Prover (compiler/prover.rs:20-21): clones refer to one SRS table and one prover
    key on the GPU, prove side by side from their own host threads, survive the context they were cloned from, and fall
    back to the 8n coset on their own (a violated gate) with the reference's error."""
    import threading
    import zkvm_amd
    from zkvm_amd import api, circuits
    c = _circuit(orc, 1024, seed=18)
    n = c["n"]
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], b"cloned", c["pi_indexes"], threads=4)
    assert op.preprocess_rc == 0
    blinders = [circuits.random_blinders(60 + k) for k in range(3)]
    wants = [op.prove(c["wires"], c["public_inputs"], bl) for bl in blinders]
    first = zkvm_amd.Context(0)
    contexts = [first]
    results, errors = {}, []
    try:
        first.srs_load(srs[:n + 7])
        gp = api.Prover(first, n, c["constraints"], c["selectors"].reshape(-1, 4), b"cloned", c["pi_indexes"])
        assert first.device_bytes_shared() == 0
        alone = first.device_bytes()
        contexts += [first.clone() for _ in range(3)]
        provers = [gp] + [gp.on(cx) for cx in contexts[1:]]
        shared = first.device_bytes_shared()
        # the table (96 + 128 bytes per point and level) and the key's vectors are held once
        assert shared >= (n + 7) * 96 + 15 * n * 32 and shared <= alone
        for cx in contexts[1:]:
            assert cx.device_bytes_shared() == shared and cx.device_bytes() - shared < alone - shared + (1 << 20)
        sxy, sinf = provers[2].selector_commitments()
        wxy, winf = op.selector_comms()
        assert np.array_equal(sinf, winf) and np.array_equal(sxy[winf == 0], wxy[winf == 0])

        def worker(t):
            try:
                for rep in range(4):
                    for k in range(3):
                        results[(t, rep, k)] = provers[t].prove(c["wires"].reshape(-1, 4), c["public_inputs"], blinders[k])
            except Exception as e:   # surfaced below
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert len(results) == 48
        for key, got in results.items():
            for f in ("challenges", "evals", "comms_inf", "comms"):
                assert np.array_equal(got[f], wants[key[2]][f]), (key, f)
            assert got["coset"] == 4
        # the source loads another key and is then destroyed: the clones keep the table and key they refer to
        c2 = _circuit(orc, 256, seed=19)
        first.srs_load(srs[:256 + 7])
        other = api.Prover(first, 256, c2["constraints"], c2["selectors"].reshape(-1, 4), b"other", c2["pi_indexes"])
        op2 = orc.Prover(256, c2["constraints"], c2["selectors"], srs[:256 + 7], b"other", c2["pi_indexes"], threads=4)
        got2, want2 = other.prove(c2["wires"].reshape(-1, 4), c2["public_inputs"], blinders[0]), op2.prove(c2["wires"], c2["public_inputs"], blinders[0])
        assert np.array_equal(got2["comms"], want2["comms"]) and np.array_equal(got2["evals"], want2["evals"])
        first.close()
        got = provers[3].prove(c["wires"].reshape(-1, 4), c["public_inputs"], blinders[1])
        assert np.array_equal(got["comms"], wants[1]["comms"]) and np.array_equal(got["evals"], wants[1]["evals"])
        # a violated gate on a clone: the proof runs over the 8n coset (vectors the clone builds for itself) and ends in
        # the oracle's error; the next good proof is exact again
        bad = c["wires"].copy()
        bad[2, 5, 0] ^= np.uint64(1)
        want_bad = op.prove(bad, c["public_inputs"], blinders[2])
        assert want_bad["rc"] == 2
        with pytest.raises(api.PolynomialDegreeTooLarge):
            provers[1].prove(bad.reshape(-1, 4), c["public_inputs"], blinders[2])
        got = provers[1].prove(c["wires"].reshape(-1, 4), c["public_inputs"], blinders[2])
        assert np.array_equal(got["comms"], wants[2]["comms"]) and np.array_equal(got["evals"], wants[2]["evals"])
    finally:
        for cx in contexts:
            cx.close()


def test_contexts_on_several_host_threads(orc, srs):
    """Independent contexts on one GPU, one host thread each (the way a service overlaps small proofs): every context
    has its own stream, buffers and host helper threads, and every proof stays equal to the oracle's."""
    import threading
    import zkvm_amd
    from zkvm_amd import api, circuits
    c = _circuit(orc, 1024, seed=16)
    n = c["n"]
    op = orc.Prover(n, c["constraints"], c["selectors"], srs[:n + 7], b"several", c["pi_indexes"], threads=4)
    assert op.preprocess_rc == 0
    blinders = [circuits.random_blinders(40 + k) for k in range(3)]
    wants = [op.prove(c["wires"], c["public_inputs"], bl) for bl in blinders]
    contexts = [zkvm_amd.Context(0) for _ in range(4)]
    results, errors = {}, []
    try:
        provers = []
        for cx in contexts:
            cx.srs_load(srs[:n + 7])
            provers.append(api.Prover(cx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"several", c["pi_indexes"]))

        def worker(t):
            try:
                for rep in range(4):
                    for k in range(3):
                        results[(t, rep, k)] = provers[t].prove(c["wires"].reshape(-1, 4), c["public_inputs"], blinders[k])
            except Exception as e:   # surfaced below
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        for cx in contexts:
            cx.close()
    assert not errors, errors
    assert len(results) == 48
    for key, got in results.items():
        want = wants[key[2]]
        for f in ("challenges", "evals", "comms_inf", "comms"):
            assert np.array_equal(got[f], want[f]), (key, f)


def test_prove_unsatisfied_circuit(ctx, orc, srs):
    """A witness violating a gate leaves a quotient of 8n coefficients; its fourth slice exceeds the SRS and the
    reference's commit returns PolynomialDegreeTooLarge (kzg10/key.rs:91-105)."""
    from zkvm_amd import api, circuits
    c = _circuit(orc, 64, seed=6)
    gp, op = _provers(ctx, orc, srs, c)
    w = c["wires"].copy()
    w[2, 5, 0] ^= np.uint64(1)
    bl = circuits.random_blinders(1)
    assert op.prove(w, c["public_inputs"], bl)["rc"] == 2
    with pytest.raises(api.PolynomialDegreeTooLarge):
        gp.prove(w.reshape(-1, 4), c["public_inputs"], bl)


def test_prove_2_14(ctx, orc):
    """A circuit large enough for multi-pass transforms on the 8n = 2^17 domain and several sort slices per range."""
    from zkvm_amd import api, circuits
    n = 1 << 14
    tau = M.random_fr_mont(1, seed=78)[0]
    srs14, inf = orc.srs_powers(orc.g1_generator(), tau, n + 7)
    assert not inf.any()
    c = _circuit(orc, n, seed=21)
    ctx.srs_load(srs14)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"parity-14", c["pi_indexes"])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs14, b"parity-14", c["pi_indexes"], threads=16)
    assert op.preprocess_rc == 0
    bl = circuits.random_blinders(21)
    want = op.prove(c["wires"], c["public_inputs"], bl)
    assert want["rc"] == 0
    got = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(got[key], want[key]), key


def _with_public_inputs(c, gates, seed):
    """Turns more gates into public-input gates: q_c[i] -= delta, PI[i] = delta keeps gate i satisfied."""
    deltas = M.random_fr_ints(len(gates), seed)
    q_c = M.fr_vec_from_mont(c["selectors"][5])
    for g, dlt in zip(gates, deltas):
        q_c[g] = (q_c[g] - dlt) % M.R_MOD
    c["selectors"][5] = M.fr_vec_to_mont(q_c)
    c["pi_indexes"] = list(c["pi_indexes"]) + list(gates)
    c["public_inputs"] = np.concatenate([c["public_inputs"], M.fr_vec_to_mont(deltas)])
    return c


@pytest.mark.parametrize("gates", [[3, 17], [1, 2, 40], [5, 9, 11, 30, 31, 50], [7, 7]])
def test_prove_several_public_inputs(ctx, orc, srs, gates):
    """Up to four distinct positions use the key-resident Lagrange vectors, more fall back to the transform; a
    repeated position keeps the later value, as the reference's dense vector does (composer.rs:1001-1014)."""
    from zkvm_amd import circuits
    c = _with_public_inputs(_circuit(orc, 64, seed=len(gates) + 40), gates, seed=9)
    gp, op = _provers(ctx, orc, srs, c)
    bl = circuits.random_blinders(3)
    want = op.prove(c["wires"], c["public_inputs"], bl)
    if len(set(gates)) == len(gates):
        assert want["rc"] == 0
    got = None
    try:
        got = gp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
    except Exception:
        assert want["rc"] != 0
    if got is not None:
        assert want["rc"] == 0
        for key in ("challenges", "evals", "comms_inf", "comms"):
            assert np.array_equal(got[key], want[key]), key
    # fewer values than positions: the remaining positions stay zero on both sides
    short = c["public_inputs"][:1]
    want = op.prove(c["wires"], short, bl)
    try:
        got = gp.prove(c["wires"].reshape(-1, 4), short, bl)
        assert want["rc"] == 0
        for key in ("challenges", "evals", "comms"):
            assert np.array_equal(got[key], want[key]), key
    except Exception:
        assert want["rc"] != 0


def test_prove_fuzz_random_composer_circuits():
    """Circuits from random sequences of Composer operations, every widget included, proved on the GPU and by the
    oracle with the same blinders (tools/fuzz_prove.py)."""
    import os
    import runpy
    import sys
    from conftest import ROOT
    argv = sys.argv
    sys.argv = ["fuzz_prove.py", "40", "11"]
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "fuzz_prove.py"), run_name="__main__")
    finally:
        sys.argv = argv
