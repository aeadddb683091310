"""Serialized prover keys (Prover::to_bytes / try_from_bytes, plonk/src/compiler/prover.rs:85-207) through the C ABI.
The reference holds no serialized key among its files, so the byte layout is pinned by the oracle's restatement of
the writers (oracle/plonk.hpp: prover_to_bytes) only - "parity unpinned" beyond that; what the tests add is that the
library's writer and reader agree with that restatement byte for byte and that a key loaded from bytes proves
exactly like the key it was written from."""
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


def _circuit(orc, n, seed):
    from zkvm_amd import circuits
    c = circuits.arithmetic_circuit(n, seed=seed)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    return c


def _srs(orc, count, seed=91):
    tau = M.random_fr_mont(1, seed=seed)[0]
    srs, inf = orc.srs_powers(orc.g1_generator(), tau, count)
    assert not inf.any()
    return srs


@pytest.mark.parametrize("log_n,seed", [(3, 31), (6, 32), (10, 33)])
def test_to_bytes_matches_oracle_and_round_trips(ctx, orc, log_n, seed):
    from zkvm_amd import api, circuits
    n = 1 << log_n
    c = _circuit(orc, n, seed)
    srs = _srs(orc, n + 7)
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"key-bytes", c["pi_indexes"], threads=4)
    assert op.preprocess_rc == 0
    want = op.to_bytes()
    ctx.srs_load(srs)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"key-bytes", c["pi_indexes"])
    got = gp.to_bytes()
    assert got.shape == want.shape and np.array_equal(got, want)

    # load the oracle's bytes into a context whose SRS is something else: the commit key travels in the bytes
    ctx.srs_load(_srs(orc, 9, seed=5))
    lp = api.Prover.try_from_bytes(ctx, want, c["pi_indexes"])
    assert lp.n == n and lp.constraints == c["constraints"]
    sxy, sinf = lp.selector_commitments()
    wxy, winf = op.selector_comms()
    assert np.array_equal(sxy, wxy) and np.array_equal(sinf, winf)
    for k in (0, 5, 7, 11, 14):
        assert np.array_equal(lp.key_poly(k)[:op.poly(k).shape[0]], op.poly(k))
        assert np.array_equal(lp.key_eval8n(k), op.eval8n(k))
    assert np.array_equal(lp.key_eval8n(15), op.eval8n(15))          # linear_evaluations
    assert np.array_equal(lp.to_bytes(), want)
    bl = circuits.random_blinders(seed)
    ref = op.prove(c["wires"], c["public_inputs"], bl)
    out = lp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(out[key], ref[key]), key


def test_verifier_key_n_differs_from_the_header(ctx, orc):
    """The serialized VerifierKey carries its own n (widget.rs:102-106), which closes the transcript seed
    (widget.rs:227), while the first separator takes the header's constraint count (transcript.rs:60): bytes whose two
    values differ must prove like the reference's deserialized prover, and be written back unchanged."""
    from zkvm_amd import api, circuits
    n = 1 << 6
    c = _circuit(orc, n, 71)
    srs = _srs(orc, n + 7)
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"vk-n", c["pi_indexes"], threads=4)
    bl = circuits.random_blinders(71)
    plain = op.prove(c["wires"], c["public_inputs"], bl)
    op.set_vk_n(c["constraints"] + 5)
    blob = op.to_bytes()
    vk_at = blob.size - (8 + 48 * 15)
    assert int.from_bytes(bytes(blob[vk_at:vk_at + 8]), "little") == c["constraints"] + 5
    ref = op.prove(c["wires"], c["public_inputs"], bl)
    assert not np.array_equal(ref["challenges"], plain["challenges"])     # the seed really changed
    lp = api.Prover.try_from_bytes(ctx, blob, c["pi_indexes"])
    out = lp.prove(c["wires"].reshape(-1, 4), c["public_inputs"], bl)
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(out[key], ref[key]), key
    assert np.array_equal(lp.to_bytes(), blob)


def test_malformed_bytes_are_rejected(ctx, orc):
    import zkvm_amd
    from zkvm_amd import api
    n = 8
    c = _circuit(orc, n, 41)
    srs = _srs(orc, n + 7)
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"bad", c["pi_indexes"])
    good = op.to_bytes()
    label_len = 3
    pk_len = int.from_bytes(bytes(good[8:16]), "big")
    ck_len = int.from_bytes(bytes(good[16:24]), "big")

    def rejected(buf):
        with pytest.raises(zkvm_amd.ZkGpuError):
            api.Prover.try_from_bytes(ctx, buf, c["pi_indexes"])

    rejected(good[:40])                                   # NotEnoughBytes: header
    rejected(good[:-1])                                   # NotEnoughBytes: pieces
    bad = good.copy()
    bad[48 + label_len + 16 + 8 + 31] = 0xff              # first q_m coefficient >= r
    rejected(bad)
    bad = good.copy()
    bad[48 + label_len + pk_len + ck_len + 8] &= 0x7f     # compression flag of the first commitment cleared
    rejected(bad)
    bad = good.copy()
    bad[48 + label_len + pk_len + ck_len + 8 + 47] ^= 1   # x no longer on the curve (or another point: wrong subgroup)
    try:
        api.Prover.try_from_bytes(ctx, bad, c["pi_indexes"])
        accepted = True
    except zkvm_amd.ZkGpuError:
        accepted = False
    # flipping the lowest bit of x gives a point on the curve only by chance, and then almost surely outside G1
    assert not accepted
    bad = good.copy()
    bad[48 + label_len + 8] ^= 1                          # evaluation size field
    rejected(bad)
    # and the good bytes still load afterwards
    api.Prover.try_from_bytes(ctx, good, c["pi_indexes"])


def test_size_budget_follows_q_m(ctx, orc):
    """ProverKey::serialization_size budgets every polynomial at q_m's length (widget.rs:265-278): with q_m = 0 the
    buffer is too small, the writer drops what does not fit, and the library reproduces exactly those bytes."""
    from zkvm_amd import api
    n = 16
    c = _circuit(orc, n, 51)
    c["selectors"][0] = 0
    srs = _srs(orc, n + 7)
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"q_m-zero", c["pi_indexes"])
    want = op.to_bytes()
    ctx.srs_load(srs)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"q_m-zero", c["pi_indexes"])
    got = gp.to_bytes()
    assert got.shape == want.shape and np.array_equal(got, want)
    import zkvm_amd
    with pytest.raises(zkvm_amd.ZkGpuError):          # such bytes are truncated: the reference's reader fails as well
        api.Prover.try_from_bytes(ctx, want, c["pi_indexes"])


def test_round_trip_2_18(ctx, orc):
    """A 1.1 GB key: written from device memory, read back, and the proof from the reloaded key is the same."""
    import torch
    from zkvm_amd import api, circuits
    n = 1 << 18
    ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n + 7)
    c = circuits.arithmetic_circuit(n, seed=61)
    circuits.fill_sigma_rows(c, ctx)
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"big", c["pi_indexes"])
    c["selectors"] = None
    wires = torch.from_numpy(c["wires"].reshape(-1, 4).view(np.int64)).cuda()
    bl = circuits.random_blinders(61)
    before = gp.prove(wires, c["public_inputs"], bl)
    blob = gp.to_bytes()
    assert blob.size == 48 + 3 + (n * 32 * 15 + (8 * n * 32 + 172) * 17 + 8 * 17) + 8 + 97 * (n + 7) + 8 + 48 * 15
    ctx.srs_load(np.asarray(ctx.srs_export(4)[0]))       # forget the SRS: it comes back from the bytes
    lp = api.Prover.try_from_bytes(ctx, blob, c["pi_indexes"])
    after = lp.prove(wires, c["public_inputs"], bl)
    for key in ("challenges", "evals", "comms_inf", "comms"):
        assert np.array_equal(after[key], before[key]), key
    again = lp.to_bytes()
    assert np.array_equal(again, blob)
