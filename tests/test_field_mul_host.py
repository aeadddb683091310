"""Host-side check (no GPU) of the Montgomery product shared by host and device code, against the oracle's
schoolbook restatement of the reference (scalar.rs:528-552, fp.rs:558-600), on random and edge operands."""
import numpy as np

import model as M


def _edge(mod, width, to_limbs):
    vals = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << (64 * width)) % mod, (1 << 32) - 1, (1 << 64) - 1,
            mod >> 1, (mod >> 29) << 29, ((1 << 29) - 1) * ((1 << 58) + (1 << 29) + 1) % mod, mod - (1 << 28)]
    vals += [(mod - 1) ^ (1 << k) if ((mod - 1) ^ (1 << k)) < mod else k for k in range(0, 64 * width - 3, 17)]
    return np.stack([to_limbs(v) for v in vals])


def test_fr_product(orc):
    from zkvm_amd import api
    e = _edge(M.R_MOD, 4, lambda v: M.int_to_limbs(v, 4))
    rnd = M.random_fr_mont(3000, seed=11)
    a = np.concatenate([np.repeat(e, len(e), axis=0), rnd[:1500]])
    b = np.concatenate([np.tile(e, (len(e), 1)), rnd[1500:]])
    got = api.field_mul_selfcheck(0, a, b)
    assert np.array_equal(got, orc.fr_op("mul", a, b))


def test_fq_product(orc):
    from zkvm_amd import api
    e = _edge(M.P_MOD, 6, lambda v: M.int_to_limbs(v, 6))
    rng = np.random.Generator(np.random.PCG64(5))
    rnd = rng.integers(0, 1 << 64, size=(2000, 6), dtype=np.uint64)
    rnd[:, 5] %= np.uint64(0x1a0111ea397fe69a)
    a = np.concatenate([np.repeat(e, len(e), axis=0), rnd[:1000]])
    b = np.concatenate([np.tile(e, (len(e), 1)), rnd[1000:]])
    got = api.field_mul_selfcheck(1, a, b)
    assert np.array_equal(got, orc.fq_op("mul", a, b))


def test_host_inversions(orc):
    """The host's inversions (Fp::invert fp.rs:383-415, Scalar::invert scalar.rs:394-477: a^(p-2) there, the binary
    extended Euclidean algorithm here) against the oracle's: same canonical result, zero for zero."""
    from zkvm_amd import api
    fr = np.concatenate([_edge(M.R_MOD, 4, lambda v: M.int_to_limbs(v, 4)), M.random_fr_mont(400, seed=12)])
    got = api.field_inv_selfcheck(0, fr)
    assert np.array_equal(got, orc.fr_op("invert", fr))
    nz = fr.any(axis=1)
    assert not got[~nz].any()
    assert np.array_equal(api.field_mul_selfcheck(0, fr[nz], got[nz]), np.tile(M.fr_to_mont(1), (int(nz.sum()), 1)))
    rng = np.random.Generator(np.random.PCG64(6))
    rnd = rng.integers(0, 1 << 64, size=(300, 6), dtype=np.uint64)
    rnd[:, 5] %= np.uint64(0x1a0111ea397fe69a)
    fq = np.concatenate([_edge(M.P_MOD, 6, lambda v: M.int_to_limbs(v, 6)), rnd])
    assert np.array_equal(api.field_inv_selfcheck(1, fq), orc.fq_op("invert", fq))
