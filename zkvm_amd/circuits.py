"""Synthetic PLONK circuits for tests and benchmarks: gate tables of the shape the reference's Composer produces
(plonk/src/composer.rs) for arithmetic-only circuits — multiplication, addition-with-fourth-wire and constant
gates, one public input, and copy constraints linking about half of the gates' left wires to the previous
gate's output.  The Poseidon-Merkle opening circuit of the reference emits only such gates
(poseidon-merkle/src/zk.rs:44-101).

Everything here is host-side input generation with Python integers; values are produced directly as
4 x 64-bit Montgomery limbs.
"""
import numpy as np

R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
MONT = 1 << 256
MONT_INV = pow(MONT, -1, R_MOD)
K = (1, 7, 13, 17)   # coset representatives of the four wire columns (composer/permutation/constants.rs:7-9)

SEL = {n: i for i, n in enumerate(["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic",
                                   "q_fixed_group_add", "q_variable_group_add", "s_sigma_1", "s_sigma_2", "s_sigma_3",
                                   "s_sigma_4"])}


# BLS12-381 base field and the standard G1 generator (coset-bls12_381/src/g1.rs:218-238), as integers
P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1


def g1_generator_limbs():
    """(12,) uint64: x | y of the generator in the Montgomery form (radix 2^384) the library's point tables use."""
    vals = [(G1_X << 384) % P_MOD, (G1_Y << 384) % P_MOD]
    buf = b"".join(v.to_bytes(48, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).copy()


def seeded_tau(seed=7):
    """A fixed SRS trapdoor for synthetic parameters (never use a known trapdoor outside tests and benchmarks)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x7a0))
    return random_mont(rng, 1)[0]


def to_mont(v):
    return (v % R_MOD) * MONT % R_MOD


def ints_to_limbs(vals):
    """list of integers < 2^256 -> (len, 4) uint64"""
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def limbs_to_ints(arr):
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def random_mont(rng, count):
    """`count` random Montgomery residues (any canonical integer is the Montgomery form of some element)"""
    a = rng.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
    a[:, 3] %= np.uint64(0x73eda753299d7d48)
    return a


def arithmetic_circuit(n, constraints=None, seed=5, copy_fraction=0.5):
    """-> dict(n, constraints, selectors (15, n, 4) with the sigma rows still empty, sigma_map (4, n) uint64,
    wires (4, n, 4), public_inputs (1, 4), pi_indexes [0]).  Gate 0 exposes a public input."""
    if constraints is None:
        constraints = n - 6 if n > 8 else n
    assert 2 <= constraints <= n
    rng = np.random.Generator(np.random.PCG64(seed))
    one, minus_one = to_mont(1), to_mont(-1)
    ra = limbs_to_ints(random_mont(rng, constraints))
    rb = limbs_to_ints(random_mont(rng, constraints))
    rd = limbs_to_ints(random_mont(rng, constraints))
    kind = rng.integers(0, 3, size=constraints)          # 0 mul, 1 add (with fourth wire), 2 mul + constant
    copy = rng.random(constraints) < copy_fraction         # gate i+1's left wire copies gate i's output
    consts = limbs_to_ints(random_mont(rng, constraints))

    a, b, c, d = [0] * n, [0] * n, [0] * n, [0] * n
    sel = np.zeros((15, n, 4), dtype=np.uint64)
    q = {k: [0] * n for k in ("q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith")}
    # gate 0: q_l * a + PI = 0
    a[0] = ra[0]
    q["q_l"][0] = one
    q["q_arith"][0] = one
    pi0 = (R_MOD - a[0]) % R_MOD
    prev_out = None
    for i in range(1, constraints):
        ai = prev_out if (prev_out is not None and copy[i - 1]) else ra[i]
        bi = rb[i]
        q["q_arith"][i] = one
        q["q_o"][i] = minus_one
        if kind[i] == 1:      # a + b + d - c = 0
            di = rd[i]
            ci = (ai + bi + di) % R_MOD
            q["q_l"][i] = one
            q["q_r"][i] = one
            q["q_f"][i] = one
            d[i] = di
        else:                 # a * b (+ k) - c = 0 ; Montgomery residues: (a~ b~ / R) is the residue of the product
            ci = ai * bi % R_MOD * MONT_INV % R_MOD
            q["q_m"][i] = one
            if kind[i] == 2:
                ci = (ci + consts[i]) % R_MOD
                q["q_c"][i] = consts[i]
        a[i], b[i], c[i] = ai, bi, ci
        prev_out = ci
    for name, vals in q.items():
        sel[SEL[name]] = ints_to_limbs(vals)
    wires = np.stack([ints_to_limbs(v) for v in (a, b, c, d)])

    # permutation: output(i) <-> left(i+1) wherever the copy was taken (composer/permutation.rs:76-109)
    sigma = np.arange(4 * n, dtype=np.uint64).reshape(4, n)
    idx = np.nonzero(copy[:constraints - 1])[0]
    idx = idx[idx >= 1]                      # gate 0 has no output
    sigma[2, idx] = 0 * n + (idx + 1)
    sigma[0, idx + 1] = 2 * n + idx
    return dict(n=n, constraints=constraints, selectors=sel, sigma_map=sigma, wires=wires,
                public_inputs=ints_to_limbs([pi0]), pi_indexes=[0])


def widget_circuit(n, repeats=8):
    """A circuit of n rows in which every widget has gates: `repeats` times a 32-bit range check, a 32-bit AND and XOR,
    a fixed-base scalar multiplication and a curve addition (built gate by gate by zkvm_amd.composer), then all-zero
    rows up to n - 6 constraints (a row whose selectors and wires are zero is satisfied).  Same layout as
    arithmetic_circuit, sigma rows still empty."""
    from . import composer as C
    c = C.Composer.initialized()
    for r in range(repeats):
        a = c.append_witness(0xbeefcafe + 977 * r)
        b = c.append_witness(0x0f0f3333 + 31 * r)
        c.component_range(a, 16)
        w_and = c.append_logic_and(a, b, 16)
        w_xor = c.append_logic_xor(a, b, 16)
        if r == 0:
            c.assert_equal(w_and, c.append_public(c[w_and]))
            c.assert_equal(w_xor, c.append_public(c[w_xor]))
        k = c.append_witness(0x1234567890abcdef1234567 + r)
        p = c.component_mul_generator(k)
        g = c.append_point(C.JUBJUB_GENERATOR)
        c.component_add_point(g, p)
    t = c.tables()
    m = t["n"]
    assert n >= 2 * m, "the widget gates need %d rows" % m
    sel = np.zeros((15, n, 4), dtype=np.uint64)
    sel[:, :m] = t["selectors"]
    wires = np.zeros((4, n, 4), dtype=np.uint64)
    wires[:, :m] = t["wires"]
    sigma = np.arange(4 * n, dtype=np.uint64).reshape(4, n)
    col, row = np.divmod(t["sigma_map"], np.uint64(m))          # entries are column * m + gate
    sigma[:, :m] = col * np.uint64(n) + row
    return dict(n=n, constraints=n - 6 if n > 8 else n, selectors=sel, sigma_map=sigma, wires=wires,
                public_inputs=t["public_inputs"], pi_indexes=t["pi_indexes"])


def fill_sigma_rows(circuit, ctx):
    """Writes the four sigma rows: sigma value = K[col'] * w^(idx') for sigma(col, i) = (col', idx')
    (composer/permutation.rs:111-141).  The roots come from the GPU: fft(K * e_1)[i] = K * w^i."""
    n = circuit["n"]
    log_n = n.bit_length() - 1
    tables = []
    for k in K:
        e1 = np.zeros((n, 4), dtype=np.uint64)
        if n > 1:
            e1[1] = ints_to_limbs([to_mont(k)])[0]
        else:
            e1[0] = ints_to_limbs([to_mont(k)])[0]
        tables.append(ctx.ntt(e1, log_n, 0))
    table = np.concatenate(tables, axis=0)          # index col' * n + idx'
    for col in range(4):
        circuit["selectors"][SEL["s_sigma_1"] + col] = table[circuit["sigma_map"][col]]
    return circuit


def random_blinders(seed=5):
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xb11d))
    return random_mont(rng, 14)
