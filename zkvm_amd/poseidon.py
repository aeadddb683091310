"""Host-side mirror of the circuits the reference proves: the Hades permutation over the BLS12-381 scalar field in
its two forms (values: coset-poseidon/src/hades/permutation/scalar.rs:33-67; gates on a Composer:
coset-poseidon/src/hades/permutation/gadget.rs:39-106; round schedule: hades/permutation.rs:27-67), the sponge hash
built on it (coset-poseidon/src/hash.rs:42-159, hash/gadget.rs:42-99), the 4-ary Poseidon-Merkle tree with its
openings (poseidon-merkle/src/lib.rs:129-160, coset-merkle/src/opening.rs:32-97,180-207) and the opening circuit
(poseidon-merkle/src/zk.rs:16-102, merkle-plonk/src/lib.rs:117-131).  It only builds gate tables and witnesses for
`zkvm_amd.composer.Composer`; proving is the GPU prover's work.

The sponge is the third-party crate dusk-safe 0.3 (coset-poseidon/Cargo.toml:10), which is not part of the
reference tree; it is restated here from its published algorithm.  The reference's known-answer hashes
(coset-poseidon/src/hades.rs:73-141, run with a zero tag) pin the permutation, the constants and the absorb /
squeeze mechanics.  No vector of the reference pins the tag derived from the call pattern, nor the operand order
of the sponge's additions inside a circuit: both only choose among equivalent circuits.

Round constants and the MDS matrix are the reference's data tables (coset-poseidon/assets/arc.bin, mds.bin: 32-byte
little-endian canonical integers, read as in hades/round_constants.rs:11-35 and hades/mds_matrix.rs:8-30).
"""
import hashlib
import os

from .composer import Composer, Constraint, R

WIDTH = 5                       # hades.rs:14
FULL_ROUNDS = 8                 # hades.rs:10
PARTIAL_ROUNDS = 60             # hades.rs:12
ROUNDS = FULL_ROUNDS + PARTIAL_ROUNDS
ARITY = 4                       # poseidon-merkle/src/lib.rs:19

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def _read_table(name, count):
    with open(os.path.join(_ASSETS, name), "rb") as f:
        raw = f.read()
    if len(raw) < 32 * count:
        raise ValueError(f"{name}: {len(raw)} bytes, {32 * count} needed")
    values = [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(count)]
    if any(v >= R for v in values):
        raise ValueError(f"{name}: entry not below the field modulus")
    return values


_arc = _read_table("poseidon_arc.bin", WIDTH * ROUNDS)
ROUND_CONSTANTS = [_arc[WIDTH * r:WIDTH * (r + 1)] for r in range(ROUNDS)]
_mds = _read_table("poseidon_mds.bin", WIDTH * WIDTH)
MDS_MATRIX = [_mds[WIDTH * r:WIDTH * (r + 1)] for r in range(WIDTH)]


def hash_to_scalar(data):
    """BlsScalar::hash_to_scalar (coset-bls12_381/src/scalar/coset.rs:260-279): 64 bytes of BLAKE2b, read as a
    little-endian integer and reduced."""
    return int.from_bytes(hashlib.blake2b(bytes(data), digest_size=64).digest(), "little") % R


class _Hades:
    """The round schedule shared by both forms (hades/permutation.rs:27-67)."""

    def apply_permutation(self, state):
        half = FULL_ROUNDS // 2
        for r in range(ROUNDS):
            full = r < half or r >= half + PARTIAL_ROUNDS
            state = self.add_round_constants(r, state)
            if full:
                state = [self.quintic_s_box(v) for v in state]
            else:
                state = state[:WIDTH - 1] + [self.quintic_s_box(state[WIDTH - 1])]
            state = self.apply_mds_matrix(r, state)
        return state

    def permute(self, state):
        assert len(state) == WIDTH
        return self.apply_permutation(list(state))


class ScalarPermutation(_Hades):
    """hades/permutation/scalar.rs:19-67, on canonical integers"""

    def tag(self, data):
        return hash_to_scalar(data)

    def add(self, right, left):
        return (right + left) % R

    def add_round_constants(self, r, state):
        return [(v + k) % R for v, k in zip(state, ROUND_CONSTANTS[r])]

    def quintic_s_box(self, v):
        return pow(v, 5, R)

    def apply_mds_matrix(self, r, state):
        return [sum(m * v for m, v in zip(row, state)) % R for row in MDS_MATRIX]


class GadgetPermutation(_Hades):
    """hades/permutation/gadget.rs:22-106, on witnesses of a Composer.  Only round 0 adds its constants with gates
    of their own; every later round's constants ride on the preceding MDS gates."""

    def __init__(self, composer):
        self.composer = composer

    def tag(self, data):
        return self.composer.append_constant(hash_to_scalar(data))

    def add(self, right, left):
        return self.composer.gate_add(Constraint().left(1).a(left).right(1).b(right))

    def add_round_constants(self, r, state):
        if r != 0:
            return state
        return [self.composer.gate_add(Constraint().left(1).a(w).constant(k))
                for w, k in zip(state, ROUND_CONSTANTS[0])]

    def quintic_s_box(self, w):
        c = self.composer
        w2 = c.gate_mul(Constraint().mult(1).a(w).b(w))
        w4 = c.gate_mul(Constraint().mult(1).a(w2).b(w2))
        return c.gate_mul(Constraint().mult(1).a(w4).b(w))

    def apply_mds_matrix(self, r, state):
        c = self.composer
        out = []
        for j in range(WIDTH):
            k = ROUND_CONSTANTS[r + 1][j] if r + 1 < ROUNDS else 0
            row = MDS_MATRIX[j]
            first = c.gate_add(Constraint().left(row[0]).a(state[0]).right(row[1]).b(state[1])
                               .fourth(row[2]).d(state[2]))
            out.append(c.gate_add(Constraint().left(row[3]).a(state[3]).right(row[4]).b(state[4])
                                  .fourth(1).d(first).constant(k)))
        return out


# ---- the sponge (dusk-safe 0.3) -------------------------------------------------------------------------------------
ABSORB, SQUEEZE = "absorb", "squeeze"


class IOPatternViolation(ValueError):
    """coset-poseidon/src/error.rs, dusk-safe's Error::IOPatternViolation"""


def tag_input(iopattern, domain_sep):
    """The bytes the tag is hashed from: neighbouring calls of one kind merged, each length as a big-endian u32
    with the top bit set for absorbing calls, then the domain separator as a big-endian u64."""
    merged = []
    for kind, length in iopattern:
        if merged and merged[-1][0] == kind:
            merged[-1][1] += length
        else:
            merged.append([kind, length])
    out = b"".join(((0x80000000 | n) if kind == ABSORB else n).to_bytes(4, "big") for kind, n in merged)
    return out + int(domain_sep).to_bytes(8, "big")


class Sponge:
    """Capacity 1, rate WIDTH - 1.  `safe` supplies permute / tag / add (ScalarPermutation or GadgetPermutation);
    `zero` is the additive identity in its value type."""
    RATE = WIDTH - 1

    def __init__(self, safe, iopattern, domain_sep, zero=0):
        iopattern = list(iopattern)
        if (not iopattern or iopattern[0][0] != ABSORB or iopattern[-1][0] != SQUEEZE
                or any(n == 0 for _, n in iopattern)):
            raise IOPatternViolation("pattern must start absorbing, end squeezing and hold no empty call")
        self.safe, self.iopattern, self.io_count = safe, iopattern, 0
        self.state = [safe.tag(tag_input(iopattern, domain_sep))] + [zero] * self.RATE
        self.pos_absorb = self.pos_squeeze = 0
        self.output = []

    def _expect(self, kind, length):
        if self.io_count >= len(self.iopattern) or self.iopattern[self.io_count] != (kind, length) or length == 0:
            raise IOPatternViolation(f"call {self.io_count}: {kind}({length}) is not what the pattern announces")
        self.io_count += 1

    def absorb(self, length, values):
        self._expect(ABSORB, length)
        if len(values) < length:
            raise IOPatternViolation("fewer values than announced")
        for v in values[:length]:
            if self.pos_absorb == self.RATE:
                self.state = self.safe.permute(self.state)
                self.pos_absorb = 0
            pos = self.pos_absorb + 1
            self.state[pos] = self.safe.add(self.state[pos], v)
            self.pos_absorb += 1
        self.pos_squeeze = self.RATE          # the next squeeze starts with a permutation

    def squeeze(self, length):
        self._expect(SQUEEZE, length)
        for _ in range(length):
            if self.pos_squeeze == self.RATE:
                self.state = self.safe.permute(self.state)
                self.pos_squeeze = self.pos_absorb = 0
            self.output.append(self.state[self.pos_squeeze + 1])
            self.pos_squeeze += 1

    def finish(self):
        if self.io_count != len(self.iopattern):
            raise IOPatternViolation("calls left over in the pattern")
        return self.output


# ---- hash.rs --------------------------------------------------------------------------------------------------------
class Domain:
    """hash.rs:15-39: the domain separators"""
    MERKLE4 = 0x0f
    MERKLE2 = 0x03
    ENCRYPTION = 1 << 32
    OTHER = 0


def _io_pattern(domain, segments, output_len):
    """hash.rs:42-67"""
    total = sum(len(s) for s in segments)
    if domain == Domain.MERKLE2 and (total != 2 or output_len != 1):
        raise IOPatternViolation("Merkle2 hashes two values into one")
    if domain == Domain.MERKLE4 and (total != 4 or output_len != 1):
        raise IOPatternViolation("Merkle4 hashes four values into one")
    return [(ABSORB, len(s)) for s in segments] + [(SQUEEZE, output_len)]


def _run_sponge(safe, domain, segments, output_len, zero):
    if domain != Domain.OTHER or output_len <= 0:          # hash.rs:86-90: only Domain::Other may widen the output
        output_len = 1
    sponge = Sponge(safe, _io_pattern(domain, segments, output_len), domain, zero)
    for s in segments:
        sponge.absorb(len(s), s)
    sponge.squeeze(output_len)
    return sponge.finish()


def digest(domain, values, output_len=1):
    """Hash::digest (hash.rs:144-148); `values` canonical integers"""
    return _run_sponge(ScalarPermutation(), domain, [[int(v) % R for v in values]], output_len, 0)


def digest_gadget(composer, domain, witnesses, output_len=1):
    """HashGadget::digest (hash/gadget.rs:79-87)"""
    return _run_sponge(GadgetPermutation(composer), domain, [list(witnesses)], output_len, Composer.ZERO)


# ---- the tree and its openings --------------------------------------------------------------------------------------
class Opening:
    """coset-merkle/src/opening.rs:19-23: root, the ARITY hashes of every level on the path (level 0 right under
    the root), and the child index taken at every level"""

    def __init__(self, root, branch, positions):
        self.root, self.branch, self.positions = root, branch, positions

    def verify(self, leaf):
        """opening.rs:68-97"""
        item = int(leaf) % R
        for level in reversed(range(len(self.branch))):
            if item != self.branch[level][self.positions[level]]:
                return False
            item = digest(Domain.MERKLE4, self.branch[level])[0]
        return item == self.root


class Tree:
    """Sparse ARITY-ary tree of the given height whose nodes carry Poseidon hashes: an inner node hashes its
    children under Domain::Merkle4 and a missing child counts as zero (poseidon-merkle/src/lib.rs:133-160)."""

    def __init__(self, height, native=False):
        self.height = height
        self.leaves = {}
        self._nodes = None
        self.native = native          # hash the inner nodes with the library's host code (same values, much faster)

    def capacity(self):
        return ARITY ** self.height

    def insert(self, position, leaf_hash):
        if not 0 <= position < self.capacity():
            raise IndexError("position outside the tree")
        self.leaves[position] = int(leaf_hash) % R
        self._nodes = None

    def _build(self):
        # nodes[h][index at depth h]; depth `height` holds the leaves, depth 0 the root
        if self._nodes is None:
            levels = [None] * (self.height + 1)
            levels[self.height] = dict(self.leaves)
            for h in reversed(range(self.height)):
                indexes = sorted({i // ARITY for i in levels[h + 1]})
                groups = [[levels[h + 1].get(ARITY * index + k, 0) for k in range(ARITY)] for index in indexes]
                hashes = merkle4_native(groups) if self.native else [digest(Domain.MERKLE4, g)[0] for g in groups]
                levels[h] = dict(zip(indexes, hashes))
            self._nodes = levels
        return self._nodes

    def root(self):
        return self._build()[0].get(0, 0)

    def opening(self, position):
        """opening.rs:32-45,180-207; None when no leaf sits at the position (tree.rs `opening`)"""
        if position not in self.leaves:
            return None
        nodes = self._build()
        branch, positions = [], []
        for h in range(self.height):
            child = position // ARITY ** (self.height - h - 1)       # index at depth h + 1
            parent = child // ARITY
            branch.append([nodes[h + 1].get(ARITY * parent + k, 0) for k in range(ARITY)])
            positions.append(child % ARITY)
        return Opening(self.root(), branch, positions)


# ---- circuits ------------------------------------------------------------------------------------------------------
def opening_gadget(composer, opening, leaf):
    """poseidon-merkle/src/zk.rs:16-102: one-hot position bits per level, the selected sibling tied to the running
    hash, and the level hashed into its parent; returns the witness of the recomputed root."""
    height = len(opening.branch)
    levels = [None] * height
    bits = [None] * height
    for h in reversed(range(height)):
        bits[h], levels[h] = [], []
        for k, item in enumerate(opening.branch[h]):
            bits[h].append(composer.append_witness(1 if k == opening.positions[h] else 0))
            levels[h].append(composer.append_witness(item))
            composer.component_boolean(bits[h][k])
        s = composer.gate_add(Constraint().left(1).a(bits[h][0]).right(1).b(bits[h][1]).fourth(1).d(bits[h][2]))
        s = composer.gate_add(Constraint().left(1).a(s).right(1).b(bits[h][3]))
        composer.assert_equal_constant(s, 1)
    current = leaf
    for h in reversed(range(height)):
        for k in range(ARITY):
            picked = composer.gate_mul(Constraint().mult(1).a(bits[h][k]).b(levels[h][k]))
            carried = composer.gate_mul(Constraint().mult(1).a(bits[h][k]).b(current))
            composer.assert_equal(picked, carried)
        current = digest_gadget(composer, Domain.MERKLE4, levels[h])[0]
    return current


def opening_circuit(opening, leaf_hash):
    """OpeningCircuit::circuit (merkle-plonk/src/lib.rs:117-131): the recomputed root against the public root"""
    composer = Composer.initialized()
    leaf = composer.append_witness(leaf_hash)
    root = opening_gadget(composer, opening, leaf)
    composer.append_gate(Constraint().left(-1).a(root).public(opening.root))
    return composer


def permutation_circuit(input_state, expected_state):
    """The reference's own test circuit for the gadget (hades/permutation/gadget.rs:143-180): permute five
    witnesses and tie the result to five claimed outputs."""
    composer = Composer.initialized()
    inputs = [composer.append_witness(v) for v in input_state]
    expected = [composer.append_witness(v) for v in expected_state]
    for got, want in zip(GadgetPermutation(composer).permute(inputs), expected):
        composer.assert_equal(got, want)
    return composer


# ---- the native generator (zkvm_amd/csrc/frontend.hip) ---------------------------------------------------------------
def _mont_table(values):
    import numpy as np
    from . import circuits
    return circuits.ints_to_limbs([circuits.to_mont(v) for v in values])


def _native_constants():
    import numpy as np
    arc = _mont_table([k for row in ROUND_CONSTANTS for k in row])
    mds = _mont_table([m for row in MDS_MATRIX for m in row])
    tag = _mont_table([hash_to_scalar(tag_input(_io_pattern(Domain.MERKLE4, [[0] * ARITY], 1), Domain.MERKLE4))])
    return np.ascontiguousarray(arc), np.ascontiguousarray(mds), np.ascontiguousarray(tag)


def merkle4_native(groups):
    """Hash::digest(Domain::Merkle4, ...) of many groups of four canonical integers by the library's host code
    -> list of canonical integers.  Same values as digest(); a tree of many leaves builds much faster with it."""
    import numpy as np
    from . import _lib, circuits
    flat = [int(v) % R for g in groups for v in g]
    assert len(flat) % ARITY == 0
    count = len(flat) // ARITY
    inp = np.ascontiguousarray(_mont_table(flat))
    out = np.zeros((count, 4), dtype=np.uint64)
    arc, mds, tag = _native_constants()
    rc = _lib.lib().zkgpu_poseidon_merkle4(inp.ctypes.data, count, arc.ctypes.data, mds.ctypes.data, tag.ctypes.data,
                                           out.ctypes.data)
    if rc != 0:
        raise _lib.ZkGpuError(rc, "poseidon_merkle4")
    return [v * circuits.MONT_INV % R for v in circuits.limbs_to_ints(out)]


def batch_opening_witness(openings, leaf_hashes, threads=8, out=None):
    """The per-proof half of batch_opening_circuit: the wire columns (4 x n) and the k public roots of the same circuit
    for other leaves, as the reference's Prover::prove regenerates them for every proof
    (plonk/src/composer/prover.rs:214-238).  `out` = (wires, public_inputs) arrays to fill again.  Returns them."""
    import ctypes as C
    import numpy as np
    from . import _lib
    k = len(openings)
    assert k >= 1 and len(leaf_hashes) == k
    height = len(openings[0].branch)
    assert all(len(o.branch) == height and len(o.positions) == height for o in openings)
    lib = _lib.lib()
    gates, n = C.c_uint64(0), C.c_uint64(0)
    rc = lib.zkgpu_merkle_batch_circuit_size(height, k, C.byref(gates), C.byref(n))
    if rc != 0:
        raise _lib.ZkGpuError(rc, "merkle_batch_circuit_size")
    n = int(n.value)
    leaves = np.ascontiguousarray(_mont_table(leaf_hashes))
    branches = np.ascontiguousarray(_mont_table([v for o in openings for level in o.branch for v in level]))
    positions = np.ascontiguousarray([p for o in openings for p in o.positions], dtype=np.uint8)
    roots = np.ascontiguousarray(_mont_table([o.root for o in openings]))
    arc, mds, tag = _native_constants()
    if out is None:
        out = (np.empty((4, n, 4), dtype=np.uint64), np.empty((k, 4), dtype=np.uint64))
    wires, pi_val = out
    assert wires.shape == (4, n, 4) and pi_val.shape == (k, 4) and wires.flags.c_contiguous and pi_val.flags.c_contiguous
    rc = lib.zkgpu_merkle_batch_witness(height, k, leaves.ctypes.data, branches.ctypes.data, positions.ctypes.data,
                                        roots.ctypes.data, arc.ctypes.data, mds.ctypes.data, tag.ctypes.data, n,
                                        wires.ctypes.data, pi_val.ctypes.data, int(threads))
    if rc != 0:
        raise _lib.ZkGpuError(rc, "merkle_batch_witness")
    return wires, pi_val


def batch_opening_circuit(openings, leaf_hashes, threads=8):
    """One circuit proving k Merkle openings (trees of one height), each tied to its public root: the reference's
    OpeningCircuit (merkle-plonk/src/lib.rs:117-131) repeated k times on one Composer, generated natively.  Returns
    the circuit dict of Composer.tables(): for k = 1 the same arrays as opening_circuit(...).tables()."""
    import ctypes as C
    import numpy as np
    from . import _lib
    k = len(openings)
    assert k >= 1 and len(leaf_hashes) == k
    height = len(openings[0].branch)
    assert all(len(o.branch) == height and len(o.positions) == height for o in openings)
    lib = _lib.lib()
    gates, n = C.c_uint64(0), C.c_uint64(0)
    rc = lib.zkgpu_merkle_batch_circuit_size(height, k, C.byref(gates), C.byref(n))
    if rc != 0:
        raise _lib.ZkGpuError(rc, "merkle_batch_circuit_size")
    n = int(n.value)
    leaves = np.ascontiguousarray(_mont_table(leaf_hashes))
    branches = np.ascontiguousarray(_mont_table([v for o in openings for level in o.branch for v in level]))
    positions = np.ascontiguousarray([p for o in openings for p in o.positions], dtype=np.uint8)
    roots = np.ascontiguousarray(_mont_table([o.root for o in openings]))
    arc, mds, tag = _native_constants()
    sel = np.zeros((15, n, 4), dtype=np.uint64)
    sigma = np.zeros((4, n), dtype=np.uint64)
    wires = np.zeros((4, n, 4), dtype=np.uint64)
    pi_idx = np.zeros(k, dtype=np.uint64)
    pi_val = np.zeros((k, 4), dtype=np.uint64)
    rc = lib.zkgpu_merkle_batch_circuit(height, k, leaves.ctypes.data, branches.ctypes.data, positions.ctypes.data,
                                        roots.ctypes.data, arc.ctypes.data, mds.ctypes.data, tag.ctypes.data, n,
                                        sel.ctypes.data, sigma.ctypes.data, wires.ctypes.data, pi_idx.ctypes.data,
                                        pi_val.ctypes.data, int(threads))
    if rc != 0:
        raise _lib.ZkGpuError(rc, "merkle_batch_circuit")
    return dict(n=n, constraints=int(gates.value), selectors=sel, sigma_map=sigma, wires=wires, public_inputs=pi_val,
                pi_indexes=[int(i) for i in pi_idx])
