"""Host-side mirror of the reference's circuit front-end: `Constraint`
(plonk/src/composer/constraint_system/constraint.rs) and `Composer` (plonk/src/composer.rs) with the gates the
Poseidon-Merkle opening circuit uses (poseidon-merkle/src/zk.rs:16-102: gate_add, gate_mul, assert_equal,
component_boolean, component_select, constants and public inputs), and the copy-constraint bookkeeping of
`Permutation` (plonk/src/composer/permutation.rs:30-109).  It produces exactly the tables the prover consumes:
15 selector rows, the sigma map, the four witness columns and the public-input positions.

The range and logic components (composer.rs:224-313, 855-934), the JubJub point addition (composer.rs:669-703) and
the fixed-base scalar multiplication (composer.rs:315-443) are mirrored as well, so that circuits exercising every
widget of the prover can be built; component_mul_point, decomposition and the select-point helpers are compositions
of these and are left out.  Values are canonical integers modulo r; tables come out as Montgomery limbs.
"""
import numpy as np

from . import circuits

R = circuits.R_MOD

_COEFFS = ("q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "pi", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
           "q_variable_group_add")
_EXTERNAL = _COEFFS[:7]          # what Constraint::from_external keeps (constraint.rs:90-104)


# ---- JubJub, the twisted Edwards curve -u^2 + v^2 = 1 + d u^2 v^2 over the BLS12-381 scalar field -----------------
EDWARDS_D = (-10240 * pow(10241, -1, R)) % R                 # coset-jubjub/src/lib.rs:368-373
assert EDWARDS_D == 0x2a9318e74bfa2b48f5fd9207e6bd7fd4292d7f6d37579d2601065fd6d6343eb1
JUBJUB_SCALAR_MODULUS = 0x0e7db4ea6533afa906673b0101343b00a6682093ccc81082d0970e5ed6f72cb7   # coset-jubjub/src/fr.rs:76-81
JUBJUB_GENERATOR = (0x3fd2814c43ac65a6f1fbf02d0fd6cce62e3ebb21fd6c54ed4df7b7ffec7beaca, 0x12)  # coset-jubjub/src/coset.rs:29-42
JUBJUB_IDENTITY = (0, 1)


def jubjub_add(p, q):
    """affine addition (complete on this curve); the reference adds in extended coordinates and normalises"""
    (u1, v1), (u2, v2) = p, q
    t = EDWARDS_D * u1 * u2 % R * v1 % R * v2 % R
    u3 = (u1 * v2 + v1 * u2) * pow(1 + t, -1, R) % R
    v3 = (v1 * v2 + u1 * u2) * pow(1 - t, -1, R) % R
    return (u3, v3)


def jubjub_neg(p):
    return ((-p[0]) % R, p[1])


def jubjub_mul(p, k):
    acc = JUBJUB_IDENTITY
    while k:
        if k & 1:
            acc = jubjub_add(acc, p)
        p = jubjub_add(p, p)
        k >>= 1
    return acc


def windowed_naf_2(scalar):
    """Fr::compute_windowed_naf(2) (coset-jubjub/src/fr/coset.rs:79-109): 256 digits in {-1, 0, 1}, least
    significant first"""
    digits = [0] * 256
    s, i = scalar, 0
    while s >= 1:
        if s & 1:
            d = s & 3
            if d >= 2:
                d -= 4
            digits[i] = d
            s -= d
        s >>= 1
        i += 1
    return digits


class Constraint:
    """constraint.rs:76-216: twelve coefficients, four wired witnesses (all Composer.ZERO by default)."""

    def __init__(self):
        self.coeff = dict.fromkeys(_COEFFS, 0)
        self.w = [0, 0, 0, 0]          # a b c d
        self.has_public_input = False

    def _copy(self):
        c = Constraint()
        c.coeff = dict(self.coeff)
        c.w = list(self.w)
        c.has_public_input = self.has_public_input
        return c

    def _set(self, name, value):
        c = self._copy()
        c.coeff[name] = int(value) % R
        return c

    def mult(self, s):
        return self._set("q_m", s)

    def left(self, s):
        return self._set("q_l", s)

    def right(self, s):
        return self._set("q_r", s)

    def output(self, s):
        return self._set("q_o", s)

    def fourth(self, s):
        return self._set("q_f", s)

    def constant(self, s):
        return self._set("q_c", s)

    def public(self, s):
        c = self._set("pi", s)
        c.has_public_input = True
        return c

    def _wire(self, k, witness):
        c = self._copy()
        c.w[k] = int(witness)
        return c

    def a(self, w):
        return self._wire(0, w)

    def b(self, w):
        return self._wire(1, w)

    def c(self, w):
        return self._wire(2, w)

    def d(self, w):
        return self._wire(3, w)

    @staticmethod
    def arithmetic(s):
        """constraint.rs:181-183: the external coefficients and witnesses, q_arith = 1"""
        c = Constraint()
        for name in _EXTERNAL:
            c.coeff[name] = s.coeff[name]
        c.w = list(s.w)
        c.has_public_input = s.has_public_input
        c.coeff["q_arith"] = 1
        return c

    @staticmethod
    def _external(s):
        c = Constraint.arithmetic(s)
        c.coeff["q_arith"] = 0
        return c

    @staticmethod
    def range(s):
        """constraint.rs:187-189"""
        c = Constraint._external(s)
        c.coeff["q_range"] = 1
        return c

    @staticmethod
    def logic(s):
        """constraint.rs:193-197: AND rows carry q_c = q_logic = 1"""
        c = Constraint._external(s)
        c.coeff["q_c"] = 1
        c.coeff["q_logic"] = 1
        return c

    @staticmethod
    def group_add_fixed_base(s):
        c = Constraint._external(s)
        c.coeff["q_fixed_group_add"] = 1
        return c

    @staticmethod
    def group_add_variable_base(s):
        c = Constraint._external(s)
        c.coeff["q_variable_group_add"] = 1
        return c

    @staticmethod
    def logic_xor(s):
        """constraint.rs:201-205: XOR rows carry q_c = q_logic = -1"""
        c = Constraint._external(s)
        c.coeff["q_c"] = R - 1
        c.coeff["q_logic"] = R - 1
        return c


class Composer:
    """composer.rs:36-1015 (see the module header for what is covered).  Witnesses are indexes into `witnesses`."""
    ZERO, ONE = 0, 1

    def __init__(self):               # Composer::uninitialized, composer.rs:154-162
        self.gates = []               # (coefficients dict, [a, b, c, d])
        self.public_inputs = {}       # gate index -> value
        self.witnesses = []
        self.wire_map = []            # per witness: list of (column, gate index), composer/permutation.rs:45-76

    @classmethod
    def initialized(cls):
        """composer.rs:139-195: the constants 0 and 1 and the two dummy gates"""
        c = cls()
        zero = c.append_witness(0)
        one = c.append_witness(1)
        c.assert_equal_constant(zero, 0)
        c.assert_equal_constant(one, 1)
        six, one_b, seven, min_twenty = (c.append_witness(v) for v in (6, 1, 7, -20))
        c.append_gate(Constraint().mult(1).left(2).right(3).fourth(1).constant(4).output(4)
                      .a(six).b(seven).d(one_b).c(min_twenty))
        c.append_gate(Constraint().mult(1).left(1).right(1).constant(127).output(1).a(min_twenty).b(six).c(seven))
        return c

    def __getitem__(self, witness):
        return self.witnesses[witness]

    def constraints(self):
        return len(self.gates)

    def append_witness(self, value):
        self.witnesses.append(int(value) % R)
        self.wire_map.append([])
        return len(self.witnesses) - 1

    def append_custom_gate(self, constraint):
        """composer.rs:76-131"""
        index = len(self.gates)
        self.gates.append((dict(constraint.coeff), list(constraint.w)))
        if constraint.has_public_input:
            self.public_inputs[index] = constraint.coeff["pi"]
        for column, witness in enumerate(constraint.w):
            self.wire_map[witness].append((column, index))

    def append_gate(self, constraint):
        self.append_custom_gate(Constraint.arithmetic(constraint))

    def append_evaluated_output(self, s):
        """composer.rs:451-501: the output witness that satisfies the gate, or None when q_o = 0"""
        a, b, d = (self.witnesses[s.w[k]] for k in (0, 1, 3))
        k = s.coeff
        value = (k["q_m"] * a * b + k["q_l"] * a + k["q_r"] * b + k["q_f"] * d + k["q_c"] + k["pi"]) % R
        if k["q_o"] == 0:
            return None
        return self.append_witness(value * (-pow(k["q_o"], -1, R)) % R)

    def append_constant(self, constant):
        w = self.append_witness(constant)
        self.assert_equal_constant(w, constant)
        return w

    def append_public(self, public):
        """composer.rs:561-572"""
        w = self.append_witness(public)
        self.append_gate(Constraint().left(-1).a(w).public(public))
        return w

    def assert_equal(self, left, right):
        self.append_gate(Constraint().left(1).right(-1).a(left).b(right))

    def assert_equal_constant(self, witness, constant, public=None):
        c = Constraint().left(-1).a(witness).constant(constant)
        if public is not None:
            c = c.public(public)
        self.append_gate(c)

    def component_boolean(self, witness):
        """composer.rs:705-716: w * w - w = 0"""
        self.append_gate(Constraint().mult(1).output(-1).a(witness).b(witness).c(witness).d(self.ZERO))

    def gate_add(self, constraint):
        """composer.rs:936-948"""
        c = Constraint.arithmetic(constraint).output(-1)
        out = self.append_evaluated_output(c)
        self.append_gate(c.c(out))
        return out

    gate_mul = gate_add               # composer.rs:950-962: the same body

    def component_select(self, bit, a, b):
        """composer.rs:781-803: bit ? a : b"""
        bit_times_a = self.gate_mul(Constraint().mult(1).a(bit).b(a))
        one_min_bit = self.gate_add(Constraint().left(-1).constant(1).a(bit))
        one_min_bit_b = self.gate_mul(Constraint().mult(1).a(one_min_bit).b(b))
        return self.gate_add(Constraint().left(1).right(1).a(one_min_bit_b).b(bit_times_a))

    def component_select_one(self, bit, value):
        """composer.rs:805-829: bit ? value : 1"""
        out = self.append_witness(1 - self[bit] + self[bit] * self[value])
        self.append_gate(Constraint().mult(1).left(-1).output(-1).constant(1).a(bit).b(value).c(out))
        return out

    def component_select_zero(self, bit, value):
        return self.gate_mul(Constraint().mult(1).a(bit).b(value))

    def component_range(self, witness, bit_pairs):
        """composer.rs:855-934: witness < 2^(2 * bit_pairs) through base-4 accumulators, four per range gate"""
        num_bits = min(bit_pairs * 2, 256)
        if num_bits == 0:
            self.append_gate(Constraint().left(1).a(witness))
            return
        value = self.witnesses[witness]
        bits = [(value >> i) & 1 for i in range(256)]        # least significant first
        num_gates = (num_bits + 7) >> 3
        num_quads = num_gates * 4
        pad = 1 + (((num_quads << 1) - num_bits) >> 1)
        rows = [Constraint.range(Constraint()) for _ in range(num_gates + 1)]
        accumulators = []
        acc = 0
        for i in range(pad, num_quads + 1):
            bit_index = (num_quads - i) << 1
            acc = (4 * acc + bits[bit_index] + 2 * bits[bit_index + 1]) % R
            var = self.append_witness(acc)
            accumulators.append(var)
            rows[i // 4].w[(3, 2, 1, 0)[i % 4]] = var        # i % 4: 0 -> d, 1 -> c, 2 -> b, 3 -> a
        rows[-1] = Constraint()
        if accumulators:
            rows[-1].w[3] = accumulators[-1]
        for row in rows:
            self.append_custom_gate(row)
        if accumulators:
            self.assert_equal(accumulators[-1], witness)

    def append_logic_component(self, a, b, bit_pairs, is_xor):
        """composer.rs:224-313: the AND / XOR of the low 2 * bit_pairs bits of a and b, one quad per logic gate"""
        num_bits = min(bit_pairs * 2, 256)
        num_quads = num_bits >> 1
        va, vb = self.witnesses[a], self.witnesses[b]
        a_bits = [(va >> (num_bits - 1 - i)) & 1 for i in range(num_bits)]   # most significant of the window first
        b_bits = [(vb >> (num_bits - 1 - i)) & 1 for i in range(num_bits)]
        left_acc = right_acc = out_acc = 0
        row = Constraint.logic_xor(Constraint()) if is_xor else Constraint.logic(Constraint())
        for i in range(num_quads):
            lq = 2 * a_bits[2 * i] + a_bits[2 * i + 1]
            rq = 2 * b_bits[2 * i] + b_bits[2 * i + 1]
            oq = (lq ^ rq) if is_xor else (lq & rq)
            left_acc = (4 * left_acc + lq) % R
            right_acc = (4 * right_acc + rq) % R
            out_acc = (4 * out_acc + oq) % R
            wa, wb, wc, wd = (self.append_witness(v) for v in (left_acc, right_acc, lq * rq, out_acc))
            row = row.c(wc)
            self.append_custom_gate(row)
            row = row.a(wa).b(wb).d(wd)
        self.append_custom_gate(Constraint().a(row.w[0]).b(row.w[1]).d(row.w[3]))
        return row.w[3]

    def append_logic_and(self, a, b, bit_pairs):
        return self.append_logic_component(a, b, bit_pairs, False)

    def append_logic_xor(self, a, b, bit_pairs):
        return self.append_logic_component(a, b, bit_pairs, True)

    def append_point(self, point):
        """composer.rs:515-525 -> (x witness, y witness)"""
        return (self.append_witness(point[0]), self.append_witness(point[1]))

    def assert_equal_public_point(self, point, public):
        """composer.rs:631-649: both coordinates against public inputs"""
        self.assert_equal_constant(point[0], 0, public[0])
        self.assert_equal_constant(point[1], 0, public[1])

    def component_add_point(self, a, b):
        """composer.rs:669-703: one curve-addition gate; the next row carries the result and x1 * y2"""
        x1, y1 = a
        x2, y2 = b
        p3 = jubjub_add((self[x1], self[y1]), (self[x2], self[y2]))
        x1_y2 = self.append_witness(self[x1] * self[y2])
        x3 = self.append_witness(p3[0])
        y3 = self.append_witness(p3[1])
        self.append_custom_gate(Constraint.group_add_variable_base(Constraint().a(x1).b(y1).c(x2).d(y2)))
        self.append_custom_gate(Constraint().a(x3).b(y3).d(x1_y2))
        return (x3, y3)

    def component_mul_generator(self, jubjub, generator=JUBJUB_GENERATOR):
        """composer.rs:315-443: [jubjub] * generator through 256 fixed-base rounds over the width-2 NAF of the scalar;
        raises ValueError where the reference returns Error::JubJubScalarMalformed"""
        bits = 256
        scalar = self[jubjub]
        if scalar >= JUBJUB_SCALAR_MODULUS:
            raise ValueError("JubJubScalarMalformed")
        multiples = [generator]
        for _ in range(1, bits):
            multiples.append(jubjub_add(multiples[-1], multiples[-1]))
        multiples.reverse()                                   # multiples[i] = 2^(255 - i) * generator
        naf = windowed_naf_2(scalar)
        scalar_acc, point_acc, xy_alphas = [0], [JUBJUB_IDENTITY], []
        for i, entry in enumerate(reversed(naf)):
            if entry == 0:
                to_add = JUBJUB_IDENTITY
            elif entry == 1:
                to_add = multiples[i]
            else:
                to_add = jubjub_neg(multiples[i])
            scalar_acc.append((2 * scalar_acc[i] + entry) % R)
            point_acc.append(jubjub_add(point_acc[i], to_add))
            xy_alphas.append(to_add[0] * to_add[1] % R)
        for i in range(bits):
            acc_x = self.append_witness(point_acc[i][0])
            acc_y = self.append_witness(point_acc[i][1])
            accumulated_bit = self.append_witness(scalar_acc[i])
            if i == 0:
                self.assert_equal_constant(acc_x, 0)
                self.assert_equal_constant(acc_y, 1)
                self.assert_equal_constant(accumulated_bit, 0)
            x_beta, y_beta = multiples[i]
            xy_alpha = self.append_witness(xy_alphas[i])
            self.append_custom_gate(Constraint.group_add_fixed_base(Constraint()).left(x_beta).right(y_beta)
                                    .constant(x_beta * y_beta).a(acc_x).b(acc_y).c(xy_alpha).d(accumulated_bit))
        acc_x = self.append_witness(point_acc[bits][0])
        acc_y = self.append_witness(point_acc[bits][1])
        last_accumulated_bit = self.append_witness(scalar_acc[bits])
        self.append_gate(Constraint().a(acc_x).b(acc_y).d(last_accumulated_bit))
        self.assert_equal(last_accumulated_bit, jubjub)
        return (acc_x, acc_y)

    def public_input_indexes(self):
        return sorted(self.public_inputs)

    def public_input_values(self):
        return [self.public_inputs[i] for i in self.public_input_indexes()]

    # ---- what Compiler::preprocess and Prover::prove read off the composer ---------------------------------
    def size(self):
        """compiler.rs:86-87: constraints.next_power_of_two()"""
        n = 1
        while n < len(self.gates):
            n <<= 1
        return n

    def tables(self):
        """-> circuit dict in the layout of zkvm_amd.circuits: selectors (15, n, 4) with empty sigma rows
        (compiler.rs:96-121), sigma_map (4, n) with entry = column * n + gate (compute_sigma_permutations,
        composer/permutation.rs:78-109), wires (4, n, 4) zero-padded (prover.rs:226-238), public inputs."""
        n = self.size()
        sel = np.zeros((15, n, 4), dtype=np.uint64)
        names = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
                 "q_variable_group_add"]
        for row, name in enumerate(names):
            vals = [circuits.to_mont(g[0][name]) for g in self.gates] + [0] * (n - len(self.gates))
            sel[row] = circuits.ints_to_limbs(vals)
        sigma = np.arange(4 * n, dtype=np.uint64).reshape(4, n)
        for wires in self.wire_map:
            for k, (column, gate) in enumerate(wires):
                nxt_column, nxt_gate = wires[(k + 1) % len(wires)]
                sigma[column, gate] = nxt_column * n + nxt_gate
        cols = []
        for column in range(4):
            vals = [circuits.to_mont(self.witnesses[g[1][column]]) for g in self.gates] + [0] * (n - len(self.gates))
            cols.append(circuits.ints_to_limbs(vals))
        pis = circuits.ints_to_limbs([circuits.to_mont(v) for v in self.public_input_values()])
        return dict(n=n, constraints=len(self.gates), selectors=sel, sigma_map=sigma, wires=np.stack(cols),
                    public_inputs=pis.reshape(-1, 4), pi_indexes=self.public_input_indexes())
