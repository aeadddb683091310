"""BLS12-381 optimal-ate pairing on Python integers: a CPU restatement of the reference's pairing stack, used only
by the tests to run the reference verifier's final check on proofs (test infrastructure, never on the product path).

  Fp2 = Fp[u]/(u^2 + 1)            coset-bls12_381/src/fp2.rs:117-240
  Fp6 = Fp2[v]/(v^3 - (u + 1))     coset-bls12_381/src/fp6.rs:129-300
  Fp12 = Fp6[w]/(w^2 - v)          coset-bls12_381/src/fp12.rs:127-222
  G2: y^2 = x^3 + 4(u + 1)         coset-bls12_381/src/g2.rs:411-483 (compressed encoding), :1109-1122 (generator)
  Miller loop, line functions      coset-bls12_381/src/pairings.rs:695-808
  final exponentiation             coset-bls12_381/src/pairings.rs:62-189

Elements are plain tuples: Fp2 = (c0, c1), Fp6 = (Fp2, Fp2, Fp2), Fp12 = (Fp6, Fp6).  Pinned by the reference's own
vectors: the 1000 compressed multiples of the G2 generator and e(G1, G2) (tests/test_oracle_pairing.py)."""

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
BLS_X = 0xd201000000010000          # the curve parameter is -BLS_X (lib.rs: BLS_X, BLS_X_IS_NEGATIVE)

# ---- Fp2 -------------------------------------------------------------------------------------------------
F2_ZERO = (0, 0)
F2_ONE = (1, 0)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f2_mul_nr(a):
    """times the cubic non-residue u + 1 (fp2.rs:173-180)"""
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


def f2_inv(a):
    t = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * t % P, (-a[1]) * t % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


def fp_sqrt(a):
    """square root in Fp (p = 3 mod 4), or None"""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


def f2_sqrt(a):
    """some square root of a, or None (the caller picks the sign, as g2.rs:445-452 does through the sort flag)"""
    if a == F2_ZERO:
        return F2_ZERO
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = fp_sqrt((-a0) % P)          # sqrt(-1) = u
        return (0, s)
    n = fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    inv2 = pow(2, -1, P)
    for cand in ((a0 + n) * inv2 % P, (a0 - n) * inv2 % P):
        x0 = fp_sqrt(cand)
        if x0 is not None and x0 != 0:
            r = (x0, a1 * pow(2 * x0, -1, P) % P)
            if f2_sqr(r) == (a0 % P, a1 % P):
                return r
    return None


def fp_lex_largest(a):
    return a > (P - 1) // 2


def f2_lex_largest(a):
    """fp2.rs:182-186"""
    return fp_lex_largest(a[1]) or (a[1] == 0 and fp_lex_largest(a[0]))


# ---- Fp6 -------------------------------------------------------------------------------------------------
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


def f6_add(a, b):
    return (f2_add(a[0], b[0]), f2_add(a[1], b[1]), f2_add(a[2], b[2]))


def f6_sub(a, b):
    return (f2_sub(a[0], b[0]), f2_sub(a[1], b[1]), f2_sub(a[2], b[2]))


def f6_neg(a):
    return (f2_neg(a[0]), f2_neg(a[1]), f2_neg(a[2]))


def f6_mul(a, b):
    """schoolbook product modulo v^3 = u + 1"""
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = f2_add(f2_mul(a0, b0), f2_mul_nr(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_nr(f2_mul(a2, b2)))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (c0, c1, c2)


def f6_mul_nr(a):
    """times v (fp6.rs:154-160)"""
    return (f2_mul_nr(a[2]), a[0], a[1])


def f6_mul_by_1(a, c1):
    """a * (c1 v)  (fp6.rs:129-135)"""
    return (f2_mul_nr(f2_mul(a[2], c1)), f2_mul(a[0], c1), f2_mul(a[1], c1))


def f6_mul_by_01(a, c0, c1):
    """a * (c0 + c1 v)  (fp6.rs:137-152)"""
    return f6_mul(a, (c0, c1, F2_ZERO))


def f6_inv(a):
    """fp6.rs:282-300"""
    a0, a1, a2 = a
    c0 = f2_sub(f2_sqr(a0), f2_mul_nr(f2_mul(a1, a2)))
    c1 = f2_sub(f2_mul_nr(f2_sqr(a2)), f2_mul(a0, a1))
    c2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    t = f2_add(f2_mul(a0, c0), f2_mul_nr(f2_add(f2_mul(a2, c1), f2_mul(a1, c2))))
    t = f2_inv(t)
    return (f2_mul(c0, t), f2_mul(c1, t), f2_mul(c2, t))


# Frobenius constants from their definitions: v^p = v xi^((p-1)/3), (v^2)^p = v^2 xi^(2(p-1)/3),
# w^p = w xi^((p-1)/6) with xi = u + 1 (the values written out in fp6.rs:168-192 and fp12.rs:159-177).
XI = (1, 1)
FROB6_C1 = f2_pow(XI, (P - 1) // 3)
FROB6_C2 = f2_pow(XI, 2 * (P - 1) // 3)
FROB12_C1 = f2_pow(XI, (P - 1) // 6)


def f6_frobenius(a):
    return (f2_conj(a[0]), f2_mul(f2_conj(a[1]), FROB6_C1), f2_mul(f2_conj(a[2]), FROB6_C2))


# ---- Fp12 ------------------------------------------------------------------------------------------------
F12_ONE = (F6_ONE, F6_ZERO)


def f12_mul(a, b):
    aa = f6_mul(a[0], b[0])
    bb = f6_mul(a[1], b[1])
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), aa), bb)
    return (f6_add(aa, f6_mul_nr(bb)), c1)


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_inv(a):
    """fp12.rs:196-203"""
    t = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mul_nr(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], t), f6_neg(f6_mul(a[1], t)))


def f12_frobenius(a):
    c1 = f6_frobenius(a[1])
    c1 = tuple(f2_mul(x, FROB12_C1) for x in c1)
    return (f6_frobenius(a[0]), c1)


def f12_mul_by_014(f, c0, c1, c4):
    """f * (c0 + c1 v + c4 v w): the sparse product by a line value (fp12.rs:127-139)"""
    aa = f6_mul_by_01(f[0], c0, c1)
    bb = f6_mul_by_1(f[1], c4)
    mid = f6_mul_by_01(f6_add(f[1], f[0]), c0, f2_add(c1, c4))
    return (f6_add(f6_mul_nr(bb), aa), f6_sub(f6_sub(mid, aa), bb))


# ---- G2 --------------------------------------------------------------------------------------------------
G2_B = (4, 4)


def g2_is_on_curve(q):
    if q is None:
        return True
    x, y = q
    return f2_sqr(y) == f2_add(f2_mul(f2_sqr(x), x), G2_B)


def g2_add(a, b):
    """affine addition; None is the identity"""
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if a[1] != b[1] or a[1] == F2_ZERO:
            return None
        lam = f2_mul(f2_scale(f2_sqr(a[0]), 3), f2_inv(f2_scale(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))


def g2_neg(a):
    return None if a is None else (a[0], f2_neg(a[1]))


def g2_mul(a, k):
    k %= R_ORDER
    r = None
    while k:
        if k & 1:
            r = g2_add(r, a)
        a = g2_add(a, a)
        k >>= 1
    return r


def g2_from_compressed(b):
    """g2.rs:411-466: x.c1 | x.c0 big-endian with the three flag bits on top; None for the identity.
    Raises ValueError on an invalid encoding."""
    if len(b) != 96 or not (b[0] >> 7) & 1:
        raise ValueError("not a compressed G2 encoding")
    infinity, sort = (b[0] >> 6) & 1, (b[0] >> 5) & 1
    xc1 = int.from_bytes(bytes([b[0] & 0x1f]) + bytes(b[1:48]), "big")
    xc0 = int.from_bytes(bytes(b[48:96]), "big")
    if xc0 >= P or xc1 >= P:
        raise ValueError("coordinate not reduced")
    if infinity:
        if sort or xc0 or xc1:
            raise ValueError("bad identity encoding")
        return None
    x = (xc0, xc1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), G2_B))
    if y is None:
        raise ValueError("not on the curve")
    if f2_lex_largest(y) != bool(sort):
        y = f2_neg(y)
    return (x, y)


def g2_to_compressed(q):
    """g2/coset.rs to_bytes"""
    if q is None:
        return bytes([0xc0]) + bytes(95)
    x, y = q
    out = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    out[0] |= 0x80 | (0x20 if f2_lex_largest(y) else 0)
    return bytes(out)


# ---- Miller loop (pairings.rs:695-808) -------------------------------------------------------------------
def _doubling_step(r):
    """r <- 2r in Jacobian coordinates; returns the tangent's three coefficients (pairings.rs:746-775)"""
    x, y, z = r
    xx, yy = f2_sqr(x), f2_sqr(y)
    yyyy = f2_sqr(yy)
    s = f2_scale(f2_sub(f2_sub(f2_sqr(f2_add(yy, x)), xx), yyyy), 2)
    m = f2_scale(xx, 3)
    mm = f2_sqr(m)
    zz = f2_sqr(z)
    x3 = f2_sub(mm, f2_scale(s, 2))
    z3 = f2_sub(f2_sub(f2_sqr(f2_add(z, y)), yy), zz)
    y3 = f2_sub(f2_mul(f2_sub(s, x3), m), f2_scale(yyyy, 8))
    c1 = f2_neg(f2_scale(f2_mul(m, zz), 2))
    c2 = f2_sub(f2_sub(f2_sub(f2_sqr(f2_add(x, m)), xx), mm), f2_scale(yy, 4))
    c0 = f2_scale(f2_mul(z3, zz), 2)
    return (x3, y3, z3), (c0, c1, c2)


def _addition_step(r, q):
    """r <- r + q (q affine); returns the chord's three coefficients (pairings.rs:778-808)"""
    x, y, z = r
    qx, qy = q
    zz = f2_sqr(z)
    qyy = f2_sqr(qy)
    u2 = f2_mul(zz, qx)
    s2 = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(qy, z)), qyy), zz), zz)
    h = f2_sub(u2, x)
    hh = f2_sqr(h)
    i = f2_scale(hh, 4)
    j = f2_mul(i, h)
    rr = f2_sub(s2, f2_scale(y, 2))
    v = f2_mul(i, x)
    x3 = f2_sub(f2_sub(f2_sqr(rr), j), f2_scale(v, 2))
    z3 = f2_sub(f2_sub(f2_sqr(f2_add(z, h)), zz), hh)
    y3 = f2_sub(f2_mul(f2_sub(v, x3), rr), f2_scale(f2_mul(y, j), 2))
    t = f2_sub(f2_sub(f2_sqr(f2_add(qy, z3)), qyy), f2_sqr(z3))
    c2 = f2_sub(f2_scale(f2_mul(rr, qx), 2), t)
    return (x3, y3, z3), (f2_scale(z3, 2), f2_scale(f2_neg(rr), 2), c2)


def _apply_line(f, coeffs, p):
    """pairings.rs:728-743: c0 scaled by p.y, c1 by p.x"""
    c0, c1, c2 = coeffs
    return f12_mul_by_014(f, c2, f2_scale(c1, p[0]), f2_scale(c0, p[1]))


def multi_miller_loop(terms):
    """terms: [(G1 affine (x, y) or None, G2 affine or None)] -> Fp12 before the final exponentiation.
    Pairs with an identity on either side contribute nothing (pairings.rs:581-589)."""
    live = [(p, q) for p, q in terms if p is not None and q is not None]
    cur = [(q[0], q[1], F2_ONE) for _, q in live]
    f = F12_ONE
    bits = bin(BLS_X >> 1)[2:]                 # from the top set bit down; the top bit itself is skipped
    for bit in bits[1:]:
        for k, (p, q) in enumerate(live):
            cur[k], line = _doubling_step(cur[k])
            f = _apply_line(f, line, p)
        if bit == "1":
            for k, (p, q) in enumerate(live):
                cur[k], line = _addition_step(cur[k], q)
                f = _apply_line(f, line, p)
        f = f12_sqr(f)
    for k, (p, q) in enumerate(live):
        cur[k], line = _doubling_step(cur[k])
        f = _apply_line(f, line, p)
    return f12_conj(f)                         # the curve parameter is negative


# ---- final exponentiation (pairings.rs:62-189) -----------------------------------------------------------
def _fp4_square(a, b):
    t0, t1 = f2_sqr(a), f2_sqr(b)
    return f2_add(f2_mul_nr(t1), t0), f2_sub(f2_sub(f2_sqr(f2_add(a, b)), t0), t1)


def _cyclotomic_square(f):
    """squaring in the cyclotomic subgroup (Granger-Scott), pairings.rs:79-124"""
    z0, z4, z3 = f[0]
    z2, z1, z5 = f[1]
    t0, t1 = _fp4_square(z0, z1)
    z0 = f2_add(f2_scale(f2_sub(t0, z0), 2), t0)
    z1 = f2_add(f2_scale(f2_add(t1, z1), 2), t1)
    t0, t1 = _fp4_square(z2, z3)
    t2, t3 = _fp4_square(z4, z5)
    z4 = f2_add(f2_scale(f2_sub(t0, z4), 2), t0)
    z5 = f2_add(f2_scale(f2_add(t1, z5), 2), t1)
    t0 = f2_mul_nr(t3)
    z2 = f2_add(f2_scale(f2_add(t0, z2), 2), t0)
    z3 = f2_add(f2_scale(f2_sub(t2, z3), 2), t2)
    return ((z0, z4, z3), (z2, z1, z5))


def _cyclotomic_exp_by_x(f):
    """f^(-BLS_X) for f in the cyclotomic subgroup (pairings.rs:126-147)"""
    acc = F12_ONE
    for k, bit in enumerate(bin(BLS_X)[2:]):
        if k:
            acc = _cyclotomic_square(acc)
        if bit == "1":
            acc = f12_mul(acc, f)
    return f12_conj(acc)


def _frob(f, times):
    for _ in range(times):
        f = f12_frobenius(f)
    return f


def final_exponentiation(f):
    """The reference's chain: the easy part f^((p^6 - 1)(p^2 + 1)), then its hard-part addition chain."""
    t2 = f12_mul(_frob(f, 6), f12_inv(f))
    t1 = t2
    t2 = f12_mul(_frob(t2, 2), t1)
    t1 = f12_conj(_cyclotomic_square(t2))
    t3 = _cyclotomic_exp_by_x(t2)
    t4 = _cyclotomic_square(t3)
    t5 = f12_mul(t1, t3)
    t1 = _cyclotomic_exp_by_x(t5)
    t0 = _cyclotomic_exp_by_x(t1)
    t6 = f12_mul(_cyclotomic_exp_by_x(t0), t4)
    t4 = _cyclotomic_exp_by_x(t6)
    t5 = f12_conj(t5)
    t4 = f12_mul(t4, f12_mul(t5, t2))
    t5 = f12_conj(t2)
    t1 = _frob(f12_mul(t1, t2), 3)
    t6 = _frob(f12_mul(t6, t5), 1)
    t3 = _frob(f12_mul(t3, t0), 2)
    t3 = f12_mul(f12_mul(t3, t1), t6)
    return f12_mul(t3, t4)


def pairing(p, q):
    """e(p, q) in GT; the identity of GT when either point is the identity (pairings.rs:628-682)"""
    return final_exponentiation(multi_miller_loop([(p, q)]))


def pairing_product_is_one(terms):
    """prod e(p_i, q_i) == 1, the form the verifier uses (plonk/src/proof_system/proof.rs:393-401)"""
    return final_exponentiation(multi_miller_loop(terms)) == F12_ONE
