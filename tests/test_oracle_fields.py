"""Pins the CPU oracle's Fr / Fq arithmetic to the known-answer tests the reference holds
(coset-bls12_381/src/scalar.rs:748-1210, fp.rs:646-926) and to the big-integer model."""
import numpy as np
import pytest

import model as M

U = np.uint64


def fr(*l):
    return np.array(l, dtype=U)


MODULUS = fr(0xffffffff00000001, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48)
R = fr(0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f)
R2 = fr(0xc999e990f3f29c6d, 0x2b6cedcb87925c23, 0x05d314967254398f, 0x0748d9d99f59ff11)
R3 = fr(0xc62c1807439b73af, 0x1b3e0d188cf06990, 0x73d13c71c7b5f418, 0x6e2a5bb9c8db33e9)
LARGEST = fr(0xffffffff00000000, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48)
ZERO = fr(0, 0, 0, 0)
TWO_INV = fr(0x00000000ffffffff, 0xac425bfd0001a401, 0xccc627f7f65e27fa, 0x0c1258acd66282b7)
ROOT_OF_UNITY = fr(0xb9b58d8c5f0e466a, 0x5b1b4c801819d7ec, 0x0af53ae352a31e64, 0x5bf3adda19e9b27b)
ROOT_OF_UNITY_INV = fr(0x4256481adcf3219a, 0x45f37b7f96b6cad3, 0xf9c3f1d75f7a3b27, 0x2d2fc049658afd43)
DELTA = fr(0x70e310d3d146f96a, 0x4b64c08919e299e6, 0x51e114186a8b970d, 0x6185d06627c067cb)
GENERATOR = fr(0x0000000efffffff1, 0x17e363d300189c0f, 0xff9c57876f8457b0, 0x351332208fc5a8c4)

NEG_ONE_BYTES = bytes([0, 0, 0, 0, 255, 255, 255, 255, 254, 91, 254, 255, 2, 164, 189, 83, 5, 216, 161, 9, 8, 216,
                       57, 51, 72, 125, 157, 41, 83, 167, 237, 115])
R2_BYTES = bytes([254, 255, 255, 255, 1, 0, 0, 0, 2, 72, 3, 0, 250, 183, 132, 88, 245, 79, 188, 236, 239, 79, 140,
                  153, 111, 5, 197, 172, 89, 177, 36, 24])


def eq(a, b):
    return np.array_equal(np.asarray(a, U).reshape(-1), np.asarray(b, U).reshape(-1))


# ---- scalar.rs:748-788 -----------------------------------------------------------------------------------
def test_fr_constants(orc):
    assert M.limbs_to_int(MODULUS) == M.R_MOD
    assert eq(orc.fr_op("mul", orc.fr_op("from_raw", fr(2, 0, 0, 0)), TWO_INV), R)
    assert eq(orc.fr_op("mul", ROOT_OF_UNITY, ROOT_OF_UNITY_INV), R)
    assert eq(orc.fr_pow(ROOT_OF_UNITY, fr(1 << 32, 0, 0, 0)), R)
    assert eq(orc.fr_pow(DELTA, fr(0xfffe5bfeffffffff, 0x09a1d80553bda402, 0x299d7d483339d808, 0x0000000073eda753)), R)
    assert M.fr_from_mont(GENERATOR) == 7
    assert M.fr_from_mont(ROOT_OF_UNITY) == M.ROOT_OF_UNITY
    assert M.fr_from_mont(R) == 1 and M.limbs_to_int(R2) == pow(2, 512, M.R_MOD) and M.limbs_to_int(R3) == pow(2, 768, M.R_MOD)


def test_fr_inv_constant():
    inv = 1
    for _ in range(63):
        inv = inv * inv % (1 << 64)
        inv = inv * int(MODULUS[0]) % (1 << 64)
    assert (-inv) % (1 << 64) == 0xfffffffeffffffff


# ---- scalar.rs:816-925 -----------------------------------------------------------------------------------
def test_fr_to_bytes(orc):
    assert orc.fr_to_bytes(ZERO) == bytes(32)
    assert orc.fr_to_bytes(R) == bytes([1] + [0] * 31)
    assert orc.fr_to_bytes(R2) == R2_BYTES
    assert orc.fr_to_bytes(orc.fr_op("neg", R)) == NEG_ONE_BYTES


def test_fr_from_bytes(orc):
    v, ok = orc.fr_from_bytes(bytes(32))
    assert ok and eq(v, ZERO)
    v, ok = orc.fr_from_bytes(bytes([1] + [0] * 31))
    assert ok and eq(v, R)
    v, ok = orc.fr_from_bytes(R2_BYTES)
    assert ok and eq(v, R2)
    assert orc.fr_from_bytes(NEG_ONE_BYTES)[1]
    for pos, val in ((0, 1), (0, 2)):
        b = bytearray(NEG_ONE_BYTES)
        b[pos] = val
        assert not orc.fr_from_bytes(b)[1]
    b = bytearray(NEG_ONE_BYTES)
    b[0], b[22] = 1, 58
    assert not orc.fr_from_bytes(b)[1]
    b = bytearray(NEG_ONE_BYTES)
    b[0], b[31] = 1, 116
    assert not orc.fr_from_bytes(b)[1]


# ---- scalar.rs:927-1003 ----------------------------------------------------------------------------------
def test_fr_wide_reduction(orc):
    assert eq(orc.fr_from_u512(list(MODULUS) + [0, 0, 0, 0]), ZERO)
    assert eq(orc.fr_from_u512([1, 0, 0, 0, 0, 0, 0, 0]), R)
    assert eq(orc.fr_from_u512([0, 0, 0, 0, 1, 0, 0, 0]), R2)
    mx = 0xffffffffffffffff
    assert eq(orc.fr_from_u512([mx] * 8), orc.fr_op("sub", R3, R))
    assert eq(orc.fr_from_bytes_wide(R2_BYTES + bytes(32)), R2)
    assert eq(orc.fr_from_bytes_wide(NEG_ONE_BYTES + bytes(32)), orc.fr_op("neg", R))
    assert eq(orc.fr_from_bytes_wide(bytes([0xff] * 64)),
              fr(0xc62c1805439b73b1, 0xc2b9551e8ced218e, 0xda44ec81daf9a422, 0x5605aa601c162e79))


# ---- scalar.rs:1005-1068 ---------------------------------------------------------------------------------
def test_fr_add_sub_neg(orc):
    assert eq(orc.fr_op("neg", ZERO), ZERO)
    assert eq(orc.fr_op("add", ZERO, ZERO), ZERO) and eq(orc.fr_op("sub", ZERO, ZERO), ZERO)
    assert eq(orc.fr_op("mul", ZERO, ZERO), ZERO)
    assert eq(orc.fr_op("add", LARGEST, LARGEST),
              fr(0xfffffffeffffffff, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48))
    assert eq(orc.fr_op("add", LARGEST, fr(1, 0, 0, 0)), ZERO)
    assert eq(orc.fr_op("neg", LARGEST), fr(1, 0, 0, 0))
    assert eq(orc.fr_op("neg", fr(1, 0, 0, 0)), LARGEST)
    assert eq(orc.fr_op("sub", LARGEST, LARGEST), ZERO)
    assert eq(orc.fr_op("sub", ZERO, LARGEST), fr(1, 0, 0, 0))


# ---- scalar.rs:1070-1120: mul / square against double-and-add over the canonical bits ----------------------
def test_fr_mul_square_recipe(orc):
    cur = LARGEST.copy()
    for _ in range(100):
        prod = orc.fr_op("mul", cur, cur)[0]
        sq = orc.fr_op("square", cur)[0]
        acc = ZERO.copy()
        bits = int.from_bytes(orc.fr_to_bytes(cur), "little")
        for i in range(255, -1, -1):
            acc = orc.fr_op("add", acc, acc)[0]
            if (bits >> i) & 1:
                acc = orc.fr_op("add", acc, cur)[0]
        assert eq(prod, acc) and eq(sq, acc)
        cur = orc.fr_op("add", cur, LARGEST)[0]


# ---- scalar.rs:1122-1165 ---------------------------------------------------------------------------------
def test_fr_inversion(orc):
    assert eq(orc.fr_op("invert", R), R)
    neg1 = orc.fr_op("neg", R)
    assert eq(orc.fr_op("invert", neg1), neg1)
    tmp = R2.copy()
    q_minus_2 = fr(0xfffffffeffffffff, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48)
    for _ in range(100):
        inv = orc.fr_op("invert", tmp)[0]
        assert eq(orc.fr_op("mul", inv, tmp), R)
        assert eq(inv, orc.fr_pow(tmp, q_minus_2))
        assert M.fr_from_mont(inv) == pow(M.fr_from_mont(tmp), -1, M.R_MOD)
        tmp = orc.fr_op("add", tmp, R2)[0]


# ---- scalar.rs:1195-1210 ---------------------------------------------------------------------------------
def test_fr_from_raw(orc):
    a = orc.fr_op("from_raw", fr(0x0001fffffffd, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f))
    b = orc.fr_op("from_raw", fr(*[0xffffffffffffffff] * 4))
    assert eq(a, b)
    assert eq(orc.fr_op("from_raw", MODULUS), ZERO)
    assert eq(orc.fr_op("from_raw", fr(1, 0, 0, 0)), R)


def test_fr_divn_and_reduce(orc):
    v = M.random_fr_ints(8, 11)
    for x in v:
        mont = M.fr_to_mont(x)
        red = orc.fr_op("reduce", mont)[0]
        assert M.limbs_to_int(red) == x
        for n in (0, 1, 16, 63, 64, 65, 128, 200, 255, 256, 300):
            assert M.limbs_to_int(orc.fr_divn(red, n)) == (x >> n if n < 256 else 0)


def test_fr_against_model(orc):
    a = M.random_fr_ints(200, 1) + [0, 1, M.R_MOD - 1, 2]
    b = M.random_fr_ints(200, 2) + [M.R_MOD - 1, 0, M.R_MOD - 1, M.R_MOD - 2]
    am, bm = M.fr_vec_to_mont(a), M.fr_vec_to_mont(b)
    assert M.fr_vec_from_mont(orc.fr_op("add", am, bm)) == [(x + y) % M.R_MOD for x, y in zip(a, b)]
    assert M.fr_vec_from_mont(orc.fr_op("sub", am, bm)) == [(x - y) % M.R_MOD for x, y in zip(a, b)]
    assert M.fr_vec_from_mont(orc.fr_op("mul", am, bm)) == [(x * y) % M.R_MOD for x, y in zip(a, b)]
    assert M.fr_vec_from_mont(orc.fr_op("neg", am)) == [(-x) % M.R_MOD for x in a]
    # every output is canonical
    for op in ("add", "sub", "mul"):
        for row in orc.fr_op(op, am, bm):
            assert M.limbs_to_int(row) < M.R_MOD


def test_batch_inversion(orc):
    vals = [1, 2, 3, 4, 5, 0, 7, 0]
    out = orc.batch_inversion(M.fr_vec_to_mont(vals))
    assert M.fr_vec_from_mont(out) == [pow(v, -1, M.R_MOD) if v else 0 for v in vals]


# ===== Fq: fp.rs:646-926 ========================================================================================
def fq(*l):
    return np.array(l, dtype=U)


def test_fq_constants():
    assert M.fq_from_mont(fq(0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745,
                             0x5c071a97a256ec6d, 0x15f65ec3fa80e493)) == 1
    inv = pow(-M.P_MOD, -1, 1 << 64)
    assert inv == 0x89f3fffcfffcfffd


def test_fq_squaring(orc):
    a = fq(0xd215d2768e83191b, 0x5085d80f8fb28261, 0xce9a032ddf393a56, 0x3e9c4fff2ca0c4bb, 0x6436b6f7f4d95dfb,
           0x10606628ad4a4d90)
    b = fq(0x33d9c42a3cb3e235, 0xdad11a094c4cd455, 0xa2f144bd729aaeba, 0xd4150932be9ffeac, 0xe27bc7c47d44ee50,
           0x14b6a78d3ec7a560)
    assert eq(orc.fq_op("square", a), b)


def test_fq_multiplication(orc):
    a = fq(0x0397a38320170cd4, 0x734c1b2c9e761d30, 0x5ed255ad9a48beb5, 0x095a3c6b22a7fcfc, 0x2294ce75d4e26a27,
           0x13338bd870011ebb)
    b = fq(0xb9c3c7c5b1196af7, 0x2580e2086ce335c1, 0xf49aed3d8a57ef42, 0x41f281e49846e878, 0xe0762346c38452ce,
           0x0652e89326e57dc0)
    c = fq(0xf96ef3d711ab5355, 0xe8d459ea00f148dd, 0x53f7354a5f00fa78, 0x9e34a4f3125c5f83, 0x3fbe0c47ca74c19e,
           0x01b06a8bbd4adfe4)
    assert eq(orc.fq_op("mul", a, b), c)


A_FQ = fq(0x5360bb5978678032, 0x7dd275ae799e128e, 0x5c5b5071ce4f4dcf, 0xcdb21f93078dbb3e, 0xc32365c5e73f474a,
          0x115a2a5489babe5b)
B_FQ = fq(0x9fd287733d23dda0, 0xb16bf2af738b3554, 0x3e57a75bd3cc6d1d, 0x900bc0bd627fd6d6, 0xd319a080efb245fe,
          0x15fdcaa4e4bb2091)


def test_fq_add_sub_neg(orc):
    assert eq(orc.fq_op("add", A_FQ, B_FQ), fq(0x393442ccb58bb327, 0x1092685f3bd547e3, 0x3382252cab6ac4c9,
                                                0xf94694cb76887f55, 0x4b215e9093a5e071, 0x0d56e30f34f5f853))
    assert eq(orc.fq_op("sub", A_FQ, B_FQ), fq(0x6d8d33e63b434d3d, 0xeb1282fdb766dd39, 0x85347bb6f133d6d5,
                                                0xa21daa5a9892f727, 0x3b256cfb3ad8ae23, 0x155d7199de7f8464))
    assert eq(orc.fq_op("neg", A_FQ), fq(0x669e44a687982a79, 0xa0d98a5037b5ed71, 0x0ad5822f2861a854,
                                          0x96c52bf1ebf75781, 0x87f841f05c0c658c, 0x08a6e795afc5283e))


def test_fq_to_bytes_hex(orc):
    # fp.rs:780-796 (Debug prints the big-endian canonical bytes)
    assert orc.fq_to_bytes(A_FQ).hex() == ("104bf052ad3bc99bcb176c24a06a6c3aad4eaf2308fc4d282e106c84a757d061"
                                           "052630515305e59bdddf8111bfdeb704")


def test_fq_from_bytes(orc):
    a = fq(0xdc906d9be3f95dc8, 0x8755caf7459691a1, 0xcff1a7f4e9583ab3, 0x9b43821f849e2284, 0xf57554f3a2974f3f,
           0x085dbea84ed47f79)
    for _ in range(100):
        a = orc.fq_op("square", a)[0]
        b, ok = orc.fq_from_bytes(orc.fq_to_bytes(a))
        assert ok and eq(a, b)
    p_minus_1 = bytes([26, 1, 17, 234, 57, 127, 230, 154, 75, 27, 167, 182, 67, 75, 172, 215, 100, 119, 75, 132, 243,
                       133, 18, 191, 103, 48, 210, 160, 246, 176, 246, 36, 30, 171, 255, 254, 177, 83, 255, 255, 185,
                       254, 255, 255, 255, 255, 170, 170])
    v, ok = orc.fq_from_bytes(p_minus_1)
    assert ok and M.fq_from_mont(v) == M.P_MOD - 1
    assert not orc.fq_from_bytes(bytes([27]) + p_minus_1[1:])[1]
    assert not orc.fq_from_bytes(bytes([0xff] * 48))[1]


def test_fq_sqrt(orc):
    four = fq(0xaa270000000cfff3, 0x53cc0032fc34000a, 0x478fe97a6b0a807f, 0xb1d37ebee6ba24d7, 0x8ec9733bbf78ab2f,
              0x09d645513d83de7e)
    two = fq(0x321300000006554f, 0xb93c0018d6c40005, 0x57605e0db0ddbb51, 0x8b256521ed1f9bcb, 0x6cf28d7901622c03,
             0x11ebab9dbb81e28c)
    assert eq(orc.fq_op("neg", orc.fq_op("sqrt", four)), two)
    assert M.fq_from_mont(four) == 4 and M.fq_from_mont(two) == 2


def test_fq_inversion(orc):
    a = fq(0x43b43a5078ac2076, 0x1ce0763046f8962b, 0x724a5276486d735c, 0x6f05c2a6282d48fd, 0x2095bd5bb4ca9331,
           0x03b35b3894b0f7da)
    b = fq(0x69ecd7040952148f, 0x985ccc2022190f55, 0xe19bba36a9ad2f41, 0x19bb16c95219dbd8, 0x14dcacfdfb478693,
           0x115ff58afff9a8e1)
    assert eq(orc.fq_op("invert", a), b)
    assert eq(orc.fq_op("invert", fq(0, 0, 0, 0, 0, 0)), fq(0, 0, 0, 0, 0, 0))


def test_fq_lexicographic_largest(orc):
    assert not orc.fq_lex_largest(fq(0, 0, 0, 0, 0, 0))
    assert not orc.fq_lex_largest(M.fq_to_mont(1))
    assert not orc.fq_lex_largest(fq(0xa1fafffffffe5557, 0x995bfff976a3fffe, 0x03f41d24d174ceb4, 0xf6547998c1995dbd,
                                     0x778a468f507a6034, 0x020559931f7f8103))
    assert orc.fq_lex_largest(fq(0x1804000000015554, 0x855000053ab00001, 0x633cb57c253c276f, 0x6e22d1ec31ebb502,
                                 0xd3916126f2d14ca2, 0x17fbb8571a006596))
    assert orc.fq_lex_largest(fq(0x43f5fffffffcaaae, 0x32b7fff2ed47fffd, 0x07e83a49a2e99d69, 0xeca8f3318332bb7a,
                                 0xef148d1ea0f4c069, 0x040ab3263eff0206))


def test_fq_against_model(orc):
    rng = np.random.Generator(np.random.PCG64(5))
    vals_a = [int.from_bytes(rng.bytes(64), "little") % M.P_MOD for _ in range(100)] + [0, 1, M.P_MOD - 1]
    vals_b = [int.from_bytes(rng.bytes(64), "little") % M.P_MOD for _ in range(100)] + [M.P_MOD - 1, M.P_MOD - 1, M.P_MOD - 1]
    am = np.stack([M.fq_to_mont(v) for v in vals_a])
    bm = np.stack([M.fq_to_mont(v) for v in vals_b])
    for op, f in (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y), ("mul", lambda x, y: x * y)):
        got = [M.fq_from_mont(r) for r in orc.fq_op(op, am, bm)]
        assert got == [f(x, y) % M.P_MOD for x, y in zip(vals_a, vals_b)]
        for row in orc.fq_op(op, am, bm):
            assert M.limbs_to_int(row) < M.P_MOD
