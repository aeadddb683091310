"""Independent big-integer model of the BLS12-381 fields, G1 and the DFT, written with plain Python
integers (no limbs, no Montgomery tricks).  It cross-checks the C++ oracle at value level and turns
limb arrays into integers and back for the tests."""
import numpy as np

R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001   # Fr modulus
P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab  # Fq modulus
FR_MONT = 1 << 256
FQ_MONT = 1 << 384
FR_MONT_INV = pow(FR_MONT, -1, R_MOD)
FQ_MONT_INV = pow(FQ_MONT, -1, P_MOD)

GEN_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
GEN_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1


def limbs_to_int(limbs):
    v = 0
    for i, l in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(-1).tolist()):
        v |= int(l) << (64 * i)
    return v


def int_to_limbs(v, n):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def fr_from_mont(limbs):
    return limbs_to_int(limbs) * FR_MONT_INV % R_MOD


def fr_to_mont(v):
    return int_to_limbs(v % R_MOD * FR_MONT % R_MOD, 4)


def fq_from_mont(limbs):
    return limbs_to_int(limbs) * FQ_MONT_INV % P_MOD


def fq_to_mont(v):
    return int_to_limbs(v % P_MOD * FQ_MONT % P_MOD, 6)


def fr_vec_to_mont(vals):
    return np.stack([fr_to_mont(v) for v in vals]) if len(vals) else np.empty((0, 4), np.uint64)


def fr_vec_from_mont(arr):
    return [fr_from_mont(r) for r in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


# ---- G1 in affine integer coordinates; None is the point at infinity -----------------------------------
def ec_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % P_MOD == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P_MOD) % P_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    return x3, (lam * (x1 - x3) - y1) % P_MOD


def ec_mul(p, k):
    k %= R_MOD
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, p)
        p = ec_add(p, p)
        k >>= 1
    return acc


def ec_msm(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        acc = ec_add(acc, ec_mul(p, s))
    return acc


def affine_from_limbs(xy, inf=0):
    if inf:
        return None
    xy = np.asarray(xy, dtype=np.uint64).reshape(12)
    return fq_from_mont(xy[:6]), fq_from_mont(xy[6:])


def affine_to_limbs(p):
    """-> (xy limbs[12], infinity flag) with the reference's identity encoding (0, 1, inf=1)."""
    if p is None:
        return np.concatenate([fq_to_mont(0), fq_to_mont(1)]), 1
    return np.concatenate([fq_to_mont(p[0]), fq_to_mont(p[1])]), 0


def projective_to_affine_int(p18):
    p18 = np.asarray(p18, dtype=np.uint64).reshape(18)
    x, y, z = fq_from_mont(p18[:6]), fq_from_mont(p18[6:12]), fq_from_mont(p18[12:])
    if z == 0:
        return None
    zi = pow(z, -1, P_MOD)
    return x * zi % P_MOD, y * zi % P_MOD


def compress(p):
    """48-byte compressed encoding: big-endian x, flag bits 7 (compressed), 6 (infinity), 5 (y > (p-1)/2)."""
    if p is None:
        b = bytearray(48)
        b[0] = 0xC0
        return bytes(b)
    b = bytearray(p[0].to_bytes(48, "big"))
    b[0] |= 0x80
    if p[1] > (P_MOD - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


# ---- DFT ----------------------------------------------------------------------------------------------
ROOT_OF_UNITY = pow(7, (R_MOD - 1) >> 32, R_MOD)


def domain_root(n):
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    return pow(ROOT_OF_UNITY, 1 << (32 - log_n), R_MOD)


def naive_dft(vals, n, inverse=False, coset=False):
    """O(n^2) evaluation: out[i] = sum_j a[j] * w^(i j); coset variants scale by 7^j before / 7^-j after."""
    w = domain_root(n)
    a = list(vals)
    if coset and not inverse:
        a = [v * pow(7, j, R_MOD) % R_MOD for j, v in enumerate(a)]
    a = (a + [0] * n)[:n]
    if inverse:
        w = pow(w, -1, R_MOD)
    out = []
    for i in range(n):
        wi = pow(w, i, R_MOD)
        acc, x = 0, 1
        for j in range(n):
            acc = (acc + a[j] * x) % R_MOD
            x = x * wi % R_MOD
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, R_MOD)
        out = [v * ninv % R_MOD for v in out]
        if coset:
            ginv = pow(7, -1, R_MOD)
            out = [v * pow(ginv, j, R_MOD) % R_MOD for j, v in enumerate(out)]
    return out


def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


# ---- seeded inputs ----------------------------------------------------------------------------------------
def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def random_fr_ints(n, seed):
    """n uniform-ish field elements: 512 pseudo-random bits reduced mod r (the from_bytes_wide recipe)."""
    out, st = [], seed & 0xFFFFFFFFFFFFFFFF
    for _ in range(n):
        v = 0
        for k in range(8):
            st, w = splitmix64(st)
            v |= w << (64 * k)
        out.append(v % R_MOD)
    return out


def random_fr_mont(n, seed):
    """Fast seeded canonical Montgomery-form limbs for large n (numpy): each element is a uniformly random
    255-bit integer with the top limb reduced below the modulus' top limb, so it is < r."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] %= np.uint64(0x73eda753299d7d48)
    return a
