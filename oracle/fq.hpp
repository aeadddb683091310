// TEST INFRASTRUCTURE ONLY (see util.hpp).
// BLS12-381 base field Fq, 6x64-bit little-endian limbs, Montgomery form R = 2^384.
// Restates coset-bls12_381/src/fp.rs.
#pragma once
#include "util.hpp"

namespace oracle {

struct Fq {
    uint64_t l[6];
    bool operator==(const Fq &o) const {
        uint64_t d = 0;
        for (int i = 0; i < 6; i++) d |= l[i] ^ o.l[i];
        return d == 0;
    }
    bool operator!=(const Fq &o) const { return !(*this == o); }
};

// fp.rs:77-84
static const uint64_t FQ_MODULUS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                       0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
// fp.rs:86
static const uint64_t FQ_INV = 0x89f3fffcfffcfffdULL;
// fp.rs:88-113
static const Fq FQ_R = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL,
                         0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
static const Fq FQ_R2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL,
                          0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};
static const Fq FQ_R3 = {{0xed48ac6bd94ca1e0ULL, 0x315f831e03a7adf8ULL, 0x9a53352a615e29ddULL, 0x34c04e5e921e1761ULL,
                          0x2512d43565724728ULL, 0x0aa6346091755d4dULL}};

static inline Fq fq_zero() { return Fq{{0, 0, 0, 0, 0, 0}}; }
static inline Fq fq_one() { return FQ_R; }
static inline bool fq_is_zero(const Fq &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }

// fp.rs:332-348 — subtract p when the value is >= p
static inline Fq fq_subtract_p(const Fq &a) {
    uint64_t borrow = 0;
    Fq r;
    for (int i = 0; i < 6; i++) r.l[i] = sbb(a.l[i], FQ_MODULUS[i], borrow, borrow);
    for (int i = 0; i < 6; i++) r.l[i] = (a.l[i] & borrow) | (r.l[i] & ~borrow);
    return r;
}

// fp.rs:351-360
static inline Fq fq_add(const Fq &a, const Fq &b) {
    uint64_t carry = 0;
    Fq d;
    for (int i = 0; i < 6; i++) d.l[i] = adc(a.l[i], b.l[i], carry, carry);
    return fq_subtract_p(d);
}

// fp.rs:363-388
static inline Fq fq_neg(const Fq &a) {
    uint64_t borrow = 0;
    Fq d;
    for (int i = 0; i < 6; i++) d.l[i] = sbb(FQ_MODULUS[i], a.l[i], borrow, borrow);
    uint64_t mask = (uint64_t)(fq_is_zero(a) ? 1 : 0) - 1;
    for (int i = 0; i < 6; i++) d.l[i] &= mask;
    return d;
}

// fp.rs:391-393
static inline Fq fq_sub(const Fq &a, const Fq &b) { return fq_add(fq_neg(b), a); }

// fp.rs:437-506
static inline Fq fq_montgomery_reduce(const uint64_t t[12]) {
    uint64_t r[13];
    for (int i = 0; i < 12; i++) r[i] = t[i];
    r[12] = 0;
    uint64_t extra = 0;  // the r7/r8/... "spill" limb carried from one round into the next
    for (int i = 0; i < 6; i++) {
        uint64_t k = r[i] * FQ_INV;
        uint64_t carry = 0;
        (void)mac(r[i], k, FQ_MODULUS[0], 0, carry);
        for (int j = 1; j < 6; j++) r[i + j] = mac(r[i + j], k, FQ_MODULUS[j], carry, carry);
        r[i + 6] = adc(r[i + 6], extra, carry, extra);
    }
    Fq out = {{r[6], r[7], r[8], r[9], r[10], r[11]}};
    return fq_subtract_p(out);
}

// fp.rs:509-555
static inline Fq fq_mul(const Fq &a, const Fq &b) {
    uint64_t t[12] = {0};
    for (int i = 0; i < 6; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 6; j++) t[i + j] = mac(t[i + j], a.l[i], b.l[j], carry, carry);
        t[i + 6] = carry;
    }
    return fq_montgomery_reduce(t);
}

// fp.rs:558-607 (same value as mul(a, a))
static inline Fq fq_square(const Fq &a) { return fq_mul(a, a); }

// fp.rs:286-298
static inline Fq fq_pow_vartime(const Fq &a, const uint64_t by[6]) {
    Fq res = fq_one();
    for (int e = 5; e >= 0; e--)
        for (int i = 63; i >= 0; i--) {
            res = fq_square(res);
            if ((by[e] >> i) & 1) res = fq_mul(res, a);
        }
    return res;
}

// fp.rs:318-329 — a^(p-2); false for zero
static inline bool fq_invert(const Fq &a, Fq &out) {
    static const uint64_t p_minus_2[6] = {0xb9feffffffffaaa9ULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                          0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    out = fq_pow_vartime(a, p_minus_2);
    return !fq_is_zero(a);
}

// fp.rs:302-313
static inline bool fq_sqrt(const Fq &a, Fq &out) {
    static const uint64_t e[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                                  0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};
    out = fq_pow_vartime(a, e);
    return fq_square(out) == a;
}

// fp.rs:213-228 — 48-byte big-endian canonical integer
static inline void fq_to_bytes(const Fq &a, uint8_t out[48]) {
    uint64_t t[12] = {a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], 0, 0, 0, 0, 0, 0};
    Fq r = fq_montgomery_reduce(t);
    for (int i = 0; i < 6; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(r.l[5 - i] >> (8 * (7 - b)));
}

// fp.rs:182-210
static inline bool fq_from_bytes(const uint8_t in[48], Fq &out) {
    Fq t;
    for (int i = 0; i < 6; i++) {
        t.l[5 - i] = 0;
        for (int b = 0; b < 8; b++) t.l[5 - i] |= (uint64_t)in[8 * i + b] << (8 * (7 - b));
    }
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) (void)sbb(t.l[i], FQ_MODULUS[i], borrow, borrow);
    out = fq_mul(t, FQ_R2);
    return (borrow & 1) == 1;
}

// fp.rs:264-278 — true when the canonical integer is > (p-1)/2
static inline bool fq_lexicographically_largest(const Fq &a) {
    static const uint64_t half[6] = {0xdcff7fffffffd556ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL,
                                     0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL};
    uint64_t t[12] = {a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], 0, 0, 0, 0, 0, 0};
    Fq r = fq_montgomery_reduce(t);
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) (void)sbb(r.l[i], half[i], borrow, borrow);
    return (borrow & 1) == 0;
}

}  // namespace oracle
