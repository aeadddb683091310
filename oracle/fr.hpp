// TEST INFRASTRUCTURE ONLY (see util.hpp).
// BLS12-381 scalar field Fr, 4x64-bit little-endian limbs, Montgomery form R = 2^256.
// Restates coset-bls12_381/src/scalar.rs and scalar/coset.rs (reduce, divn).
#pragma once
#include "util.hpp"
#include <cstring>

namespace oracle {

struct Fr {
    uint64_t l[4];
    bool operator==(const Fr &o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
    bool operator!=(const Fr &o) const { return !(*this == o); }
};

// scalar.rs:80-85
static const Fr FR_MODULUS = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
// scalar.rs:156
static const uint64_t FR_INV = 0xfffffffeffffffffULL;
// scalar.rs:158-177
static const Fr FR_R = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static const Fr FR_R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
static const Fr FR_R3 = {{0xc62c1807439b73afULL, 0x1b3e0d188cf06990ULL, 0x73d13c71c7b5f418ULL, 0x6e2a5bb9c8db33e9ULL}};
// scalar.rs:101-106 (multiplicative generator 7) and :191-196 (2^32-th root of unity)
static const Fr FR_GENERATOR = {{0x0000000efffffff1ULL, 0x17e363d300189c0fULL, 0xff9c57876f8457b0ULL, 0x351332208fc5a8c4ULL}};
static const Fr FR_ROOT_OF_UNITY = {{0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL, 0x5bf3adda19e9b27bULL}};
static const uint32_t FR_TWO_ADACITY = 32;  // scalar.rs:187

static inline Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
static inline Fr fr_one() { return FR_R; }  // scalar.rs:231-233
static inline bool fr_is_zero(const Fr &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

// scalar.rs:554-567
static inline Fr fr_sub(const Fr &a, const Fr &b) {
    uint64_t borrow = 0, carry = 0;
    Fr d;
    for (int i = 0; i < 4; i++) d.l[i] = sbb(a.l[i], b.l[i], borrow, borrow);
    for (int i = 0; i < 4; i++) d.l[i] = adc(d.l[i], FR_MODULUS.l[i] & borrow, carry, carry);
    return d;
}

// scalar.rs:570-578
static inline Fr fr_add(const Fr &a, const Fr &b) {
    uint64_t carry = 0;
    Fr d;
    for (int i = 0; i < 4; i++) d.l[i] = adc(a.l[i], b.l[i], carry, carry);
    return fr_sub(d, FR_MODULUS);
}

// scalar.rs:581-592
static inline Fr fr_neg(const Fr &a) {
    uint64_t borrow = 0;
    Fr d;
    for (int i = 0; i < 4; i++) d.l[i] = sbb(FR_MODULUS.l[i], a.l[i], borrow, borrow);
    uint64_t mask = (uint64_t)(fr_is_zero(a) ? 1 : 0) - 1;
    for (int i = 0; i < 4; i++) d.l[i] &= mask;
    return d;
}

// scalar.rs:484-525 — 4 rounds, each clears the lowest live limb; then one conditional subtract.
static inline Fr fr_montgomery_reduce(const uint64_t t[8]) {
    uint64_t r[9];
    for (int i = 0; i < 8; i++) r[i] = t[i];
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t k = r[i] * FR_INV;
        uint64_t carry = 0;
        (void)mac(r[i], k, FR_MODULUS.l[0], 0, carry);
        for (int j = 1; j < 4; j++) r[i + j] = mac(r[i + j], k, FR_MODULUS.l[j], carry, carry);
        r[i + 4] = adc(r[i + 4], carry2, carry, carry2);
    }
    Fr out = {{r[4], r[5], r[6], r[7]}};
    return fr_sub(out, FR_MODULUS);
}

// scalar.rs:528-551
static inline Fr fr_mul(const Fr &a, const Fr &b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) t[i + j] = mac(t[i + j], a.l[i], b.l[j], carry, carry);
        t[i + 4] = carry;
    }
    return fr_montgomery_reduce(t);
}

// scalar.rs:311-339 (same value as mul(a, a))
static inline Fr fr_square(const Fr &a) { return fr_mul(a, a); }

// scalar/coset.rs:242-246 — out of Montgomery form
static inline Fr fr_reduce(const Fr &a) {
    uint64_t t[8] = {a.l[0], a.l[1], a.l[2], a.l[3], 0, 0, 0, 0};
    return fr_montgomery_reduce(t);
}

// scalar.rs:306-308
static inline Fr fr_from_raw(const uint64_t v[4]) {
    Fr t = {{v[0], v[1], v[2], v[3]}};
    return fr_mul(t, FR_R2);
}

// scalar.rs:47-51
static inline Fr fr_from_u64(uint64_t v) {
    Fr t = {{v, 0, 0, 0}};
    return fr_mul(t, FR_R2);
}

// scalar.rs:267-279 — 32-byte little-endian canonical integer
static inline void fr_to_bytes(const Fr &a, uint8_t out[32]) {
    Fr t = fr_reduce(a);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(t.l[i] >> (8 * b));
}

// scalar.rs:242-264 — returns false when the integer is >= r
static inline bool fr_from_bytes(const uint8_t in[32], Fr &out) {
    Fr t;
    for (int i = 0; i < 4; i++) {
        t.l[i] = 0;
        for (int b = 0; b < 8; b++) t.l[i] |= (uint64_t)in[8 * i + b] << (8 * b);
    }
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) (void)sbb(t.l[i], FR_MODULUS.l[i], borrow, borrow);
    out = fr_mul(t, FR_R2);
    return (borrow & 1) == 1;
}

// scalar.rs:295-303 — d0*R2 + d1*R3
static inline Fr fr_reduce_u512_words(const uint64_t w[8]) {
    Fr d0 = {{w[0], w[1], w[2], w[3]}};
    Fr d1 = {{w[4], w[5], w[6], w[7]}};
    return fr_add(fr_mul(d0, FR_R2), fr_mul(d1, FR_R3));
}

// scalar.rs:282-293
static inline Fr fr_from_bytes_wide(const uint8_t in[64]) {
    uint64_t w[8];
    for (int i = 0; i < 8; i++) {
        w[i] = 0;
        for (int b = 0; b < 8; b++) w[i] |= (uint64_t)in[8 * i + b] << (8 * b);
    }
    return fr_reduce_u512_words(w);
}

// scalar.rs:342-353 / :358-370 — square-and-multiply from the top bit of by[3]
static inline Fr fr_pow(const Fr &a, const uint64_t by[4]) {
    Fr res = fr_one();
    for (int e = 3; e >= 0; e--)
        for (int i = 63; i >= 0; i--) {
            res = fr_square(res);
            if ((by[e] >> i) & 1) res = fr_mul(res, a);
        }
    return res;
}
static inline Fr fr_pow_u64(const Fr &a, uint64_t e) {
    uint64_t by[4] = {e, 0, 0, 0};
    return fr_pow(a, by);
}

// scalar.rs:373-481 — the reference walks a fixed addition chain for a^(r-2); the value is
// the same as pow(r-2) (the reference's own test_invert_is_pow, scalar.rs:1141-1165).
// Returns false for zero (reference: None).
static inline bool fr_invert(const Fr &a, Fr &out) {
    if (fr_is_zero(a)) {
        out = fr_zero();
        return false;
    }
    static const uint64_t r_minus_2[4] = {0xfffffffeffffffffULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                          0x73eda753299d7d48ULL};
    out = fr_pow(a, r_minus_2);
    return true;
}
static inline Fr fr_inv_unwrap(const Fr &a) {
    Fr o;
    (void)fr_invert(a, o);
    return o;
}

// scalar/coset.rs:282-305 — logical right shift of the 256-bit limb vector
static inline void fr_divn(Fr &a, uint32_t n) {
    if (n >= 256) {
        a = fr_zero();  // reference: Self::from(0)
        return;
    }
    while (n >= 64) {
        uint64_t t = 0;
        for (int i = 3; i >= 0; i--) {
            uint64_t s = a.l[i];
            a.l[i] = t;
            t = s;
        }
        n -= 64;
    }
    if (n > 0) {
        uint64_t t = 0;
        for (int i = 3; i >= 0; i--) {
            uint64_t t2 = a.l[i] << (64 - n);
            a.l[i] >>= n;
            a.l[i] |= t;
            t = t2;
        }
    }
}

}  // namespace oracle
