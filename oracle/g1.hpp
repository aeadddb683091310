// TEST INFRASTRUCTURE ONLY (see util.hpp).
// BLS12-381 G1: y^2 = x^3 + 4, affine and homogeneous-projective points with the complete
// Renes-Costello-Batina formulas. Restates coset-bls12_381/src/g1.rs and g1/coset.rs.
#pragma once
#include "fq.hpp"
#include "fr.hpp"

namespace oracle {

struct G1Affine {
    Fq x, y;
    uint8_t infinity;  // 0 or 1
};
struct G1Projective {
    Fq x, y, z;
};

// g1.rs:198-205 — curve constant b = 4 in Montgomery form
static const Fq G1_B = {{0xaa270000000cfff3ULL, 0x53cc0032fc34000aULL, 0x478fe97a6b0a807fULL, 0xb1d37ebee6ba24d7ULL,
                         0x8ec9733bbf78ab2fULL, 0x09d645513d83de7eULL}};
// g1.rs:218-238
static const Fq G1_GEN_X = {{0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL, 0xf0ae6acdf3d0e747ULL,
                             0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL}};
static const Fq G1_GEN_Y = {{0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL, 0x51ac582950405194ULL,
                             0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL}};

// g1.rs:209-215
static inline G1Affine g1a_identity() { return G1Affine{fq_zero(), fq_one(), 1}; }
static inline G1Affine g1a_generator() { return G1Affine{G1_GEN_X, G1_GEN_Y, 0}; }
// g1.rs:592-598
static inline G1Projective g1p_identity() { return G1Projective{fq_zero(), fq_one(), fq_zero()}; }
static inline G1Projective g1p_generator() { return G1Projective{G1_GEN_X, G1_GEN_Y, fq_one()}; }
static inline bool g1p_is_identity(const G1Projective &p) { return fq_is_zero(p.z); }  // g1.rs:812-814

// g1.rs:86-97
static inline bool g1a_eq(const G1Affine &a, const G1Affine &b) {
    if (a.infinity && b.infinity) return true;
    return !a.infinity && !b.infinity && a.x == b.x && a.y == b.y;
}

// g1.rs:584-588
static inline Fq mul_by_curve_b3(Fq a) {
    a = fq_add(a, a);
    a = fq_add(a, a);
    return fq_add(fq_add(a, a), a);
}

// g1.rs:624-655
static inline G1Projective g1p_double(const G1Projective &p) {
    Fq t0 = fq_square(p.y);
    Fq z3 = fq_add(t0, t0);
    z3 = fq_add(z3, z3);
    z3 = fq_add(z3, z3);
    Fq t1 = fq_mul(p.y, p.z);
    Fq t2 = fq_square(p.z);
    t2 = mul_by_curve_b3(t2);
    Fq x3 = fq_mul(t2, z3);
    Fq y3 = fq_add(t0, t2);
    z3 = fq_mul(t1, z3);
    t1 = fq_add(t2, t2);
    t2 = fq_add(t1, t2);
    t0 = fq_sub(t0, t2);
    y3 = fq_mul(t0, y3);
    y3 = fq_add(x3, y3);
    t1 = fq_mul(p.x, p.y);
    x3 = fq_mul(t0, t1);
    x3 = fq_add(x3, x3);
    if (g1p_is_identity(p)) return g1p_identity();
    return G1Projective{x3, y3, z3};
}

// g1.rs:658-698
static inline G1Projective g1p_add(const G1Projective &a, const G1Projective &b) {
    Fq t0 = fq_mul(a.x, b.x);
    Fq t1 = fq_mul(a.y, b.y);
    Fq t2 = fq_mul(a.z, b.z);
    Fq t3 = fq_add(a.x, a.y);
    Fq t4 = fq_add(b.x, b.y);
    t3 = fq_mul(t3, t4);
    t4 = fq_add(t0, t1);
    t3 = fq_sub(t3, t4);
    t4 = fq_add(a.y, a.z);
    Fq x3 = fq_add(b.y, b.z);
    t4 = fq_mul(t4, x3);
    x3 = fq_add(t1, t2);
    t4 = fq_sub(t4, x3);
    x3 = fq_add(a.x, a.z);
    Fq y3 = fq_add(b.x, b.z);
    x3 = fq_mul(x3, y3);
    y3 = fq_add(t0, t2);
    y3 = fq_sub(x3, y3);
    x3 = fq_add(t0, t0);
    t0 = fq_add(x3, t0);
    t2 = mul_by_curve_b3(t2);
    Fq z3 = fq_add(t1, t2);
    t1 = fq_sub(t1, t2);
    y3 = mul_by_curve_b3(y3);
    x3 = fq_mul(t4, y3);
    t2 = fq_mul(t3, t1);
    x3 = fq_sub(t2, x3);
    y3 = fq_mul(y3, t0);
    t1 = fq_mul(t1, z3);
    y3 = fq_add(t1, y3);
    t0 = fq_mul(t0, t3);
    z3 = fq_mul(z3, t4);
    z3 = fq_add(z3, t0);
    return G1Projective{x3, y3, z3};
}

// g1.rs:701-736
static inline G1Projective g1p_add_mixed(const G1Projective &a, const G1Affine &b) {
    Fq t0 = fq_mul(a.x, b.x);
    Fq t1 = fq_mul(a.y, b.y);
    Fq t3 = fq_add(b.x, b.y);
    Fq t4 = fq_add(a.x, a.y);
    t3 = fq_mul(t3, t4);
    t4 = fq_add(t0, t1);
    t3 = fq_sub(t3, t4);
    t4 = fq_mul(b.y, a.z);
    t4 = fq_add(t4, a.y);
    Fq y3 = fq_mul(b.x, a.z);
    y3 = fq_add(y3, a.x);
    Fq x3 = fq_add(t0, t0);
    t0 = fq_add(x3, t0);
    Fq t2 = mul_by_curve_b3(a.z);
    Fq z3 = fq_add(t1, t2);
    t1 = fq_sub(t1, t2);
    y3 = mul_by_curve_b3(y3);
    x3 = fq_mul(t4, y3);
    t2 = fq_mul(t3, t1);
    x3 = fq_sub(t2, x3);
    y3 = fq_mul(y3, t0);
    t1 = fq_mul(t1, z3);
    y3 = fq_add(t1, y3);
    t0 = fq_mul(t0, t3);
    z3 = fq_mul(z3, t4);
    z3 = fq_add(z3, t0);
    if (b.infinity) return a;
    return G1Projective{x3, y3, z3};
}

// g1.rs:60-78
static inline G1Affine g1_to_affine(const G1Projective &p) {
    Fq zinv;
    if (!fq_invert(p.z, zinv)) zinv = fq_zero();
    if (fq_is_zero(zinv)) return g1a_identity();
    return G1Affine{fq_mul(p.x, zinv), fq_mul(p.y, zinv), 0};
}

static inline G1Projective g1_from_affine(const G1Affine &a) {
    if (a.infinity) return g1p_identity();
    return G1Projective{a.x, a.y, fq_one()};
}

// g1.rs:121-136
static inline G1Affine g1a_neg(const G1Affine &a) {
    return G1Affine{a.x, a.infinity ? fq_one() : fq_neg(a.y), a.infinity};
}
static inline G1Projective g1p_neg(const G1Projective &a) { return G1Projective{a.x, fq_neg(a.y), a.z}; }

// g1.rs:738-756 — double-and-add over the 32 little-endian bytes of the scalar, skipping the top bit
static inline G1Projective g1p_mul(const G1Projective &p, const Fr &s) {
    uint8_t by[32];
    fr_to_bytes(s, by);
    G1Projective acc = g1p_identity();
    bool first = true;
    for (int i = 31; i >= 0; i--)
        for (int b = 7; b >= 0; b--) {
            if (first) {
                first = false;
                continue;
            }
            acc = g1p_double(acc);
            if ((by[i] >> b) & 1) acc = g1p_add(acc, p);
        }
    return acc;
}

// g1.rs:816-820
static inline bool g1p_is_on_curve(const G1Projective &p) {
    if (fq_is_zero(p.z)) return true;
    Fq lhs = fq_mul(fq_square(p.y), p.z);
    Fq rhs = fq_add(fq_mul(fq_square(p.x), p.x), fq_mul(fq_mul(fq_square(p.z), p.z), G1_B));
    return lhs == rhs;
}

// g1/coset.rs:54-71 — 48-byte compressed encoding
static inline void g1a_to_compressed(const G1Affine &a, uint8_t out[48]) {
    fq_to_bytes(a.infinity ? fq_zero() : a.x, out);
    out[0] |= 1u << 7;
    if (a.infinity) out[0] |= 1u << 6;
    if (!a.infinity && fq_lexicographically_largest(a.y)) out[0] |= 1u << 5;
}

// g1/coset.rs:8-24 — 97-byte raw encoding: 12 LE u64 Montgomery limbs (x then y) + infinity byte
static inline void g1a_to_raw(const G1Affine &a, uint8_t out[97]) {
    for (int i = 0; i < 6; i++)
        for (int b = 0; b < 8; b++) {
            out[8 * i + b] = (uint8_t)(a.x.l[i] >> (8 * b));
            out[48 + 8 * i + b] = (uint8_t)(a.y.l[i] >> (8 * b));
        }
    out[96] = a.infinity;
}

// g1.rs:784-809 — the values are those of g1_to_affine applied per point
static inline void g1_batch_normalize(const G1Projective *p, G1Affine *q, size_t n) {
    Fq acc = fq_one();
    for (size_t i = 0; i < n; i++) {
        q[i].x = acc;
        if (!g1p_is_identity(p[i])) acc = fq_mul(acc, p[i].z);
    }
    Fq inv;
    (void)fq_invert(acc, inv);
    acc = inv;
    for (size_t i = n; i-- > 0;) {
        bool skip = g1p_is_identity(p[i]);
        Fq tmp = fq_mul(q[i].x, acc);
        if (!skip) acc = fq_mul(acc, p[i].z);
        if (skip) {
            q[i] = g1a_identity();
        } else {
            q[i].x = fq_mul(p[i].x, tmp);
            q[i].y = fq_mul(p[i].y, tmp);
            q[i].infinity = 0;
        }
    }
}

}  // namespace oracle
