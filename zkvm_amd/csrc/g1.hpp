// BLS12-381 G1 (y^2 = x^3 + 4) point arithmetic for device and host: affine and homogeneous projective
// coordinates with the complete Renes-Costello-Batina formulas, i.e. the same group law and the same
// identity conventions as coset-bls12_381/src/g1.rs:592-736 (identity = (0 : 1 : 0)).  Complete formulas
// have no exceptional cases, so bucket accumulation needs no branches for P + P or P - P.
#pragma once
#include "ff.hpp"

namespace zk {

// Affine point as stored on the device: x | y, 96 bytes, the first 96 bytes of the reference's raw
// encoding (coset-bls12_381/src/g1/coset.rs:8-24).  The point at infinity is stored as (0, 0), which is
// not on the curve, so it needs no separate flag array.
struct G1Affine {
    Fq x, y;
    FF_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

struct G1Proj {
    Fq x, y, z;
    FF_HD static G1Proj identity() {
        G1Proj r;
        r.x = Fq::zero();
        r.y = Fq::one();
        r.z = Fq::zero();
        return r;
    }
    FF_HD bool is_identity() const { return z.is_zero(); }
};

// On the device the 12-limb Montgomery product is several hundred instructions and the group law calls it
// 11-12 times; keeping it out of line keeps the accumulation loops inside the instruction cache.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __noinline__ Fq fq_mul_outlined(Fq a, Fq b) { return Fq::mul(a, b); }
__device__ __forceinline__ Fq fmul(const Fq &a, const Fq &b) { return fq_mul_outlined(a, b); }
#else
inline Fq fmul(const Fq &a, const Fq &b) { return Fq::mul(a, b); }
#endif
FF_HD Fq fsqr(const Fq &a) { return fmul(a, a); }
FF_HD Fq fadd(const Fq &a, const Fq &b) { return Fq::add(a, b); }
FF_HD Fq fsub(const Fq &a, const Fq &b) { return Fq::sub(a, b); }

FF_HD Fq mul_by_3b(const Fq &a) {  // b3 = 12 (g1.rs:584-588)
    Fq t = fadd(a, a);
    t = fadd(t, t);
    return fadd(fadd(t, t), t);
}

// g1.rs:658-698 — complete addition, 12 multiplications
FF_HD G1Proj g1_add(const G1Proj &a, const G1Proj &b) {
    Fq t0 = fmul(a.x, b.x);
    Fq t1 = fmul(a.y, b.y);
    Fq t2 = fmul(a.z, b.z);
    Fq t3 = fmul(fadd(a.x, a.y), fadd(b.x, b.y));
    t3 = fsub(t3, fadd(t0, t1));
    Fq t4 = fmul(fadd(a.y, a.z), fadd(b.y, b.z));
    t4 = fsub(t4, fadd(t1, t2));
    Fq y3 = fmul(fadd(a.x, a.z), fadd(b.x, b.z));
    y3 = fsub(y3, fadd(t0, t2));
    Fq x3 = fadd(t0, t0);
    t0 = fadd(x3, t0);
    t2 = mul_by_3b(t2);
    Fq z3 = fadd(t1, t2);
    t1 = fsub(t1, t2);
    y3 = mul_by_3b(y3);
    x3 = fmul(t4, y3);
    t2 = fmul(t3, t1);
    x3 = fsub(t2, x3);
    y3 = fmul(y3, t0);
    t1 = fmul(t1, z3);
    y3 = fadd(t1, y3);
    t0 = fmul(t0, t3);
    z3 = fmul(z3, t4);
    z3 = fadd(z3, t0);
    G1Proj r;
    r.x = x3;
    r.y = y3;
    r.z = z3;
    return r;
}

// g1.rs:701-736 — complete mixed addition, 11 multiplications; `negate` adds -b instead
FF_HD G1Proj g1_add_mixed(const G1Proj &a, const G1Affine &b, bool negate = false) {
    if (b.is_identity()) return a;
    Fq by = negate ? Fq::neg(b.y) : b.y;
    Fq t0 = fmul(a.x, b.x);
    Fq t1 = fmul(a.y, by);
    Fq t3 = fmul(fadd(b.x, by), fadd(a.x, a.y));
    t3 = fsub(t3, fadd(t0, t1));
    Fq t4 = fadd(fmul(by, a.z), a.y);
    Fq y3 = fadd(fmul(b.x, a.z), a.x);
    Fq x3 = fadd(t0, t0);
    t0 = fadd(x3, t0);
    Fq t2 = mul_by_3b(a.z);
    Fq z3 = fadd(t1, t2);
    t1 = fsub(t1, t2);
    y3 = mul_by_3b(y3);
    x3 = fmul(t4, y3);
    t2 = fmul(t3, t1);
    x3 = fsub(t2, x3);
    y3 = fmul(y3, t0);
    t1 = fmul(t1, z3);
    y3 = fadd(t1, y3);
    t0 = fmul(t0, t3);
    z3 = fmul(z3, t4);
    z3 = fadd(z3, t0);
    G1Proj r;
    r.x = x3;
    r.y = y3;
    r.z = z3;
    return r;
}

// g1.rs:624-655
FF_HD G1Proj g1_double(const G1Proj &p) {
    Fq t0 = fsqr(p.y);
    Fq z3 = fadd(t0, t0);
    z3 = fadd(z3, z3);
    z3 = fadd(z3, z3);
    Fq t1 = fmul(p.y, p.z);
    Fq t2 = mul_by_3b(fsqr(p.z));
    Fq x3 = fmul(t2, z3);
    Fq y3 = fadd(t0, t2);
    z3 = fmul(t1, z3);
    t1 = fadd(t2, t2);
    t2 = fadd(t1, t2);
    t0 = fsub(t0, t2);
    y3 = fmul(t0, y3);
    y3 = fadd(x3, y3);
    t1 = fmul(p.x, p.y);
    x3 = fmul(t0, t1);
    x3 = fadd(x3, x3);
    G1Proj r;
    r.x = x3;
    r.y = y3;
    r.z = z3;
    // (0 : y : 0) doubles to (0 : y^4 : 0), still the identity; normalise it as the reference does
    return p.is_identity() ? G1Proj::identity() : r;
}

// g1.rs:60-78 — one inversion; the identity maps to the reference's affine identity (0, 1, infinity)
FF_HD void g1_to_affine(const G1Proj &p, Fq &x, Fq &y, bool &infinity) {
    Fq zinv = Fq::inv(p.z);
    if (zinv.is_zero()) {
        x = Fq::zero();
        y = Fq::one();
        infinity = true;
        return;
    }
    x = fmul(p.x, zinv);
    y = fmul(p.y, zinv);
    infinity = false;
}

// Affine forms of up to 8 projective points with a single field inversion (prefix products of the z coordinates,
// one inverse, and back): the same canonical values as g1_to_affine point by point.
inline void g1_to_affine_many(const G1Proj *p, uint32_t m, Fq *x, Fq *y, bool *infinity) {
    Fq prefix[8];
    Fq run = Fq::one();
    for (uint32_t j = 0; j < m && j < 8; j++) {
        infinity[j] = p[j].z.is_zero();
        prefix[j] = run;
        if (!infinity[j]) run = Fq::mul(run, p[j].z);
    }
    Fq inv = Fq::inv(run);
    for (uint32_t j = (m < 8 ? m : 8); j-- > 0;) {
        if (infinity[j]) {
            x[j] = Fq::zero();
            y[j] = Fq::one();
            continue;
        }
        const Fq zinv = Fq::mul(inv, prefix[j]);
        inv = Fq::mul(inv, p[j].z);
        x[j] = Fq::mul(p[j].x, zinv);
        y[j] = Fq::mul(p[j].y, zinv);
    }
}

}  // namespace zk
