// Fq in 14 unsaturated limbs of 28 bits, for the inner loops of the MSM.
//
// The 32-bit-limb Montgomery product spends half of its instructions on carries: every 32x32 product needs a
// v_mad_u64_u32 plus an add-with-carry into a third accumulator word.  With 28-bit limbs a whole column of
// 14 + 14 products (56..60 bits each) fits a 64-bit accumulator, so a product is ONE v_mad_u64_u32, and additions
// and subtractions are limb-wise without carry chains.  Values are kept only loosely reduced:
//   * "N"  (normalised): limbs < 2^28, value < 2p          — what mul() returns;
//   * lazy: limbs < 2^30, value < 32p                      — what mul() accepts (see the bounds at mul()).
// The Montgomery radix is 2^392, so an element x is held as x * 2^392 mod p (+ a small multiple of p); conversions
// from and to the reference's representation (x * 2^384 mod p in 6 x 64-bit limbs, coset-bls12_381/src/fp.rs:23-88)
// cost one product each and happen only at the edges (table build, bucket output).
//
// On top: bucket accumulation in extended Jacobian ("XYZZ") coordinates, x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2:
// a mixed addition costs 8 products + 2 squarings against 11 products + two multiplications by 12 of the complete
// projective formulas (coset-bls12_381/src/g1.rs:701-736).  The formulas are not complete; the two exceptional
// cases (adding a point to itself, adding a point to its negative) are detected from ZZ3 = 0 and resolved on a
// slow path, so the sum is the same group element in every case.
#pragma once
#include "g1.hpp"

namespace zk {

// Host builds of the bucket arithmetic count every broken precondition here (limb bounds of products, operands of
// the biased subtractions); the device never calls it.
inline long &fq28_violations() {
    static long count = 0;
    return count;
}
inline void note_violation() { fq28_violations()++; }

struct Fq28 {
    static constexpr int N = 14;
    static constexpr uint32_t MASK = 0x0fffffffu;
    static constexpr uint32_t PINV = 0x0ffcfffdu;  // -p^-1 mod 2^28
    uint32_t v[14];

    FF_HD static uint32_t P(int i) {
        constexpr uint32_t M[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                                    0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
        return M[i];
    }
    FF_HD static uint32_t ONE(int i) {  // 2^392 mod p
        constexpr uint32_t M[14] = {0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu,
                                    0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au};
        return M[i];
    }
    FF_HD static uint32_t TO(int i) {  // 2^400 mod p: (x 2^384) -> x 2^392
        constexpr uint32_t M[14] = {0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u,
                                    0xa4e6fe9u, 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau};
        return M[i];
    }
    FF_HD static uint32_t FROM(int i) {  // 2^384 mod p: (x 2^392) -> x 2^384
        constexpr uint32_t M[14] = {0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u,
                                    0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u};
        return M[i];
    }
    // 4p and 16p with every limb but the top one raised by 2^28 (borrowed from the limb above): limb-wise
    // "bias - b" stays non-negative for a normalised b below 2p (B4) or below 16p - 2^364 (B16).
    FF_HD static uint32_t B4(int i) {
        constexpr uint32_t M[14] = {0x1ffeaaacu, 0x1fbffffeu, 0x1ffffee6u, 0x1fffac53u, 0x18907aaeu, 0x183dac3cu, 0x1d9cc349u,
                                    0x1ce144aeu, 0x11dd2e12u, 0x12eb35d8u, 0x1e9ed90cu, 0x19a692c5u, 0x17a8e5feu, 0x0068043u};
        return M[i];
    }
    FF_HD static uint32_t B16(int i) {
        constexpr uint32_t M[14] = {0x1ffaaab0u, 0x1efffffeu, 0x1ffffb9eu, 0x1ffeb152u, 0x1241eabeu, 0x10f6b0f5u, 0x16730d29u,
                                    0x138512beu, 0x1774b84eu, 0x1bacd763u, 0x1a7b6433u, 0x169a4b1au, 0x1ea397fdu, 0x01a0110u};
        return M[i];
    }

    FF_HD static Fq28 zero() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.v[i] = 0;
        return r;
    }
    FF_HD static Fq28 one() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.v[i] = ONE(i);
        return r;
    }
    FF_HD bool limbs_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) o |= v[i];
        return o == 0;
    }
    // for an N value (< 2p, limbs < 2^28): is it 0 modulo p?
    FF_HD bool is_zero_mod_p() const {
        uint32_t o = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            o |= v[i];
            q |= v[i] ^ P(i);
        }
        return o == 0 || q == 0;
    }

    // Montgomery product a * b * 2^-392 mod p, returned normalised and below 2p.
    // Preconditions: every column sum fits 64 bits — limbs of a and b below 2^30 with the product of the two limb
    // bounds at most 2^59.2 (e.g. 2^30 x 2^29, or 1.5 * 2^29 both) — and value(a) * value(b) < 2^392 * p, i.e. the
    // product of the two multiples of p at most 2^11 (e.g. 18p x 18p).
    FF_HD static Fq28 mul(const Fq28 &a, const Fq28 &b) {
        Fq28 r;
        uint32_t m[14];
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 14; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) c += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) c += (uint64_t)m[i] * P(k - i);
            m[k] = ((uint32_t)c * PINV) & MASK;
            c += (uint64_t)m[k] * P(0);
            c >>= 28;
        }
#pragma unroll
        for (int k = 14; k < 27; k++) {
#pragma unroll
            for (int i = k - 13; i < 14; i++) c += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = k - 13; i < 14; i++) c += (uint64_t)m[i] * P(k - i);
            r.v[k - 14] = (uint32_t)c & MASK;
            c >>= 28;
        }
        r.v[13] = (uint32_t)c;
        return r;
    }

    // (a * b + c * d) * 2^-392 mod p with ONE reduction, normalised and below 2p.  Preconditions: the column sums of
    // both products together fit 64 bits (e.g. limbs 1.5 * 2^29 x 1.5 * 2^29 plus 2^28 x 2^28) and
    // value(a) value(b) + value(c) value(d) < 2^392 * p.
    FF_HD static Fq28 mul_add(const Fq28 &a, const Fq28 &b, const Fq28 &c2, const Fq28 &d) {
        Fq28 r;
        uint32_t m[14];
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 14; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) c += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i <= k; i++) c += (uint64_t)c2.v[i] * d.v[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) c += (uint64_t)m[i] * P(k - i);
            m[k] = ((uint32_t)c * PINV) & MASK;
            c += (uint64_t)m[k] * P(0);
            c >>= 28;
        }
#pragma unroll
        for (int k = 14; k < 27; k++) {
#pragma unroll
            for (int i = k - 13; i < 14; i++) c += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = k - 13; i < 14; i++) c += (uint64_t)c2.v[i] * d.v[k - i];
#pragma unroll
            for (int i = k - 13; i < 14; i++) c += (uint64_t)m[i] * P(k - i);
            r.v[k - 14] = (uint32_t)c & MASK;
            c >>= 28;
        }
        r.v[13] = (uint32_t)c;
        return r;
    }

    // a * a * 2^-392: the off-diagonal partial products are taken once against the doubled operand (105 products
    // instead of 196 before the reduction).  Same preconditions as mul(a, a).
    FF_HD static Fq28 sqr(const Fq28 &a) {
        Fq28 r;
        uint32_t m[14], d[14];
#pragma unroll
        for (int i = 0; i < 14; i++) d[i] = a.v[i] << 1;
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 14; k++) {
#pragma unroll
            for (int i = 0; 2 * i < k; i++) c += (uint64_t)a.v[i] * d[k - i];
            if ((k & 1) == 0) c += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = 0; i < k; i++) c += (uint64_t)m[i] * P(k - i);
            m[k] = ((uint32_t)c * PINV) & MASK;
            c += (uint64_t)m[k] * P(0);
            c >>= 28;
        }
#pragma unroll
        for (int k = 14; k < 27; k++) {
#pragma unroll
            for (int i = k - 13; 2 * i < k; i++) c += (uint64_t)a.v[i] * d[k - i];
            if ((k & 1) == 0) c += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = k - 13; i < 14; i++) c += (uint64_t)m[i] * P(k - i);
            r.v[k - 14] = (uint32_t)c & MASK;
            c >>= 28;
        }
        r.v[13] = (uint32_t)c;
        return r;
    }

    FF_HD static Fq28 add(const Fq28 &a, const Fq28 &b) {  // limb-wise, no carries
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.v[i] = a.v[i] + b.v[i];
        return r;
    }
    // a - b + 4p for a normalised b < 2p;  a - b + 16p for a normalised b < 16p - 2^364
    FF_HD static Fq28 sub4(const Fq28 &a, const Fq28 &b) {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.v[i] = a.v[i] + (B4(i) - b.v[i]);
#if !defined(__HIP_DEVICE_COMPILE__)
        for (int i = 0; i < 14; i++)
            if (b.v[i] > B4(i)) note_violation();   // host builds check the "normalised b" precondition
#endif
        return r;
    }
    FF_HD static Fq28 sub16(const Fq28 &a, const Fq28 &b) {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.v[i] = a.v[i] + (B16(i) - b.v[i]);
#if !defined(__HIP_DEVICE_COMPILE__)
        for (int i = 0; i < 14; i++)
            if (b.v[i] > B16(i)) note_violation();
#endif
        return r;
    }
    // carry propagation: same value, limbs < 2^28 except the top one
    FF_HD static Fq28 normalize(const Fq28 &a) {
        Fq28 r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            uint32_t t = a.v[i] + c;
            r.v[i] = t & MASK;
            c = t >> 28;
        }
        r.v[13] = a.v[13] + c;
        return r;
    }

    // ---- conversions to and from the 12 x 32-bit Montgomery form (radix 2^384) ----
    FF_HD static Fq28 unpack(const Fq &x) {  // same integer, 28-bit limbs
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int bit = 28 * i, w = bit >> 5, s = bit & 31;
            uint32_t lo = x.v[w] >> s;
            if (s > 4 && w + 1 < 12) lo |= x.v[w + 1] << (32 - s);
            r.v[i] = lo & MASK;
        }
        return r;
    }
    FF_HD static Fq pack(const Fq28 &a) {  // normalised limbs, value < 2^384
        Fq r;
#pragma unroll
        for (int w = 0; w < 12; w++) {
            const int bit = 32 * w, i = bit / 28, s = bit - 28 * i;
            uint32_t lo = a.v[i] >> s;
            if (i + 1 < 14) lo |= a.v[i + 1] << (28 - s);
            if (s > 24 && i + 2 < 14) lo |= a.v[i + 2] << (56 - s);
            r.v[w] = lo;
        }
        return r;
    }
    FF_HD static Fq28 from_fq(const Fq &x) { return mul(unpack(x), constant_to()); }
    // lazy value (mul preconditions against a constant below p) -> fully reduced reference form
    FF_HD static Fq to_fq(const Fq28 &a) {
        Fq28 t = mul(a, constant_from());  // < 2p, normalised
        // subtract p once if t >= p
        Fq28 d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            uint32_t s = t.v[i] - P(i) - borrow;
            borrow = s >> 31;
            d.v[i] = s & MASK;
        }
        return pack(borrow ? t : d);
    }
    FF_HD static Fq28 constant_to() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.v[i] = TO(i);
        return r;
    }
    FF_HD static Fq28 constant_from() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.v[i] = FROM(i);
        return r;
    }
};

// Affine table entry in the unsaturated form: x | y, 112 bytes; the point at infinity is all zero.
struct G1Affine28 {
    Fq28 x, y;
    FF_HD bool is_identity() const { return x.limbs_zero() && y.limbs_zero(); }
};

// Bucket accumulator: X and Y normalised (below 14p and 6p), ZZ and ZZZ as returned by mul(); identity = all zero.
struct G1Xyzz {
    Fq28 x, y, zz, zzz;
    FF_HD static G1Xyzz identity() {
        G1Xyzz r;
        r.x = Fq28::zero();
        r.y = Fq28::zero();
        r.zz = Fq28::zero();
        r.zzz = Fq28::zero();
        return r;
    }
    FF_HD bool is_identity() const { return zz.limbs_zero(); }
};


#if defined(__HIP_DEVICE_COMPILE__)
// Inlined on purpose: at ~510 instructions a product is small enough for the whole mixed addition (10 of them) to
// stay inside the instruction cache, and a call would move 28 operand registers and pin the register allocation
// (measured: 161 VGPRs / 3 waves per SIMD inlined against 209 VGPRs + spills / 2 waves outlined, 24% faster MSM).
static __device__ __forceinline__ Fq28 fq28_mul_inlined(const Fq28 &a, const Fq28 &b) { return Fq28::mul(a, b); }
__device__ __forceinline__ Fq28 m28(const Fq28 &a, const Fq28 &b) { return fq28_mul_inlined(a, b); }
__device__ __forceinline__ Fq28 s28(const Fq28 &a) { return Fq28::sqr(a); }
__device__ __forceinline__ Fq28 ma28(const Fq28 &a, const Fq28 &b, const Fq28 &c, const Fq28 &d) { return Fq28::mul_add(a, b, c, d); }
#else
// Host build: every product checks the preconditions of Fq28::mul and counts violations, so that the host-side
// self-check of the bucket arithmetic also proves the limb and magnitude bounds on the inputs it sees.
inline Fq28 m28(const Fq28 &a, const Fq28 &b) {
    for (int k = 0; k < 27; k++) {
        unsigned __int128 col = (unsigned __int128)1 << 37;   // carry from the column below
        for (int i = 0; i < 14; i++) {
            int j = k - i;
            if (j < 0 || j > 13) continue;
            col += (unsigned __int128)a.v[i] * b.v[j] + (unsigned __int128)Fq28::MASK * Fq28::P(j);
        }
        if (col >> 64) fq28_violations()++;
    }
    const double ta = (double)a.v[13] + (double)(a.v[12] >> 28) + 1.0, tb = (double)b.v[13] + (double)(b.v[12] >> 28) + 1.0;
    if (ta * tb >= 2.9e13) fq28_violations()++;   // value(a) * value(b) must stay below 2^392 * p
    return Fq28::mul(a, b);
}
inline Fq28 ma28(const Fq28 &a, const Fq28 &b, const Fq28 &c, const Fq28 &d) {
    for (int k = 0; k < 27; k++) {
        unsigned __int128 col = (unsigned __int128)1 << 37;
        for (int i = 0; i < 14; i++) {
            int j = k - i;
            if (j < 0 || j > 13) continue;
            col += (unsigned __int128)a.v[i] * b.v[j] + (unsigned __int128)c.v[i] * d.v[j] + (unsigned __int128)Fq28::MASK * Fq28::P(j);
        }
        if (col >> 64) fq28_violations()++;
    }
    auto top = [](const Fq28 &x) { return (double)x.v[13] + (double)(x.v[12] >> 28) + 1.0; };
    if (top(a) * top(b) + top(c) * top(d) >= 2.9e13) fq28_violations()++;
    return Fq28::mul_add(a, b, c, d);
}
inline Fq28 s28(const Fq28 &a) {
    Fq28 checked = m28(a, a);   // bounds check + reference result
    Fq28 r = Fq28::sqr(a);
    for (int i = 0; i < 14; i++)
        if (r.v[i] != checked.v[i]) fq28_violations()++;
    return r;
}
#endif

// 2 * (x, y) for an affine point with y below 4p (lazy): "dbl-2008-s-1" with Z = 1
FF_HD G1Xyzz xyzz_double_affine(const Fq28 &x, const Fq28 &y) {
    G1Xyzz r;
    Fq28 u = Fq28::add(y, y);                    // < 8p, limbs < 2^30
    Fq28 v = s28(u);
    Fq28 w = m28(u, v);
    Fq28 s = m28(x, v);
    Fq28 xx = s28(x);
    Fq28 m = Fq28::add(Fq28::add(xx, xx), xx);   // < 6p, limbs < 3 * 2^28
    Fq28 mm = s28(m);
    Fq28 d = Fq28::sub4(Fq28::zero(), s);        // 4p - s
    r.x = Fq28::normalize(Fq28::add(mm, Fq28::add(d, d)));        // mm - 2s, < 10p
    Fq28 t = Fq28::sub16(s, r.x);
    const Fq28 ny = Fq28::normalize(Fq28::sub16(Fq28::zero(), Fq28::normalize(y)));   // y may come with lazy limbs
    r.y = ma28(m, t, w, ny);                                       // < 2p
    r.zz = v;
    r.zzz = w;
    return r;
}

// acc + (+-)p: "madd-2008-s" (8 products + 2 squarings) with the exceptional cases resolved exactly
FF_HD G1Xyzz xyzz_add_mixed(const G1Xyzz &acc, const G1Affine28 &p, bool negate) {
    if (p.is_identity()) return acc;
    const Fq28 y2 = negate ? Fq28::sub4(Fq28::zero(), p.y) : p.y;   // < 4p, limbs < 2^29
    if (acc.is_identity()) {
        G1Xyzz r;
        r.x = p.x;
        r.y = negate ? Fq28::normalize(y2) : y2;
        r.zz = Fq28::one();
        r.zzz = Fq28::one();
        return r;
    }
    const Fq28 u2 = m28(p.x, acc.zz);
    const Fq28 s2 = m28(y2, acc.zzz);
    const Fq28 pd = Fq28::sub16(u2, acc.x);      // < 18p, limbs < 1.5 * 2^29
    const Fq28 rd = Fq28::sub16(s2, acc.y);
    const Fq28 pp = s28(pd);
    const Fq28 zz3 = m28(acc.zz, pp);
    if (zz3.is_zero_mod_p()) {
        // pd = 0 mod p: the two points share their x coordinate
        if (m28(rd, Fq28::one()).is_zero_mod_p()) return xyzz_double_affine(p.x, y2);   // P + P
        return G1Xyzz::identity();                                                        // P + (-P)
    }
    const Fq28 ppp = m28(pd, pp);
    const Fq28 q = m28(acc.x, pp);
    const Fq28 rr = s28(rd);
    G1Xyzz r;
    const Fq28 nq = Fq28::sub4(Fq28::zero(), q);
    r.x = Fq28::normalize(Fq28::add(Fq28::sub4(rr, ppp), Fq28::add(nq, nq)));   // rr - ppp - 2q, < 14p
    const Fq28 t = Fq28::sub16(q, r.x);
    const Fq28 ny = Fq28::normalize(Fq28::sub16(Fq28::zero(), acc.y));         // 16p - y, limbs < 2^28
    r.y = ma28(rd, t, ny, ppp);                                                // rd t - y ppp with one reduction, < 2p
    r.zz = zz3;
    r.zzz = m28(acc.zzz, ppp);
    return r;
}

// 2 * a for a non-identity accumulator: "dbl-2008-s-1"
FF_HD G1Xyzz xyzz_double(const G1Xyzz &a) {
    G1Xyzz r;
    Fq28 u = Fq28::add(a.y, a.y);                // < 12p, limbs < 2^29
    Fq28 v = s28(u);
    Fq28 w = m28(u, v);
    Fq28 s = m28(a.x, v);
    Fq28 xx = s28(a.x);
    Fq28 m = Fq28::add(Fq28::add(xx, xx), xx);   // < 6p, limbs < 3 * 2^28
    Fq28 mm = s28(m);
    Fq28 d = Fq28::sub4(Fq28::zero(), s);
    r.x = Fq28::normalize(Fq28::add(mm, Fq28::add(d, d)));        // < 10p
    Fq28 t = Fq28::sub16(s, r.x);
    const Fq28 ny = Fq28::normalize(Fq28::sub16(Fq28::zero(), a.y));
    r.y = ma28(m, t, w, ny);                                       // < 2p
    r.zz = m28(v, a.zz);
    r.zzz = m28(w, a.zzz);
    return r;
}

// a + b for two accumulators: "add-2008-s" (12 products + 2 squarings), exceptional cases resolved exactly
FF_HD G1Xyzz xyzz_add(const G1Xyzz &a, const G1Xyzz &b) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    const Fq28 u1 = m28(a.x, b.zz);
    const Fq28 u2 = m28(b.x, a.zz);
    const Fq28 s1 = m28(a.y, b.zzz);
    const Fq28 s2 = m28(b.y, a.zzz);
    const Fq28 pd = Fq28::sub4(u2, u1);          // < 6p, limbs < 1.5 * 2^29
    const Fq28 rd = Fq28::sub4(s2, s1);
    const Fq28 pp = s28(pd);
    const Fq28 zz12 = m28(a.zz, b.zz);
    const Fq28 zz3 = m28(zz12, pp);
    if (zz3.is_zero_mod_p()) {
        if (m28(rd, Fq28::one()).is_zero_mod_p()) return xyzz_double(a);
        return G1Xyzz::identity();
    }
    const Fq28 ppp = m28(pd, pp);
    const Fq28 q = m28(u1, pp);
    const Fq28 rr = s28(rd);
    G1Xyzz r;
    const Fq28 nq = Fq28::sub4(Fq28::zero(), q);
    r.x = Fq28::normalize(Fq28::add(Fq28::sub4(rr, ppp), Fq28::add(nq, nq)));   // < 14p
    const Fq28 t = Fq28::sub16(q, r.x);
    const Fq28 ns1 = Fq28::normalize(Fq28::sub4(Fq28::zero(), s1));            // 4p - s1, limbs < 2^28
    r.y = ma28(rd, t, ns1, ppp);                                               // < 2p
    r.zz = zz3;
    r.zzz = m28(m28(a.zzz, b.zzz), ppp);
    return r;
}

// The accumulator as a homogeneous projective point in the reference form: (X ZZZ : Y ZZ : ZZ ZZZ)
FF_HD G1Proj xyzz_to_proj(const G1Xyzz &a) {
    if (a.is_identity()) return G1Proj::identity();
    G1Proj r;
    r.x = Fq28::to_fq(m28(a.x, a.zzz));
    r.y = Fq28::to_fq(m28(a.y, a.zz));
    r.z = Fq28::to_fq(m28(a.zz, a.zzz));
    return r;
}

// Table entry from the reference form; (0, 0) -> all limbs zero
FF_HD G1Affine28 affine_to_28(const G1Affine &p) {
    G1Affine28 r;
    r.x = Fq28::from_fq(p.x);
    r.y = Fq28::from_fq(p.y);
    return r;
}

}  // namespace zk
