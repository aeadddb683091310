// Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (and, through the same source, for the
// host side of the library).  N limbs little-endian; the in-memory image of an element is identical to
// the reference's 64-bit-limb Montgomery representation (coset-bls12_381/src/scalar.rs:28, fp.rs:23), so
// buffers cross the C boundary without conversion.  Every result is fully reduced to [0, p), as the
// reference's are (scalar.rs:524,577; fp.rs:332-348) — results are compared as raw limbs.
//
// Additions and subtractions work on the 32-bit words with the hardware carry; the Montgomery product unpacks
// to unsaturated limbs (see Fp_::mul) so that its columns need no carry instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FF_HD __host__ __device__ __forceinline__

namespace zk {

// P supplies: N, mod(i), one(i) (= R mod p), r2(i), and the unsaturated form UL, UN, USH, UINV, umod(i)
template <class P>
struct Fp_ {
    static constexpr int N = P::N;
    uint32_t v[N];

    FF_HD static uint32_t mod_limb(int i) { return P::mod(i); }

    FF_HD static Fp_ zero() {
        Fp_ r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    FF_HD static Fp_ one() {
        Fp_ r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = P::one(i);
        return r;
    }
    FF_HD static Fp_ r2() {
        Fp_ r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = P::r2(i);
        return r;
    }
    FF_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= v[i];
        return o == 0;
    }
    FF_HD bool operator==(const Fp_ &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    FF_HD bool operator!=(const Fp_ &b) const { return !(*this == b); }
    // the limbs, read as an integer, are below the modulus
    FF_HD bool below_modulus() const {
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) borrow = (((uint64_t)v[i] - P::mod(i) - borrow) >> 32) & 1;
        return borrow != 0;
    }

    // r = a - p if a >= p else a   (a < 2p)
    FF_HD static Fp_ reduce_once(const Fp_ &a) {
        Fp_ d;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a.v[i] - P::mod(i) - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (t >> 32) & 1;
        }
        Fp_ r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
        return r;
    }

    FF_HD static Fp_ add(const Fp_ &a, const Fp_ &b) {
        Fp_ s;
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a.v[i] + b.v[i] + carry;
            s.v[i] = (uint32_t)t;
            carry = t >> 32;
        }
        // both moduli leave the top bit(s) of the top limb free, so a + b < 2p never carries out
        return reduce_once(s);
    }
    FF_HD static Fp_ sub(const Fp_ &a, const Fp_ &b) {
        Fp_ d;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (t >> 32) & 1;
        }
        uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)d.v[i] + (P::mod(i) & mask) + carry;
            d.v[i] = (uint32_t)t;
            carry = t >> 32;
        }
        return d;
    }
    FF_HD static Fp_ neg(const Fp_ &a) {
        Fp_ d;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)P::mod(i) - a.v[i] - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (t >> 32) & 1;
        }
        uint32_t mask = a.is_zero() ? 0u : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < N; i++) d.v[i] &= mask;
        return d;
    }
    FF_HD static Fp_ dbl(const Fp_ &a) { return add(a, a); }

    // ---- Montgomery product a*b*R^-1 mod p, canonical ---------------------------------------------------------
    // Computed on unsaturated limbs: UN limbs of UL bits (9 x 29 for Fr, 14 x 28 for Fq).  A column of UN + UN
    // partial products then fits one 64-bit accumulator, so every product is a single v_mad_u64_u32 with no carry
    // instruction behind it (the 32-bit-limb form needs an add-with-carry per product).  The wider radix
    // 2^(UL*UN) = R * 2^USH is compensated by taking b * 2^USH while it is unpacked: a * (b 2^USH) / 2^(UL*UN) =
    // a b / R exactly, below 2p, and one conditional subtraction makes it canonical again.
    FF_HD static uint32_t unpack_limb(const Fp_ &x, int start) {   // bits [start, start + UL) of x, zero outside
        constexpr uint32_t MASK = (1u << P::UL) - 1u;
        if (start < 0) return (x.v[0] << (-start)) & MASK;
        const int w = start >> 5, s = start & 31;
        uint32_t lo = w < N ? x.v[w] >> s : 0u;
        if (s + P::UL > 32 && w + 1 < N) lo |= x.v[w + 1] << (32 - s);
        return lo & MASK;
    }
    // x[i] = bits [UL i - shift, UL i - shift + UL) of a: the integer a * 2^shift in UN limbs of UL bits
    FF_HD static void unpack(const Fp_ &a, uint32_t *x, int shift) {
#pragma unroll
        for (int i = 0; i < P::UN; i++) x[i] = unpack_limb(a, P::UL * i - shift);
    }
    // r = x * y / 2^(UL*UN) mod p on limbs: r normalised (limbs < 2^UL) and below x y / 2^(UL*UN) + p.
    // Needs every column sum below 2^64: UN * (max x limb) * (max y limb) + UN * 2^(2 UL) < 2^64.
    FF_HD static void mul_limbs(const uint32_t *x, const uint32_t *y, uint32_t *r) {
        constexpr int UL = P::UL, UN = P::UN;
        constexpr uint32_t MASK = (1u << UL) - 1u;
        uint32_t m[UN];
        uint64_t c = 0;
        if constexpr (P::UINV == MASK) {
            // The modulus is 1 modulo 2^UL: m = -c mod 2^UL, and c + m * 1 is c rounded up to a multiple of 2^UL.  With
            // c' = c + (2^UL - 1) the carry is c' >> UL and m = ~c' mod 2^UL: one 64-bit addition of a constant in place
            // of a negation, a 64-bit addition of m and their dependency.
#pragma unroll
            for (int k = 0; k < UN; k++) {
                c += MASK;
#pragma unroll
                for (int i = 0; i <= k; i++) c += (uint64_t)x[i] * y[k - i];
#pragma unroll
                for (int i = 0; i < k; i++) c += (uint64_t)m[i] * P::umod(k - i);
                m[k] = ~(uint32_t)c & MASK;
                c >>= UL;
            }
        } else {
#pragma unroll
            for (int k = 0; k < UN; k++) {
#pragma unroll
                for (int i = 0; i <= k; i++) c += (uint64_t)x[i] * y[k - i];
#pragma unroll
                for (int i = 0; i < k; i++) c += (uint64_t)m[i] * P::umod(k - i);
                m[k] = ((uint32_t)c * P::UINV) & MASK;
                c += (uint64_t)m[k] * P::umod(0);
                c >>= UL;
            }
        }
#pragma unroll
        for (int k = UN; k < 2 * UN - 1; k++) {
#pragma unroll
            for (int i = k - UN + 1; i < UN; i++) c += (uint64_t)x[i] * y[k - i];
#pragma unroll
            for (int i = k - UN + 1; i < UN; i++) c += (uint64_t)m[i] * P::umod(k - i);
            r[k - UN] = (uint32_t)c & MASK;
            c >>= UL;
        }
        r[UN - 1] = (uint32_t)c;
    }
    // normalised r < 2p -> r mod p
    FF_HD static void cond_sub_limbs(uint32_t *r) {
        constexpr int UL = P::UL, UN = P::UN;
        constexpr uint32_t MASK = (1u << UL) - 1u;
        uint32_t d[UN], borrow = 0;
#pragma unroll
        for (int i = 0; i < UN; i++) {
            uint32_t t = r[i] - P::umod(i) - borrow;
            borrow = t >> 31;
            d[i] = t & MASK;
        }
#pragma unroll
        for (int i = 0; i < UN; i++) r[i] = borrow ? r[i] : d[i];
    }
    // normalised limbs of a value below 2^(32 N) -> 32-bit words
    FF_HD static Fp_ pack(const uint32_t *r) {
        constexpr int UL = P::UL, UN = P::UN;
        Fp_ o;
#pragma unroll
        for (int w = 0; w < N; w++) {
            const int bit = 32 * w, i = bit / UL, s = bit - UL * i;
            uint32_t lo = r[i] >> s;
            if (i + 1 < UN) lo |= r[i + 1] << (UL - s);
            if (2 * UL - s < 32 && i + 2 < UN) lo |= r[i + 2] << (2 * UL - s);
            o.v[w] = lo;
        }
        return o;
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    // Host side: the same canonical product on 64-bit words (operand scanning with the reduction interleaved, 128-bit
    // partial products).  The unsaturated form above is shaped for v_mad_u64_u32; on a CPU core this one is several
    // times faster, and the host does serial work between launches (final sums of an MSM, inversions, challenges).
    static inline Fp_ mul_words64(const Fp_ &a, const Fp_ &b) {
        constexpr int W = N / 2;
        typedef unsigned __int128 u128;
        uint64_t x[W], y[W], p[W];
        __builtin_memcpy(x, a.v, sizeof x);   // the host is little-endian: two 32-bit limbs are one 64-bit word
        __builtin_memcpy(y, b.v, sizeof y);
#pragma unroll
        for (int i = 0; i < W; i++) p[i] = P::mod(2 * i) | (uint64_t)P::mod(2 * i + 1) << 32;
        uint64_t inv = 1;   // 1 / p mod 2^64 by Newton's iteration (the correct bits double each time), then negated:
#pragma unroll
        for (int k = 0; k < 6; k++) inv *= 2 - p[0] * inv;   // constants only, folded at compile time
        inv = 0 - inv;
        uint64_t t[W + 2];
#pragma unroll
        for (int i = 0; i < W + 2; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            u128 c = 0;
#pragma unroll
            for (int j = 0; j < W; j++) {
                c += (u128)x[j] * y[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[W];
            t[W] = (uint64_t)c;
            t[W + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv;
            c = ((u128)m * p[0] + t[0]) >> 64;
#pragma unroll
            for (int j = 1; j < W; j++) {
                c += (u128)m * p[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[W];
            t[W - 1] = (uint64_t)c;
            t[W] = t[W + 1] + (uint64_t)(c >> 64);
        }
        // below 2p, and 2p fits the words (both moduli leave the top bit free): one conditional subtraction
        uint64_t d[W], borrow = 0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            const u128 diff = (u128)t[i] - p[i] - borrow;
            d[i] = (uint64_t)diff;
            borrow = (uint64_t)(diff >> 64) & 1;
        }
#pragma unroll
        for (int i = 0; i < W; i++) d[i] = borrow ? t[i] : d[i];
        Fp_ r;
        __builtin_memcpy(r.v, d, sizeof d);
        return r;
    }
#endif
    FF_HD static Fp_ mul(const Fp_ &a, const Fp_ &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t x[P::UN], y[P::UN], r[P::UN];
        unpack(a, x, 0);
        unpack(b, y, P::USH);
        mul_limbs(x, y, r);   // < 2p
        cond_sub_limbs(r);
        return pack(r);
#else
        return mul_words64(a, b);
#endif
    }
    FF_HD static Fp_ sqr(const Fp_ &a) { return mul(a, a); }

    // value * R^-1 : leaves Montgomery form (scalar/coset.rs:242-246 `reduce`)
    FF_HD static Fp_ from_mont(const Fp_ &a) {
        Fp_ o = zero();
        o.v[0] = 1;
        return mul(a, o);
    }
    FF_HD static Fp_ to_mont(const Fp_ &a) { return mul(a, r2()); }
    // small integer -> field element
    FF_HD static Fp_ from_u32(uint32_t x) {
        Fp_ o = zero();
        o.v[0] = x;
        return to_mont(o);
    }

    FF_HD Fp_ operator+(const Fp_ &b) const { return add(*this, b); }
    FF_HD Fp_ operator-(const Fp_ &b) const { return sub(*this, b); }
    FF_HD Fp_ operator*(const Fp_ &b) const { return mul(*this, b); }
    FF_HD Fp_ operator-() const { return neg(*this); }

    // a^e for a multi-limb little-endian exponent (square-and-multiply from the top bit)
    FF_HD static Fp_ pow(const Fp_ &a, const uint32_t *e, int n_limbs) {
        Fp_ res = one();
        for (int l = n_limbs - 1; l >= 0; l--)
            for (int b = 31; b >= 0; b--) {
                res = sqr(res);
                if ((e[l] >> b) & 1) res = mul(res, a);
            }
        return res;
    }
    FF_HD static Fp_ pow_u64(const Fp_ &a, uint64_t e) {
        if (e == 0) return one();
        int top = 63;   // squarings start below the highest set bit: small exponents (thread indexes) cost few products
        while (!((e >> top) & 1)) top--;
        Fp_ res = a;
        for (int b = top - 1; b >= 0; b--) {
            res = sqr(res);
            if ((e >> b) & 1) res = mul(res, a);
        }
        return res;
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    // Host side: the inverse by the binary extended Euclidean algorithm on 64-bit words (u, v shrink by shifts and
    // subtractions while x1 a = u and x2 a = v hold modulo p), about four times faster than the power a^(p-2) below.
    // The host inverts once per commitment batch and per permutation, between two launches.  The argument is a R, its
    // integer inverse a^-1 R^-1; two products by R^2 lead back to a^-1 R.  Zero maps to zero.
    static inline Fp_ inv_words64(const Fp_ &a) {
        constexpr int W = N / 2;
        if (a.is_zero()) return a;
        uint64_t u[W], v[W], x1[W], x2[W], p[W];
        __builtin_memcpy(u, a.v, sizeof u);
        for (int i = 0; i < W; i++) {
            p[i] = P::mod(2 * i) | (uint64_t)P::mod(2 * i + 1) << 32;
            v[i] = p[i];
            x1[i] = x2[i] = 0;
        }
        x1[0] = 1;
        auto is_one = [](const uint64_t *t) {
            uint64_t o = t[0] ^ 1;
            for (int i = 1; i < W; i++) o |= t[i];
            return o == 0;
        };
        auto shr1 = [](uint64_t *t) {
            for (int i = 0; i < W - 1; i++) t[i] = t[i] >> 1 | t[i + 1] << 63;
            t[W - 1] >>= 1;
        };
        auto halve = [&](uint64_t *x) {   // x / 2 modulo p: x + p stays within the words (both moduli leave the top bit free)
            if (x[0] & 1) {
                unsigned __int128 c = 0;
                for (int i = 0; i < W; i++) {
                    c += (unsigned __int128)x[i] + p[i];
                    x[i] = (uint64_t)c;
                    c >>= 64;
                }
            }
            shr1(x);
        };
        auto geq = [](const uint64_t *s, const uint64_t *t) {
            for (int i = W - 1; i >= 0; i--)
                if (s[i] != t[i]) return s[i] > t[i];
            return true;
        };
        auto sub = [](uint64_t *s, const uint64_t *t) {   // s -= t, returns the borrow
            uint64_t borrow = 0;
            for (int i = 0; i < W; i++) {
                const unsigned __int128 d = (unsigned __int128)s[i] - t[i] - borrow;
                s[i] = (uint64_t)d;
                borrow = (uint64_t)(d >> 64) & 1;
            }
            return borrow;
        };
        auto sub_mod = [&](uint64_t *s, const uint64_t *t) {   // s = s - t modulo p for s, t below p
            if (sub(s, t)) {
                unsigned __int128 c = 0;
                for (int i = 0; i < W; i++) {
                    c += (unsigned __int128)s[i] + p[i];
                    s[i] = (uint64_t)c;
                    c >>= 64;
                }
            }
        };
        while (!is_one(u) && !is_one(v)) {
            while (!(u[0] & 1)) {
                shr1(u);
                halve(x1);
            }
            while (!(v[0] & 1)) {
                shr1(v);
                halve(x2);
            }
            if (geq(u, v)) {
                sub(u, v);
                sub_mod(x1, x2);
            } else {
                sub(v, u);
                sub_mod(x2, x1);
            }
        }
        Fp_ r;
        __builtin_memcpy(r.v, is_one(u) ? x1 : x2, sizeof x1);
        return mul(mul(r, r2()), r2());
    }
#endif
    // a^(p-2); zero maps to zero
    FF_HD static Fp_ inv(const Fp_ &a) {
#if !defined(__HIP_DEVICE_COMPILE__)
        return inv_words64(a);
#endif
        uint32_t e[N];
        uint64_t borrow = 2;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)P::mod(i) - borrow;
            e[i] = (uint32_t)t;
            borrow = (t >> 32) & 1;
        }
        return pow(a, e, N);
    }
};

// ---- BLS12-381 scalar field (coset-bls12_381/src/scalar.rs:80-85,156-170) --------------------------------
struct FrParams {
    static constexpr int N = 8;
    // unsaturated form used inside mul: 9 limbs of 29 bits, radix 2^261 = 2^256 * 2^5
    static constexpr int UL = 29, UN = 9, USH = 5;
    static constexpr uint32_t UINV = 0x1fffffffu;   // -r^-1 mod 2^29
    FF_HD static constexpr uint32_t umod(int i) {
        constexpr uint32_t M[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu,
                                   0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
        return M[i];
    }
    FF_HD static uint32_t mod(int i) {
        constexpr uint32_t M[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return M[i];
    }
    FF_HD static uint32_t one(int i) {
        constexpr uint32_t M[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return M[i];
    }
    FF_HD static uint32_t r2(int i) {
        constexpr uint32_t M[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return M[i];
    }
};
// ---- BLS12-381 base field (coset-bls12_381/src/fp.rs:77-104) -----------------------------------------------
struct FqParams {
    static constexpr int N = 12;
    // unsaturated form used inside mul: 14 limbs of 28 bits, radix 2^392 = 2^384 * 2^8
    static constexpr int UL = 28, UN = 14, USH = 8;
    static constexpr uint32_t UINV = 0x0ffcfffdu;   // -p^-1 mod 2^28
    FF_HD static uint32_t umod(int i) {
        constexpr uint32_t M[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                                    0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
        return M[i];
    }
    FF_HD static uint32_t mod(int i) {
        constexpr uint32_t M[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                    0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return M[i];
    }
    FF_HD static uint32_t one(int i) {
        constexpr uint32_t M[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                    0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return M[i];
    }
    FF_HD static uint32_t r2(int i) {
        constexpr uint32_t M[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                    0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return M[i];
    }
};

typedef Fp_<FrParams> Fr;
typedef Fp_<FqParams> Fq;

}  // namespace zk
