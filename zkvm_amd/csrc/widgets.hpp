// Gate identities of the PLONK widgets as functions of the wire values (a, b, c, d), the next-row values
// (a_w, b_w, d_w) and selector values.  Every widget is a list of sub-identities ("parts") that the prover weights
// with powers of kappa = separation_challenge^2.  The same formulas serve three callers, so they are written over a
// ring R with value type R::V:
//   * the host, once per proof on the 15 evaluations at z (widget `compute_linearization`): field elements;
//   * the row-wise check of a witness on the domain (does every part vanish on every row that selects the widget?):
//     field elements, on the device;
//   * the top coefficients of a widget's term in the quotient's numerator: truncated series (prover.hip).
// The quotient kernels evaluate the same identities in their own limb form (poly.hip).
// References: plonk/src/proof_system/widget/{range,logic,ecc/scalar_mul/fixed_base,ecc/curve_addition}/proverkey.rs.
#pragma once
#include "ff.hpp"

namespace zk {

struct InlineMul {
    FF_HD static Fr mul(const Fr &a, const Fr &b) { return Fr::mul(a, b); }
};

// k * a for a small compile-time k by doubling and adding (cheaper than a Montgomery product)
template <uint32_t K>
FF_HD Fr mul_small(const Fr &a) {
    if constexpr (K == 0) {
        return Fr::zero();
    } else if constexpr (K == 1) {
        return a;
    } else if constexpr ((K & 1) == 0) {
        return Fr::dbl(mul_small<K / 2>(a));
    } else {
        return Fr::add(Fr::dbl(mul_small<K / 2>(a)), a);
    }
}

// coset-jubjub/src/lib.rs:368-373 (EDWARDS_D = -10240 / 10241), in Montgomery form
FF_HD Fr edwards_d() {
    Fr d;
    const uint32_t w[8] = {0xb974f6b0u, 0x2a522455u, 0x0d9acab3u, 0xfc6cc9efu, 0xc27628d1u, 0x7a08fb94u, 0xfe0e262eu, 0x57f8f6a8u};
#pragma unroll
    for (int i = 0; i < 8; i++) d.v[i] = w[i];
    return d;
}

// Field elements as the ring, with the product routine M (out of line on the device, inline on the host)
template <class M>
struct FrRing {
    typedef Fr V;
    FF_HD static Fr add(const Fr &a, const Fr &b) { return Fr::add(a, b); }
    FF_HD static Fr sub(const Fr &a, const Fr &b) { return Fr::sub(a, b); }
    FF_HD static Fr mul(const Fr &a, const Fr &b) { return M::mul(a, b); }
    FF_HD static Fr scale(const Fr &a, const Fr &k) { return M::mul(a, k); }   // by a constant of the field
    template <uint32_t K>
    FF_HD static Fr small(const Fr &a) {
        return mul_small<K>(a);
    }
    FF_HD static Fr cst(uint32_t k) { return k == 1 ? Fr::one() : Fr::from_u32(k); }
};

// Powers of kappa = separation_challenge^2, the weights of a widget's parts
struct SepPowers {
    Fr k1, k2, k3, k4;
};
template <class M>
FF_HD SepPowers sep_powers(const Fr &sep) {
    SepPowers k;
    k.k1 = M::mul(sep, sep);
    k.k2 = M::mul(k.k1, k.k1);
    k.k3 = M::mul(k.k2, k.k1);
    k.k4 = M::mul(k.k3, k.k1);
    return k;
}
// parts[0] + kappa parts[1] + kappa^2 parts[2] + ...
template <class R>
FF_HD typename R::V w_combine(const typename R::V *parts, int count, const SepPowers &k) {
    const Fr *kp[4] = {&k.k1, &k.k2, &k.k3, &k.k4};
    typename R::V s = parts[0];
    for (int j = 1; j < count; j++) s = R::add(s, R::scale(parts[j], *kp[j - 1]));
    return s;
}

// range/proverkey.rs:92-97 and logic/proverkey.rs:124-129: f (f-1) (f-2) (f-3)
template <class R>
FF_HD typename R::V w_delta(const typename R::V &f) {
    typedef typename R::V V;
    const V one = R::cst(1);
    const V f1 = R::sub(f, one), f2 = R::sub(f1, one), f3 = R::sub(f2, one);
    return R::mul(R::mul(f, f1), R::mul(f2, f3));
}

// range/proverkey.rs:44-56 (quotient) and :69-84 (linearization): four parts
static const int W_RANGE_PARTS = 4;
template <class R>
FF_HD void w_range_parts(const typename R::V &a, const typename R::V &b, const typename R::V &c, const typename R::V &d,
                         const typename R::V &d_w, typename R::V *p) {
    p[0] = w_delta<R>(R::sub(c, R::template small<4>(d)));
    p[1] = w_delta<R>(R::sub(b, R::template small<4>(c)));
    p[2] = w_delta<R>(R::sub(a, R::template small<4>(b)));
    p[3] = w_delta<R>(R::sub(d_w, R::template small<4>(a)));
}

// logic/proverkey.rs:132-155
template <class R>
FF_HD typename R::V w_delta_xor_and(const typename R::V &a, const typename R::V &b, const typename R::V &w, const typename R::V &c,
                                    const typename R::V &q_c) {
    typedef typename R::V V;
    const V ab = R::add(a, b);
    const V inner = R::add(R::sub(R::template small<4>(w), R::template small<18>(ab)), R::cst(81));
    V F = R::add(R::mul(w, inner), R::template small<18>(R::add(R::mul(a, a), R::mul(b, b))));
    F = R::add(R::sub(F, R::template small<81>(ab)), R::cst(83));
    F = R::mul(w, F);
    const V E = R::sub(R::template small<3>(R::add(ab, c)), R::template small<2>(F));
    const V B = R::mul(q_c, R::sub(R::template small<9>(c), R::template small<3>(ab)));
    return R::add(B, E);
}

// logic/proverkey.rs:51-77 and :88-118: c_0 ... c_4
static const int W_LOGIC_PARTS = 5;
template <class R>
FF_HD void w_logic_parts(const typename R::V &a, const typename R::V &a_w, const typename R::V &b, const typename R::V &b_w,
                         const typename R::V &c, const typename R::V &d, const typename R::V &d_w, const typename R::V &q_c,
                         typename R::V *p) {
    typedef typename R::V V;
    const V ai = R::sub(a_w, R::template small<4>(a)), bi = R::sub(b_w, R::template small<4>(b));
    const V di = R::sub(d_w, R::template small<4>(d));
    p[0] = w_delta<R>(ai);
    p[1] = w_delta<R>(bi);
    p[2] = w_delta<R>(di);
    p[3] = R::sub(c, R::mul(ai, bi));
    p[4] = w_delta_xor_and<R>(ai, bi, c, di, q_c);
}

// ecc/scalar_mul/fixed_base/proverkey.rs:53-98 and :110-154: bit, x y, x accumulator, y accumulator
static const int W_FIXED_PARTS = 4;
template <class R>
FF_HD void w_fixed_base_parts(const typename R::V &a, const typename R::V &a_w, const typename R::V &b, const typename R::V &b_w,
                              const typename R::V &c, const typename R::V &d, const typename R::V &d_w, const typename R::V &q_l,
                              const typename R::V &q_r, const typename R::V &q_c, typename R::V *p) {
    typedef typename R::V V;
    const V one = R::cst(1);
    const V bit = R::sub(R::sub(d_w, d), d);                                          // extract_bit :160-162
    p[0] = R::mul(R::mul(bit, R::sub(bit, one)), R::add(bit, one));                   // :164-167
    const V y_alpha = R::add(R::mul(R::mul(bit, bit), R::sub(q_r, one)), one);
    const V x_alpha = R::mul(bit, q_l);
    p[1] = R::sub(R::mul(bit, q_c), c);
    const V dxy = R::scale(R::mul(R::mul(c, a), b), edwards_d());
    const V x_lhs = R::add(a_w, R::mul(a_w, dxy));
    const V x_rhs = R::add(R::mul(a, y_alpha), R::mul(b, x_alpha));
    p[2] = R::sub(x_lhs, x_rhs);
    const V y_lhs = R::sub(b_w, R::mul(b_w, dxy));
    const V y_rhs = R::add(R::mul(b, y_alpha), R::mul(a, x_alpha));
    p[3] = R::sub(y_lhs, y_rhs);
}

// ecc/curve_addition/proverkey.rs:46-80 and :92-124 ; x1 = a, x3 = a_w, y1 = b, y3 = b_w, x2 = c, y2 = d, x1*y2 = d_w
static const int W_CURVE_PARTS = 3;
template <class R>
FF_HD void w_curve_add_parts(const typename R::V &a, const typename R::V &a_w, const typename R::V &b, const typename R::V &b_w,
                             const typename R::V &c, const typename R::V &d, const typename R::V &d_w, typename R::V *p) {
    typedef typename R::V V;
    p[0] = R::sub(R::mul(a, d), d_w);
    const V y1x2 = R::mul(b, c), y1y2 = R::mul(b, d), x1x2 = R::mul(a, c);
    const V dd = R::mul(R::scale(d_w, edwards_d()), y1x2);
    p[1] = R::sub(R::add(d_w, y1x2), R::add(a_w, R::mul(a_w, dd)));
    p[2] = R::sub(R::add(y1y2, x1x2), R::sub(b_w, R::mul(b_w, dd)));
}

// The sums before "* selector * separation challenge" on field elements (the linearization's use)
template <class M>
FF_HD Fr w_range_identity(const SepPowers &k, const Fr &a, const Fr &b, const Fr &c, const Fr &d, const Fr &d_w) {
    Fr p[W_RANGE_PARTS];
    w_range_parts<FrRing<M>>(a, b, c, d, d_w, p);
    return w_combine<FrRing<M>>(p, W_RANGE_PARTS, k);
}
template <class M>
FF_HD Fr w_logic_identity(const SepPowers &k, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c, const Fr &d,
                          const Fr &d_w, const Fr &q_c) {
    Fr p[W_LOGIC_PARTS];
    w_logic_parts<FrRing<M>>(a, a_w, b, b_w, c, d, d_w, q_c, p);
    return w_combine<FrRing<M>>(p, W_LOGIC_PARTS, k);
}
template <class M>
FF_HD Fr w_fixed_base_identity(const SepPowers &k, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c,
                               const Fr &d, const Fr &d_w, const Fr &q_l, const Fr &q_r, const Fr &q_c) {
    Fr p[W_FIXED_PARTS];
    w_fixed_base_parts<FrRing<M>>(a, a_w, b, b_w, c, d, d_w, q_l, q_r, q_c, p);
    return w_combine<FrRing<M>>(p, W_FIXED_PARTS, k);
}
template <class M>
FF_HD Fr w_curve_add_identity(const SepPowers &k, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c,
                              const Fr &d, const Fr &d_w) {
    Fr p[W_CURVE_PARTS];
    w_curve_add_parts<FrRing<M>>(a, a_w, b, b_w, c, d, d_w, p);
    return w_combine<FrRing<M>>(p, W_CURVE_PARTS, k);
}

}  // namespace zk
