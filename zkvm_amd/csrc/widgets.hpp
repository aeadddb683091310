// Gate identities of the PLONK widgets as functions of the wire values (a, b, c, d), the next-row values
// (a_w, b_w, d_w) and selector values.  The same formulas serve two callers:
//   * the quotient kernel, per point of the 8n coset (widget `compute_quotient_i`), and
//   * the host, once per proof on the 15 evaluations at z (widget `compute_linearization`),
// so they are templated on the product routine M (out-of-line on the device, inline on the host).
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
template <class M>
FF_HD Fr edwards_d() {
    Fr d;
    const uint32_t w[8] = {0xb974f6b0u, 0x2a522455u, 0x0d9acab3u, 0xfc6cc9efu, 0xc27628d1u, 0x7a08fb94u, 0xfe0e262eu, 0x57f8f6a8u};
#pragma unroll
    for (int i = 0; i < 8; i++) d.v[i] = w[i];
    return d;
}

// Powers of kappa = separation_challenge^2, the weights of a widget's sub-identities: the same for every point of a
// proof, so the quotient kernels receive them ready-made.
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

// range/proverkey.rs:92-97 and logic/proverkey.rs:124-129: f (f-1) (f-2) (f-3)
template <class M>
FF_HD Fr w_delta(const Fr &f) {
    const Fr one = Fr::one();
    Fr f1 = Fr::sub(f, one), f2 = Fr::sub(f1, one), f3 = Fr::sub(f2, one);
    return M::mul(M::mul(f, f1), M::mul(f2, f3));
}

// range/proverkey.rs:44-56 (quotient) and :69-84 (linearization): the sum before "* q_range * separator"
template <class M>
FF_HD Fr w_range_identity(const SepPowers &k, const Fr &a, const Fr &b, const Fr &c, const Fr &d, const Fr &d_w) {
    const Fr &kappa = k.k1, &kappa_sq = k.k2, &kappa_cu = k.k3;
    Fr s = w_delta<M>(Fr::sub(c, mul_small<4>(d)));
    s = Fr::add(s, M::mul(w_delta<M>(Fr::sub(b, mul_small<4>(c))), kappa));
    s = Fr::add(s, M::mul(w_delta<M>(Fr::sub(a, mul_small<4>(b))), kappa_sq));
    s = Fr::add(s, M::mul(w_delta<M>(Fr::sub(d_w, mul_small<4>(a))), kappa_cu));
    return s;
}

// logic/proverkey.rs:132-155
template <class M>
FF_HD Fr w_delta_xor_and(const Fr &a, const Fr &b, const Fr &w, const Fr &c, const Fr &q_c) {
    Fr ab = Fr::add(a, b);
    Fr inner = Fr::add(Fr::sub(mul_small<4>(w), mul_small<18>(ab)), Fr::from_u32(81));
    Fr F = Fr::add(M::mul(w, inner), mul_small<18>(Fr::add(M::mul(a, a), M::mul(b, b))));
    F = Fr::add(Fr::sub(F, mul_small<81>(ab)), Fr::from_u32(83));
    F = M::mul(w, F);
    Fr E = Fr::sub(mul_small<3>(Fr::add(ab, c)), Fr::dbl(F));
    Fr B = M::mul(q_c, Fr::sub(mul_small<9>(c), mul_small<3>(ab)));
    return Fr::add(B, E);
}

// logic/proverkey.rs:51-77 and :88-118: c_0 + ... + c_4
template <class M>
FF_HD Fr w_logic_identity(const SepPowers &k, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c, const Fr &d,
                          const Fr &d_w, const Fr &q_c) {
    const Fr &kappa = k.k1, &kappa_sq = k.k2, &kappa_cu = k.k3, &kappa_qu = k.k4;
    Fr ai = Fr::sub(a_w, mul_small<4>(a)), bi = Fr::sub(b_w, mul_small<4>(b)), di = Fr::sub(d_w, mul_small<4>(d));
    Fr c0 = w_delta<M>(ai);
    Fr c1 = M::mul(w_delta<M>(bi), kappa);
    Fr c2 = M::mul(w_delta<M>(di), kappa_sq);
    Fr c3 = M::mul(Fr::sub(c, M::mul(ai, bi)), kappa_cu);
    Fr c4 = M::mul(w_delta_xor_and<M>(ai, bi, c, di, q_c), kappa_qu);
    return Fr::add(Fr::add(Fr::add(Fr::add(c3, c0), c1), c2), c4);
}

// ecc/scalar_mul/fixed_base/proverkey.rs:53-98 and :110-154
template <class M>
FF_HD Fr w_fixed_base_identity(const SepPowers &k, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c,
                               const Fr &d, const Fr &d_w, const Fr &q_l, const Fr &q_r, const Fr &q_c) {
    const Fr one = Fr::one();
    const Fr &kappa = k.k1, &kappa_sq = k.k2, &kappa_cu = k.k3;
    Fr bit = Fr::sub(Fr::sub(d_w, d), d);                                            // extract_bit :160-162
    Fr bit_cons = M::mul(M::mul(bit, Fr::sub(bit, one)), Fr::add(bit, one));          // :164-167
    Fr y_alpha = Fr::add(M::mul(M::mul(bit, bit), Fr::sub(q_r, one)), one);
    Fr x_alpha = M::mul(bit, q_l);
    Fr xy_cons = M::mul(Fr::sub(M::mul(bit, q_c), c), kappa);
    Fr dxy = M::mul(M::mul(M::mul(c, a), b), edwards_d<M>());
    Fr x_lhs = Fr::add(a_w, M::mul(a_w, dxy));
    Fr x_rhs = Fr::add(M::mul(a, y_alpha), M::mul(b, x_alpha));
    Fr x_acc = M::mul(Fr::sub(x_lhs, x_rhs), kappa_sq);
    Fr y_lhs = Fr::sub(b_w, M::mul(b_w, dxy));
    Fr y_rhs = Fr::add(M::mul(b, y_alpha), M::mul(a, x_alpha));
    Fr y_acc = M::mul(Fr::sub(y_lhs, y_rhs), kappa_cu);
    return Fr::add(Fr::add(Fr::add(bit_cons, x_acc), y_acc), xy_cons);
}

// ecc/curve_addition/proverkey.rs:46-80 and :92-124 ; x1 = a, x3 = a_w, y1 = b, y3 = b_w, x2 = c, y2 = d, x1*y2 = d_w
template <class M>
FF_HD Fr w_curve_add_identity(const SepPowers &k, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c,
                              const Fr &d, const Fr &d_w) {
    const Fr &kappa = k.k1;
    Fr xy_cons = Fr::sub(M::mul(a, d), d_w);
    Fr y1x2 = M::mul(b, c), y1y2 = M::mul(b, d), x1x2 = M::mul(a, c);
    Fr dd = M::mul(M::mul(edwards_d<M>(), d_w), y1x2);
    Fr x3c = M::mul(Fr::sub(Fr::add(d_w, y1x2), Fr::add(a_w, M::mul(a_w, dd))), kappa);
    Fr y3c = M::mul(Fr::sub(Fr::add(y1y2, x1x2), Fr::sub(b_w, M::mul(b_w, dd))), k.k2);
    return Fr::add(Fr::add(xy_cons, x3c), y3c);
}

}  // namespace zk
