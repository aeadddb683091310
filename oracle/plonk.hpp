// TEST INFRASTRUCTURE ONLY (see util.hpp).
// Restates the prover-side protocol code of the plonk crate:
//   widget quotient / linearization terms   plonk/src/proof_system/widget/**/proverkey.rs
//   ProverKey construction                   plonk/src/compiler.rs:78-365 (preprocess)
//   permutation accumulator                  plonk/src/composer/permutation.rs:181-245
//   quotient polynomial                      plonk/src/proof_system/quotient_poly.rs
//   linearization polynomial                 plonk/src/proof_system/linearization_poly.rs:146-272
//   KZG commit / aggregate witness           plonk/src/commitment_scheme/kzg10/key.rs:108-146
//   Prover::prove                            plonk/src/compiler/prover.rs:64-83,210-509
#pragma once
#include "poly.hpp"
#include "msm.hpp"
#include "transcript.hpp"
#include <array>
#include <atomic>
#include <thread>
#include <vector>
#include <string>

namespace oracle {

// plonk/src/composer/permutation/constants.rs:7-9
static inline Fr K1() { return fr_from_u64(7); }
static inline Fr K2() { return fr_from_u64(13); }
static inline Fr K3() { return fr_from_u64(17); }
// coset-jubjub/src/lib.rs:368-373
static inline Fr EDWARDS_D() {
    static const uint64_t raw[4] = {0x01065fd6d6343eb1ULL, 0x292d7f6d37579d26ULL, 0xf5fd9207e6bd7fd4ULL,
                                    0x2a9318e74bfa2b48ULL};
    return fr_from_raw(raw);
}

// Selector / sigma order used everywhere in this file.
enum Sel { Q_M, Q_L, Q_R, Q_O, Q_F, Q_C, Q_ARITH, Q_RANGE, Q_LOGIC, Q_FIXED, Q_VAR, S1, S2, S3, S4, N_SEL };

// linearization_poly.rs:29-74 (field order of the struct)
struct ProofEvaluations {
    Fr a_eval, b_eval, c_eval, d_eval, a_w_eval, b_w_eval, d_w_eval, q_arith_eval, q_c_eval, q_l_eval, q_r_eval,
        s_sigma_1_eval, s_sigma_2_eval, s_sigma_3_eval, z_eval;
};

struct ProverKey {  // widget.rs:238-262 flattened
    size_t n;
    Poly poly[N_SEL];     // coefficient forms
    FrVec eval8n[N_SEL];  // coset evaluations on the 8n domain
    FrVec linear_evaluations, v_h_coset_8n;
};

struct Challenges {
    Fr alpha, beta, gamma, range, logic, fixed_base, var_base;
};

// ---- widgets ----------------------------------------------------------------------------------
// range/proverkey.rs:92-97 and logic/proverkey.rs:124-129
static inline Fr delta(const Fr &f) {
    Fr f1 = fr_sub(f, fr_one()), f2 = fr_sub(f, fr_from_u64(2)), f3 = fr_sub(f, fr_from_u64(3));
    return fr_mul(fr_mul(fr_mul(f, f1), f2), f3);
}
// logic/proverkey.rs:132-155
static inline Fr delta_xor_and(const Fr &a, const Fr &b, const Fr &w, const Fr &c, const Fr &q_c) {
    Fr nine = fr_from_u64(9), two = fr_from_u64(2), three = fr_from_u64(3), four = fr_from_u64(4),
       eighteen = fr_from_u64(18), eighty_one = fr_from_u64(81), eighty_three = fr_from_u64(83);
    Fr ab = fr_add(a, b);
    Fr inner = fr_add(fr_sub(fr_mul(four, w), fr_mul(eighteen, ab)), eighty_one);
    Fr F = fr_mul(w, fr_add(fr_sub(fr_add(fr_mul(w, inner), fr_mul(eighteen, fr_add(fr_square(a), fr_square(b)))),
                                   fr_mul(eighty_one, ab)),
                            eighty_three));
    Fr E = fr_sub(fr_mul(three, fr_add(ab, c)), fr_mul(two, F));
    Fr B = fr_mul(q_c, fr_sub(fr_mul(nine, c), fr_mul(three, ab)));
    return fr_add(B, E);
}

// arithmetic/proverkey.rs:43-73
static inline Fr arith_quotient_i(const ProverKey &pk, size_t i, const Fr &a, const Fr &b, const Fr &c, const Fr &d) {
    Fr t = fr_mul(fr_mul(a, b), pk.eval8n[Q_M][i]);
    t = fr_add(t, fr_mul(a, pk.eval8n[Q_L][i]));
    t = fr_add(t, fr_mul(b, pk.eval8n[Q_R][i]));
    t = fr_add(t, fr_mul(c, pk.eval8n[Q_O][i]));
    t = fr_add(t, fr_mul(d, pk.eval8n[Q_F][i]));
    t = fr_add(t, pk.eval8n[Q_C][i]);
    return fr_mul(t, pk.eval8n[Q_ARITH][i]);
}
// the scalar part shared by range quotient and linearization (range/proverkey.rs:44-58,69-86)
static inline Fr range_identity(const Fr &sep, const Fr &a, const Fr &b, const Fr &c, const Fr &d, const Fr &d_w) {
    Fr four = fr_from_u64(4);
    Fr kappa = fr_square(sep), kappa_sq = fr_square(kappa), kappa_cu = fr_mul(kappa_sq, kappa);
    Fr t = delta(fr_sub(c, fr_mul(four, d)));
    t = fr_add(t, fr_mul(delta(fr_sub(b, fr_mul(four, c))), kappa));
    t = fr_add(t, fr_mul(delta(fr_sub(a, fr_mul(four, b))), kappa_sq));
    t = fr_add(t, fr_mul(delta(fr_sub(d_w, fr_mul(four, a))), kappa_cu));
    return t;
}
// range/proverkey.rs:31-59
static inline Fr range_quotient_i(const ProverKey &pk, size_t i, const Fr &sep, const Fr &a, const Fr &b, const Fr &c,
                                  const Fr &d, const Fr &d_w) {
    return fr_mul(fr_mul(range_identity(sep, a, b, c, d, d_w), pk.eval8n[Q_RANGE][i]), sep);
}
// logic/proverkey.rs:51-77,88-118 (the sum c_0..c_4)
static inline Fr logic_identity(const Fr &sep, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c,
                                const Fr &d, const Fr &d_w, const Fr &q_c) {
    Fr four = fr_from_u64(4);
    Fr kappa = fr_square(sep), kappa_sq = fr_square(kappa), kappa_cu = fr_mul(kappa_sq, kappa),
       kappa_qu = fr_mul(kappa_cu, kappa);
    Fr a_in = fr_sub(a_w, fr_mul(four, a));
    Fr c_0 = delta(a_in);
    Fr b_in = fr_sub(b_w, fr_mul(four, b));
    Fr c_1 = fr_mul(delta(b_in), kappa);
    Fr d_in = fr_sub(d_w, fr_mul(four, d));
    Fr c_2 = fr_mul(delta(d_in), kappa_sq);
    Fr c_3 = fr_mul(fr_sub(c, fr_mul(a_in, b_in)), kappa_cu);
    Fr c_4 = fr_mul(delta_xor_and(a_in, b_in, c, d_in, q_c), kappa_qu);
    return fr_add(fr_add(fr_add(fr_add(c_3, c_0), c_1), c_2), c_4);
}
// logic/proverkey.rs:34-78
static inline Fr logic_quotient_i(const ProverKey &pk, size_t i, const Fr &sep, const Fr &a, const Fr &a_w, const Fr &b,
                                  const Fr &b_w, const Fr &c, const Fr &d, const Fr &d_w) {
    return fr_mul(fr_mul(pk.eval8n[Q_LOGIC][i], logic_identity(sep, a, a_w, b, b_w, c, d, d_w, pk.eval8n[Q_C][i])), sep);
}
// fixed_base/proverkey.rs:53-98,110-154 (bit/x/y/xy consistency sum)
static inline Fr fixed_base_identity(const Fr &sep, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c,
                                     const Fr &d, const Fr &d_w, const Fr &q_l, const Fr &q_r, const Fr &q_c) {
    Fr one = fr_one();
    Fr kappa = fr_square(sep), kappa_sq = fr_square(kappa), kappa_cu = fr_mul(kappa_sq, kappa);
    const Fr &x_beta = q_l, &y_beta = q_r, &acc_x = a, &acc_x_w = a_w, &acc_y = b, &acc_y_w = b_w, &xy_alpha = c;
    Fr bit = fr_sub(fr_sub(d_w, d), d);  // extract_bit :160-162
    Fr bit_consistency = fr_mul(fr_mul(bit, fr_sub(bit, one)), fr_add(bit, one));  // :164-167
    Fr y_alpha = fr_add(fr_mul(fr_square(bit), fr_sub(y_beta, one)), one);
    Fr x_alpha = fr_mul(bit, x_beta);
    Fr xy_consistency = fr_mul(fr_sub(fr_mul(bit, q_c), xy_alpha), kappa);
    Fr dxy = fr_mul(fr_mul(fr_mul(xy_alpha, acc_x), acc_y), EDWARDS_D());
    Fr x_lhs = fr_add(acc_x_w, fr_mul(acc_x_w, dxy));
    Fr x_rhs = fr_add(fr_mul(acc_x, y_alpha), fr_mul(acc_y, x_alpha));
    Fr x_acc = fr_mul(fr_sub(x_lhs, x_rhs), kappa_sq);
    Fr y_lhs = fr_sub(acc_y_w, fr_mul(acc_y_w, dxy));
    Fr y_rhs = fr_add(fr_mul(acc_y, y_alpha), fr_mul(acc_x, x_alpha));
    Fr y_acc = fr_mul(fr_sub(y_lhs, y_rhs), kappa_cu);
    return fr_add(fr_add(fr_add(bit_consistency, x_acc), y_acc), xy_consistency);
}
// fixed_base/proverkey.rs:38-101
static inline Fr fixed_base_quotient_i(const ProverKey &pk, size_t i, const Fr &sep, const Fr &a, const Fr &a_w,
                                       const Fr &b, const Fr &b_w, const Fr &c, const Fr &d, const Fr &d_w) {
    Fr id = fixed_base_identity(sep, a, a_w, b, b_w, c, d, d_w, pk.eval8n[Q_L][i], pk.eval8n[Q_R][i], pk.eval8n[Q_C][i]);
    return fr_mul(fr_mul(id, pk.eval8n[Q_FIXED][i]), sep);
}
// curve_addition/proverkey.rs:46-80,92-124
static inline Fr curve_add_identity(const Fr &sep, const Fr &a, const Fr &a_w, const Fr &b, const Fr &b_w, const Fr &c,
                                    const Fr &d, const Fr &d_w) {
    Fr kappa = fr_square(sep);
    const Fr &x1 = a, &x3 = a_w, &y1 = b, &y3 = b_w, &x2 = c, &y2 = d, &x1y2 = d_w;
    Fr xy_consistency = fr_sub(fr_mul(x1, y2), x1y2);
    Fr y1x2 = fr_mul(y1, x2), y1y2 = fr_mul(y1, y2), x1x2 = fr_mul(x1, x2);
    Fr dd = fr_mul(fr_mul(EDWARDS_D(), x1y2), y1x2);
    Fr x3_lhs = fr_add(x1y2, y1x2);
    Fr x3_rhs = fr_add(x3, fr_mul(x3, dd));
    Fr x3_c = fr_mul(fr_sub(x3_lhs, x3_rhs), kappa);
    Fr y3_lhs = fr_add(y1y2, x1x2);
    Fr y3_rhs = fr_sub(y3, fr_mul(y3, dd));
    Fr y3_c = fr_mul(fr_sub(y3_lhs, y3_rhs), fr_square(kappa));
    return fr_add(fr_add(xy_consistency, x3_c), y3_c);
}
// curve_addition/proverkey.rs:32-83
static inline Fr curve_add_quotient_i(const ProverKey &pk, size_t i, const Fr &sep, const Fr &a, const Fr &a_w,
                                      const Fr &b, const Fr &b_w, const Fr &c, const Fr &d, const Fr &d_w) {
    return fr_mul(fr_mul(curve_add_identity(sep, a, a_w, b, b_w, c, d, d_w), pk.eval8n[Q_VAR][i]), sep);
}
// permutation/proverkey.rs:39-119
static inline Fr perm_quotient_i(const ProverKey &pk, size_t i, const Fr &a, const Fr &b, const Fr &c, const Fr &d,
                                 const Fr &z, const Fr &z_w, const Fr &alpha, const Fr &l1_alpha_sq, const Fr &beta,
                                 const Fr &gamma) {
    const Fr &x = pk.linear_evaluations[i];
    Fr id = fr_add(fr_add(a, fr_mul(beta, x)), gamma);
    id = fr_mul(id, fr_add(fr_add(b, fr_mul(fr_mul(beta, K1()), x)), gamma));
    id = fr_mul(id, fr_add(fr_add(c, fr_mul(fr_mul(beta, K2()), x)), gamma));
    id = fr_mul(id, fr_add(fr_add(d, fr_mul(fr_mul(beta, K3()), x)), gamma));
    id = fr_mul(fr_mul(id, z), alpha);
    Fr cp = fr_add(fr_add(a, fr_mul(beta, pk.eval8n[S1][i])), gamma);
    cp = fr_mul(cp, fr_add(fr_add(b, fr_mul(beta, pk.eval8n[S2][i])), gamma));
    cp = fr_mul(cp, fr_add(fr_add(c, fr_mul(beta, pk.eval8n[S3][i])), gamma));
    cp = fr_mul(cp, fr_add(fr_add(d, fr_mul(beta, pk.eval8n[S4][i])), gamma));
    cp = fr_neg(fr_mul(fr_mul(cp, z_w), alpha));
    Fr one_check = fr_mul(fr_sub(z, fr_one()), l1_alpha_sq);
    return fr_add(fr_add(id, cp), one_check);
}

// ---- preprocess: compiler.rs:89-147 (selector iFFTs), 239-306 (8n coset evaluations), 353-354 ----
// `selector_evals[k]` holds the n gate-wise values of selector k (S1..S4 = sigma values, i.e. the
// output of build_permutation_lagrange_mapping, permutation.rs:111-141).
// The selectors are independent of each other: with n_threads > 1 they are transformed side by side (the values do
// not depend on it; the reference runs them one after the other).
static inline ProverKey preprocess(size_t n, const FrVec selector_evals[N_SEL], unsigned n_threads = 1) {
    ProverKey pk;
    EvaluationDomain domain(n - 1 ? n - 1 : 1);
    pk.n = domain.size;
    EvaluationDomain domain_8n(8 * domain.size);
    auto one = [&](int k) {
        if (k < N_SEL) {
            pk.poly[k] = Poly::from_coefficients_vec(domain.ifft(selector_evals[k]));
            pk.eval8n[k] = domain_8n.coset_fft(pk.poly[k].c);
        } else if (k == N_SEL) {
            pk.linear_evaluations = domain_8n.coset_fft(FrVec{fr_zero(), fr_one()});
        } else {
            pk.v_h_coset_8n = domain_8n.compute_vanishing_poly_over_coset(domain.size);
        }
    };
    const int jobs = N_SEL + 2;
    if (n_threads <= 1) {
        for (int k = 0; k < jobs; k++) one(k);
    } else {
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads && t < (unsigned)jobs; t++)
            pool.emplace_back([&] {
                for (int k = next++; k < jobs; k = next++) one(k);
            });
        for (auto &th : pool) th.join();
    }
    return pk;
}

// permutation.rs:111-141 — sigma[w][i] encodes (wire column 0..3, gate index) as col * n + index
static inline FrVec sigma_to_lagrange(const uint64_t *sigma_map, size_t n, const EvaluationDomain &domain) {
    FrVec roots = domain.elements();
    Fr k[4] = {fr_one(), K1(), K2(), K3()};
    FrVec out(n);
    for (size_t i = 0; i < n; i++) {
        size_t col = sigma_map[i] / n, idx = sigma_map[i] % n;
        out[i] = col == 0 ? roots[idx] : fr_mul(k[col], roots[idx]);
    }
    return out;
}

// ---- permutation.rs:181-245 ---------------------------------------------------------------------
static inline FrVec compute_permutation_vec(const EvaluationDomain &domain, const FrVec wires[4], const Fr &beta,
                                            const Fr &gamma, const Poly *sigma_polys[4]) {
    size_t n = domain.size;
    Fr ks[4] = {fr_one(), K1(), K2(), K3()};
    FrVec sig[4];
    for (int w = 0; w < 4; w++) sig[w] = domain.fft(sigma_polys[w]->c);
    FrVec roots = domain.elements();
    // izip! stops at the shortest input; wires may be shorter than n only if the caller passed fewer.
    size_t m = n;
    for (int w = 0; w < 4; w++) m = wires[w].size() < m ? wires[w].size() : m;
    FrVec acc;
    acc.reserve(n + 1);
    Fr state = fr_one();
    acc.push_back(state);
    for (size_t i = 0; i < m; i++) {
        Fr num = fr_one(), den = fr_one();
        for (int w = 0; w < 4; w++) {
            num = fr_mul(num, fr_add(fr_add(wires[w][i], fr_mul(fr_mul(beta, ks[w]), roots[i])), gamma));
            den = fr_mul(den, fr_add(fr_add(wires[w][i], fr_mul(beta, sig[w][i])), gamma));
        }
        state = fr_mul(state, fr_mul(num, fr_inv_unwrap(den)));
        acc.push_back(state);
    }
    acc.erase(acc.begin() + n);  // permutation.rs:240
    return acc;
}

// ---- quotient_poly.rs ------------------------------------------------------------------------------
// quotient_poly.rs:237-245
static inline Poly build_scaled_first_lagrange_poly(const EvaluationDomain &domain, const Fr &scale) {
    FrVec x(domain.size, fr_zero());
    x[0] = scale;
    return Poly::from_coefficients_vec(domain.ifft(x));
}

// quotient_poly.rs:15-100. n_threads only splits the three pointwise maps (rayon in the reference).
static inline Poly build_quotient_polynomial(const EvaluationDomain &domain, const ProverKey &pk, const Poly &z_poly,
                                             const Poly &a_poly, const Poly &b_poly, const Poly &c_poly,
                                             const Poly &d_poly, const Poly &pi_poly, const Challenges &ch,
                                             unsigned n_threads = 1) {
    EvaluationDomain d8(8 * domain.size);
    size_t N = d8.size;
    FrVec z = d8.coset_fft(z_poly.c), a = d8.coset_fft(a_poly.c), b = d8.coset_fft(b_poly.c), c = d8.coset_fft(c_poly.c),
          d = d8.coset_fft(d_poly.c);
    for (int i = 0; i < 8; i++) {  // :53-59
        z.push_back(z[i]);
        a.push_back(a[i]);
        b.push_back(b[i]);
        d.push_back(d[i]);
    }
    FrVec pi = d8.coset_fft(pi_poly.c);                                                                       // :120
    FrVec l1 = d8.coset_fft(build_scaled_first_lagrange_poly(domain, fr_square(ch.alpha)).c);  // :208-210
    FrVec quotient(N);
    auto body = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            // :128-190
            Fr t = arith_quotient_i(pk, i, a[i], b[i], c[i], d[i]);
            t = fr_add(t, range_quotient_i(pk, i, ch.range, a[i], b[i], c[i], d[i], d[i + 8]));
            t = fr_add(t, logic_quotient_i(pk, i, ch.logic, a[i], a[i + 8], b[i], b[i + 8], c[i], d[i], d[i + 8]));
            t = fr_add(t, fixed_base_quotient_i(pk, i, ch.fixed_base, a[i], a[i + 8], b[i], b[i + 8], c[i], d[i], d[i + 8]));
            t = fr_add(t, curve_add_quotient_i(pk, i, ch.var_base, a[i], a[i + 8], b[i], b[i + 8], c[i], d[i], d[i + 8]));
            t = fr_add(t, pi[i]);
            // :218-234
            Fr p = perm_quotient_i(pk, i, a[i], b[i], c[i], d[i], z[i], z[i + 8], ch.alpha, l1[i], ch.beta, ch.gamma);
            // :88-95
            Fr numerator = fr_add(t, p);
            quotient[i] = fr_mul(numerator, fr_inv_unwrap(pk.v_h_coset_8n[i]));
        }
    };
    if (n_threads <= 1) {
        body(0, N);
    } else {
        std::vector<std::thread> pool;
        size_t chunk = (N + n_threads - 1) / n_threads;
        for (unsigned t = 0; t < n_threads; t++) {
            size_t lo = t * chunk, hi = lo + chunk < N ? lo + chunk : N;
            if (lo < hi) pool.emplace_back(body, lo, hi);
        }
        for (auto &th : pool) th.join();
    }
    return Poly::from_coefficients_vec(d8.coset_ifft(quotient));  // :97-99
}

// ---- linearization_poly.rs:146-272 ----------------------------------------------------------------------
// proof.rs:511-561
static inline Fr compute_barycentric_eval(const FrVec &evaluations, const Fr &point, const EvaluationDomain &domain) {
    Fr numerator = fr_mul(fr_sub(fr_pow_u64(point, domain.size), fr_one()), domain.size_inv);
    std::vector<size_t> nz;
    for (size_t i = 0; i < evaluations.size(); i++)
        if (!fr_is_zero(evaluations[i])) nz.push_back(i);
    FrVec den(nz.size());
    for (size_t k = 0; k < nz.size(); k++)
        den[k] = fr_sub(fr_mul(fr_pow_u64(domain.group_gen_inv, nz[k]), point), fr_one());
    batch_inversion(den.data(), den.size());
    Fr result = fr_zero();
    for (size_t k = 0; k < nz.size(); k++) result = fr_add(result, fr_mul(den[k], evaluations[nz[k]]));
    return fr_mul(result, numerator);
}

static inline Poly build_linearization_polynomial(const ProverKey &pk, const Challenges &ch, const Fr &z_challenge,
                                                  const Poly &z_poly, const ProofEvaluations &ev,
                                                  const EvaluationDomain &domain, const Poly &t_low, const Poly &t_mid,
                                                  const Poly &t_high, const Poly &t_fourth, const FrVec &pub_inputs) {
    // arithmetic/proverkey.rs:75-107
    Poly lin = poly_add(poly_scale(pk.poly[Q_M], fr_mul(ev.a_eval, ev.b_eval)), poly_scale(pk.poly[Q_L], ev.a_eval));
    lin = poly_add(lin, poly_scale(pk.poly[Q_R], ev.b_eval));
    lin = poly_add(lin, poly_scale(pk.poly[Q_O], ev.c_eval));
    lin = poly_add(lin, poly_scale(pk.poly[Q_F], ev.d_eval));
    lin = poly_add(lin, pk.poly[Q_C]);
    Poly arith = poly_scale(lin, ev.q_arith_eval);
    // range/proverkey.rs:61-89
    Poly range = poly_scale(pk.poly[Q_RANGE],
                            fr_mul(range_identity(ch.range, ev.a_eval, ev.b_eval, ev.c_eval, ev.d_eval, ev.d_w_eval), ch.range));
    // logic/proverkey.rs:80-121
    Poly logic = poly_scale(pk.poly[Q_LOGIC], fr_mul(logic_identity(ch.logic, ev.a_eval, ev.a_w_eval, ev.b_eval, ev.b_w_eval,
                                                                    ev.c_eval, ev.d_eval, ev.d_w_eval, ev.q_c_eval),
                                                     ch.logic));
    // fixed_base/proverkey.rs:103-157
    Poly fixed = poly_scale(pk.poly[Q_FIXED],
                            fr_mul(fixed_base_identity(ch.fixed_base, ev.a_eval, ev.a_w_eval, ev.b_eval, ev.b_w_eval, ev.c_eval,
                                                       ev.d_eval, ev.d_w_eval, ev.q_l_eval, ev.q_r_eval, ev.q_c_eval),
                                   ch.fixed_base));
    // curve_addition/proverkey.rs:85-127
    Poly var = poly_scale(pk.poly[Q_VAR], fr_mul(curve_add_identity(ch.var_base, ev.a_eval, ev.a_w_eval, ev.b_eval, ev.b_w_eval,
                                                                    ev.c_eval, ev.d_eval, ev.d_w_eval),
                                                 ch.var_base));
    // linearization_poly.rs:266-271
    Poly circuit = poly_add(arith, range);
    poly_add_assign(circuit, logic);
    poly_add_assign(circuit, fixed);
    poly_add_assign(circuit, var);
    // :187-190 — NOTE: the reference passes the sparse public-input list here, not the dense vector
    Fr pi_eval = compute_barycentric_eval(pub_inputs, z_challenge, domain);
    circuit = poly_add_scalar(circuit, pi_eval);

    // permutation/proverkey.rs:121-245
    Fr beta_z = fr_mul(ch.beta, z_challenge);
    Fr acc = fr_add(fr_add(ev.a_eval, beta_z), ch.gamma);
    acc = fr_mul(acc, fr_add(fr_add(ev.b_eval, fr_mul(K1(), beta_z)), ch.gamma));
    acc = fr_mul(acc, fr_add(fr_add(ev.c_eval, fr_mul(K2(), beta_z)), ch.gamma));
    acc = fr_mul(acc, fr_add(fr_add(ev.d_eval, fr_mul(K3(), beta_z)), ch.gamma));
    acc = fr_mul(acc, ch.alpha);
    Poly perm_id = poly_scale(z_poly, acc);
    Fr cacc = fr_add(fr_add(ev.a_eval, fr_mul(ch.beta, ev.s_sigma_1_eval)), ch.gamma);
    cacc = fr_mul(cacc, fr_add(fr_add(ev.b_eval, fr_mul(ch.beta, ev.s_sigma_2_eval)), ch.gamma));
    cacc = fr_mul(cacc, fr_add(fr_add(ev.c_eval, fr_mul(ch.beta, ev.s_sigma_3_eval)), ch.gamma));
    cacc = fr_mul(cacc, fr_mul(ch.beta, ev.z_eval));
    cacc = fr_mul(cacc, ch.alpha);
    Poly perm_copy = poly_scale(pk.poly[S4], fr_neg(cacc));
    EvaluationDomain dz(z_poly.degree() - 2);  // permutation/proverkey.rs:156
    Fr l_1_z = dz.evaluate_all_lagrange_coefficients(z_challenge)[0];
    Poly perm_one = poly_scale(z_poly, fr_mul(l_1_z, fr_square(ch.alpha)));
    Poly perm = poly_add(poly_add(perm_id, perm_copy), perm_one);

    // linearization_poly.rs:211-233
    uint64_t n = domain.size;
    Fr z_n = fr_pow_u64(z_challenge, n), z_2n = fr_pow_u64(z_challenge, 2 * n), z_3n = fr_pow_u64(z_challenge, 3 * n);
    Poly q = poly_add(poly_add(t_low, poly_scale(t_mid, z_n)), poly_scale(t_high, z_2n));
    q = poly_add(q, poly_scale(t_fourth, z_3n));
    Fr z_h_eval = fr_neg(domain.evaluate_vanishing_polynomial(z_challenge));
    q = poly_scale(q, z_h_eval);
    return poly_add(poly_add(circuit, perm), q);
}

// ---- kzg10/key.rs ------------------------------------------------------------------------------------
struct CommitKey {
    std::vector<G1Affine> powers_of_g;
    size_t max_degree() const { return powers_of_g.size() - 1; }  // key.rs:84-86
    // key.rs:108-129. Returns 0 ok, 1 PolynomialDegreeIsZero, 2 PolynomialDegreeTooLarge.
    int commit(const Poly &p, G1Affine &out, unsigned n_threads = 1) const {
        size_t deg = p.degree();
        if (deg == 0) return 1;
        if (deg > max_degree()) return 2;
        out = g1_to_affine(
            msm_variable_base(powers_of_g.data(), p.c.data(), p.c.size(), powers_of_g.size(), n_threads));
        return 0;
    }
};

// key.rs:131-146
static inline Poly compute_aggregate_witness(const std::vector<const Poly *> &polys, const Fr &point, const Fr &v) {
    FrVec powers = powers_of(v, polys.size() - 1);
    Poly num;
    for (size_t i = 0; i < polys.size(); i++) num = poly_add(num, poly_scale(*polys[i], powers[i]));
    return num.ruffini(point);
}

// ---- proof_system/proof.rs:29-65 ------------------------------------------------------------------
struct Proof {
    G1Affine a_comm, b_comm, c_comm, d_comm, z_comm, t_low_comm, t_mid_comm, t_high_comm, t_fourth_comm, w_z_chall_comm,
        w_z_chall_w_comm;
    ProofEvaluations evaluations;
};

// widget.rs:199-228 (VerifierKey::seed_transcript) on top of transcript.rs:51-65 (Transcript::base).
// comms are in Sel order; zero selector polynomials commit to the identity (compiler.rs:152-166).
// `vk_n` is VerifierKey.n, which seed_transcript appends last (widget.rs:227); Compiler::preprocess sets it to the
// constraint count (compiler.rs:211-212), a deserialized key carries its own (widget.rs:102-106).
static inline Transcript transcript_base(const std::string &label, const G1Affine comms[N_SEL], uint64_t constraints,
                                         uint64_t vk_n) {
    Transcript t(label);
    t.circuit_domain_sep(constraints);
    static const struct {
        const char *label;
        int sel;
    } order[] = {{"q_m", Q_M},         {"q_l", Q_L},         {"q_r", Q_R},
                 {"q_o", Q_O},         {"q_c", Q_C},         {"q_f", Q_F},
                 {"q_arith", Q_ARITH}, {"q_range", Q_RANGE}, {"q_logic", Q_LOGIC},
                 {"q_variable_group_add", Q_VAR},            {"q_fixed_group_add", Q_FIXED},
                 {"s_sigma_1", S1},    {"s_sigma_2", S2},    {"s_sigma_3", S3},
                 {"s_sigma_4", S1}};  // sic: widget.rs:224-225 appends s_sigma_1 under the label "s_sigma_4"
    for (auto &o : order) t.append_commitment(o.label, comms[o.sel]);
    t.circuit_domain_sep(vk_n);
    return t;
}

struct Prover {  // prover.rs:22-30
    ProverKey pk;
    CommitKey ck;
    G1Affine selector_comms[N_SEL];
    std::string label;
    size_t size, constraints;
    uint64_t vk_n = 0;   // VerifierKey.n; equals `constraints` for a compiled prover (compiler.rs:211-212)
    std::vector<size_t> public_input_indexes;

    // compiler.rs:152-171: unwrap_or_default for the 11 selectors, `?` for the sigmas
    // With n_threads > 1 the fifteen commitments run side by side, one thread each (same points: they do not depend
    // on each other), instead of one after the other with the windows of each spread over the threads.
    int commit_selectors(unsigned n_threads) {
        int rcs[N_SEL];
        if (n_threads <= 1) {
            for (int k = 0; k < N_SEL; k++) rcs[k] = ck.commit(pk.poly[k], selector_comms[k], 1);
        } else {
            std::atomic<int> next(0);
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < n_threads && t < (unsigned)N_SEL; t++)
                pool.emplace_back([&] {
                    for (int k = next++; k < N_SEL; k = next++) rcs[k] = ck.commit(pk.poly[k], selector_comms[k], 1);
                });
            for (auto &th : pool) th.join();
        }
        for (int k = 0; k < N_SEL; k++)
            if (rcs[k] != 0) {
                if (k >= S1) return rcs[k];
                selector_comms[k] = g1a_identity();
            }
        return 0;
    }
};

// prover.rs:64-83; blinders are consumed in draw order
static inline Poly blind_poly(const Fr *&blinders, const FrVec &witnesses, size_t hiding_degree,
                              const EvaluationDomain &domain) {
    FrVec w = domain.ifft(witnesses);
    for (size_t i = 0; i < hiding_degree + 1; i++) {
        Fr b = *blinders++;
        w[i] = fr_sub(w[i], b);
        w.push_back(b);
    }
    return Poly::from_coefficients_vec(std::move(w));
}

// prover.rs:210-509. `wires[k]` has `size` entries (a, b, c, d gate-wise values); `blinders` holds the 14
// scalars the reference draws from its RNG, in draw order: a(2) b(2) c(2) d(2) z(3) b12 b13 b14.
// Returns 0, or the commit error code (1/2) of the first failing commitment.
static inline int prove(const Prover &pr, const FrVec wires[4], const FrVec &public_inputs, const Fr blinders_in[14],
                        Proof &proof, unsigned n_threads = 1, std::vector<Fr> *challenges_out = nullptr) {
    const Fr *bl = blinders_in;
    EvaluationDomain domain(pr.constraints);
    Transcript transcript = transcript_base(pr.label, pr.selector_comms, pr.constraints, pr.vk_n);
    FrVec dense(pr.size, fr_zero());  // composer.rs:1001-1014
    for (size_t i = 0; i < pr.public_input_indexes.size() && i < public_inputs.size(); i++)
        dense[pr.public_input_indexes[i]] = public_inputs[i];
    for (auto &pi : public_inputs) transcript.append_scalar("pi", pi);

    Poly a_poly = blind_poly(bl, wires[0], 1, domain), b_poly = blind_poly(bl, wires[1], 1, domain),
         c_poly = blind_poly(bl, wires[2], 1, domain), d_poly = blind_poly(bl, wires[3], 1, domain);
    int rc;
    if ((rc = pr.ck.commit(a_poly, proof.a_comm, n_threads))) return rc;
    if ((rc = pr.ck.commit(b_poly, proof.b_comm, n_threads))) return rc;
    if ((rc = pr.ck.commit(c_poly, proof.c_comm, n_threads))) return rc;
    if ((rc = pr.ck.commit(d_poly, proof.d_comm, n_threads))) return rc;
    transcript.append_commitment("a_comm", proof.a_comm);
    transcript.append_commitment("b_comm", proof.b_comm);
    transcript.append_commitment("c_comm", proof.c_comm);
    transcript.append_commitment("d_comm", proof.d_comm);

    Challenges ch;
    ch.beta = transcript.challenge_scalar("beta");
    transcript.append_scalar("beta", ch.beta);  // prover.rs:271-272
    ch.gamma = transcript.challenge_scalar("gamma");
    const Poly *sig[4] = {&pr.pk.poly[S1], &pr.pk.poly[S2], &pr.pk.poly[S3], &pr.pk.poly[S4]};
    FrVec perm = compute_permutation_vec(domain, wires, ch.beta, ch.gamma, sig);
    Poly z_poly = blind_poly(bl, perm, 2, domain);
    if ((rc = pr.ck.commit(z_poly, proof.z_comm, n_threads))) return rc;
    transcript.append_commitment("z_comm", proof.z_comm);

    ch.alpha = transcript.challenge_scalar("alpha");
    ch.range = transcript.challenge_scalar("range separation challenge");
    ch.logic = transcript.challenge_scalar("logic separation challenge");
    ch.fixed_base = transcript.challenge_scalar("fixed base separation challenge");
    ch.var_base = transcript.challenge_scalar("variable base separation challenge");

    Poly pi_poly = Poly::from_coefficients_vec(domain.ifft(dense));
    Poly t_poly = build_quotient_polynomial(domain, pr.pk, z_poly, a_poly, b_poly, c_poly, d_poly, pi_poly, ch, n_threads);

    size_t n = domain.size;
    // prover.rs:329-332 — slicing panics in the reference when t_poly is shorter than 3n
    if (t_poly.c.size() < 3 * n) return 3;
    FrVec t_low(t_poly.c.begin(), t_poly.c.begin() + n), t_mid(t_poly.c.begin() + n, t_poly.c.begin() + 2 * n),
        t_high(t_poly.c.begin() + 2 * n, t_poly.c.begin() + 3 * n), t_fourth(t_poly.c.begin() + 3 * n, t_poly.c.end());
    Fr b_12 = *bl++, b_13 = *bl++, b_14 = *bl++;
    if (t_fourth.empty()) return 3;  // t_fourth_vec[0] would panic
    t_low.push_back(b_12);
    t_mid[0] = fr_sub(t_mid[0], b_12);
    t_mid.push_back(b_13);
    t_high[0] = fr_sub(t_high[0], b_13);
    t_high.push_back(b_14);
    t_fourth[0] = fr_sub(t_fourth[0], b_14);
    Poly t_low_poly = Poly::from_coefficients_vec(t_low), t_mid_poly = Poly::from_coefficients_vec(t_mid),
         t_high_poly = Poly::from_coefficients_vec(t_high), t_fourth_poly = Poly::from_coefficients_vec(t_fourth);
    if ((rc = pr.ck.commit(t_low_poly, proof.t_low_comm, n_threads))) return rc;
    if ((rc = pr.ck.commit(t_mid_poly, proof.t_mid_comm, n_threads))) return rc;
    if ((rc = pr.ck.commit(t_high_poly, proof.t_high_comm, n_threads))) return rc;
    if ((rc = pr.ck.commit(t_fourth_poly, proof.t_fourth_comm, n_threads))) return rc;
    transcript.append_commitment("t_low_comm", proof.t_low_comm);
    transcript.append_commitment("t_mid_comm", proof.t_mid_comm);
    transcript.append_commitment("t_high_comm", proof.t_high_comm);
    transcript.append_commitment("t_fourth_comm", proof.t_fourth_comm);

    Fr z_ch = transcript.challenge_scalar("z_challenge");
    Fr z_w = fr_mul(z_ch, domain.group_gen);
    ProofEvaluations &ev = proof.evaluations;
    ev.a_eval = a_poly.evaluate(z_ch);
    ev.b_eval = b_poly.evaluate(z_ch);
    ev.c_eval = c_poly.evaluate(z_ch);
    ev.d_eval = d_poly.evaluate(z_ch);
    ev.s_sigma_1_eval = pr.pk.poly[S1].evaluate(z_ch);
    ev.s_sigma_2_eval = pr.pk.poly[S2].evaluate(z_ch);
    ev.s_sigma_3_eval = pr.pk.poly[S3].evaluate(z_ch);
    ev.z_eval = z_poly.evaluate(z_w);
    transcript.append_scalar("a_eval", ev.a_eval);
    transcript.append_scalar("b_eval", ev.b_eval);
    transcript.append_scalar("c_eval", ev.c_eval);
    transcript.append_scalar("d_eval", ev.d_eval);
    transcript.append_scalar("s_sigma_1_eval", ev.s_sigma_1_eval);
    transcript.append_scalar("s_sigma_2_eval", ev.s_sigma_2_eval);
    transcript.append_scalar("s_sigma_3_eval", ev.s_sigma_3_eval);
    transcript.append_scalar("z_eval", ev.z_eval);
    ev.a_w_eval = a_poly.evaluate(z_w);
    ev.b_w_eval = b_poly.evaluate(z_w);
    ev.d_w_eval = d_poly.evaluate(z_w);
    ev.q_arith_eval = pr.pk.poly[Q_ARITH].evaluate(z_ch);
    ev.q_c_eval = pr.pk.poly[Q_C].evaluate(z_ch);
    ev.q_l_eval = pr.pk.poly[Q_L].evaluate(z_ch);
    ev.q_r_eval = pr.pk.poly[Q_R].evaluate(z_ch);
    transcript.append_scalar("a_w_eval", ev.a_w_eval);
    transcript.append_scalar("b_w_eval", ev.b_w_eval);
    transcript.append_scalar("d_w_eval", ev.d_w_eval);
    transcript.append_scalar("q_arith_eval", ev.q_arith_eval);
    transcript.append_scalar("q_c_eval", ev.q_c_eval);
    transcript.append_scalar("q_l_eval", ev.q_l_eval);
    transcript.append_scalar("q_r_eval", ev.q_r_eval);

    Fr v_ch = transcript.challenge_scalar("v_challenge");
    Poly r_poly = build_linearization_polynomial(pr.pk, ch, z_ch, z_poly, ev, domain, t_low_poly, t_mid_poly, t_high_poly,
                                                 t_fourth_poly, public_inputs);
    Poly agg = compute_aggregate_witness(
        {&r_poly, &a_poly, &b_poly, &c_poly, &d_poly, &pr.pk.poly[S1], &pr.pk.poly[S2], &pr.pk.poly[S3]}, z_ch, v_ch);
    if ((rc = pr.ck.commit(agg, proof.w_z_chall_comm, n_threads))) return rc;
    Fr v_w_ch = transcript.challenge_scalar("v_w_challenge");
    Poly sagg = compute_aggregate_witness({&z_poly, &a_poly, &b_poly, &d_poly}, z_w, v_w_ch);
    if ((rc = pr.ck.commit(sagg, proof.w_z_chall_w_comm, n_threads))) return rc;
    if (challenges_out) *challenges_out = {ch.beta, ch.gamma, ch.alpha, ch.range, ch.logic, ch.fixed_base, ch.var_base, z_ch, v_ch, v_w_ch};
    return 0;
}

// The transcript half of verification: Verifier::verify (compiler/verifier.rs:187-217: the key-seeded transcript, then
// every public input under "pi") followed by Proof::verify's replay of the proof (proof_system/proof.rs:186-255).
// Returns beta gamma alpha range logic fixed var z v v_w u.
static inline std::vector<Fr> verifier_challenges(const std::string &label, const G1Affine comms[N_SEL], uint64_t constraints,
                                                  const FrVec &public_inputs, const Proof &proof) {
    Transcript t = transcript_base(label, comms, constraints, constraints);
    for (auto &pi : public_inputs) t.append_scalar("pi", pi);
    t.append_commitment("a_comm", proof.a_comm);
    t.append_commitment("b_comm", proof.b_comm);
    t.append_commitment("c_comm", proof.c_comm);
    t.append_commitment("d_comm", proof.d_comm);
    std::vector<Fr> out;
    out.push_back(t.challenge_scalar("beta"));
    t.append_scalar("beta", out[0]);
    out.push_back(t.challenge_scalar("gamma"));
    t.append_commitment("z_comm", proof.z_comm);
    for (const char *name : {"alpha", "range separation challenge", "logic separation challenge",
                             "fixed base separation challenge", "variable base separation challenge"})
        out.push_back(t.challenge_scalar(name));
    t.append_commitment("t_low_comm", proof.t_low_comm);
    t.append_commitment("t_mid_comm", proof.t_mid_comm);
    t.append_commitment("t_high_comm", proof.t_high_comm);
    t.append_commitment("t_fourth_comm", proof.t_fourth_comm);
    out.push_back(t.challenge_scalar("z_challenge"));
    const ProofEvaluations &e = proof.evaluations;
    const struct {
        const char *name;
        const Fr *v;
    } order[] = {{"a_eval", &e.a_eval},       {"b_eval", &e.b_eval},
                 {"c_eval", &e.c_eval},       {"d_eval", &e.d_eval},
                 {"s_sigma_1_eval", &e.s_sigma_1_eval}, {"s_sigma_2_eval", &e.s_sigma_2_eval},
                 {"s_sigma_3_eval", &e.s_sigma_3_eval}, {"z_eval", &e.z_eval},
                 {"a_w_eval", &e.a_w_eval},   {"b_w_eval", &e.b_w_eval},
                 {"d_w_eval", &e.d_w_eval},   {"q_arith_eval", &e.q_arith_eval},
                 {"q_c_eval", &e.q_c_eval},   {"q_l_eval", &e.q_l_eval},
                 {"q_r_eval", &e.q_r_eval}};
    for (auto &o : order) t.append_scalar(o.name, *o.v);
    out.push_back(t.challenge_scalar("v_challenge"));
    out.push_back(t.challenge_scalar("v_w_challenge"));
    t.append_commitment("w_z_chall_comm", proof.w_z_chall_comm);
    t.append_commitment("w_z_chall_w_comm", proof.w_z_chall_w_comm);
    out.push_back(t.challenge_scalar("u_challenge"));
    return out;
}

// ---- key serialization (Prover::to_bytes, compiler/prover.rs:85-134) ------------------------------------------
// A byte sink with the semantics of coset-bytes' `Write for &mut [u8]` (bytes/src/serialize.rs:105-118): a write
// that does not fit is dropped as a whole, the cursor stays, and the callers ignore the error.
struct SliceWriter {
    std::vector<uint8_t> &buf;
    size_t pos = 0;
    void write(const uint8_t *d, size_t n) {
        if (n > buf.size() - pos) return;
        if (n) memcpy(buf.data() + pos, d, n);
        pos += n;
    }
    void u64le(uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
        write(b, 8);
    }
};
static inline void put_u64be(std::vector<uint8_t> &out, uint64_t v) {
    for (int i = 7; i >= 0; i--) out.push_back((uint8_t)(v >> (8 * i)));
}
// Polynomial::to_var_bytes (fft/polynomial.rs:121-129): coefficients 0..=degree, canonical little-endian
static inline std::vector<uint8_t> poly_to_var_bytes(const Poly &p) {
    std::vector<uint8_t> out;
    const size_t deg = p.degree();
    for (size_t i = 0; i < p.c.size() && i <= deg; i++) {
        uint8_t b[32];
        fr_to_bytes(p.c[i], b);
        out.insert(out.end(), b, b + 32);
    }
    return out;
}
// EvaluationDomain::to_bytes (fft/domain.rs:46-66): 8 + 4 + 5 * 32 = 172 bytes
static const size_t DOMAIN_BYTES = 172;
static inline void domain_to_bytes(const EvaluationDomain &d, uint8_t out[DOMAIN_BYTES]) {
    for (int i = 0; i < 8; i++) out[i] = (uint8_t)(d.size >> (8 * i));
    for (int i = 0; i < 4; i++) out[8 + i] = (uint8_t)(d.log_size_of_group >> (8 * i));
    const Fr *f[5] = {&d.size_as_field_element, &d.size_inv, &d.group_gen, &d.group_gen_inv, &d.generator_inv};
    for (int k = 0; k < 5; k++) fr_to_bytes(*f[k], out + 12 + 32 * k);
}
// Evaluations::to_var_bytes (fft/evaluations.rs:40-49): the domain, then every value
static inline std::vector<uint8_t> evals_to_var_bytes(const FrVec &e, const EvaluationDomain &d) {
    std::vector<uint8_t> out(DOMAIN_BYTES + 32 * e.size());
    domain_to_bytes(d, out.data());
    for (size_t i = 0; i < e.size(); i++) fr_to_bytes(e[i], out.data() + DOMAIN_BYTES + 32 * i);
    return out;
}
// Order in which ProverKey::to_var_bytes and VerifierKey::to_bytes walk the selectors (widget.rs:82-97,294-365):
// q_logic comes before q_range there.
static const int SERIAL_ORDER[N_SEL] = {Q_M, Q_L, Q_R, Q_O, Q_F, Q_C, Q_ARITH, Q_LOGIC, Q_RANGE, Q_FIXED, Q_VAR, S1, S2, S3, S4};

// ProverKey::to_var_bytes (proof_system/widget.rs:265-372), including its size rule: every polynomial is budgeted
// at q_m's coefficient count, so the buffer ends in zero padding when others are shorter, and pieces that would
// overrun it are dropped.
static inline std::vector<uint8_t> prover_key_to_var_bytes(const ProverKey &pk) {
    const EvaluationDomain d8(8 * pk.n);
    const size_t eval_size = pk.eval8n[Q_M].size() * 32 + DOMAIN_BYTES;
    const size_t size = pk.poly[Q_M].c.size() * 32 * 15 + eval_size * 17 + 8 * 17;
    std::vector<uint8_t> bytes(size, 0);
    SliceWriter w{bytes};
    w.u64le(pk.n);
    w.u64le(eval_size);
    for (int k : SERIAL_ORDER) {
        w.u64le(pk.poly[k].c.size());
        std::vector<uint8_t> pb = poly_to_var_bytes(pk.poly[k]);
        w.write(pb.data(), pb.size());
        std::vector<uint8_t> eb = evals_to_var_bytes(pk.eval8n[k], d8);
        w.write(eb.data(), eb.size());
    }
    std::vector<uint8_t> lb = evals_to_var_bytes(pk.linear_evaluations, d8);
    w.write(lb.data(), lb.size());
    std::vector<uint8_t> vb = evals_to_var_bytes(pk.v_h_coset_8n, d8);
    w.write(vb.data(), vb.size());
    return bytes;
}
// VerifierKey::to_bytes (widget.rs:73-100): n (the constraint count for a compiled key, compiler.rs:211-212), 15
// compressed points
static inline std::vector<uint8_t> verifier_key_to_bytes(uint64_t vk_n, const G1Affine comms[N_SEL]) {
    std::vector<uint8_t> out(8 + 48 * N_SEL);
    for (int i = 0; i < 8; i++) out[i] = (uint8_t)(vk_n >> (8 * i));
    for (int j = 0; j < N_SEL; j++) g1a_to_compressed(comms[SERIAL_ORDER[j]], out.data() + 8 + 48 * j);
    return out;
}
// CommitKey::to_raw_var_bytes (kzg10/key.rs:38-52)
static inline std::vector<uint8_t> commit_key_to_raw_var_bytes(const CommitKey &ck) {
    std::vector<uint8_t> out(8 + 97 * ck.powers_of_g.size());
    const uint64_t len = ck.powers_of_g.size();
    for (int i = 0; i < 8; i++) out[i] = (uint8_t)(len >> (8 * i));
    for (size_t i = 0; i < ck.powers_of_g.size(); i++) g1a_to_raw(ck.powers_of_g[i], out.data() + 8 + 97 * i);
    return out;
}
// Prover::to_bytes (compiler/prover.rs:109-134): six big-endian u64 (label, prover key, commit key and verifier key
// lengths, size, constraints), then the four pieces
static inline std::vector<uint8_t> prover_to_bytes(const Prover &pr) {
    std::vector<uint8_t> pk = prover_key_to_var_bytes(pr.pk), ck = commit_key_to_raw_var_bytes(pr.ck),
                         vk = verifier_key_to_bytes(pr.vk_n, pr.selector_comms);
    std::vector<uint8_t> out;
    out.reserve(48 + pr.label.size() + pk.size() + ck.size() + vk.size());
    put_u64be(out, pr.label.size());
    put_u64be(out, pk.size());
    put_u64be(out, ck.size());
    put_u64be(out, vk.size());
    put_u64be(out, pr.size);
    put_u64be(out, pr.constraints);
    out.insert(out.end(), pr.label.begin(), pr.label.end());
    out.insert(out.end(), pk.begin(), pk.end());
    out.insert(out.end(), ck.begin(), ck.end());
    out.insert(out.end(), vk.begin(), vk.end());
    return out;
}

}  // namespace oracle
