// TEST INFRASTRUCTURE ONLY (see util.hpp).
// Restates plonk/src/fft/polynomial.rs: coefficient-form polynomial whose leading zeros are
// truncated by from_coefficients_vec and by every add/sub, plus evaluate and ruffini.
#pragma once
#include "domain.hpp"

namespace oracle {

struct Poly {
    FrVec c;

    size_t len() const { return c.size(); }
    // polynomial.rs:61-64
    bool is_zero() const {
        for (auto &x : c)
            if (!fr_is_zero(x)) return false;
        return true;
    }
    // polynomial.rs:95-103
    void truncate_leading_zeros() {
        while (!c.empty() && fr_is_zero(c.back())) c.pop_back();
    }
    // polynomial.rs:66-77
    static Poly from_coefficients_vec(FrVec v) {
        Poly p{std::move(v)};
        p.truncate_leading_zeros();
        return p;
    }
    // polynomial.rs:79-93
    size_t degree() const {
        if (is_zero()) return 0;
        for (size_t i = c.size(); i-- > 0;)
            if (!fr_is_zero(c[i])) return i;
        return 0;
    }
    // polynomial.rs:105-119 — sum_i c_i * x^i accumulated from i = 0 upward
    Fr evaluate(const Fr &x) const {
        if (is_zero()) return fr_zero();
        FrVec pw = powers_of(x, c.size());
        Fr sum = fr_zero();
        for (size_t i = 0; i < c.size(); i++) sum = fr_add(sum, fr_mul(pw[i], c[i]));
        return sum;
    }
    // polynomial.rs:343-357 — synthetic division by (X - root); the remainder is dropped
    Poly ruffini(const Fr &root) const {
        FrVec q;
        q.reserve(c.size());
        Fr running = fr_zero();
        for (size_t i = c.size(); i-- > 0;) {
            Fr u = fr_add(c[i], running);
            q.push_back(u);
            running = fr_mul(root, u);
        }
        if (!q.empty()) q.pop_back();
        FrVec rev(q.rbegin(), q.rend());
        return from_coefficients_vec(std::move(rev));
    }
};

// polynomial.rs:164-193
static inline Poly poly_add(const Poly &a, const Poly &b) {
    Poly r;
    if (a.is_zero()) {
        r = b;
    } else if (b.is_zero()) {
        r = a;
    } else if (a.degree() >= b.degree()) {
        r = a;
        size_t m = r.c.size() < b.c.size() ? r.c.size() : b.c.size();
        for (size_t i = 0; i < m; i++) r.c[i] = fr_add(r.c[i], b.c[i]);
    } else {
        r = b;
        size_t m = r.c.size() < a.c.size() ? r.c.size() : a.c.size();
        for (size_t i = 0; i < m; i++) r.c[i] = fr_add(r.c[i], a.c[i]);
    }
    r.truncate_leading_zeros();
    return r;
}

// polynomial.rs:195-218 (AddAssign<&Polynomial>)
static inline void poly_add_assign(Poly &a, const Poly &b) {
    if (a.is_zero()) {
        a.c = b.c;
    } else if (b.is_zero()) {
    } else if (a.degree() >= b.degree()) {
        size_t m = a.c.size() < b.c.size() ? a.c.size() : b.c.size();
        for (size_t i = 0; i < m; i++) a.c[i] = fr_add(a.c[i], b.c[i]);
    } else {
        a.c.resize(b.c.size(), fr_zero());
        for (size_t i = 0; i < b.c.size(); i++) a.c[i] = fr_add(a.c[i], b.c[i]);
    }
    a.truncate_leading_zeros();
}

// polynomial.rs:385-397 (Mul<&BlsScalar>)
static inline Poly poly_scale(const Poly &a, const Fr &k) {
    if (a.is_zero() || fr_is_zero(k)) return Poly{};
    FrVec v(a.c.size());
    for (size_t i = 0; i < v.size(); i++) v[i] = fr_mul(a.c[i], k);
    return Poly::from_coefficients_vec(std::move(v));
}

// polynomial.rs:399-415 (Add<&BlsScalar>)
static inline Poly poly_add_scalar(const Poly &a, const Fr &k) {
    if (a.is_zero()) return Poly::from_coefficients_vec(FrVec{k});
    Poly r = a;
    if (fr_is_zero(k)) return r;
    r.c[0] = fr_add(r.c[0], k);
    return r;
}

// polynomial.rs:151-162 (Sum)
static inline Poly poly_sum(const std::vector<Poly> &ps) {
    Poly r;
    for (auto &p : ps) r = poly_add(r, p);
    return r;
}

}  // namespace oracle
