// TEST INFRASTRUCTURE ONLY (see util.hpp).
// Restates plonk/src/fft/domain.rs (EvaluationDomain, serial radix-2 FFT) and
// plonk/src/util.rs:31-41,64-93 (powers_of, batch_inversion).
#pragma once
#include "fr.hpp"
#include <vector>
#include <stdexcept>

namespace oracle {

typedef std::vector<Fr> FrVec;

// util.rs:31-41 — [1, s, s^2, ..., s^max_degree]
static inline FrVec powers_of(const Fr &s, size_t max_degree) {
    FrVec p;
    p.reserve(max_degree + 1);
    p.push_back(fr_one());
    for (size_t i = 1; i <= max_degree; i++) p.push_back(fr_mul(p[i - 1], s));
    return p;
}

// util.rs:64-93 — zeros are skipped and stay zero
static inline void batch_inversion(Fr *v, size_t n) {
    FrVec prefix;
    prefix.reserve(n);
    Fr running = fr_one();
    for (size_t i = 0; i < n; i++)
        if (!fr_is_zero(v[i])) {
            running = fr_mul(running, v[i]);
            prefix.push_back(running);
        }
    running = fr_inv_unwrap(running);
    size_t k = prefix.size();
    for (size_t i = n; i-- > 0;) {
        if (fr_is_zero(v[i])) continue;
        Fr pp = (k >= 2) ? prefix[k - 2] : fr_one();
        k--;
        Fr next = fr_mul(running, v[i]);
        v[i] = fr_mul(running, pp);
        running = next;
    }
}

// domain.rs:291-298
static inline uint32_t bitreverse(uint32_t n, uint32_t l) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < l; i++) {
        r = (r << 1) | (n & 1);
        n >>= 1;
    }
    return r;
}

// domain.rs:300-340 — in-place, natural order in and out
static inline void serial_fft(Fr *a, size_t len, const Fr &omega, uint32_t log_n) {
    uint32_t n = (uint32_t)len;
    if (n != (1u << log_n)) throw std::runtime_error("serial_fft: length is not 2^log_n");
    for (uint32_t k = 0; k < n; k++) {
        uint32_t rk = bitreverse(k, log_n);
        if (k < rk) {
            Fr t = a[rk];
            a[rk] = a[k];
            a[k] = t;
        }
    }
    uint32_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        Fr root_step = fr_pow_u64(omega, (uint64_t)(n / (2 * m)));
        for (uint32_t start = 0; start < n; start += 2 * m) {
            Fr w = fr_one();
            for (uint32_t off = 0; off < m; off++) {
                Fr right = fr_mul(a[start + off + m], w);
                Fr left = a[start + off];
                a[start + off + m] = fr_sub(left, right);
                a[start + off] = fr_add(left, right);
                w = fr_mul(w, root_step);
            }
        }
        m *= 2;
    }
}

struct EvaluationDomain {  // domain.rs:23-44
    uint64_t size;
    uint32_t log_size_of_group;
    Fr size_as_field_element, size_inv, group_gen, group_gen_inv, generator_inv;

    // domain.rs:106-133 ; throws where the reference returns Err(InvalidEvalDomainSize)
    explicit EvaluationDomain(size_t num_coeffs) {
        uint64_t s = 1;
        while (s < num_coeffs) s <<= 1;  // next_power_of_two (0 and 1 -> 1)
        size = s;
        log_size_of_group = (uint32_t)__builtin_ctzll(s);
        if (log_size_of_group >= FR_TWO_ADACITY) throw std::runtime_error("InvalidEvalDomainSize");
        group_gen = FR_ROOT_OF_UNITY;
        for (uint32_t i = log_size_of_group; i < FR_TWO_ADACITY; i++) group_gen = fr_square(group_gen);
        size_as_field_element = fr_from_u64(size);
        size_inv = fr_inv_unwrap(size_as_field_element);
        group_gen_inv = fr_inv_unwrap(group_gen);
        generator_inv = fr_inv_unwrap(FR_GENERATOR);
    }

    // domain.rs:139-148 — resize (zero-pad or TRUNCATE) to the domain size, then transform
    FrVec fft(const FrVec &coeffs) const {
        FrVec v = coeffs;
        v.resize(size, fr_zero());
        serial_fft(v.data(), v.size(), group_gen, log_size_of_group);
        return v;
    }
    // domain.rs:150-166
    FrVec ifft(const FrVec &evals) const {
        FrVec v = evals;
        v.resize(size, fr_zero());
        serial_fft(v.data(), v.size(), group_gen_inv, log_size_of_group);
        for (auto &x : v) x = fr_mul(x, size_inv);
        return v;
    }
    // domain.rs:168-174
    static void distribute_powers(FrVec &c, const Fr &g) {
        Fr pow = fr_one();
        for (auto &x : c) {
            x = fr_mul(x, pow);
            pow = fr_mul(pow, g);
        }
    }
    // domain.rs:176-185 — powers are distributed over the UN-padded coefficients
    FrVec coset_fft(const FrVec &coeffs) const {
        FrVec v = coeffs;
        distribute_powers(v, FR_GENERATOR);
        return fft(v);
    }
    // domain.rs:187-196
    FrVec coset_ifft(const FrVec &evals) const {
        FrVec v = ifft(evals);
        distribute_powers(v, generator_inv);
        return v;
    }
    // domain.rs:343-361 (Elements iterator)
    FrVec elements() const {
        FrVec e;
        e.reserve(size);
        Fr cur = fr_one();
        for (uint64_t i = 0; i < size; i++) {
            e.push_back(cur);
            cur = fr_mul(cur, group_gen);
        }
        return e;
    }
    // domain.rs:249-254
    Fr evaluate_vanishing_polynomial(const Fr &tau) const { return fr_sub(fr_pow_u64(tau, size), fr_one()); }
    // domain.rs:256-275
    FrVec compute_vanishing_poly_over_coset(uint64_t poly_degree) const {
        if (!(size > poly_degree)) throw std::runtime_error("vanishing poly: domain too small");
        Fr coset_gen = fr_pow_u64(FR_GENERATOR, poly_degree);
        FrVec v(size);
        for (uint64_t i = 0; i < size; i++)
            v[i] = fr_sub(fr_mul(coset_gen, fr_pow_u64(group_gen, poly_degree * i)), fr_one());
        return v;
    }
    // domain.rs:200-246
    FrVec evaluate_all_lagrange_coefficients(const Fr &tau) const {
        Fr t_size = fr_pow_u64(tau, size);
        Fr one = fr_one();
        FrVec u(size, fr_zero());
        if (t_size == one) {
            Fr omega_i = one;
            for (uint64_t i = 0; i < size; i++) {
                if (omega_i == tau) {
                    u[i] = one;
                    break;
                }
                omega_i = fr_mul(omega_i, group_gen);
            }
            return u;
        }
        Fr l = fr_mul(fr_sub(t_size, one), size_inv);
        Fr r = one;
        FrVec ls(size);
        for (uint64_t i = 0; i < size; i++) {
            u[i] = fr_sub(tau, r);
            ls[i] = l;
            l = fr_mul(l, group_gen);
            r = fr_mul(r, group_gen);
        }
        batch_inversion(u.data(), u.size());
        for (uint64_t i = 0; i < size; i++) u[i] = fr_mul(ls[i], u[i]);
        return u;
    }
};

}  // namespace oracle
