// TEST INFRASTRUCTURE ONLY (see util.hpp).
// Restates coset-bls12_381/src/coset/multiscalar_mul.rs:143-232 (msm_variable_base):
// unsigned fixed-window bucket method, one task per window (rayon there, std::thread here).
#pragma once
#include "g1.hpp"
#include <thread>
#include <vector>

namespace oracle {

// multiscalar_mul.rs:225-232
static inline uint32_t msm_log2(size_t value) {
    if (value <= 1) return 0;
    return 64 - (uint32_t)__builtin_clzll((unsigned long long)value);
}
// multiscalar_mul.rs:222-224
static inline size_t ln_without_floats(size_t value) { return (size_t)(msm_log2(value) * 69 / 100); }

// multiscalar_mul.rs:150-154
static inline size_t msm_window_size(size_t n_scalars) { return n_scalars < 32 ? 3 : ln_without_floats(n_scalars) + 2; }

// One window of multiscalar_mul.rs:168-204.
static inline G1Projective msm_window_sum(const G1Affine *points, const Fr *scalars, size_t n, size_t window_start,
                                          size_t window_size) {
    const Fr one = fr_one();
    G1Projective window_sum = g1p_identity();
    std::vector<G1Projective> buckets(((size_t)1 << window_size) - 1, g1p_identity());
    for (size_t i = 0; i < n; i++) {
        const Fr &scalar = scalars[i];
        if (fr_is_zero(scalar)) continue;
        if (scalar == one) {
            if (window_start == 0) window_sum = g1p_add_mixed(window_sum, points[i]);
        } else {
            Fr reduced = fr_reduce(scalar);
            fr_divn(reduced, (uint32_t)window_start);
            uint64_t v = reduced.l[0] % ((uint64_t)1 << window_size);
            if (v != 0) buckets[v - 1] = g1p_add_mixed(buckets[v - 1], points[i]);
        }
    }
    G1Projective running_sum = g1p_identity();
    for (size_t b = buckets.size(); b-- > 0;) {
        running_sum = g1p_add(running_sum, buckets[b]);
        window_sum = g1p_add(window_sum, running_sum);
    }
    return window_sum;
}

// multiscalar_mul.rs:143-220. `n` = min(points.len(), scalars.len()) (the zip at :172-174).
// n_threads = 0 -> one thread per window (the reference's rayon behaviour on a wide machine).
static inline G1Projective msm_variable_base(const G1Affine *points, const Fr *scalars, size_t n_scalars, size_t n_points,
                                             unsigned n_threads = 1) {
    size_t window_size = msm_window_size(n_scalars);  // uses scalars.len(), as the reference does
    size_t n = n_scalars < n_points ? n_scalars : n_points;
    const size_t num_bits = 255;
    std::vector<size_t> window_starts;
    for (size_t s = 0; s < num_bits; s += window_size) window_starts.push_back(s);
    std::vector<G1Projective> window_sums(window_starts.size());

    if (n_threads == 0 || n_threads > window_starts.size()) n_threads = (unsigned)window_starts.size();
    if (n_threads <= 1) {
        for (size_t w = 0; w < window_starts.size(); w++)
            window_sums[w] = msm_window_sum(points, scalars, n, window_starts[w], window_size);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; t++)
            pool.emplace_back([&, t]() {
                for (size_t w = t; w < window_starts.size(); w += n_threads)
                    window_sums[w] = msm_window_sum(points, scalars, n, window_starts[w], window_size);
            });
        for (auto &th : pool) th.join();
    }

    // multiscalar_mul.rs:207-219
    G1Projective lowest = window_sums[0];
    G1Projective total = g1p_identity();
    for (size_t w = window_sums.size(); w-- > 1;) {
        total = g1p_add(total, window_sums[w]);
        for (size_t k = 0; k < window_size; k++) total = g1p_double(total);
    }
    return g1p_add(total, lowest);
}

}  // namespace oracle
