// Device kernels for the prover's polynomial work around the NTT and the MSM: the permutation accumulator,
// the fused quotient evaluation over the 8n coset, evaluation at a point, linear combinations and synthetic
// division.  All pointers are device pointers; everything is asynchronous on the engine's stream.
#pragma once
#include "common.hpp"
#include "widgets.hpp"
#include "ff.hpp"

namespace zk {

// Selector / sigma order shared with the oracle and the C boundary.
enum Sel { Q_M, Q_L, Q_R, Q_O, Q_F, Q_C, Q_ARITH, Q_RANGE, Q_LOGIC, Q_FIXED, Q_VAR, S1, S2, S3, S4, N_SEL };

struct QuotientArgs {
    const Fr *z, *a, *b, *c, *d, *pi;  // 8n coset evaluations of the round polynomials
    const Fr *sel[N_SEL];              // 8n coset evaluations of the selector and sigma polynomials
    const Fr *linear;                  // X on the coset (ProverKey.permutation.linear_evaluations)
    const Fr *l1;                      // L_1 on the coset (unscaled)
    Fr *out;                           // numerator / Z_H
    Fr alpha, alpha_sq, beta, gamma, range_ch, logic_ch, fixed_ch, var_ch;
    Fr vh_inv[8];                      // 1 / Z_H on the coset has period 8
    Fr beta_x32, gamma_x32;            // 32 beta, 32 gamma: filled in by PolyEngine::quotient
    uint32_t n8;                       // 8n, or the 8n / world points this GPU owns when the coset is dealt out
    // With the 8n points shared by `world` GPUs, this GPU owns the points i = first + world * k, k < n8 (world divides
    // 8): the "next row" i + 8 is the local k + rot with rot = 8 / world, and Z_H's period-8 table is read at
    // (first + world * k) mod 8.  One GPU: rot = 8, world = 1, first = 0.
    uint32_t rot, world, first;
    uint32_t widgets;                  // bit 0 range, 1 logic, 2 fixed-base, 3 curve-addition: selector is non-zero
    // Constants of the widget kernels, filled in by PolyEngine::quotient.  Those kernels carry every value x as 32 x
    // (so that a limb product of two of them is again 32 times the Montgomery product), hence the factors 32:
    struct WidgetConsts {
        Fr k32[4];                     // 32 kappa^j, j = 1..4, kappa = the widget's separation challenge squared
        Fr k1024;                      // 1024 kappa: takes a value that is not scaled into a scaled product with kappa
        Fr ch32;                       // 32 times the separation challenge itself
    } wc[4];                           // range, logic, fixed-base, curve addition
    Fr one32, two32, c81_32, c83_32, edwards_d32;   // 32 * {1, 2, 81, 83, the curve's d}
    const Fr *widget_sum;              // sum of the active widgets' terms per point (filled in by PolyEngine::quotient), or null
};

struct EvalJob {
    const Fr *coeffs;
    uint64_t len;
    Fr x;
};

struct LinTerm {
    const Fr *coeffs;
    uint64_t len;
    Fr scalar;
};
static const int MAX_LIN_TERMS = 18;
struct LinArgs {
    LinTerm t[MAX_LIN_TERMS];
    int n_terms;
    Fr constant;  // added to coefficient 0
    Fr *out;
    uint64_t out_len;
};

class PolyEngine {
   public:
    explicit PolyEngine(hipStream_t s) : stream_(s) {}
    ~PolyEngine();

    // z[0] = 1, z[i+1] = z[i] * prod_w (wire_w[i] + beta k_w root^i + gamma) / prod_w (wire_w[i] + beta sigma_w[i] + gamma)
    // (plonk/src/composer/permutation.rs:181-245).  `roots` = w^i for i < n/2 (the rest is -w^(i-n/2)).
    // `closed`, when given, tells whether the product of all n numerators equals that of all n denominators (and is
    // not zero): the accumulator then returns to 1 after the last row, which is what makes the permutation terms of the
    // quotient's numerator vanish on the domain.
    void permutation_z(const Fr *const wires[4], const Fr *const sigma_evals[4], const Fr &beta, const Fr &gamma,
                       const Fr *roots_half, uint32_t n, Fr *z_out, bool *closed = nullptr);
    // Row-wise check of the arithmetic gate (plonk/src/proof_system/widget/arithmetic/proverkey.rs:43-73 on the domain
    // itself): *flag_dev is set to 1 when (a b q_m + a q_l + b q_r + c q_o + d q_4 + q_c) q_arith + PI differs from
    // zero in some row.  wires = 4 x n values, sel_h = the 7 x n values of q_m q_l q_r q_o q_4 q_c q_arith, pi = n dense
    // public inputs.  The caller clears the flag.
    void gate_check(const Fr *wires, const Fr *sel_h, const Fr *pi, uint32_t n, uint32_t *flag_dev);
    // The same for the widget gates (`widgets`: bit 0 range, 1 logic, 2 fixed-base, 3 curve addition): *flag_dev is set
    // to 1 when, in a row whose selector of such a widget is not zero, some part of that widget's identity differs from
    // zero.  sel_h = the 11 x n values of q_m ... q_variable_group_add in the order of Sel.
    void widget_check(const Fr *wires, const Fr *sel_h, uint32_t n, uint32_t widgets, uint32_t *flag_dev);
    // dst[j * count + i] = src[j][first[j] + i] for j < m <= 8, i < count <= 8: a few coefficients of several vectors
    // gathered for one copy to the host
    void gather_windows(uint32_t m, const Fr *const *src, const uint64_t *first, uint32_t count, Fr *dst);
    // dst[i] = src[2 i] for i < count
    void take_even(const Fr *src, uint64_t count, Fr *dst);
    // Pointwise quotient over the 8n coset (plonk/src/proof_system/quotient_poly.rs:88-95,128-190,218-234).
    void quotient(const QuotientArgs &args);
    // Milliseconds the most recent quotient launch took on the device (HIP events recorded around every launch);
    // waits for that launch.  Throws ZKGPU_ERR_NOT_LOADED before the first launch.
    float last_quotient_ms();
    // out[j] = sum_i coeffs_j[i] * x_j^i  (plonk/src/fft/polynomial.rs:105-119), results in device memory
    void evaluate(const EvalJob *jobs, int n_jobs, Fr *out_dev);
    // out[i] = sum_k scalar_k * P_k[i] (+ constant at i = 0)
    void linear_combination(const LinArgs &args);
    // quotient of division by (X - root), remainder dropped (plonk/src/fft/polynomial.rs:343-357); out has len-1 entries
    void ruffini(const Fr *coeffs, uint64_t len, const Fr &root, Fr *out);
    // small edits of single coefficients: p[idx] = p[idx] - v (sub = true) or p[idx] = v
    void patch(Fr *p, const uint64_t *idx, const Fr *vals, const bool *sub, int count);
    // The quotient's split into its four committed pieces with their blinders (plonk/src/compiler/prover.rs:327-351) in
    // one pass: low / mid / high = t[0, n), t[n, 2n), t[2n, 3n) copied out with room for one more coefficient,
    //   low.push(b[0]);  mid[0] -= b[0], mid.push(b[1]);  high[0] -= b[1], high.push(b[2]);  t[3n] -= b[2]
    // (the fourth piece stays where it is, from t + 3n)
    void split_quotient(Fr *t, uint64_t n, Fr *low, Fr *mid, Fr *high, const Fr b[3]);

    // number of coefficients once leading zeros are dropped (Polynomial::truncate_leading_zeros,
    // plonk/src/fft/polynomial.rs:95-103); synchronises the stream
    // dst[0..len) = 0, then dst[idx[k]] = vals[k] for distinct indexes below len (host arrays)
    void zero_and_scatter(Fr *dst, uint64_t len, const uint64_t *idx_host, const Fr *vals_host, uint32_t count);
    // BlsScalar::from_bytes / to_bytes element-wise (coset-bls12_381/src/scalar.rs:256-305): `raw` holds count
    // canonical integers of 32 little-endian bytes; out receives their Montgomery forms followed by zeros up to
    // pad_to.  Returns false when a value is not below the modulus (the reference's Err on such bytes).
    bool from_canonical(const Fr *raw, uint64_t count, Fr *out, uint64_t pad_to);
    void to_canonical(const Fr *in, uint64_t count, Fr *out_raw);
    uint64_t length_without_leading_zeros(const Fr *coeffs, uint64_t len);
    // the same for up to 8 vectors with a single synchronisation
    void lengths_without_leading_zeros(uint32_t m, const Fr *const *coeffs, const uint64_t *lens, uint64_t *out);
    // the same without the wait: the m lengths arrive in host_out (pinned host memory, 8 bytes each) once the stream
    // has run past this call
    void lengths_enqueue(uint32_t m, const Fr *const *coeffs, const uint64_t *lens, unsigned long long *host_out);

    size_t device_bytes() const { return s0_.cap + s1_.cap + s2_.cap + s3_.cap + jobs_.cap + widget_sum_.cap; }

   private:
    hipStream_t stream_;
    DevBuf s0_, s1_, s2_, s3_, jobs_;
    DevBuf widget_sum_;   // per coset point: the range / logic / curve terms of a key that has such gates
    hipEvent_t q_ev_[2] = {nullptr, nullptr};   // around the quotient kernel, created at the first launch
    bool q_recorded_ = false;
};

}  // namespace zk
