// TEST INFRASTRUCTURE ONLY (see util.hpp).
// C entry points of the CPU oracle, loaded with ctypes by oracle/pyoracle.py.
// Layouts: Fr = 4 x u64 LE limbs (Montgomery), Fq = 6 x u64, affine point = x|y (12 x u64)
// plus a separate infinity byte, projective point = x|y|z (18 x u64).
#include "fr.hpp"
#include "fq.hpp"
#include "g1.hpp"
#include "msm.hpp"
#include "domain.hpp"
#include "plonk.hpp"
#include <cstring>
#include <chrono>

using namespace oracle;

static inline Fr ld_fr(const uint64_t *p) { return Fr{{p[0], p[1], p[2], p[3]}}; }
static inline void st_fr(uint64_t *p, const Fr &a) { memcpy(p, a.l, 32); }
static inline Fq ld_fq(const uint64_t *p) { return Fq{{p[0], p[1], p[2], p[3], p[4], p[5]}}; }
static inline void st_fq(uint64_t *p, const Fq &a) { memcpy(p, a.l, 48); }
static inline G1Affine ld_aff(const uint64_t *xy, uint8_t inf) { return G1Affine{ld_fq(xy), ld_fq(xy + 6), inf}; }
static inline G1Projective ld_proj(const uint64_t *p) { return G1Projective{ld_fq(p), ld_fq(p + 6), ld_fq(p + 12)}; }
static inline void st_proj(uint64_t *p, const G1Projective &a) {
    st_fq(p, a.x);
    st_fq(p + 6, a.y);
    st_fq(p + 12, a.z);
}
static inline FrVec ld_vec(const uint64_t *p, size_t n) {
    FrVec v(n);
    if (n) memcpy(v.data(), p, n * 32);
    return v;
}

extern "C" {

// ---- Fr -------------------------------------------------------------------------------------
// op: 0 add, 1 sub, 2 mul, 3 neg(a), 4 square(a), 5 invert(a) (0 -> 0), 6 reduce(a) [out of Montgomery],
//     7 from_raw(a) [into Montgomery]
void orc_fr_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fr x = ld_fr(a + 4 * i), y = b ? ld_fr(b + 4 * i) : fr_zero(), r;
        switch (op) {
            case 0: r = fr_add(x, y); break;
            case 1: r = fr_sub(x, y); break;
            case 2: r = fr_mul(x, y); break;
            case 3: r = fr_neg(x); break;
            case 4: r = fr_square(x); break;
            case 5: r = fr_inv_unwrap(x); break;
            case 6: r = fr_reduce(x); break;
            default: r = fr_from_raw(x.l); break;
        }
        st_fr(out + 4 * i, r);
    }
}
void orc_fr_pow(const uint64_t *a, const uint64_t *by4, uint64_t *out) { st_fr(out, fr_pow(ld_fr(a), by4)); }
void orc_fr_to_bytes(const uint64_t *a, uint8_t *out32) { fr_to_bytes(ld_fr(a), out32); }
int orc_fr_from_bytes(const uint8_t *in32, uint64_t *out) {
    Fr r;
    bool ok = fr_from_bytes(in32, r);
    st_fr(out, r);
    return ok ? 1 : 0;
}
void orc_fr_from_bytes_wide(const uint8_t *in64, uint64_t *out) { st_fr(out, fr_from_bytes_wide(in64)); }
void orc_fr_from_u512(const uint64_t *w8, uint64_t *out) { st_fr(out, fr_reduce_u512_words(w8)); }
void orc_fr_divn(uint64_t *a, uint32_t n) {
    Fr x = ld_fr(a);
    fr_divn(x, n);
    st_fr(a, x);
}
void orc_batch_inversion(uint64_t *v, size_t n) { batch_inversion(reinterpret_cast<Fr *>(v), n); }

// ---- Fq -------------------------------------------------------------------------------------
// op: 0 add, 1 sub, 2 mul, 3 neg, 4 square, 5 invert (0 -> 0), 6 sqrt (0 on failure)
void orc_fq_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fq x = ld_fq(a + 6 * i), y = b ? ld_fq(b + 6 * i) : fq_zero(), r;
        switch (op) {
            case 0: r = fq_add(x, y); break;
            case 1: r = fq_sub(x, y); break;
            case 2: r = fq_mul(x, y); break;
            case 3: r = fq_neg(x); break;
            case 4: r = fq_square(x); break;
            case 5:
                if (!fq_invert(x, r)) r = fq_zero();
                break;
            default:
                if (!fq_sqrt(x, r)) r = fq_zero();
                break;
        }
        st_fq(out + 6 * i, r);
    }
}
void orc_fq_to_bytes(const uint64_t *a, uint8_t *out48) { fq_to_bytes(ld_fq(a), out48); }
int orc_fq_from_bytes(const uint8_t *in48, uint64_t *out) {
    Fq r;
    bool ok = fq_from_bytes(in48, r);
    st_fq(out, r);
    return ok ? 1 : 0;
}
int orc_fq_lex_largest(const uint64_t *a) { return fq_lexicographically_largest(ld_fq(a)) ? 1 : 0; }

// ---- G1 -------------------------------------------------------------------------------------
void orc_g1_generator(uint64_t *out_proj18) { st_proj(out_proj18, g1p_generator()); }
void orc_g1_identity(uint64_t *out_proj18) { st_proj(out_proj18, g1p_identity()); }
void orc_g1_add(const uint64_t *a18, const uint64_t *b18, uint64_t *out18) { st_proj(out18, g1p_add(ld_proj(a18), ld_proj(b18))); }
void orc_g1_double(const uint64_t *a18, uint64_t *out18) { st_proj(out18, g1p_double(ld_proj(a18))); }
void orc_g1_add_mixed(const uint64_t *a18, const uint64_t *bxy12, uint8_t binf, uint64_t *out18) {
    st_proj(out18, g1p_add_mixed(ld_proj(a18), ld_aff(bxy12, binf)));
}
void orc_g1_mul(const uint64_t *a18, const uint64_t *scalar4, uint64_t *out18) {
    st_proj(out18, g1p_mul(ld_proj(a18), ld_fr(scalar4)));
}
void orc_g1_to_affine(const uint64_t *a18, uint64_t *out_xy12, uint8_t *out_inf) {
    G1Affine q = g1_to_affine(ld_proj(a18));
    st_fq(out_xy12, q.x);
    st_fq(out_xy12 + 6, q.y);
    *out_inf = q.infinity;
}
int orc_g1_is_on_curve(const uint64_t *a18) { return g1p_is_on_curve(ld_proj(a18)) ? 1 : 0; }
void orc_g1_compress(const uint64_t *xy12, uint8_t inf, uint8_t *out48) { g1a_to_compressed(ld_aff(xy12, inf), out48); }

// Consecutive multiples start*G, (start+1)*G, ... as affine points (tests and synthetic SRS for small cases).
void orc_g1_multiples_of_generator(uint64_t start, size_t n, uint64_t *out_xy, uint8_t *out_inf) {
    G1Projective e = g1p_mul(g1p_generator(), fr_from_u64(start));
    std::vector<G1Projective> ps(n);
    for (size_t i = 0; i < n; i++) {
        ps[i] = e;
        e = g1p_add(e, g1p_generator());
    }
    std::vector<G1Affine> qs(n);
    g1_batch_normalize(ps.data(), qs.data(), n);
    for (size_t i = 0; i < n; i++) {
        st_fq(out_xy + 12 * i, qs[i].x);
        st_fq(out_xy + 12 * i + 6, qs[i].y);
        out_inf[i] = qs[i].infinity;
    }
}

// SRS powers g, tau*g, tau^2*g, ... following plonk/src/commitment_scheme/kzg10/srs.rs:50-60
// (powers_of(tau) then one scalar multiplication per power, then batch_normalize).
void orc_srs_powers(const uint64_t *g_proj18, const uint64_t *tau4, size_t n, uint64_t *out_xy, uint8_t *out_inf) {
    G1Projective g = ld_proj(g_proj18);
    FrVec pw = n ? powers_of(ld_fr(tau4), n - 1) : FrVec();
    std::vector<G1Projective> ps(n);
    for (size_t i = 0; i < n; i++) ps[i] = g1p_mul(g, pw[i]);
    std::vector<G1Affine> qs(n);
    g1_batch_normalize(ps.data(), qs.data(), n);
    for (size_t i = 0; i < n; i++) {
        st_fq(out_xy + 12 * i, qs[i].x);
        st_fq(out_xy + 12 * i + 6, qs[i].y);
        out_inf[i] = qs[i].infinity;
    }
}

// ---- MSM ------------------------------------------------------------------------------------
// Returns seconds spent inside msm_variable_base. inf may be NULL (no point at infinity).
double orc_msm(const uint64_t *points_xy, const uint8_t *inf, const uint64_t *scalars, size_t n_scalars, size_t n_points,
               unsigned n_threads, uint64_t *out_proj18) {
    std::vector<G1Affine> pts(n_points);
    for (size_t i = 0; i < n_points; i++) pts[i] = ld_aff(points_xy + 12 * i, inf ? inf[i] : 0);
    const Fr *sc = reinterpret_cast<const Fr *>(scalars);
    auto t0 = std::chrono::steady_clock::now();
    G1Projective r = msm_variable_base(pts.data(), sc, n_scalars, n_points, n_threads);
    auto t1 = std::chrono::steady_clock::now();
    st_proj(out_proj18, r);
    return std::chrono::duration<double>(t1 - t0).count();
}
size_t orc_msm_window_size(size_t n) { return msm_window_size(n); }

// ---- FFT ------------------------------------------------------------------------------------
// mode: 0 fft, 1 ifft, 2 coset_fft, 3 coset_ifft. `in` has len_in elements (any length), `out`
// receives domain(size_hint).size elements. Returns seconds spent in the transform; <0 on error.
double orc_fft(int mode, const uint64_t *in, size_t len_in, size_t domain_size_hint, uint64_t *out) {
    try {
        EvaluationDomain d(domain_size_hint);
        FrVec v = ld_vec(in, len_in);
        auto t0 = std::chrono::steady_clock::now();
        FrVec r = mode == 0 ? d.fft(v) : mode == 1 ? d.ifft(v) : mode == 2 ? d.coset_fft(v) : d.coset_ifft(v);
        auto t1 = std::chrono::steady_clock::now();
        memcpy(out, r.data(), r.size() * 32);
        return std::chrono::duration<double>(t1 - t0).count();
    } catch (...) {
        return -1.0;
    }
}
// Fills the seven fields of EvaluationDomain::new(n): returns log_size, or -1 on InvalidEvalDomainSize.
// out5 = size_as_field_element | size_inv | group_gen | group_gen_inv | generator_inv (5 x 4 limbs)
int orc_domain_new(size_t num_coeffs, uint64_t *out_size, uint64_t *out5) {
    try {
        EvaluationDomain d(num_coeffs);
        *out_size = d.size;
        st_fr(out5, d.size_as_field_element);
        st_fr(out5 + 4, d.size_inv);
        st_fr(out5 + 8, d.group_gen);
        st_fr(out5 + 12, d.group_gen_inv);
        st_fr(out5 + 16, d.generator_inv);
        return (int)d.log_size_of_group;
    } catch (...) {
        return -1;
    }
}
void orc_vanishing_poly_over_coset(size_t domain_size, uint64_t poly_degree, uint64_t *out) {
    EvaluationDomain d(domain_size);
    FrVec v = d.compute_vanishing_poly_over_coset(poly_degree);
    memcpy(out, v.data(), v.size() * 32);
}
void orc_lagrange_coefficients(size_t domain_size, const uint64_t *tau, uint64_t *out) {
    EvaluationDomain d(domain_size);
    FrVec v = d.evaluate_all_lagrange_coefficients(ld_fr(tau));
    memcpy(out, v.data(), v.size() * 32);
}

}  // extern "C"

#include "plonk_capi.inc"
