/* zkgpu — C boundary of the MI355X PLONK prover core.
 *
 * The reference (zk590/zkvm, a Rust workspace) has no FFI for this path: its hot functions are private Rust
 * methods.  Each entry point below names the reference function it replaces; INTEGRATION.md shows the
 * `extern "C"` block and the call sites a maintainer adds to the `plonk` crate.
 *
 * Data formats (identical to the reference's in-memory representations, so no conversion is needed):
 *   Fr element      4 x uint64_t little-endian limbs, Montgomery form, canonical (< r)   = Scalar.0
 *                   (coset-bls12_381/src/scalar.rs:28)
 *   G1 affine point 12 x uint64_t = x limbs then y limbs, Montgomery form                = the first 96 bytes of
 *                   G1Affine::to_raw_bytes() (coset-bls12_381/src/g1/coset.rs:8-24); the infinity flag travels
 *                   as a separate byte
 * Every function returns 0 on success and a negative ZKGPU_ERR_* code otherwise; zkgpu_last_error() then
 * holds a message.  Nothing throws or aborts across this boundary.  Calls on one context are serialised by
 * the caller (one context per thread or an external lock); different contexts are independent.
 */
#ifndef ZKGPU_H
#define ZKGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKGPU_OK 0
#define ZKGPU_ERR_INVALID_ARG (-1)
#define ZKGPU_ERR_HIP (-2)
#define ZKGPU_ERR_NO_DEVICE (-3)
#define ZKGPU_ERR_DOMAIN_SIZE (-4)   /* plonk Error::InvalidEvalDomainSize   (plonk/src/error.rs) */
#define ZKGPU_ERR_DEGREE_ZERO (-5)   /* plonk Error::PolynomialDegreeIsZero  */
#define ZKGPU_ERR_DEGREE_LARGE (-6)  /* plonk Error::PolynomialDegreeTooLarge */
#define ZKGPU_ERR_NOT_LOADED (-7)
#define ZKGPU_ERR_OOM (-8)
#define ZKGPU_ERR_QUOTIENT_SHAPE (-9)
#define ZKGPU_ERR_INTERNAL (-10)

/* Transform selector of zkgpu_ntt*: the four public methods of EvaluationDomain
 * (plonk/src/fft/domain.rs:139-196). */
#define ZKGPU_FFT 0
#define ZKGPU_IFFT 1
#define ZKGPU_COSET_FFT 2
#define ZKGPU_COSET_IFFT 3

typedef struct zkgpu_ctx zkgpu_ctx;

/* Mirror of EvaluationDomain (plonk/src/fft/domain.rs:23-44). */
typedef struct zkgpu_domain {
    uint64_t size;
    uint32_t log_size_of_group;
    uint64_t size_as_field_element[4];
    uint64_t size_inv[4];
    uint64_t group_gen[4];
    uint64_t group_gen_inv[4];
    uint64_t generator_inv[4];
} zkgpu_domain;

/* ---- context -------------------------------------------------------------------------------------------- */
/* Creates a context bound to HIP device `device_id` with its own stream.  Fails with ZKGPU_ERR_NO_DEVICE when
 * no GPU is present: there is no CPU fallback. */
int zkgpu_init(zkgpu_ctx **out, int device_id);
void zkgpu_destroy(zkgpu_ctx *ctx);
/* Message of the last failing call on this context (never NULL).  With ctx == NULL: the last zkgpu_init failure. */
const char *zkgpu_last_error(const zkgpu_ctx *ctx);
/* Blocks until all work queued on the context's stream has finished. */
int zkgpu_sync(zkgpu_ctx *ctx);
/* Device bytes currently held by the context (tables, point table, scratch). */
size_t zkgpu_device_bytes(const zkgpu_ctx *ctx);
/* HIP-event stopwatch on the context's stream: start, queue work, then stop returns the elapsed milliseconds. */
int zkgpu_timer_start(zkgpu_ctx *ctx);
int zkgpu_timer_stop(zkgpu_ctx *ctx, float *elapsed_ms);

/* ---- EvaluationDomain ----------------------------------------------------------------------------------- */
/* EvaluationDomain::new(num_coeffs) (domain.rs:106-133).  Host-only arithmetic; needs no context.
 * ZKGPU_ERR_DOMAIN_SIZE when log2(size) >= 32, as the reference's Err(InvalidEvalDomainSize). */
int zkgpu_domain_new(uint64_t num_coeffs, zkgpu_domain *out);

/* fft / ifft / coset_fft / coset_ifft (domain.rs:139-196), which all end in best_fft (domain.rs:286).
 * `data` is a HOST buffer with room for 2^log_n elements of which the first len_in are the input; like the
 * reference's Vec::resize the input is zero-padded, or truncated when len_in > 2^log_n.  The result
 * (2^log_n elements, natural order) overwrites `data`. */
int zkgpu_ntt(zkgpu_ctx *ctx, uint64_t *data, size_t len_in, uint32_t log_n, int mode);
/* Same on DEVICE memory, asynchronous on the context's stream; in_dev and out_dev may alias.
 * out_dev must hold 2^log_n elements. */
int zkgpu_ntt_dev(zkgpu_ctx *ctx, const void *in_dev, size_t len_in, void *out_dev, uint32_t log_n, int mode);

/* ---- CommitKey ------------------------------------------------------------------------------------------ */
/* Uploads CommitKey.powers_of_g (plonk/src/commitment_scheme/kzg10/key.rs:32-35) once; it stays resident, as
 * the reference's Prover holds its commit_key by value (plonk/src/compiler/prover.rs:22-30).
 * `infinity` may be NULL. */
int zkgpu_srs_load(zkgpu_ctx *ctx, const uint64_t *xy, const uint8_t *infinity, size_t n_points);
/* As above from a device buffer of n_points x 96 bytes (copied). */
int zkgpu_srs_load_dev(zkgpu_ctx *ctx, const void *xy_dev, size_t n_points);

/* msm_variable_base(&powers_of_g, scalars) followed by Commitment::from(G1Projective)
 * (coset-bls12_381/src/coset/multiscalar_mul.rs:143-220, kzg10/commitment.rs:34-38): the affine sum of
 * scalars[i] * powers_of_g[i] over i < min(n, n_points), as the reference's zip truncates.  `scalars` is a HOST
 * buffer.  out_xy/out_infinity use the reference's identity encoding (x = 0, y = 1, infinity = 1). */
int zkgpu_msm(zkgpu_ctx *ctx, const uint64_t *scalars, size_t n, uint64_t out_xy[12], uint8_t *out_infinity);
/* Same with the scalars in DEVICE memory.  Synchronous (the result is needed on the host). */
int zkgpu_msm_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n, uint64_t out_xy[12], uint8_t *out_infinity);
/* Point-range shard: affine-free partial sum of scalars[i] * powers_of_g[base + i], i < n, returned as a
 * projective point (18 x uint64_t, x | y | z) so that shards from several GPUs can be added before the single
 * affine conversion. */
int zkgpu_msm_partial_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n, size_t base, uint64_t out_xyz[18]);
/* Adds `count` projective points (host, 18 limbs each) and converts the sum to affine — the combining step of
 * a sharded commitment. */
int zkgpu_g1_sum_to_affine(const uint64_t *xyz, size_t count, uint64_t out_xy[12], uint8_t *out_infinity);
/* Device phase of the MSM only (no host tail, no result): used to time the kernels. */
int zkgpu_msm_kernels_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n);

/* CommitKey::commit (kzg10/key.rs:108-129): degree check, then the MSM.  `coeffs` is a host buffer of `len`
 * coefficients as held by the reference's Polynomial (leading zeros allowed; the degree is computed as
 * Polynomial::degree does).  ZKGPU_ERR_DEGREE_ZERO / ZKGPU_ERR_DEGREE_LARGE mirror the reference's errors. */
int zkgpu_commit(zkgpu_ctx *ctx, const uint64_t *coeffs, size_t len, uint64_t out_xy[12], uint8_t *out_infinity);

/* G1Affine::to_bytes (coset-bls12_381/src/g1/coset.rs:54-71): the 48-byte compressed encoding that the
 * transcript absorbs.  Host-only. */
int zkgpu_g1_compress(const uint64_t xy[12], uint8_t infinity, uint8_t out[48]);

#ifdef __cplusplus
}
#endif
#endif /* ZKGPU_H */
