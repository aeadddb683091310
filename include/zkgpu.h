/* zkgpu — C boundary of the MI355X PLONK prover core.
 *
 * The reference (zk590/zkvm, a Rust workspace) has no FFI for this path: its hot functions are private Rust
 * methods.  Each entry point below names the reference function it replaces; rust/zkgpu-sys holds the
 * `extern "C"` block a maintainer adds to the `plonk` crate to bind them.
 *
 * Data formats (identical to the reference's in-memory representations, so no conversion is needed):
 *   Fr element      4 x uint64_t little-endian limbs, Montgomery form, canonical (< r)   = Scalar.0
 *                   (coset-bls12_381/src/scalar.rs:28)
 *   G1 affine point 12 x uint64_t = x limbs then y limbs, Montgomery form                = the first 96 bytes of
 *                   G1Affine::to_raw_bytes() (coset-bls12_381/src/g1/coset.rs:8-24); the infinity flag travels
 *                   as a separate byte
 * Every function returns 0 on success and a negative ZKGPU_ERR_* code otherwise; zkgpu_last_error() then
 * holds a message.  Nothing throws or aborts across this boundary.  A context may be shared by several host
 * threads, as the reference's Prover::prove takes &self (plonk/src/compiler/prover.rs:21,210): every call locks the
 * context for its duration, so calls on one context run one after the other; different contexts are independent and
 * run concurrently (each has its own stream and buffers).  zkgpu_last_error returns the calling thread's own last
 * failure on that context.
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
/* #[derive(Clone)] Prover (plonk/src/compiler/prover.rs:20-21): a second context on the same GPU, with a stream and
 * per-proof buffers of its own, that refers to the SRS table and the prover key `src` holds instead of copying them (at
 * 2^20 gates 4.8 GiB of table and key together stay single).  Both are read-only after their load, so the clones
 * prove concurrently, one host thread each, with no lock between them; the shared data lives until the last context
 * that refers to it is destroyed or loads something else (a load replaces only that context's reference).  Coset
 * vectors `src` built lazily after its key load, twiddle tables and all scratch are per context.  Not for a key
 * loaded under zkgpu_set_collective (ZKGPU_ERR_INVALID_ARG). */
int zkgpu_ctx_clone(zkgpu_ctx *src, zkgpu_ctx **out);
/* Of zkgpu_device_bytes(ctx): the bytes this context holds together with other contexts (0 without clones). */
size_t zkgpu_device_bytes_shared(const zkgpu_ctx *ctx);
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

/* zkgpu_ntt_dev for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  It calls no hipMalloc / hipFree and no allocating routine of
 * any library: the buffer between the passes of the transform is a file-scope device array of
 * 2^ZKGPU_NOALLOC_MAX_LOG_N elements (32 MiB per GPU, held until the process ends), so log_n above that maximum is
 * rejected with ZKGPU_ERR_INVALID_ARG; and the twiddle tables of this mode and size must already be resident
 * (ZKGPU_ERR_NOT_LOADED otherwise): any earlier zkgpu_ntt / zkgpu_ntt_dev call on this context with the same mode at
 * this size (for coset_ifft) or at this or a larger size (the other three modes) leaves them behind.  All contexts on
 * one GPU share that GPU's array, each on a stream of its own: unordered, overlapping calls would race on it, so the
 * library orders them itself.  One event per device is waited for before the first pass and recorded behind the last
 * (hipStreamWaitEvent / hipEventRecord on an event made in zkgpu_init: no allocation), so a call from any context on
 * that GPU starts only when the previous one's passes are done; calls from several threads and contexts are safe and
 * run one after the other.  The array is not cleared between calls; every pass overwrites all the elements the next
 * one reads. */
#define ZKGPU_NOALLOC_MAX_LOG_N 20
int zkgpu_ntt_dev_noalloc(zkgpu_ctx *ctx, const void *in_dev, size_t len_in, void *out_dev, uint32_t log_n, int mode);

/* Tuning knobs of the transform, for experiments and tests: log2 of the LDS tile in elements (6..12, 0 = default)
 * and log2 of the elements each thread carries per stage (1..3, 0 = default).  Results do not depend on them. */
int zkgpu_ntt_set_params(zkgpu_ctx *ctx, uint32_t tile_log, uint32_t elems_per_thread_log);

/* ---- CommitKey ------------------------------------------------------------------------------------------ */
/* Uploads CommitKey.powers_of_g (plonk/src/commitment_scheme/kzg10/key.rs:32-35) once; it stays resident, as
 * the reference's Prover holds its commit_key by value (plonk/src/compiler/prover.rs:22-30).
 * `infinity` may be NULL. */
int zkgpu_srs_load(zkgpu_ctx *ctx, const uint64_t *xy, const uint8_t *infinity, size_t n_points);
/* As above from a device buffer of n_points x 96 bytes (copied). */
int zkgpu_srs_load_dev(zkgpu_ctx *ctx, const void *xy_dev, size_t n_points);

/* PublicParameters::setup (plonk/src/commitment_scheme/kzg10/srs.rs:40-71) with the two random draws made explicit:
 * powers_of_g[i] = tau^i * g for i < n_points, generated on the device and kept as the context's SRS (n_points =
 * max_degree + 6 + 1 in the reference's terms).  `g_xy` is an affine point in the layout of zkgpu_srs_load. */
int zkgpu_srs_setup(zkgpu_ctx *ctx, const uint64_t tau[4], const uint64_t g_xy[12], size_t n_points);
/* Copies the first n points of the resident SRS back to the host (x | y, and the infinity flags when not NULL). */
int zkgpu_srs_export(zkgpu_ctx *ctx, uint64_t *out_xy, uint8_t *out_infinity, size_t n);

/* CommitKey::from_slice_unchecked / to_raw_var_bytes (plonk/src/commitment_scheme/kzg10/key.rs:38-66): a u64
 * little-endian point count followed by 97 bytes per point (x and y as six little-endian u64 Montgomery limbs each,
 * then the infinity flag; coset-bls12_381/src/g1/coset.rs:8-48).  In a PublicParameters::to_raw_var_bytes file the
 * commit key starts after the 240-byte OpeningKey (srs.rs:73-79).  As the reference, the loader takes
 * min(count, whole points present); zkgpu_srs_export_raw writes the resident SRS in the same format
 * (8 + 97 * n_points bytes; *written receives that size, and nothing is written when cap is smaller). */
int zkgpu_srs_load_raw(zkgpu_ctx *ctx, const uint8_t *bytes, size_t len);
int zkgpu_srs_export_raw(zkgpu_ctx *ctx, uint8_t *out, size_t cap, size_t *written);

/* Tuning knobs of the bucket method, for experiments and tests: the signed-digit window width used by the NEXT
 * zkgpu_srs_load* (4..22, 0 = chosen from the table size) and the log2 segment width of the bucket reduction
 * (1..6, 0 = default).  Results do not depend on them. */
int zkgpu_msm_set_params(zkgpu_ctx *ctx, uint32_t window_bits, uint32_t segment_bits);

/* Window width and number of table levels (signed digits per scalar) chosen for the loaded SRS. */
int zkgpu_msm_get_params(zkgpu_ctx *ctx, uint32_t *window_bits, uint32_t *n_windows);

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

/* As zkgpu_msm_kernels_dev with HIP events on the context's stream at the phase boundaries (milliseconds):
 *   0 digits            de-Montgomery and signed recoding
 *   1 range partition   above 2^15 buckets: the pairs partitioned by bucket range (0 otherwise)
 *   2 count pass        per-slice bucket histograms
 *   3 offsets           bucket totals, their scan, per-slice cursors
 *   4 scatter pass      table indices written bucket by bucket
 *   5 schedule          buckets ordered by size
 *   6 accumulate        msm_accumulate_kernel alone
 *   7 oversized         buckets cut into workgroup tasks
 *   8 reduce            bucket reduction */
#define ZKGPU_MSM_PHASES 9
int zkgpu_msm_profile_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n, float phase_ms[ZKGPU_MSM_PHASES]);

/* CommitKey::commit (kzg10/key.rs:108-129): degree check, then the MSM.  `coeffs` is a host buffer of `len`
 * coefficients as held by the reference's Polynomial (leading zeros allowed; the degree is computed as
 * Polynomial::degree does).  ZKGPU_ERR_DEGREE_ZERO / ZKGPU_ERR_DEGREE_LARGE mirror the reference's errors. */
int zkgpu_commit(zkgpu_ctx *ctx, const uint64_t *coeffs, size_t len, uint64_t out_xy[12], uint8_t *out_infinity);
/* The commitments of one proof round in one pass, as the reference's prover makes them one after the other
 * (compiler/prover.rs:262-265 the four wires, :344-347 the four quotient parts, :463-485 the two openings): up to 4
 * polynomials over the same SRS share the sort and reduction launches of a batched MSM.  coeffs[j] is a host buffer
 * of lens[j] coefficients; out_xy receives 12 limbs and out_infinity one flag per polynomial.  The degree checks of
 * zkgpu_commit run in order before any work and the first failure is returned, as the reference returns its first
 * error. */
int zkgpu_commit_many(zkgpu_ctx *ctx, uint32_t count, const uint64_t *const *coeffs, const size_t *lens, uint64_t *out_xy,
                      uint8_t *out_infinity);

/* Host-only self-check of the Montgomery product the kernels are built on (the same source compiles for host and
 * device): field 0 = Fr (4 limbs per element), 1 = Fq (6 limbs); out[i] = a[i] * b[i] for count elements. */
int zkgpu_field_mul_selfcheck(int field, const uint64_t *a, const uint64_t *b, size_t count, uint64_t *out);
/* The same for the host's inversion (binary extended Euclidean algorithm on 64-bit words; the host inverts once per
 * commitment batch and per permutation, where the reference calls Fp::invert / Scalar::invert, fp.rs:383-415 and
 * scalar.rs:394-477): out[i] = a[i]^-1 in Montgomery form, zero for zero. */
int zkgpu_field_inv_selfcheck(int field, const uint64_t *a, size_t count, uint64_t *out);

/* Host-only self-check of the bucket arithmetic used inside the MSM kernels (28-bit unsaturated limbs, extended
 * Jacobian mixed additions): adds the n affine points in order, point i negated when negate[i] != 0, and returns the
 * sum as a projective point.  *violations receives the number of products whose operands broke the limb or
 * magnitude bounds that the device code relies on (0 on a correct build). */
int zkgpu_g1_bucket_selfcheck(const uint64_t *xy, const uint8_t *infinity, const uint8_t *negate, size_t n,
                              uint64_t out_xyz[18], uint64_t *violations);
/* Same for the accumulator + accumulator additions of the bucket reduction: every point becomes an accumulator
 * and the accumulators are summed pairwise, level by level, as the device's workgroup trees do. */
int zkgpu_g1_bucket_selfcheck_tree(const uint64_t *xy, const uint8_t *infinity, const uint8_t *negate, size_t n,
                                   uint64_t out_xyz[18], uint64_t *violations);

/* G1Affine::to_bytes (coset-bls12_381/src/g1/coset.rs:54-71): the 48-byte compressed encoding that the
 * transcript absorbs.  Host-only. */
int zkgpu_g1_compress(const uint64_t xy[12], uint8_t infinity, uint8_t out[48]);
/* G1Affine::from_bytes (coset-bls12_381/src/g1/coset.rs:73-140): compressed x with its three flag bits, y from the
 * curve equation and the sort flag, subgroup check.  ZKGPU_ERR_INVALID_ARG where the reference returns Err.  Host-only. */
int zkgpu_g1_decompress(const uint8_t in[48], uint64_t out_xy[12], uint8_t *out_infinity);

/* Proof::to_bytes / Proof::from_bytes (plonk/src/proof_system/proof.rs:95-157): the 1008-byte proof the reference's
 * service writes to its proof files (merkle-plonk/src/lib.rs:335-352) — the eleven commitments a b c d z t_low t_mid
 * t_high t_fourth w_z w_zw in the compressed 48-byte form, then the fifteen evaluations in ProofEvaluations order
 * (proof_system/linearization_poly.rs:29-74,76-137) as canonical little-endian scalars.  The arrays are those of
 * zkgpu_prove (commitments x | y of 12 limbs with their infinity flags, evaluations in Montgomery form).  from_bytes
 * returns ZKGPU_ERR_INVALID_ARG where the reference returns an Err: a point encoding that is not a compressed
 * subgroup point, a scalar not below the modulus.  Host-only. */
#define ZKGPU_PROOF_BYTES 1008
int zkgpu_proof_to_bytes(const uint64_t *comms_xy, const uint8_t *comms_infinity, const uint64_t *evals,
                         uint8_t out[ZKGPU_PROOF_BYTES]);
int zkgpu_proof_from_bytes(const uint8_t in[ZKGPU_PROOF_BYTES], uint64_t *comms_xy, uint8_t *comms_infinity,
                           uint64_t *evals);

/* ---- ProverKey / Prover ---------------------------------------------------------------------------------- */
/* Order of the 15 selector / sigma vectors in every array below:
 *   q_m q_l q_r q_o q_f q_c q_arith q_range q_logic q_fixed_group_add q_variable_group_add s_sigma_1..4 */
#define ZKGPU_N_SELECTORS 15

/* Compiler::preprocess (plonk/src/compiler.rs:78-390) on the GPU: `selector_evals` holds 15 x n gate-wise
 * values (n a power of two; the four sigma rows are the values produced by build_permutation_lagrange_mapping,
 * plonk/src/composer/permutation.rs:111-141).  Builds the resident ProverKey (15 coefficient vectors, their 8n
 * coset evaluations, X and L_1 on the coset, 1/Z_H) and the 15 commitments that seed the transcript
 * (plonk/src/proof_system/widget.rs:199-228).  The SRS must be loaded first.  `constraints` is the gate
 * count the reference passes to Transcript::base and EvaluationDomain::new; `pi_indexes` are the gate indexes of
 * the public inputs in ascending order (composer.rs:985-992). */
int zkgpu_prover_key_load(zkgpu_ctx *ctx, uint32_t n, uint64_t constraints, const uint64_t *selector_evals,
                          const char *label, size_t label_len, const uint64_t *pi_indexes, size_t n_pi);
/* Prover::try_from_bytes (plonk/src/compiler/prover.rs:137-207): loads a serialized prover — label, ProverKey
 * (plonk/src/proof_system/widget.rs:374-519: 15 coefficient vectors and their 8n coset evaluations as canonical
 * little-endian scalars), CommitKey in the raw form (kzg10/key.rs:54-66; it replaces the loaded SRS) and VerifierKey
 * (widget.rs:102-123: 15 compressed commitments) — into device memory, ready for zkgpu_prove.  The reference keeps
 * the public-input positions in the circuit (composer.rs:985-992), not in these bytes, so they are passed beside
 * them.  *n_out receives the key's gate count.  ZKGPU_ERR_INVALID_ARG where the reference returns an Err
 * (NotEnoughBytes, an invalid point or scalar). */
int zkgpu_prover_load_bytes(zkgpu_ctx *ctx, const uint8_t *bytes, size_t len, const uint64_t *pi_indexes, size_t n_pi,
                            uint32_t *n_out);
/* Prover::to_bytes (plonk/src/compiler/prover.rs:85-134) of the loaded key and SRS, byte for byte the reference's
 * layout, including ProverKey::serialization_size's rule of budgeting every polynomial at q_m's length
 * (widget.rs:265-278).  *written receives the size; bytes are written when `out` is not NULL and cap suffices. */
int zkgpu_prover_to_bytes(zkgpu_ctx *ctx, uint8_t *out, size_t cap, size_t *written);
/* The 15 commitments of the loaded key (x | y, 12 limbs each) and their infinity flags. */
int zkgpu_prover_key_commitments(zkgpu_ctx *ctx, uint64_t *out_xy, uint8_t *out_infinity);
/* Copies of key members for inspection: coefficient vector `which` (n elements), and the 8n coset evaluations
 * `which` in 0..14 (selectors), 15 (linear_evaluations) or 16 (L_1, unscaled); for a key loaded under
 * zkgpu_set_collective the latter are this rank's 8n / world elements i = rank + world * k. */
int zkgpu_prover_key_poly(zkgpu_ctx *ctx, int which, uint64_t *out);
int zkgpu_prover_key_eval8n(zkgpu_ctx *ctx, int which, uint64_t *out);
/* Elements zkgpu_prover_key_eval8n writes: 8n, or 8n / world for a shared key; 0 when no key is loaded. */
size_t zkgpu_prover_key_eval_len(const zkgpu_ctx *ctx);
/* Which coset zkgpu_prove evaluates the quotient's numerator on.  The reference always takes the 8n-point coset
 * (plonk/src/proof_system/quotient_poly.rs:28,46-51).  The quotient has at most 4n + 7 coefficients (the numerator's
 * permutation term reaches degree 5n + 6, a widget's term 5n + 3), and from n = 16 up this library evaluates on the 4n
 * points g w_4n^k (every second point of the 8n coset) and restores the seven top coefficients from those of the round
 * and key polynomials: the same coefficients at half the transform and pointwise work.  A proof whose witness violates
 * a gate (arithmetic or widget) or the permutation (the numerator then does not vanish on the domain) is detected
 * beforehand and runs over the 8n coset, so it ends in the reference's error; so does every proof of a key loaded
 * under zkgpu_set_collective.  zkgpu_prover_force_full_coset(ctx, 1) makes keys loaded afterwards use the 8n coset always;
 * zkgpu_prover_last_proof_coset returns 4 or 8 for the most recent zkgpu_prove on this context (0 before the first).
 * No counterpart in the reference. */
int zkgpu_prover_force_full_coset(zkgpu_ctx *ctx, int on);
int zkgpu_prover_last_proof_coset(const zkgpu_ctx *ctx);

/* Multi-GPU proving, one context per GPU: this context's SRS holds the points [shard_base, shard_base + its point
 * count) of an SRS of total_points points.  Every commitment of zkgpu_prover_key_load and zkgpu_prove then sums this
 * context's range only and calls `exchange(user, mine, m, all)`, which must all-gather the m partial sums (18 limbs
 * each) of every rank into all[rank][m][18] and return 0; the sums are added in rank order, so all ranks obtain the
 * same proof.  Everything else (transforms, quotient) runs replicated.  exchange == NULL switches sharding off.
 * No counterpart in the reference, which proves on one host. */
typedef int (*zkgpu_exchange_fn)(void *user, const uint64_t *mine, size_t m, uint64_t *all);
int zkgpu_prover_set_sharding(zkgpu_ctx *ctx, uint64_t shard_base, uint64_t total_points, uint32_t world,
                              zkgpu_exchange_fn exchange, void *user);

/* ---- one transform, one quotient over several GPUs ----------------------------------------------------------------
 * No counterpart in the reference, whose best_fft (plonk/src/fft/domain.rs:286-340) and quotient loops
 * (quotient_poly.rs:86-95,128-191,218-234) run on one host.  One context per GPU, `world` = 2, 4 or 8 of them.
 *
 * The 2^log_n evaluation points are dealt out cyclically: rank c owns the points k = c + world * k'.  They form the
 * coset (g) w^c <w^world>, so a rank computes its evaluations of a polynomial it holds in full with one local
 * transform of 2^log_n / world points and no exchange (zkgpu_ntt_slice_dev).  The way back
 * (zkgpu_intt_sharded_dev) is a four-step transform whose small dimension lies across the GPUs: local inverse
 * transform, all-to-all transpose (2^log_n * 32 / world^2 bytes to every peer), a world-point inverse DFT on every
 * rank's index range, and an all-gather that leaves all coefficients on every rank.  Because world divides 8, the
 * quotient's "next row" i + 8 (quotient_poly.rs:53-59) stays inside a rank's slice: no halo is exchanged.
 *
 * The exchanges go through the caller: `fn(user, op, send_dev, recv_dev, block_bytes)` runs one collective on DEVICE
 * buffers and returns 0 once recv_dev holds the data (the library has synchronised its stream before the call and
 * launches nothing during it):
 *   op ZKGPU_ALL_TO_ALL   send_dev holds world blocks of block_bytes, block r for rank r; recv_dev receives block r
 *                         from rank r
 *   op ZKGPU_ALL_GATHER   send_dev holds one block; recv_dev receives the blocks of all ranks in rank order
 * zkgpu_set_collective registers it (fn == NULL removes it).  A prover key loaded afterwards
 * (zkgpu_prover_key_load) keeps only this rank's slice of every 8n-point vector (1 / world of the ProverKey's
 * memory), and zkgpu_prove / zkgpu_quotient evaluate the quotient on that slice; all ranks must then call them
 * together, and all obtain the same proof.  Combine with zkgpu_prover_set_sharding to share the commitments too. */
#define ZKGPU_ALL_TO_ALL 0
#define ZKGPU_ALL_GATHER 1
typedef int (*zkgpu_collective_fn)(void *user, int op, const void *send_dev, void *recv_dev, size_t block_bytes);
int zkgpu_set_collective(zkgpu_ctx *ctx, uint32_t rank, uint32_t world, zkgpu_collective_fn fn, void *user);
/* This rank's slice of fft / coset_fft (mode ZKGPU_FFT or ZKGPU_COSET_FFT) of the len_in coefficients at in_dev over the
 * 2^log_n domain: out_dev receives 2^log_n / world elements, out[k'] = the transform's element rank + world * k'.
 * world is a power of two; asynchronous on the context's stream; needs no collective. */
int zkgpu_ntt_slice_dev(zkgpu_ctx *ctx, const void *in_dev, size_t len_in, void *out_dev, uint32_t log_n, int mode,
                        uint32_t rank, uint32_t world);
/* ifft / coset_ifft (mode ZKGPU_IFFT or ZKGPU_COSET_IFFT) of evaluations held as slices by the ranks of
 * zkgpu_set_collective: slice_dev holds this rank's 2^log_n / world evaluations, out_dev receives all 2^log_n
 * coefficients on every rank.  Collective: every rank calls it.  Needs 2^log_n >= world^2. */
int zkgpu_intt_sharded_dev(zkgpu_ctx *ctx, const void *slice_dev, void *out_dev, uint32_t log_n, int mode);

/* build_quotient_polynomial (plonk/src/proof_system/quotient_poly.rs:15-100).  `polys` packs the coefficient
 * vectors z, a, b, c, d, pi back to back with lens[6]; challenges7 = alpha beta gamma range logic fixed_base
 * var_base.  t_out receives 8n coefficients of which *t_len remain after dropping leading zeros. */
int zkgpu_quotient(zkgpu_ctx *ctx, const uint64_t *polys, const size_t lens[6], const uint64_t *challenges7,
                   uint64_t *t_out, size_t *t_len);
/* Device time in milliseconds of the pointwise quotient kernel (the three loops of quotient_poly.rs:88-95,128-191,
 * 218-234 fused) in the most recent zkgpu_quotient or zkgpu_prove on this context, from HIP events around its launch. */
int zkgpu_quotient_kernel_ms(zkgpu_ctx *ctx, float *ms);
/* Permutation::compute_permutation_vec (plonk/src/composer/permutation.rs:181-245): wires and sigma_evals hold
 * 4 x n values each (sigma_evals = the FFT of the sigma polynomials, i.e. the sigma values). */
int zkgpu_perm_z(zkgpu_ctx *ctx, uint32_t n, const uint64_t *wires, const uint64_t *sigma_evals, const uint64_t beta[4],
                 const uint64_t gamma[4], uint64_t *z_out);
/* Polynomial::evaluate (plonk/src/fft/polynomial.rs:105-119) and Polynomial::ruffini (:343-357; out receives
 * len - 1 coefficients). */
int zkgpu_poly_eval(zkgpu_ctx *ctx, const uint64_t *coeffs, size_t len, const uint64_t x[4], uint64_t out[4]);
int zkgpu_poly_ruffini(zkgpu_ctx *ctx, const uint64_t *coeffs, size_t len, const uint64_t root[4], uint64_t *out);

/* build_linearization_polynomial (plonk/src/proof_system/linearization_poly.rs:146-234) for the loaded key.  `polys`
 * packs the coefficient vectors z, t_low, t_mid, t_high, t_fourth back to back with lens[5]; challenges7 = alpha beta
 * gamma range logic fixed_base var_base; evals15 in ProofEvaluations field order (linearization_poly.rs:29-74);
 * public_inputs = the sparse list the reference hands to compute_barycentric_eval (proof_system/proof.rs:511-561).
 * r_out (room for r_cap >= max(n, lens) coefficients) receives r(X); *r_len its length once leading zeros are dropped. */
int zkgpu_linearization(zkgpu_ctx *ctx, const uint64_t *polys, const size_t lens[5], const uint64_t *challenges7,
                        const uint64_t z_challenge[4], const uint64_t *evals15, const uint64_t *public_inputs, size_t n_pi,
                        uint64_t *r_out, size_t r_cap, size_t *r_len);
/* CommitKey::compute_aggregate_witness (plonk/src/commitment_scheme/kzg10/key.rs:131-146): (sum_j v^j p_j) / (X - point)
 * with the remainder dropped, for up to 18 polynomials packed back to back with lens[n_polys].  out receives at most
 * max(lens) - 1 coefficients, *out_len the count once leading zeros are dropped.  Needs no key. */
int zkgpu_aggregate_witness(zkgpu_ctx *ctx, const uint64_t *polys, const size_t *lens, size_t n_polys, const uint64_t point[4],
                            const uint64_t v[4], uint64_t *out, size_t *out_len);

/* Prover::prove (plonk/src/compiler/prover.rs:210-509) for the loaded key.  `wires` holds the 4 x n gate-wise
 * values a, b, c, d (host memory, or a device pointer when the witness is already in HBM); `public_inputs` the n_pi public input values in index order; `blinders14` the 14 scalars
 * the reference draws from its RNG, in draw order a(2) b(2) c(2) d(2) z(3) b12 b13 b14.  Outputs: the 11
 * commitments in the order of Proof (a b c d z t_low t_mid t_high t_fourth w_z w_zw), the 15 evaluations in
 * ProofEvaluations field order, the ten challenges (beta gamma alpha range logic fixed var z v v_w) and,
 * when timings8 is not NULL, milliseconds per phase.  ZKGPU_ERR_DEGREE_* mirror the reference's commit errors;
 * ZKGPU_ERR_QUOTIENT_SHAPE is returned where the reference would panic on a too-short quotient. */
int zkgpu_prove(zkgpu_ctx *ctx, const uint64_t *wires, const uint64_t *public_inputs, size_t n_pi,
                const uint64_t *blinders14, uint64_t *out_comms_xy, uint8_t *out_comms_infinity, uint64_t *out_evals15,
                uint64_t *out_challenges10, float *timings8);

/* The reference's service proves one fresh witness after the other (merkle-plonk/src/lib.rs:275-332: one opening
 * per proof).  zkgpu_witness_prefetch starts the upload of a COMING proof's witness (4 x n values in host memory) on a
 * copy stream of the context's own, into one of two device buffers, and returns at once; the zkgpu_prove call that is
 * later given the same `wires` pointer reads the uploaded copy instead of copying itself.  Called right before
 * zkgpu_prove(witness i) with witness i + 1, the upload (128 MiB at 2^20 gates) runs beside proof i.  The host buffer
 * should be pinned (zkgpu_host_alloc, or the caller's own hipHostMalloc / hipHostRegister): from pageable memory the
 * copy is staged by the runtime and the call returns only when it is done.  At most two uploads are outstanding (a
 * third reuses the older buffer); the host values must stay unchanged until the proof that consumes them has
 * returned.  No counterpart in the reference (it has no device). */
int zkgpu_witness_prefetch(zkgpu_ctx *ctx, const uint64_t *wires_host);
/* Pinned host memory for witnesses (hipHostMalloc); NULL when the allocation fails. */
void *zkgpu_host_alloc(size_t bytes);
void zkgpu_host_free(void *p);

/* ---- circuit front-end of the reference's own workload (host only, no context) --------------------------------------
 * OpeningCircuit (merkle-plonk/src/lib.rs:117-131) over opening_gadget (poseidon-merkle/src/zk.rs:16-102) and the
 * Hades permutation gadget (coset-poseidon/src/hades/permutation/gadget.rs:39-106), built gate by gate as the
 * reference's Composer does (plonk/src/composer.rs:76-131,139-195,451-501,705-716,936-962), for k openings of trees
 * of the same height in ONE circuit (one public root per opening), written straight into the tables
 * zkgpu_prover_key_load and zkgpu_prove consume.  The reference proves one opening per circuit
 * (merkle-plonk/src/lib.rs:275-332); k = 1 gives exactly that circuit, larger k the batch circuit of many openings.
 *
 * All field elements are Montgomery limbs.  leaves: k elements; branches: k x height x 4 sibling hashes, level 0
 * right under the root (coset-merkle/src/opening.rs:19-23); positions: k x height child indexes below 4; roots: the k
 * public roots; arc / mds: the 340 round constants and the 5 x 5 MDS matrix of coset-poseidon/assets; tag: the sponge's
 * tag for four absorbed values and one squeezed under Domain::Merkle4 (coset-poseidon/src/hash.rs:15-67).
 * Outputs for the power of two n returned by the _size call: selectors 15 x n (the four sigma rows left zero: they are
 * the values of build_permutation_lagrange_mapping at sigma_map, composer/permutation.rs:111-141), sigma_map 4 x n
 * with entry = column * n + gate (compute_sigma_permutations, :78-109), wires 4 x n gate-wise witness values, and the
 * k public-input gate indexes with their values.  `threads` host threads generate the openings in parallel. */
int zkgpu_merkle_batch_circuit_size(uint32_t height, uint32_t k, uint64_t *constraints, uint64_t *n);
int zkgpu_merkle_batch_circuit(uint32_t height, uint32_t k, const uint64_t *leaves, const uint64_t *branches,
                               const uint8_t *positions, const uint64_t *roots, const uint64_t *arc, const uint64_t *mds,
                               const uint64_t tag[4], uint64_t n, uint64_t *selectors, uint64_t *sigma_map, uint64_t *wires,
                               uint64_t *pi_indexes, uint64_t *public_inputs, uint32_t threads);
/* The witness-dependent tables of the same circuit alone: wires 4 x n and the k public-input values.  This is what the
 * reference's Prover::prove recomputes per proof by running the circuit on a fresh Composer
 * (plonk/src/composer/prover.rs:214-238) while the selector and permutation tables stay with the key; same inputs and
 * same values as zkgpu_merkle_batch_circuit writes to those two outputs. */
int zkgpu_merkle_batch_witness(uint32_t height, uint32_t k, const uint64_t *leaves, const uint64_t *branches,
                               const uint8_t *positions, const uint64_t *roots, const uint64_t *arc, const uint64_t *mds,
                               const uint64_t tag[4], uint64_t n, uint64_t *wires, uint64_t *public_inputs, uint32_t threads);
/* Hash::digest(Domain::Merkle4, four values) (coset-poseidon/src/hash.rs:144-148 over hades/permutation/scalar.rs:33-67)
 * for `count` groups of four inputs: the node hash of the Poseidon-Merkle tree (poseidon-merkle/src/lib.rs:133-160). */
int zkgpu_poseidon_merkle4(const uint64_t *inputs, size_t count, const uint64_t *arc, const uint64_t *mds, const uint64_t tag[4],
                           uint64_t *out);

/* Transcript::new(label); append_message(msg_label, msg); challenge_bytes(ch_label, out): conformance hook for the
 * host-side Merlin implementation the prover's transcript uses (plonk/src/transcript.rs). Host-only. */
int zkgpu_transcript_challenge(const char *label, const char *msg_label, const uint8_t *msg, size_t msg_len,
                               const char *ch_label, uint8_t *out, size_t out_len);

#ifdef __cplusplus
}
#endif
#endif /* ZKGPU_H */
