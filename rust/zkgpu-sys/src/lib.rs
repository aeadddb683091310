//! Raw `extern "C"` declarations of include/zkgpu.h, one to one.
//!
//! Fr elements travel as `[u64; 4]` — exactly `Scalar.0` (coset-bls12_381/src/scalar.rs:28, Montgomery limbs) —
//! and affine G1 points as `[u64; 12]` = x limbs then y limbs (`Fp.0`, coset-bls12_381/src/fp.rs:23), so the
//! plonk crate can pass `polynomial.as_ptr() as *const u64` without converting anything.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct zkgpu_ctx {
    _private: [u8; 0],
}

/// Mirror of `EvaluationDomain` (plonk/src/fft/domain.rs:23-44).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct zkgpu_domain {
    pub size: u64,
    pub log_size_of_group: u32,
    pub size_as_field_element: [u64; 4],
    pub size_inv: [u64; 4],
    pub group_gen: [u64; 4],
    pub group_gen_inv: [u64; 4],
    pub generator_inv: [u64; 4],
}

pub const ZKGPU_OK: c_int = 0;
pub const ZKGPU_ERR_INVALID_ARG: c_int = -1;
pub const ZKGPU_ERR_HIP: c_int = -2;
pub const ZKGPU_ERR_NO_DEVICE: c_int = -3;
pub const ZKGPU_ERR_DOMAIN_SIZE: c_int = -4;
pub const ZKGPU_ERR_DEGREE_ZERO: c_int = -5;
pub const ZKGPU_ERR_DEGREE_LARGE: c_int = -6;
pub const ZKGPU_ERR_NOT_LOADED: c_int = -7;
pub const ZKGPU_ERR_OOM: c_int = -8;
pub const ZKGPU_ERR_QUOTIENT_SHAPE: c_int = -9;
pub const ZKGPU_ERR_INTERNAL: c_int = -10;

pub const ZKGPU_FFT: c_int = 0;
pub const ZKGPU_IFFT: c_int = 1;
pub const ZKGPU_COSET_FFT: c_int = 2;
pub const ZKGPU_COSET_IFFT: c_int = 3;
pub const ZKGPU_N_SELECTORS: usize = 15;
pub const ZKGPU_NOALLOC_MAX_LOG_N: u32 = 20;
pub const ZKGPU_MSM_PHASES: usize = 9;
pub const ZKGPU_ALL_TO_ALL: c_int = 0;
pub const ZKGPU_ALL_GATHER: c_int = 1;

extern "C" {
    pub fn zkgpu_init(out: *mut *mut zkgpu_ctx, device_id: c_int) -> c_int;
    pub fn zkgpu_destroy(ctx: *mut zkgpu_ctx);
    pub fn zkgpu_last_error(ctx: *const zkgpu_ctx) -> *const c_char;
    pub fn zkgpu_sync(ctx: *mut zkgpu_ctx) -> c_int;
    pub fn zkgpu_device_bytes(ctx: *const zkgpu_ctx) -> usize;
    pub fn zkgpu_timer_start(ctx: *mut zkgpu_ctx) -> c_int;
    pub fn zkgpu_timer_stop(ctx: *mut zkgpu_ctx, elapsed_ms: *mut f32) -> c_int;

    pub fn zkgpu_domain_new(num_coeffs: u64, out: *mut zkgpu_domain) -> c_int;
    pub fn zkgpu_ntt(ctx: *mut zkgpu_ctx, data: *mut u64, len_in: usize, log_n: u32, mode: c_int) -> c_int;
    pub fn zkgpu_ntt_dev(ctx: *mut zkgpu_ctx, in_dev: *const c_void, len_in: usize, out_dev: *mut c_void, log_n: u32,
                         mode: c_int) -> c_int;
    /// The same transform without any allocation inside the library (static 2^20-element scratch per GPU).
    pub fn zkgpu_ntt_dev_noalloc(ctx: *mut zkgpu_ctx, in_dev: *const c_void, len_in: usize, out_dev: *mut c_void,
                                 log_n: u32, mode: c_int) -> c_int;

    pub fn zkgpu_ntt_set_params(ctx: *mut zkgpu_ctx, tile_log: u32, elems_per_thread_log: u32) -> c_int;

    pub fn zkgpu_srs_load(ctx: *mut zkgpu_ctx, xy: *const u64, infinity: *const u8, n_points: usize) -> c_int;
    pub fn zkgpu_srs_load_dev(ctx: *mut zkgpu_ctx, xy_dev: *const c_void, n_points: usize) -> c_int;
    pub fn zkgpu_srs_setup(ctx: *mut zkgpu_ctx, tau: *const u64, g_xy: *const u64, n_points: usize) -> c_int;
    pub fn zkgpu_srs_export(ctx: *mut zkgpu_ctx, out_xy: *mut u64, out_infinity: *mut u8, n: usize) -> c_int;
    pub fn zkgpu_srs_load_raw(ctx: *mut zkgpu_ctx, bytes: *const u8, len: usize) -> c_int;
    pub fn zkgpu_srs_export_raw(ctx: *mut zkgpu_ctx, out: *mut u8, cap: usize, written: *mut usize) -> c_int;
    pub fn zkgpu_msm_set_params(ctx: *mut zkgpu_ctx, window_bits: u32, segment_bits: u32) -> c_int;
    pub fn zkgpu_msm_get_params(ctx: *mut zkgpu_ctx, window_bits: *mut u32, n_windows: *mut u32) -> c_int;
    pub fn zkgpu_msm(ctx: *mut zkgpu_ctx, scalars: *const u64, n: usize, out_xy: *mut u64, out_infinity: *mut u8) -> c_int;
    pub fn zkgpu_msm_dev(ctx: *mut zkgpu_ctx, scalars_dev: *const c_void, n: usize, out_xy: *mut u64,
                         out_infinity: *mut u8) -> c_int;
    pub fn zkgpu_msm_partial_dev(ctx: *mut zkgpu_ctx, scalars_dev: *const c_void, n: usize, base: usize,
                                 out_xyz: *mut u64) -> c_int;
    pub fn zkgpu_g1_sum_to_affine(xyz: *const u64, count: usize, out_xy: *mut u64, out_infinity: *mut u8) -> c_int;
    pub fn zkgpu_msm_kernels_dev(ctx: *mut zkgpu_ctx, scalars_dev: *const c_void, n: usize) -> c_int;
    pub fn zkgpu_msm_profile_dev(ctx: *mut zkgpu_ctx, scalars_dev: *const c_void, n: usize, phase_ms: *mut f32) -> c_int;
    pub fn zkgpu_commit(ctx: *mut zkgpu_ctx, coeffs: *const u64, len: usize, out_xy: *mut u64, out_infinity: *mut u8)
                        -> c_int;
    pub fn zkgpu_commit_many(ctx: *mut zkgpu_ctx, count: u32, coeffs: *const *const u64, lens: *const usize, out_xy: *mut u64,
                             out_infinity: *mut u8) -> c_int;
    pub fn zkgpu_field_mul_selfcheck(field: c_int, a: *const u64, b: *const u64, count: usize, out: *mut u64) -> c_int;
    pub fn zkgpu_field_inv_selfcheck(field: c_int, a: *const u64, count: usize, out: *mut u64) -> c_int;
    pub fn zkgpu_g1_bucket_selfcheck(xy: *const u64, infinity: *const u8, negate: *const u8, n: usize, out_xyz: *mut u64,
                                     violations: *mut u64) -> c_int;
    pub fn zkgpu_g1_bucket_selfcheck_tree(xy: *const u64, infinity: *const u8, negate: *const u8, n: usize,
                                          out_xyz: *mut u64, violations: *mut u64) -> c_int;
    pub fn zkgpu_g1_compress(xy: *const u64, infinity: u8, out: *mut u8) -> c_int;

    pub fn zkgpu_prover_set_sharding(ctx: *mut zkgpu_ctx, shard_base: u64, total_points: u64, world: u32,
                                     exchange: Option<unsafe extern "C" fn(*mut c_void, *const u64, usize, *mut u64) -> c_int>,
                                     user: *mut c_void) -> c_int;
    pub fn zkgpu_set_collective(ctx: *mut zkgpu_ctx, rank: u32, world: u32,
                                fn_: Option<unsafe extern "C" fn(*mut c_void, c_int, *const c_void, *mut c_void, usize) -> c_int>,
                                user: *mut c_void) -> c_int;
    pub fn zkgpu_ntt_slice_dev(ctx: *mut zkgpu_ctx, in_dev: *const c_void, len_in: usize, out_dev: *mut c_void, log_n: u32,
                               mode: c_int, rank: u32, world: u32) -> c_int;
    pub fn zkgpu_intt_sharded_dev(ctx: *mut zkgpu_ctx, slice_dev: *const c_void, out_dev: *mut c_void, log_n: u32,
                                  mode: c_int) -> c_int;
    pub fn zkgpu_prover_key_load(ctx: *mut zkgpu_ctx, n: u32, constraints: u64, selector_evals: *const u64,
                                 label: *const c_char, label_len: usize, pi_indexes: *const u64, n_pi: usize) -> c_int;
    pub fn zkgpu_prover_load_bytes(ctx: *mut zkgpu_ctx, bytes: *const u8, len: usize, pi_indexes: *const u64, n_pi: usize,
                                   n_out: *mut u32) -> c_int;
    pub fn zkgpu_prover_to_bytes(ctx: *mut zkgpu_ctx, out: *mut u8, cap: usize, written: *mut usize) -> c_int;
    pub fn zkgpu_prover_key_commitments(ctx: *mut zkgpu_ctx, out_xy: *mut u64, out_infinity: *mut u8) -> c_int;
    pub fn zkgpu_prover_key_poly(ctx: *mut zkgpu_ctx, which: c_int, out: *mut u64) -> c_int;
    pub fn zkgpu_prover_key_eval8n(ctx: *mut zkgpu_ctx, which: c_int, out: *mut u64) -> c_int;
    pub fn zkgpu_prover_key_eval_len(ctx: *const zkgpu_ctx) -> usize;
    pub fn zkgpu_g1_decompress(input: *const u8, out_xy: *mut u64, out_infinity: *mut u8) -> c_int;
    pub fn zkgpu_proof_to_bytes(comms_xy: *const u64, comms_infinity: *const u8, evals: *const u64, out: *mut u8) -> c_int;
    pub fn zkgpu_proof_from_bytes(input: *const u8, comms_xy: *mut u64, comms_infinity: *mut u8, evals: *mut u64) -> c_int;
    pub fn zkgpu_prover_force_full_coset(ctx: *mut zkgpu_ctx, on: c_int) -> c_int;
    pub fn zkgpu_prover_last_proof_coset(ctx: *const zkgpu_ctx) -> c_int;
    pub fn zkgpu_quotient(ctx: *mut zkgpu_ctx, polys: *const u64, lens: *const usize, challenges7: *const u64,
                          t_out: *mut u64, t_len: *mut usize) -> c_int;
    pub fn zkgpu_quotient_kernel_ms(ctx: *mut zkgpu_ctx, ms: *mut f32) -> c_int;
    pub fn zkgpu_perm_z(ctx: *mut zkgpu_ctx, n: u32, wires: *const u64, sigma_evals: *const u64, beta: *const u64,
                        gamma: *const u64, z_out: *mut u64) -> c_int;
    pub fn zkgpu_poly_eval(ctx: *mut zkgpu_ctx, coeffs: *const u64, len: usize, x: *const u64, out: *mut u64) -> c_int;
    pub fn zkgpu_poly_ruffini(ctx: *mut zkgpu_ctx, coeffs: *const u64, len: usize, root: *const u64, out: *mut u64)
                              -> c_int;
    pub fn zkgpu_linearization(ctx: *mut zkgpu_ctx, polys: *const u64, lens: *const usize, challenges7: *const u64,
                               z_challenge: *const u64, evals15: *const u64, public_inputs: *const u64, n_pi: usize,
                               r_out: *mut u64, r_cap: usize, r_len: *mut usize) -> c_int;
    pub fn zkgpu_aggregate_witness(ctx: *mut zkgpu_ctx, polys: *const u64, lens: *const usize, n_polys: usize,
                                   point: *const u64, v: *const u64, out: *mut u64, out_len: *mut usize) -> c_int;
    pub fn zkgpu_prove(ctx: *mut zkgpu_ctx, wires: *const u64, public_inputs: *const u64, n_pi: usize,
                       blinders14: *const u64, out_comms_xy: *mut u64, out_comms_infinity: *mut u8,
                       out_evals15: *mut u64, out_challenges10: *mut u64, timings8: *mut f32) -> c_int;
    pub fn zkgpu_ctx_clone(src: *mut zkgpu_ctx, out: *mut *mut zkgpu_ctx) -> c_int;
    pub fn zkgpu_device_bytes_shared(ctx: *const zkgpu_ctx) -> usize;
    pub fn zkgpu_witness_prefetch(ctx: *mut zkgpu_ctx, wires_host: *const u64) -> c_int;
    pub fn zkgpu_host_alloc(bytes: usize) -> *mut c_void;
    pub fn zkgpu_host_free(p: *mut c_void);
    pub fn zkgpu_merkle_batch_circuit_size(height: u32, k: u32, constraints: *mut u64, n: *mut u64) -> c_int;
    pub fn zkgpu_merkle_batch_circuit(height: u32, k: u32, leaves: *const u64, branches: *const u64, positions: *const u8,
                                      roots: *const u64, arc: *const u64, mds: *const u64, tag: *const u64, n: u64,
                                      selectors: *mut u64, sigma_map: *mut u64, wires: *mut u64, pi_indexes: *mut u64,
                                      public_inputs: *mut u64, threads: u32) -> c_int;
    pub fn zkgpu_merkle_batch_witness(height: u32, k: u32, leaves: *const u64, branches: *const u64, positions: *const u8,
                                      roots: *const u64, arc: *const u64, mds: *const u64, tag: *const u64, n: u64,
                                      wires: *mut u64, public_inputs: *mut u64, threads: u32) -> c_int;
    pub fn zkgpu_poseidon_merkle4(inputs: *const u64, count: usize, arc: *const u64, mds: *const u64, tag: *const u64,
                                  out: *mut u64) -> c_int;
    pub fn zkgpu_transcript_challenge(label: *const c_char, msg_label: *const c_char, msg: *const u8, msg_len: usize,
                                      ch_label: *const c_char, out: *mut u8, out_len: usize) -> c_int;
}
