// GPU-resident restatement of the reference's prover orchestration: Compiler::preprocess
// (plonk/src/compiler.rs:78-390) builds the device-resident ProverKey, Prover::prove
// (plonk/src/compiler/prover.rs:210-509) runs the five rounds with the transcript, blinders and the 15 scalar
// evaluations on the host and every polynomial on the device.
#pragma once
#include "common.hpp"
#include "g1.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include "poly.hpp"
#include "transcript.hpp"
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace zk {

struct AffineOut {
    Fq x, y;
    bool infinity;
};

// G1Affine::from_bytes (coset-bls12_381/src/g1/coset.rs:73-140) on the host: false where the reference returns Err
bool g1_decompress(const uint8_t in[48], AffineOut &out);

struct ProofOut {
    AffineOut comms[11];  // a b c d z t_low t_mid t_high t_fourth w_z w_zw   (proof_system/proof.rs:29-65)
    Fr evals[15];         // ProofEvaluations field order (linearization_poly.rs:29-74)
    Fr challenges[10];    // beta gamma alpha range logic fixed var z v v_w
};

struct ProveTimings {
    float ms[8];  // wire commits, permutation, quotient, quotient commits, evaluations, linearization, openings, total
};

class Prover {
   public:
    Prover(hipStream_t s, NttEngine &ntt, MsmEngine &msm, PolyEngine &poly) : stream_(s), ntt_(ntt), msm_(msm), poly_(poly) {}
    ~Prover();

    // Compiler::preprocess for a circuit given by its gate-wise selector values: selector_evals holds N_SEL x n
    // Fr elements in Sel order (S1..S4 = the sigma values of build_permutation_lagrange_mapping,
    // composer/permutation.rs:111-141).  Requires the SRS to be loaded.
    void load_key(uint32_t n, uint64_t constraints, const uint64_t *selector_evals_host, const std::string &label,
                  const uint64_t *pi_indexes, size_t n_pi);
    // Prover::try_from_bytes (compiler/prover.rs:137-207) straight into device memory: label, ProverKey
    // (widget.rs:374-519), CommitKey (kzg10/key.rs:54-66, replaces the loaded SRS) and VerifierKey
    // (widget.rs:102-123).  The reference keeps the public input positions in the circuit, not in these bytes, so
    // they are passed beside them.  Malformed input throws ZKGPU_ERR_INVALID_ARG where the reference returns Err.
    void load_key_bytes(const uint8_t *bytes, size_t len, const uint64_t *pi_indexes, size_t n_pi);
    // Prover::to_bytes (compiler/prover.rs:85-134) of the loaded key; returns the size and writes when cap suffices.
    size_t to_bytes(uint8_t *out, size_t cap);
    // Point-range sharding of every commitment over several contexts (one per GPU): `fn(user, mine, m, all)` must
    // all-gather m partial sums of 18 limbs each into all[rank][m][18]; nullptr switches sharding off.
    typedef int (*ExchangeFn)(void *user, const uint64_t *mine, size_t m, uint64_t *all);
    void set_sharding(uint64_t shard_base, uint64_t total_points, uint32_t world, ExchangeFn fn, void *user);
    // Shares the 8n-point work of a proof among `world` contexts (one per GPU): every key vector over the 8n coset and
    // every round polynomial's coset evaluations are held as the cyclic slice i = rank + world * k, the quotient is
    // evaluated on that slice, and its inverse transform goes through the caller's all-to-all and all-gather
    // (NttEngine::inverse_sharded).  Takes effect at the next load_key; 2, 4 or 8 ranks, otherwise the 8n work stays
    // replicated.  fn == nullptr switches it off.
    void set_collective(uint32_t rank, uint32_t world, Collective::Fn fn, void *user);
    // number of elements of one key vector over the coset held by this context: 8n, or 8n / world when shared
    size_t eval_len() const { return ev_len_; }
    bool loaded() const { return n_ != 0; }
    uint32_t n() const { return n_; }
    const AffineOut *selector_comms() const { return sel_comms_; }
    // device pointers to the key (for the piecewise entry points and tests)
    const Fr *key_poly(int k) const { return reinterpret_cast<const Fr *>(polys_.p) + (size_t)k * n_; }
    // k < N_SEL selectors, N_SEL linear evaluations, N_SEL + 1 L1.  A key that proves over the 4n coset (see
    // half_coset()) builds its 8n vectors at the first call that needs them and keeps them from then on.
    const Fr *key_eval8n(int k);
    // True when proofs of the loaded key evaluate the quotient's numerator on the 4n-point coset g<w_4n> (every second
    // point of the reference's 8n coset) instead of all 8n points: keys whose range, logic and curve selectors are
    // identically zero, held by one context, n >= 16.  See Prover::quotient_half_coset.
    bool half_coset() const { return half_coset_; }
    // which of the two the most recent prove() took (false after a proof that fell back to the 8n coset)
    bool last_proof_half_coset() const { return last_half_; }
    bool proved_once() const { return proved_once_; }
    // on = true: keys loaded from now on prove over the full 8n coset whatever their gates (the reference's own route;
    // measurements and tests compare the two)
    void force_full_coset(bool on) { force_full_ = on; }
    bool force_full() const { return force_full_; }

    // build_quotient_polynomial (quotient_poly.rs:15-100) on device polynomials; t_out must hold 8n elements.
    // Returns the coefficient count after truncating leading zeros.
    // With pi == nullptr the public-input polynomial on the coset is sum_j pi_values[j] * L_idx(j) over the resident
    // Lagrange vectors of the key's public-input positions (same values, no transform).
    uint64_t quotient(const Fr *z, uint64_t z_len, const Fr *const wires[4], const uint64_t wire_len[4], const Fr *pi,
                      uint64_t pi_len, const Fr challenges7[7], Fr *t_out, const Fr *pi_values = nullptr);

    // build_linearization_polynomial (linearization_poly.rs:146-272) and compute_aggregate_witness
    // (kzg10/key.rs:131-146) on device polynomials; see prover.hip
    uint64_t linearization(const Fr ch7[7], const Fr &z_ch, const Fr *evals15, const Fr *p_z, uint64_t z_len,
                           const Fr *const t_parts[4], const uint64_t t_lens[4], const std::vector<Fr> &public_inputs, Fr *r_out,
                           uint64_t r_cap, uint64_t z_len_without_leading_zeros = 0);
    uint64_t aggregate_witness(const Fr *const *polys, const uint64_t *lens, int count, const Fr &point, const Fr &v, Fr *sum_tmp,
                               Fr *out);
    PolyEngine &poly() { return poly_; }

    // Prover::prove.  wires_host: 4 x n values; blinders: 14 scalars in the reference's draw order.
    // Returns 0 or a ZKGPU_ERR_* code for the reference's error paths.
    int prove(const uint64_t *wires_host, const uint64_t *public_inputs_host, size_t n_pi, const uint64_t *blinders14,
              ProofOut &out, ProveTimings *timings);

    // Starts the upload of a coming proof's witness (4 x n wire values in host memory, pinned for the copy to run
    // beside the device's work) on a copy stream of the prover's own, into one of two device buffers; returns at once.
    // The prove() call that is later given the same pointer takes the uploaded copy instead of copying itself.  The
    // reference proves one fresh witness after the other (merkle-plonk/src/lib.rs:275-332): prefetching witness i + 1
    // right before prove(witness i) hides its upload behind that proof.  At most two uploads are outstanding (a
    // third reuses the older buffer); the host values must stay unchanged until the proof that consumes them returns.
    void prefetch_witness(const uint64_t *wires_host);

    size_t device_bytes() const;
    // Makes this prover use the key `src` has loaded, by reference: coefficient vectors, coset vectors, sigma values and
    // resident public-input vectors are read-only after the load, so the contexts of one GPU share one copy
    // (#[derive(Clone)] Prover, plonk/src/compiler/prover.rs:20-21); everything a proof writes stays per prover.  Coset
    // vectors that `src` built after its load (ensure_evals8n) are not shared: this prover builds its own when it needs
    // them.  Not for keys loaded under set_collective.  Same device only.
    void share_key(const Prover &src);
    // bytes of the key vectors this prover refers to together with others
    size_t shared_key_bytes() const;

   private:
    // CommitKey::commit (kzg10/key.rs:108-129) of a device polynomial
    int commit_dev(const Fr *coeffs, uint64_t len, AffineOut &out);
    // `meanwhile`, when given, is called once the sums are on the stream: device work it launches runs while the host
    // waits for the sums and finishes them
    int commit_many(uint32_t m, const Fr *const *coeffs, const uint64_t *lens, AffineOut *out,
                    const std::function<void()> &meanwhile = nullptr);
    void blind_poly(const Fr *evals_dev, int hiding_degree, const Fr *blinders, Fr *out_dev);
    void build_vanishing();
    // coset evaluations of the 15 key polynomials, of X and of L1 into evals[k * stride]: over the 4n coset (half) or as
    // coset_fft_8n does it; tmp holds at least n elements
    void build_key_evals(bool half, Fr *evals, size_t stride, Fr *tmp, bool with_selectors, bool with_linear);
    void ensure_evals8n();
    void choose_coset();
    void read_sigma_tops();
    void build_pi_lagrange();
    void coset_fft_4n(const Fr *coeffs, uint64_t len, Fr *out);
    uint64_t quotient_half_coset(const Fr *z, const Fr *const wires[4], const Fr ch7[7], const Fr *pi_values, const Fr *pi_poly,
                                 Fr *t_out);
    // coset_fft over the 8n domain (fft/domain.rs:176-185) of a polynomial held in full: all 8n evaluations, or this
    // context's cyclic slice of them when the key was loaded under set_collective
    void coset_fft_8n(const Fr *coeffs, uint64_t len, Fr *out);
    bool shared() const { return key_world_ > 1; }
    // z, the four wires and the public inputs on this context's coset points, plus the quotient's slice when shared
    size_t coset_buffer_bytes() const { return (6 + (shared() ? 1 : 0)) * ev_len_ * sizeof(Fr); }

    hipStream_t stream_;
    NttEngine &ntt_;
    MsmEngine &msm_;
    PolyEngine &poly_;
    // prefetch_witness: two witness buffers filled on copy_stream_; ready = behind the upload, consumed = behind the
    // last kernel of the proof that read the buffer (the next upload into it waits for that)
    hipStream_t copy_stream_ = nullptr;
    DevBuf wit_buf_[2];
    const uint64_t *wit_src_[2] = {nullptr, nullptr};
    hipEvent_t wit_ready_[2] = {nullptr, nullptr}, wit_consumed_[2] = {nullptr, nullptr};
    int wit_next_ = 0;
    ExchangeFn exchange_ = nullptr;
    void *exchange_user_ = nullptr;
    uint64_t shard_base_ = 0, total_points_ = 0;
    uint32_t world_ = 1;
    Collective coll_;                     // as set by set_collective
    uint32_t key_world_ = 1, key_rank_ = 0;   // how the loaded key's coset vectors are dealt out
    size_t ev_len_ = 0;
    uint32_t n_ = 0, log_n_ = 0;
    uint64_t constraints_ = 0;
    uint64_t vk_n_ = 0;   // VerifierKey.n: the constraint count for a compiled key (compiler.rs:211-212), the stored value
                          // for one loaded from bytes (widget.rs:102-106); closes the transcript seed (widget.rs:227)
    std::string label_;
    std::unique_ptr<MerlinTranscript> seed_tr_;   // the transcript after seed_transcript (widget.rs:199-228) for the loaded key
    std::vector<uint64_t> pi_indexes_;
    uint64_t poly_len_[N_SEL] = {0};
    uint32_t widgets_ = 0;
    Fr vh_[8], vh_inv_[8];   // the 8 distinct values of Z_H on the 8n coset and their inverses
    AffineOut sel_comms_[N_SEL];
    // The key's device vectors are owned by key_mem_ once a load has succeeded (polys_ ... sel_h_ below are views of its
    // buffers) and may be shared with other provers; evals_private_ marks 8n vectors this prover built for itself later.
    struct KeyMem {
        DevBuf polys, evals, sigma, pi_lagrange, evals4, sel_h;
        ~KeyMem();
    };
    std::shared_ptr<KeyMem> key_mem_;
    bool evals_private_ = false;
    void detach_key();    // before a load: the views stop referring to a published key
    void publish_key();   // after a load: the buffers the views own move into a fresh key_mem_
    DevBuf polys_;   // N_SEL x n coefficients
    DevBuf evals_;   // (N_SEL + 2) x 8n: selector coset evaluations, X on the coset, L1 on the coset
    DevBuf sigma_;   // 4 x n sigma values (their FFT is what compute_permutation_vec recomputes)
    DevBuf work_;    // per-proof buffers
    DevBuf small_;   // 15 evaluations etc.
    hipEvent_t sums_ready_ = nullptr;   // behind the copy of a commitment's group sums
    // coset evaluations prove() has already launched into small_ for the coming quotient(): bit 0 z, bit 1 the wires
    uint32_t ev_ready_ = 0;
    // Proving over the 4n coset
    bool half_coset_ = false, last_half_ = false, force_full_ = false, proved_once_ = false;
    bool evals8_built_ = false;   // evals_ holds the 8n vectors (always true for keys that do not use the 4n coset)
    DevBuf evals4_;               // (N_SEL + 2) x 4n: the key's vectors over the 4n coset
    DevBuf sel_h_;                // 7 x n: q_m q_l q_r q_o q_4 q_c q_arith on the domain itself (the row-wise gate check);
                                  // 11 x n with the four widget selectors for a key that has widget gates
    DevBuf tops_dev_;             // gathered top coefficients + the gate check's flag, copied to tops_host_ in one piece
    Fr *tops_host_ = nullptr;     // pinned: 5 windows of 7 coefficients, then the flag word
    Fr sigma_top_[4][7];          // coefficients n+1-j (j < 7) of the four sigma polynomials
    // coefficients n-1-j (j < 7) of the selector polynomials that widget terms multiply (keys with widget gates)
    static const int N_SEL_TOPS = 7;
    static constexpr int SEL_TOPS[N_SEL_TOPS] = {Q_RANGE, Q_LOGIC, Q_FIXED, Q_VAR, Q_L, Q_R, Q_C};
    Fr sel_top_[N_SEL_TOPS][7];
    DevBuf pi_lagrange_;                  // for keys with few public inputs: L_j on the coset the key proves over, one vector per position
    std::vector<uint64_t> pi_lagrange_idx_;   // the distinct positions, ascending; empty = not resident
};

}  // namespace zk
