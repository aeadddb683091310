// Host orchestration of preprocess and prove over the device engines (see prover.hpp).
#include "prover.hpp"
#include <map>
#include "widgets.hpp"
#include <algorithm>
#include <chrono>
#include <cstring>

namespace zk {

// ---- host field helpers ---------------------------------------------------------------------------------------
static Fr fr_from_words(const uint64_t w[4]) {
    Fr r;
    memcpy(r.v, w, 32);
    return r;
}
// Scalar::from_bytes_wide (coset-bls12_381/src/scalar.rs:282-303): d0 * R^2 + d1 * R^3
static Fr fr_from_bytes_wide(const uint8_t b[64]) {
    Fr d0, d1, r3;
    memcpy(d0.v, b, 32);
    memcpy(d1.v, b + 32, 32);
    const uint64_t R3[4] = {0xc62c1807439b73afULL, 0x1b3e0d188cf06990ULL, 0x73d13c71c7b5f418ULL, 0x6e2a5bb9c8db33e9ULL};
    memcpy(r3.v, R3, 32);
    return Fr::add(Fr::mul(d0, Fr::r2()), Fr::mul(d1, r3));
}
static Fr fr_u64(uint64_t x) {
    Fr o = Fr::zero();
    o.v[0] = (uint32_t)x;
    o.v[1] = (uint32_t)(x >> 32);
    return Fr::to_mont(o);
}

// plonk/src/transcript.rs:30-66
struct PlonkTranscript {
    MerlinTranscript t;
    explicit PlonkTranscript(const std::string &label) : t(label) {}
    explicit PlonkTranscript(const MerlinTranscript &from) : t(from) {}
    void append_commitment(const char *label, const AffineOut &c) {
        uint64_t xy[12];
        uint8_t out[48];
        memcpy(xy, c.x.v, 48);
        memcpy(xy + 6, c.y.v, 48);
        zkgpu_g1_compress(xy, c.infinity ? 1 : 0, out);
        t.append_message(label, out, 48);
    }
    void append_scalar(const char *label, const Fr &s) {
        Fr c = Fr::from_mont(s);  // Scalar::to_bytes: canonical little-endian
        t.append_message(label, c.v, 32);
    }
    Fr challenge_scalar(const char *label) {
        uint8_t buf[64];
        t.challenge_bytes(label, buf, 64);
        return fr_from_bytes_wide(buf);
    }
    void circuit_domain_sep(uint64_t n) {
        t.append_message("dom-sep", "circuit_size", 12);
        t.append_u64("n", n);
    }
};

static AffineOut to_affine(const G1Proj &p) {
    AffineOut o;
    g1_to_affine(p, o.x, o.y, o.infinity);
    return o;
}
static AffineOut affine_identity() {
    AffineOut o;
    o.x = Fq::zero();
    o.y = Fq::one();
    o.infinity = true;
    return o;
}

// util::batch_inversion semantics on a handful of host scalars (zeros stay zero)
static void host_batch_inverse(std::vector<Fr> &v) {
    for (auto &x : v) x = Fr::inv(x);
}

// ---------------------------------------------------------------------------------------------------------------
Prover::KeyMem::~KeyMem() {
    polys.release();
    evals.release();
    sigma.release();
    pi_lagrange.release();
    evals4.release();
    sel_h.release();
}

void Prover::detach_key() {
    if (!key_mem_) return;   // the views own their buffers (a load that failed half-way): reused or freed by the next one
    if (evals_private_) evals_.release();
    evals_private_ = false;
    polys_ = evals_ = sigma_ = pi_lagrange_ = evals4_ = sel_h_ = DevBuf();
    key_mem_.reset();
}

void Prover::publish_key() {
    key_mem_ = std::make_shared<KeyMem>();
    key_mem_->polys = polys_;
    key_mem_->evals = evals_;
    key_mem_->sigma = sigma_;
    key_mem_->pi_lagrange = pi_lagrange_;
    key_mem_->evals4 = evals4_;
    key_mem_->sel_h = sel_h_;
    evals_private_ = false;
}

void Prover::share_key(const Prover &src) {
    ZK_REQUIRE(src.loaded(), ZKGPU_ERR_NOT_LOADED, "clone: the source context has no prover key");
    ZK_REQUIRE(!src.shared() && src.key_mem_, ZKGPU_ERR_INVALID_ARG, "clone: a key loaded under set_collective is not shared");
    detach_key();
    polys_.release();   // what a failed load may have left in the views
    evals_.release();
    sigma_.release();
    pi_lagrange_.release();
    evals4_.release();
    sel_h_.release();
    n_ = 0;
    key_mem_ = src.key_mem_;
    polys_ = src.polys_;
    sigma_ = src.sigma_;
    pi_lagrange_ = src.pi_lagrange_;
    evals4_ = src.evals4_;
    sel_h_ = src.sel_h_;
    evals_private_ = false;
    if (src.evals_private_) {   // built by src for itself after its load
        evals_ = DevBuf();
        evals8_built_ = false;
    } else {
        evals_ = src.evals_;
        evals8_built_ = src.evals8_built_;
    }
    key_world_ = 1;
    key_rank_ = 0;
    ev_len_ = src.ev_len_;
    log_n_ = src.log_n_;
    constraints_ = src.constraints_;
    vk_n_ = src.vk_n_;
    label_ = src.label_;
    seed_tr_.reset(src.seed_tr_ ? new MerlinTranscript(*src.seed_tr_) : nullptr);
    pi_indexes_ = src.pi_indexes_;
    for (int k = 0; k < N_SEL; k++) {
        poly_len_[k] = src.poly_len_[k];
        sel_comms_[k] = src.sel_comms_[k];
    }
    widgets_ = src.widgets_;
    for (int k = 0; k < 8; k++) {
        vh_[k] = src.vh_[k];
        vh_inv_[k] = src.vh_inv_[k];
    }
    half_coset_ = src.half_coset_;
    force_full_ = src.force_full_;
    last_half_ = false;
    proved_once_ = false;
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 7; j++) sigma_top_[k][j] = src.sigma_top_[k][j];
    for (int k = 0; k < N_SEL_TOPS; k++)
        for (int j = 0; j < 7; j++) sel_top_[k][j] = src.sel_top_[k][j];
    pi_lagrange_idx_ = src.pi_lagrange_idx_;
    n_ = src.n_;
}

size_t Prover::shared_key_bytes() const {
    if (!key_mem_ || key_mem_.use_count() < 2) return 0;
    return polys_.cap + sigma_.cap + pi_lagrange_.cap + evals4_.cap + sel_h_.cap + (evals_private_ ? 0 : evals_.cap);
}

Prover::~Prover() {
    if (key_mem_) {
        detach_key();
    } else {
        polys_.release();
        evals_.release();
        sigma_.release();
        pi_lagrange_.release();
        evals4_.release();
        sel_h_.release();
    }
    work_.release();
    small_.release();
    tops_dev_.release();
    if (tops_host_) (void)hipHostFree(tops_host_);
    if (sums_ready_) (void)hipEventDestroy(sums_ready_);
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    for (int s = 0; s < 2; s++) {
        wit_buf_[s].release();
        if (wit_ready_[s]) (void)hipEventDestroy(wit_ready_[s]);
        if (wit_consumed_[s]) (void)hipEventDestroy(wit_consumed_[s]);
    }
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
}

void Prover::prefetch_witness(const uint64_t *wires_host) {
    ZK_REQUIRE(loaded(), ZKGPU_ERR_NOT_LOADED, "witness_prefetch: no prover key loaded");
    ZK_REQUIRE(wires_host != nullptr, ZKGPU_ERR_INVALID_ARG, "witness_prefetch: null input");
    if (!copy_stream_) ZK_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    const int s = wit_next_;
    wit_next_ ^= 1;
    if (!wit_ready_[s]) {
        ZK_HIP(hipEventCreateWithFlags(&wit_ready_[s], hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&wit_consumed_[s], hipEventDisableTiming));
    }
    wit_src_[s] = nullptr;
    const size_t bytes = 4 * (size_t)n_ * sizeof(Fr);
    if (wit_buf_[s].cap < bytes) {
        ZK_HIP(hipStreamSynchronize(copy_stream_));   // an upload into the buffer about to be replaced
        ZK_HIP(hipStreamSynchronize(stream_));        // a proof still reading it
        wit_buf_[s].reserve(bytes);
    }
    ZK_HIP(hipStreamWaitEvent(copy_stream_, wit_consumed_[s], 0));   // no-op until a proof has read this buffer
    ZK_HIP(hipMemcpyAsync(wit_buf_[s].p, wires_host, bytes, hipMemcpyHostToDevice, copy_stream_));
    ZK_HIP(hipEventRecord(wit_ready_[s], copy_stream_));
    wit_src_[s] = wires_host;
}

size_t Prover::device_bytes() const {
    return polys_.cap + evals_.cap + sigma_.cap + work_.cap + small_.cap + pi_lagrange_.cap + evals4_.cap + sel_h_.cap +
           tops_dev_.cap + wit_buf_[0].cap + wit_buf_[1].cap;
}

const Fr *Prover::key_eval8n(int k) {
    ensure_evals8n();
    return reinterpret_cast<const Fr *>(evals_.p) + (size_t)k * ev_len_;
}

void Prover::coset_fft_4n(const Fr *coeffs, uint64_t len, Fr *out) {
    ntt_.transform(NTT_COSET_FFT, coeffs, len, out, log_n_ + 2);
}

void Prover::set_collective(uint32_t rank, uint32_t world, Collective::Fn fn, void *user) {
    ZK_REQUIRE(fn == nullptr || (world >= 1 && rank < world), ZKGPU_ERR_INVALID_ARG, "set_collective: rank outside the world");
    coll_.fn = fn;
    coll_.user = user;
    coll_.rank = fn ? rank : 0;
    coll_.world = fn ? world : 1;
}

void Prover::coset_fft_8n(const Fr *coeffs, uint64_t len, Fr *out) {
    if (shared())
        ntt_.slice_forward(true, coeffs, len, out, log_n_ + 3, key_rank_, key_world_);
    else
        ntt_.transform(NTT_COSET_FFT, coeffs, len, out, log_n_ + 3);
}

int Prover::commit_dev(const Fr *coeffs, uint64_t len, AffineOut &out) { return commit_many(1, &coeffs, &len, &out); }

// Several commitments over the same SRS in one batched MSM.  The reference commits one polynomial after the other
// and returns the first error (prover.rs:262-265, 344-347): the degree checks run in that order before any work.
// With sharding set, this context holds the points [shard_base_, shard_base_ + n_local) of an SRS of total_points_
// points: it sums its own range of every polynomial, the partial sums of all ranks are exchanged through the
// caller's function and added in rank order, so every rank obtains the same commitments.
int Prover::commit_many(uint32_t m, const Fr *const *coeffs, const uint64_t *lens, AffineOut *out,
                        const std::function<void()> &meanwhile) {
    // The lengths without leading zeros decide the reference's degree errors; they are read back together with the
    // sums, which are taken over the given lengths meanwhile (zero coefficients add nothing to a sum).
    unsigned long long *eff = static_cast<unsigned long long *>(msm_.pinned_scratch());
    poly_.lengths_enqueue(m, coeffs, lens, eff);
    const uint64_t total_points = exchange_ ? total_points_ : msm_.n_points();
    // this context's share of every sum
    const uint64_t lo = exchange_ ? shard_base_ : 0, local_points = msm_.n_points();
    const Fr *ptr[MsmEngine::MAX_BATCH];
    size_t cnt[MsmEngine::MAX_BATCH];
    uint32_t slot[MsmEngine::MAX_BATCH], k = 0;
    G1Proj sums[MsmEngine::MAX_BATCH];
    for (uint32_t j = 0; j < m; j++) {
        sums[j] = G1Proj::identity();
        uint64_t hi = lens[j] < lo + local_points ? lens[j] : lo + local_points;
        if (hi <= lo) continue;
        ptr[k] = coeffs[j] + lo;
        cnt[k] = (size_t)(hi - lo);
        slot[k++] = j;
    }
    uint32_t groups = 0, log_seg = 0;
    if (k) {
        msm_.msm_batch_device(k, ptr, cnt, 0, groups, log_seg);
        msm_.enqueue_result_copy(k, groups);
    }
    if (!sums_ready_) ZK_HIP(hipEventCreateWithFlags(&sums_ready_, hipEventDisableTiming));
    ZK_HIP(hipEventRecord(sums_ready_, stream_));
    if (meanwhile) meanwhile();
    ZK_HIP(hipEventSynchronize(sums_ready_));
    for (uint32_t j = 0; j < m; j++) {
        const uint64_t degree = eff[j] ? eff[j] - 1 : 0;
        if (degree == 0) return ZKGPU_ERR_DEGREE_ZERO;
        if (degree > total_points - 1) return ZKGPU_ERR_DEGREE_LARGE;
    }
    if (k) {
        G1Proj part[MsmEngine::MAX_BATCH];
        msm_.finish_host(k, groups, log_seg, part);
        for (uint32_t i = 0; i < k; i++) sums[slot[i]] = part[i];
    }
    if (exchange_) {
        std::vector<uint64_t> mine((size_t)m * 18), all((size_t)world_ * m * 18);
        for (uint32_t j = 0; j < m; j++) memcpy(mine.data() + 18 * j, &sums[j], sizeof(G1Proj));
        const int rc = exchange_(exchange_user_, mine.data(), m, all.data());
        ZK_REQUIRE(rc == 0, ZKGPU_ERR_INTERNAL, "commit: the partial-sum exchange failed");
        for (uint32_t j = 0; j < m; j++) {
            G1Proj total = G1Proj::identity();
            for (uint32_t r = 0; r < world_; r++) {
                G1Proj p;
                memcpy(&p, all.data() + ((size_t)r * m + j) * 18, sizeof p);
                total = g1_add(total, p);
            }
            sums[j] = total;
        }
    }
    // one inversion for the whole batch
    Fq xs[MsmEngine::MAX_BATCH], ys[MsmEngine::MAX_BATCH];
    bool inf[MsmEngine::MAX_BATCH];
    g1_to_affine_many(sums, m, xs, ys, inf);
    for (uint32_t j = 0; j < m; j++) {
        out[j].x = xs[j];
        out[j].y = ys[j];
        out[j].infinity = inf[j];
    }
    return 0;
}

void Prover::set_sharding(uint64_t shard_base, uint64_t total_points, uint32_t world, ExchangeFn fn, void *user) {
    ZK_REQUIRE(fn == nullptr || (world >= 1 && shard_base + msm_.n_points() <= total_points), ZKGPU_ERR_INVALID_ARG,
               "prover_set_sharding: this context's range lies outside the SRS");
    exchange_ = fn;
    exchange_user_ = user;
    shard_base_ = shard_base;
    total_points_ = total_points;
    world_ = fn ? world : 1;
}

// compiler.rs:241-301: the selector and sigma polynomials over the coset; compiler.rs:303-306: linear_evaluations =
// coset_fft([0, 1]); L1 = coset_fft(ifft([1, 0, 0, ...])) (quotient_poly.rs:237-245 builds alpha^2 * L1 per proof;
// alpha^2 is applied inside the quotient kernel).
void Prover::build_key_evals(bool half, Fr *evals, size_t stride, Fr *tmp, bool with_selectors, bool with_linear) {
    const uint32_t n = n_;
    auto forward = [&](const Fr *coeffs, uint64_t len, Fr *out) {
        if (half)
            coset_fft_4n(coeffs, len, out);
        else
            coset_fft_8n(coeffs, len, out);
    };
    if (with_selectors)
        for (int k = 0; k < N_SEL; k++) forward(key_poly(k), poly_len_[k], evals + (size_t)k * stride);
    if (with_linear) {
        std::vector<Fr> two(2);
        two[0] = Fr::zero();
        two[1] = Fr::one();
        ZK_HIP(hipMemcpyAsync(tmp, two.data(), 2 * sizeof(Fr), hipMemcpyHostToDevice, stream_));
        ZK_HIP(hipStreamSynchronize(stream_));
        forward(tmp, 2, evals + (size_t)N_SEL * stride);
    }
    Fr one = Fr::one();
    ZK_HIP(hipMemcpyAsync(tmp, &one, sizeof(Fr), hipMemcpyHostToDevice, stream_));
    ZK_HIP(hipStreamSynchronize(stream_));
    ntt_.transform(NTT_IFFT, tmp, 1, tmp, log_n_);
    forward(tmp, n, evals + (size_t)(N_SEL + 1) * stride);
}

// compute_vanishing_poly_over_coset (domain.rs:256-275): v_h[i] = g^n * w_8n^(n i) - 1, period 8.
void Prover::build_vanishing() {
    const uint32_t n = n_;
    Domain d8 = Domain::make((size_t)8 * n);
    Fr g = Fr::inv(d8.generator_inv);
    Fr gn = Fr::pow_u64(g, n);
    Fr w8 = Fr::pow_u64(d8.group_gen, n);
    Fr cur = Fr::one();
    for (int i = 0; i < 8; i++) {
        vh_[i] = Fr::sub(Fr::mul(gn, cur), Fr::one());
        vh_inv_[i] = Fr::inv(vh_[i]);
        cur = Fr::mul(cur, w8);
    }
}

// Which coset the loaded key's proofs evaluate the quotient on.  The numerator has degree at most 5n + 6 (the
// permutation term: z of degree n + 2 times four factors of degree n + 1; a widget's term is a selector times at most
// four wire factors, 5n + 3) and the quotient 4n + 6: its 4n evaluations over g<w_4n> fix it up to its seven top
// coefficients, and those follow from the top coefficients of the round and key polynomials (quotient_half_coset).
void Prover::choose_coset() {
    half_coset_ = !force_full_ && !shared() && n_ >= 16;
    last_half_ = false;
    if (!half_coset_) {
        evals4_.release();
        sel_h_.release();
    }
}

// coefficients n+1-j, j < 7, of the sigma polynomials (they have n coefficients: the first two are zero)
void Prover::read_sigma_tops() {
    const uint32_t n = n_;
    Fr buf[4][5];
    for (int k = 0; k < 4; k++)
        ZK_HIP(hipMemcpyAsync(buf[k], key_poly(S1 + k) + (n - 5), 5 * sizeof(Fr), hipMemcpyDeviceToHost, stream_));
    ZK_HIP(hipStreamSynchronize(stream_));
    for (int k = 0; k < 4; k++) {
        sigma_top_[k][0] = sigma_top_[k][1] = Fr::zero();
        for (int j = 2; j < 7; j++) sigma_top_[k][j] = buf[k][4 - (j - 2)];   // coefficient n+1-j = buf index (n+1-j) - (n-5)
    }
    // coefficients n-1-j, j < 7, of the selector polynomials the widget terms multiply
    if (widgets_) {
        Fr sb[N_SEL_TOPS][7];
        for (int k = 0; k < N_SEL_TOPS; k++)
            ZK_HIP(hipMemcpyAsync(sb[k], key_poly(SEL_TOPS[k]) + (n - 7), 7 * sizeof(Fr), hipMemcpyDeviceToHost, stream_));
        ZK_HIP(hipStreamSynchronize(stream_));
        for (int k = 0; k < N_SEL_TOPS; k++)
            for (int j = 0; j < 7; j++) sel_top_[k][j] = sb[k][6 - j];
    }
}

// The 8n vectors of a key that proves over the 4n coset: needed by the 8n quotient (a proof whose numerator does not
// vanish on the domain, the piecewise entry point) and by to_bytes.  Uses small_ as scratch.
void Prover::ensure_evals8n() {
    if (evals8_built_) return;
    ZK_REQUIRE(loaded(), ZKGPU_ERR_NOT_LOADED, "no prover key loaded");
    if (key_mem_) {   // a published key is read-only (other provers may refer to it): these vectors are this prover's own
        evals_ = DevBuf();
        evals_private_ = true;
    }
    evals_.reserve((size_t)(N_SEL + 2) * ev_len_ * sizeof(Fr));
    small_.reserve(coset_buffer_bytes());
    build_key_evals(false, evals_.as<Fr>(), ev_len_, small_.as<Fr>(), true, true);
    evals8_built_ = true;
}

// The public-input polynomial is sum_j pi_j L_j (composer.rs:1001-1014 + prover.rs:305-306); its 8n coset evaluations
// are what the quotient needs (quotient_poly.rs:46-51).  A circuit has few public inputs at fixed gates, so for up to
// four of them (and at most 8 GiB) the coset evaluations of their Lagrange polynomials stay resident with the key and a
// proof forms the combination instead of running an n-point iFFT and an 8n-point coset FFT.
void Prover::build_pi_lagrange() {
    pi_lagrange_idx_.clear();
    std::vector<uint64_t> idx;
    for (uint64_t i : pi_indexes_)
        if (i < n_) idx.push_back(i);
    std::sort(idx.begin(), idx.end());
    idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
    const size_t N8 = half_coset_ ? (size_t)4 * n_ : ev_len_;   // the coset points this key's proofs evaluate
    if (idx.empty() || idx.size() > 4 || idx.size() * N8 * sizeof(Fr) > (8ull << 30)) return;
    pi_lagrange_.reserve(idx.size() * N8 * sizeof(Fr));
    Fr *tmp = work_.as<Fr>();
    const Fr one = Fr::one();
    for (size_t j = 0; j < idx.size(); j++) {
        poly_.zero_and_scatter(tmp, n_, &idx[j], &one, 1);
        ntt_.transform(NTT_IFFT, tmp, n_, tmp, log_n_);
        if (half_coset_)
            coset_fft_4n(tmp, n_, pi_lagrange_.as<Fr>() + j * N8);
        else
            coset_fft_8n(tmp, n_, pi_lagrange_.as<Fr>() + j * N8);
    }
    ZK_HIP(hipStreamSynchronize(stream_));
    pi_lagrange_idx_ = idx;
}

void Prover::load_key(uint32_t n, uint64_t constraints, const uint64_t *selector_evals_host, const std::string &label,
                      const uint64_t *pi_indexes, size_t n_pi) {
    ZK_REQUIRE(n >= 1 && (n & (n - 1)) == 0, ZKGPU_ERR_INVALID_ARG, "prover_key_load: n must be a power of two");
    ZK_REQUIRE(selector_evals_host != nullptr, ZKGPU_ERR_INVALID_ARG, "prover_key_load: null selectors");
    ZK_REQUIRE(msm_.n_points() > 0, ZKGPU_ERR_NOT_LOADED, "prover_key_load: load the SRS first");
    ZK_REQUIRE((uint64_t)n * 8 < (1ull << 32), ZKGPU_ERR_DOMAIN_SIZE, "InvalidEvalDomainSize: 8n must be below 2^32");
    n_ = 0;
    detach_key();
    seed_tr_.reset();
    uint32_t log_n = 0;
    while ((1u << log_n) < n) log_n++;
    const size_t N8 = (size_t)8 * n;
    // the 8n-point work is shared when a collective is set for 2, 4 or 8 ranks and every rank's share of the
    // inverse transform's transpose is at least one element
    key_world_ = 1;
    key_rank_ = 0;
    if (coll_.usable() && N8 >= (size_t)coll_.world * coll_.world) {
        key_world_ = coll_.world;
        key_rank_ = coll_.rank;
    }
    ev_len_ = N8 / key_world_;
    polys_.reserve((size_t)N_SEL * n * sizeof(Fr));
    sigma_.reserve((size_t)4 * n * sizeof(Fr));
    work_.reserve((size_t)N_SEL * n * sizeof(Fr));
    Fr *stage = work_.as<Fr>();
    Fr *polys = polys_.as<Fr>();
    ZK_HIP(hipMemcpyAsync(stage, selector_evals_host, (size_t)N_SEL * n * sizeof(Fr), hipMemcpyHostToDevice, stream_));
    ZK_HIP(hipMemcpyAsync(sigma_.p, stage + (size_t)S1 * n, (size_t)4 * n * sizeof(Fr), hipMemcpyDeviceToDevice, stream_));
    n_ = n;
    log_n_ = log_n;
    constraints_ = constraints;
    vk_n_ = constraints;   // compiler.rs:211-212
    label_ = label;
    pi_indexes_.assign(pi_indexes, pi_indexes + n_pi);
    widgets_ = 0;
    for (int k = 0; k < N_SEL; k++) {
        // compiler.rs:123-147 (selectors) and permutation.rs:164-171 (sigmas): ifft, leading zeros dropped
        ntt_.transform(NTT_IFFT, stage + (size_t)k * n, n, polys + (size_t)k * n, log_n);
        poly_len_[k] = poly_.length_without_leading_zeros(polys + (size_t)k * n, n);
    }
    if (poly_len_[Q_RANGE]) widgets_ |= 1u;
    if (poly_len_[Q_LOGIC]) widgets_ |= 2u;
    if (poly_len_[Q_FIXED]) widgets_ |= 4u;
    if (poly_len_[Q_VAR]) widgets_ |= 8u;
    choose_coset();
    if (half_coset_) {
        // the selector values on the domain itself stay for the row-wise gate check of every proof
        // (a key with widget gates: also q_range q_logic q_fixed_group_add q_variable_group_add, the next four in Sel)
        const size_t n_sel_h = widgets_ ? 11 : 7;
        sel_h_.reserve(n_sel_h * n * sizeof(Fr));
        ZK_HIP(hipMemcpyAsync(sel_h_.p, stage, n_sel_h * n * sizeof(Fr), hipMemcpyDeviceToDevice, stream_));
        evals4_.reserve((size_t)(N_SEL + 2) * 4 * n * sizeof(Fr));
        build_key_evals(true, evals4_.as<Fr>(), (size_t)4 * n, stage, true, true);
        read_sigma_tops();
        evals8_built_ = false;
        evals_.release();   // the 8n vectors of an earlier key; this key's are built when something asks for them
    } else {
        evals_.reserve((size_t)(N_SEL + 2) * ev_len_ * sizeof(Fr));
        build_key_evals(false, evals_.as<Fr>(), ev_len_, stage, true, true);
        evals8_built_ = true;
    }
    build_vanishing();
    // compiler.rs:152-171: selector commitments default to the identity on error, sigma commitments must succeed
    for (int k = 0; k < N_SEL; k++) {
        int rc = commit_dev(polys + (size_t)k * n, poly_len_[k], sel_comms_[k]);
        if (rc != 0) {
            if (k >= S1) {
                n_ = 0;
                throw ZkError(rc, "prover_key_load: sigma polynomial cannot be committed");
            }
            sel_comms_[k] = affine_identity();
        }
    }
    build_pi_lagrange();
    publish_key();
}

// ---- key (de)serialization --------------------------------------------------------------------------------------
static uint64_t rd_be64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
static uint64_t rd_le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}
static void wr_be64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * (7 - i)));
}
static void wr_le64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}
// Order in which ProverKey::to_var_bytes and VerifierKey::to_bytes walk the selectors (widget.rs:82-97,294-365):
// q_logic before q_range.
static const int SERIAL_ORDER[N_SEL] = {Q_M, Q_L, Q_R, Q_O, Q_F, Q_C, Q_ARITH, Q_LOGIC, Q_RANGE, Q_FIXED, Q_VAR, S1, S2, S3, S4};
static const size_t DOMAIN_BYTES = 172;   // EvaluationDomain::SIZE (fft/domain.rs:46): u64 + u32 + 5 scalars

// y > (p - 1) / 2 for a canonical y (fp.rs:264-278)
static bool fq_lex_largest(const Fq &y_mont) {
    const Fq y = Fq::from_mont(y_mont);
    // 2y > p  <=>  y > (p - 1) / 2 for odd p: compare 2y against p limb by limb
    uint32_t twice[13];
    uint32_t carry = 0;
    for (int i = 0; i < 12; i++) {
        twice[i] = (y.v[i] << 1) | carry;
        carry = y.v[i] >> 31;
    }
    twice[12] = carry;
    if (twice[12]) return true;
    for (int i = 11; i >= 0; i--) {
        if (twice[i] != Fq::mod_limb(i)) return twice[i] > Fq::mod_limb(i);
    }
    return false;
}
// G1Affine::from_bytes (coset-bls12_381/src/g1/coset.rs:73-140): compressed x with the three flag bits, y from the
// curve equation, sign from the sort flag, then the subgroup check.  Returns false where the reference returns Err.
bool g1_decompress(const uint8_t in[48], AffineOut &out) {
    const bool compressed = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (!compressed) return false;
    Fq xc;
    for (int i = 0; i < 12; i++) {
        uint32_t limb = 0;
        for (int b = 0; b < 4; b++) {
            uint8_t byte = in[4 * i + b];
            if (i == 0 && b == 0) byte &= 0x1f;
            limb = (limb << 8) | byte;
        }
        xc.v[11 - i] = limb;
    }
    if (!xc.below_modulus()) return false;
    if (infinity) {
        if (sort || !xc.is_zero()) return false;
        out = affine_identity();
        return true;
    }
    const Fq x = Fq::to_mont(xc);
    Fq four = Fq::zero();
    four.v[0] = 4;
    const Fq rhs = Fq::add(Fq::mul(Fq::mul(x, x), x), Fq::to_mont(four));
    // p = 3 mod 4: the root, if any, is rhs^((p + 1) / 4)
    uint32_t e[12];
    uint64_t carry = 1;
    for (int i = 0; i < 12; i++) {
        uint64_t t = (uint64_t)Fq::mod_limb(i) + carry;
        e[i] = (uint32_t)t;
        carry = t >> 32;
    }
    for (int i = 0; i < 12; i++) e[i] = (e[i] >> 2) | (i + 1 < 12 ? e[i + 1] << 30 : 0);
    Fq y = Fq::pow(rhs, e, 12);
    if (Fq::mul(y, y) != rhs) return false;
    if (fq_lex_largest(y) != sort) y = Fq::neg(y);
    // subgroup check: [r] P is the identity
    G1Affine pa;
    pa.x = x;
    pa.y = y;
    G1Proj acc = G1Proj::identity();
    for (int i = 254; i >= 0; i--) {
        acc = g1_double(acc);
        if ((Fr::mod_limb(i >> 5) >> (i & 31)) & 1) acc = g1_add_mixed(acc, pa);
    }
    if (!acc.z.is_zero()) return false;
    out.x = x;
    out.y = y;
    out.infinity = false;
    return true;
}

void Prover::load_key_bytes(const uint8_t *b, size_t len, const uint64_t *pi_indexes, size_t n_pi) {
    ZK_REQUIRE(b != nullptr && len >= 48, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: NotEnoughBytes (48-byte header)");
    ZK_REQUIRE(!coll_.usable(), ZKGPU_ERR_INVALID_ARG,
               "prover_load_bytes: a serialized key holds whole coset vectors; load it on a context without set_collective");
    const uint64_t label_len = rd_be64(b), pk_len = rd_be64(b + 8), ck_len = rd_be64(b + 16), vk_len = rd_be64(b + 24),
                   size = rd_be64(b + 32), constraints = rd_be64(b + 40);
    const uint64_t rest = len - 48;
    ZK_REQUIRE(label_len <= rest && pk_len <= rest - label_len && ck_len <= rest - label_len - pk_len &&
                   vk_len <= rest - label_len - pk_len - ck_len,
               ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: NotEnoughBytes (pieces longer than the input)");
    const uint8_t *label = b + 48, *pk = label + label_len, *ck = pk + pk_len, *vk = ck + ck_len;

    // VerifierKey::from_slice (widget.rs:102-123)
    ZK_REQUIRE(vk_len >= 8 + 48 * N_SEL, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: verifier key too short");
    AffineOut comms[N_SEL];
    for (int j = 0; j < N_SEL; j++)
        ZK_REQUIRE(g1_decompress(vk + 8 + 48 * j, comms[SERIAL_ORDER[j]]), ZKGPU_ERR_INVALID_ARG,
                   "prover_load_bytes: a verifier-key commitment is not a valid compressed G1 point");

    // ProverKey::from_slice (widget.rs:374-519): header, then walk the pieces to validate before touching the device
    ZK_REQUIRE(pk_len >= 16, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: prover key too short");
    const uint64_t n64 = rd_le64(pk), eval_size = rd_le64(pk + 8);
    ZK_REQUIRE(n64 >= 1 && (n64 & (n64 - 1)) == 0 && n64 * 8 < (1ull << 32), ZKGPU_ERR_DOMAIN_SIZE,
               "prover_load_bytes: the key's size is not a power of two below 2^29");
    ZK_REQUIRE(size == n64, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: header size and prover key size differ");
    const uint32_t n = (uint32_t)n64;
    const size_t N8 = (size_t)8 * n;
    ZK_REQUIRE(eval_size == DOMAIN_BYTES + N8 * 32, ZKGPU_ERR_INVALID_ARG,
               "prover_load_bytes: evaluation vectors are not over the 8n domain");
    const uint8_t *poly_at[N_SEL], *eval_at[N_SEL + 2];
    uint64_t coeffs[N_SEL];
    uint64_t pos = 16;
    for (int j = 0; j < N_SEL; j++) {
        ZK_REQUIRE(pk_len - pos >= 8, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: prover key truncated");
        coeffs[j] = rd_le64(pk + pos);
        pos += 8;
        ZK_REQUIRE(coeffs[j] <= n && (pk_len - pos) / 32 >= coeffs[j] && pk_len - pos - coeffs[j] * 32 >= eval_size,
                   ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: prover key truncated or a polynomial longer than n");
        poly_at[j] = pk + pos;
        pos += coeffs[j] * 32;
        eval_at[j] = pk + pos;
        pos += eval_size;
    }
    for (int j = N_SEL; j < N_SEL + 2; j++) {
        ZK_REQUIRE(pk_len - pos >= eval_size, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: prover key truncated");
        eval_at[j] = pk + pos;
        pos += eval_size;
    }
    for (int j = 0; j < N_SEL + 2; j++)
        ZK_REQUIRE(rd_le64(eval_at[j]) == N8, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: an evaluation vector names another domain");

    // CommitKey::from_slice_unchecked (kzg10/key.rs:54-66)
    n_ = 0;
    detach_key();
    seed_tr_.reset();
    key_world_ = 1;
    key_rank_ = 0;
    ev_len_ = N8;
    msm_.load_points_raw(ck, ck_len);
    ZK_REQUIRE(msm_.n_points() > 0, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: empty commit key");

    uint32_t log_n = 0;
    while ((1u << log_n) < n) log_n++;
    polys_.reserve((size_t)N_SEL * n * sizeof(Fr));
    evals_.reserve((size_t)(N_SEL + 2) * N8 * sizeof(Fr));
    sigma_.reserve((size_t)4 * n * sizeof(Fr));
    work_.reserve((size_t)N_SEL * n * sizeof(Fr));   // >= 8n elements of staging
    Fr *stage = work_.as<Fr>();
    Fr *polys = polys_.as<Fr>();
    Fr *evals = evals_.as<Fr>();
    log_n_ = log_n;
    bool canonical = true;
    for (int j = 0; j < N_SEL + 1; j++) {
        const int k = j < N_SEL ? SERIAL_ORDER[j] : N_SEL;   // the last one is linear_evaluations
        if (j < N_SEL) {
            if (coeffs[j]) ZK_HIP(hipMemcpyAsync(stage, poly_at[j], coeffs[j] * 32, hipMemcpyHostToDevice, stream_));
            canonical &= poly_.from_canonical(stage, coeffs[j], polys + (size_t)k * n, n);
            // Polynomial::from_slice drops leading zeros (fft/polynomial.rs:131-143)
            poly_len_[k] = coeffs[j] ? poly_.length_without_leading_zeros(polys + (size_t)k * n, coeffs[j]) : 0;
        }
        ZK_HIP(hipMemcpyAsync(stage, eval_at[j] + DOMAIN_BYTES, N8 * 32, hipMemcpyHostToDevice, stream_));
        canonical &= poly_.from_canonical(stage, N8, evals + (size_t)k * N8, N8);
    }
    ZK_REQUIRE(canonical, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: a scalar is not below the field modulus");
    n_ = n;
    constraints_ = constraints;
    vk_n_ = rd_le64(vk);   // VerifierKey::from_bytes keeps the stored n (widget.rs:102-106)
    label_.assign(reinterpret_cast<const char *>(label), label_len);
    pi_indexes_.assign(pi_indexes, pi_indexes + n_pi);
    widgets_ = 0;
    if (poly_len_[Q_RANGE]) widgets_ |= 1u;
    if (poly_len_[Q_LOGIC]) widgets_ |= 2u;
    if (poly_len_[Q_FIXED]) widgets_ |= 4u;
    if (poly_len_[Q_VAR]) widgets_ |= 8u;
    for (int k = 0; k < N_SEL; k++) sel_comms_[k] = comms[k];
    // the sigma values over the n domain, which compute_permutation_vec re-derives from the polynomials
    // (composer/permutation.rs:189-199)
    for (int k = 0; k < 4; k++)
        ntt_.transform(NTT_FFT, polys + (size_t)(S1 + k) * n, poly_len_[S1 + k], sigma_.as<Fr>() + (size_t)k * n, log_n);
    // L1 on the coset is not part of the serialized key (quotient_poly.rs:237-245 rebuilds it per proof)
    build_key_evals(false, evals, N8, stage, false, false);
    evals8_built_ = true;
    build_vanishing();
    choose_coset();
    if (half_coset_) {
        // the 4n coset is every second point of the 8n coset: the stored vectors' even entries
        evals4_.reserve((size_t)(N_SEL + 2) * 4 * n * sizeof(Fr));
        for (int k = 0; k < N_SEL + 2; k++) poly_.take_even(evals + (size_t)k * N8, (size_t)4 * n, evals4_.as<Fr>() + (size_t)k * 4 * n);
        const int n_sel_h = widgets_ ? 11 : 7;
        sel_h_.reserve((size_t)n_sel_h * n * sizeof(Fr));
        for (int k = 0; k < n_sel_h; k++)
            ntt_.transform(NTT_FFT, polys + (size_t)k * n, poly_len_[k], sel_h_.as<Fr>() + (size_t)k * n, log_n);
        read_sigma_tops();
    }
    // v_h_coset_8n has period 8 (domain.rs:256-275); the stored vector must be the one of this domain
    for (int i = 0; i < 8; i++) {
        Fr raw;
        memcpy(raw.v, eval_at[N_SEL + 1] + DOMAIN_BYTES + 32 * i, 32);
        if (!raw.below_modulus() || Fr::to_mont(raw) != vh_[i]) {
            n_ = 0;
            throw ZkError(ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: v_h_coset_8n does not belong to the 8n coset");
        }
    }
    ZK_HIP(hipStreamSynchronize(stream_));
    build_pi_lagrange();
    publish_key();
}

size_t Prover::to_bytes(uint8_t *out, size_t cap) {
    ZK_REQUIRE(n_ != 0, ZKGPU_ERR_NOT_LOADED, "prover_to_bytes: no key loaded");
    ZK_REQUIRE(!shared(), ZKGPU_ERR_INVALID_ARG,
               "prover_to_bytes: this context holds a slice of every coset vector; serialize from an unshared context");
    const uint32_t n = n_;
    const size_t N8 = (size_t)8 * n;
    const size_t eval_size = DOMAIN_BYTES + N8 * 32;
    if (out != nullptr) ensure_evals8n();
    // ProverKey::serialization_size (widget.rs:265-278): every polynomial is budgeted at q_m's length
    const size_t pk_size = (size_t)poly_len_[Q_M] * 32 * 15 + eval_size * 17 + 8 * 17;
    const size_t ck_size = msm_.raw_size(), vk_size = 8 + 48 * N_SEL;
    const size_t total = 48 + label_.size() + pk_size + ck_size + vk_size;
    if (out == nullptr || cap < total) return total;
    wr_be64(out, label_.size());
    wr_be64(out + 8, pk_size);
    wr_be64(out + 16, ck_size);
    wr_be64(out + 24, vk_size);
    wr_be64(out + 32, n);
    wr_be64(out + 40, constraints_);
    memcpy(out + 48, label_.data(), label_.size());
    uint8_t *pk = out + 48 + label_.size();
    memset(pk, 0, pk_size);
    // writes follow coset-bytes' slice writer (bytes/src/serialize.rs:105-118): a piece that does not fit in what
    // is left of the budget is dropped as a whole
    size_t pos = 0;
    auto fits = [&](size_t bytes) { return bytes <= pk_size - pos; };
    auto put_u64 = [&](uint64_t v) {
        if (!fits(8)) return;
        wr_le64(pk + pos, v);
        pos += 8;
    };
    uint8_t dom[DOMAIN_BYTES];
    {
        Domain d8 = Domain::make(N8);
        wr_le64(dom, d8.size);
        for (int i = 0; i < 4; i++) dom[8 + i] = (uint8_t)(d8.log_size >> (8 * i));
        const Fr *f[5] = {&d8.size_as_field_element, &d8.size_inv, &d8.group_gen, &d8.group_gen_inv, &d8.generator_inv};
        for (int k = 0; k < 5; k++) {
            Fr c = Fr::from_mont(*f[k]);
            memcpy(dom + 12 + 32 * k, c.v, 32);
        }
    }
    work_.reserve((size_t)N_SEL * n * sizeof(Fr));
    Fr *stage = work_.as<Fr>();
    auto put_scalars = [&](const Fr *dev, size_t count, bool with_domain) {
        const size_t bytes = count * 32 + (with_domain ? DOMAIN_BYTES : 0);
        if (!fits(bytes)) return;
        if (with_domain) {
            memcpy(pk + pos, dom, DOMAIN_BYTES);
            pos += DOMAIN_BYTES;
        }
        if (count) {
            poly_.to_canonical(dev, count, stage);
            ZK_HIP(hipMemcpyAsync(pk + pos, stage, count * 32, hipMemcpyDeviceToHost, stream_));
            ZK_HIP(hipStreamSynchronize(stream_));
            pos += count * 32;
        }
    };
    put_u64(n);
    put_u64(eval_size);
    const Fr *polys = polys_.as<Fr>();
    const Fr *evals = evals_.as<Fr>();
    for (int j = 0; j < N_SEL; j++) {
        const int k = SERIAL_ORDER[j];
        put_u64(poly_len_[k]);
        put_scalars(polys + (size_t)k * n, poly_len_[k], false);
        put_scalars(evals + (size_t)k * N8, N8, true);
    }
    put_scalars(evals + (size_t)N_SEL * N8, N8, true);
    if (fits(eval_size)) {   // v_h_coset_8n: the eight values, repeated
        memcpy(pk + pos, dom, DOMAIN_BYTES);
        pos += DOMAIN_BYTES;
        uint8_t period[8 * 32];
        for (int i = 0; i < 8; i++) {
            Fr c = Fr::from_mont(vh_[i]);
            memcpy(period + 32 * i, c.v, 32);
        }
        for (size_t i = 0; i < N8; i += 8) {
            const size_t chunk = N8 - i < 8 ? N8 - i : 8;
            memcpy(pk + pos, period, chunk * 32);
            pos += chunk * 32;
        }
    }
    uint8_t *ck = pk + pk_size;
    msm_.export_points_raw(ck);
    uint8_t *vk = ck + ck_size;
    wr_le64(vk, vk_n_);
    for (int j = 0; j < N_SEL; j++) {
        const AffineOut &c = sel_comms_[SERIAL_ORDER[j]];
        uint64_t xy[12];
        memcpy(xy, c.x.v, 48);
        memcpy(xy + 6, c.y.v, 48);
        zkgpu_g1_compress(xy, c.infinity ? 1 : 0, vk + 8 + 48 * j);
    }
    return total;
}

void Prover::blind_poly(const Fr *evals_dev, int hiding_degree, const Fr *blinders, Fr *out_dev) {
    // prover.rs:64-83
    ntt_.transform(NTT_IFFT, evals_dev, n_, out_dev, log_n_);
    uint64_t idx[8];
    Fr vals[8];
    bool sub[8];
    int k = 0;
    for (int i = 0; i <= hiding_degree; i++) {
        idx[k] = i;
        vals[k] = blinders[i];
        sub[k++] = true;
        idx[k] = (uint64_t)n_ + i;
        vals[k] = blinders[i];
        sub[k++] = false;
    }
    poly_.patch(out_dev, idx, vals, sub, k);
}

uint64_t Prover::quotient(const Fr *z, uint64_t z_len, const Fr *const wires[4], const uint64_t wire_len[4], const Fr *pi,
                          uint64_t pi_len, const Fr ch[7], Fr *t_out, const Fr *pi_values) {
    ZK_REQUIRE(loaded(), ZKGPU_ERR_NOT_LOADED, "quotient: no prover key loaded");
    const uint32_t ready = ev_ready_;   // ensure_evals8n may use the coset buffer as scratch: only on a path that set no mark
    ev_ready_ = 0;
    ensure_evals8n();
    const size_t N8 = (size_t)8 * n_, L = ev_len_;   // L = the coset points this context evaluates
    ZK_REQUIRE(!shared() || (coll_.usable() && coll_.world == key_world_ && coll_.rank == key_rank_), ZKGPU_ERR_INVALID_ARG,
               "quotient: the key was loaded for another set of ranks than set_collective names now");
    small_.reserve(coset_buffer_bytes());
    Fr *ev = small_.as<Fr>();
    const uint32_t lg8 = log_n_ + 3;
    // quotient_poly.rs:46-51,120; prove() may have launched some of these transforms already, while it waited for a
    // commitment
    if (!(ready & 1u)) coset_fft_8n(z, z_len, ev);
    if (!(ready & 2u))
        for (int k = 0; k < 4; k++) coset_fft_8n(wires[k], wire_len[k], ev + (size_t)(k + 1) * L);
    if (pi != nullptr) {
        coset_fft_8n(pi, pi_len, ev + 5 * L);
    } else {
        ZK_REQUIRE(pi_values != nullptr && !pi_lagrange_idx_.empty() && !half_coset_, ZKGPU_ERR_INTERNAL,
                   "quotient: no public-input vectors over the 8n coset");
        LinArgs lin;
        lin.n_terms = (int)pi_lagrange_idx_.size();
        for (int j = 0; j < lin.n_terms; j++) {
            lin.t[j].coeffs = pi_lagrange_.as<Fr>() + (size_t)j * L;
            lin.t[j].len = L;
            lin.t[j].scalar = pi_values[j];
        }
        lin.constant = Fr::zero();
        lin.out = ev + 5 * L;
        lin.out_len = L;
        poly_.linear_combination(lin);
    }
    QuotientArgs q;
    q.z = ev;
    q.a = ev + L;
    q.b = ev + 2 * L;
    q.c = ev + 3 * L;
    q.d = ev + 4 * L;
    q.pi = ev + 5 * L;
    for (int k = 0; k < N_SEL + 2; k++) {
        const Fr *v = evals_.as<Fr>() + (size_t)k * ev_len_;
        if (k < N_SEL)
            q.sel[k] = v;
        else if (k == N_SEL)
            q.linear = v;
        else
            q.l1 = v;
    }
    q.out = shared() ? ev + 6 * L : t_out;
    q.alpha = ch[0];
    q.alpha_sq = Fr::sqr(ch[0]);
    q.beta = ch[1];
    q.gamma = ch[2];
    q.range_ch = ch[3];
    q.logic_ch = ch[4];
    q.fixed_ch = ch[5];
    q.var_ch = ch[6];
    for (int i = 0; i < 8; i++) q.vh_inv[i] = vh_inv_[i];
    q.n8 = (uint32_t)L;
    // a slice i = rank + world * k keeps the rotation by 8 local: world divides 8
    q.world = key_world_;
    q.first = key_rank_;
    q.rot = 8 / key_world_;
    q.widgets = widgets_;
    poly_.quotient(q);
    // quotient_poly.rs:97
    if (shared())
        ntt_.inverse_sharded(true, q.out, t_out, lg8, coll_);
    else
        ntt_.transform(NTT_COSET_IFFT, t_out, N8, t_out, lg8);
    return poly_.length_without_leading_zeros(t_out, N8);    // Polynomial::from_coefficients_vec
}

// a <- a * f modulo Y^7
static void series_mul7(Fr a[7], const Fr f[7]) {
    Fr out[7];
    for (int m = 0; m < 7; m++) {
        Fr acc = Fr::zero();
        for (int j = 0; j <= m; j++) acc = Fr::add(acc, Fr::mul(a[j], f[m - j]));
        out[m] = acc;
    }
    for (int m = 0; m < 7; m++) a[m] = out[m];
}

// The top seven coefficients of a polynomial of degree at most `deg`: c[j] belongs to X^(deg - j), and is zero where
// deg - j is negative.  Sums and products of such series are exact in their own top seven coefficients, so running a
// widget's identity (widgets.hpp) over the series of the wire and selector polynomials yields the top coefficients of
// that widget's term in the numerator.
struct Top7 {
    Fr c[7];
    int64_t deg;
};
struct TopRing {
    typedef Top7 V;
    static Top7 add(const Top7 &a, const Top7 &b) {
        const Top7 &hi = a.deg >= b.deg ? a : b, &lo = a.deg >= b.deg ? b : a;
        Top7 r = hi;
        for (int64_t j = hi.deg - lo.deg; j < 7; j++) r.c[j] = Fr::add(r.c[j], lo.c[j - (hi.deg - lo.deg)]);
        return r;
    }
    static Top7 neg(const Top7 &a) {
        Top7 r = a;
        for (int j = 0; j < 7; j++) r.c[j] = Fr::neg(a.c[j]);
        return r;
    }
    static Top7 sub(const Top7 &a, const Top7 &b) { return add(a, neg(b)); }
    static Top7 mul(const Top7 &a, const Top7 &b) {
        Top7 r;
        r.deg = a.deg + b.deg;
        for (int m = 0; m < 7; m++) {
            Fr acc = Fr::zero();
            for (int j = 0; j <= m; j++) acc = Fr::add(acc, Fr::mul(a.c[j], b.c[m - j]));
            r.c[m] = acc;
        }
        return r;
    }
    static Top7 scale(const Top7 &a, const Fr &k) {
        Top7 r = a;
        for (int j = 0; j < 7; j++) r.c[j] = Fr::mul(a.c[j], k);
        return r;
    }
    template <uint32_t K>
    static Top7 small(const Top7 &a) {
        Top7 r = a;
        for (int j = 0; j < 7; j++) r.c[j] = mul_small<K>(a.c[j]);
        return r;
    }
    static Top7 cst(uint32_t k) {
        Top7 r;
        r.deg = 0;
        r.c[0] = k == 1 ? Fr::one() : Fr::from_u32(k);
        for (int j = 1; j < 7; j++) r.c[j] = Fr::zero();
        return r;
    }
};
// the series of p(w X) from that of p: the coefficient of X^m gains w^m, and w^n = 1
static Top7 top_next_row(const Top7 &p, const Domain &dn) {
    Top7 r = p;
    Fr pw = Fr::pow_u64(dn.group_gen, (uint64_t)(p.deg % (int64_t)dn.size));
    for (int j = 0; j < 7; j++) {
        r.c[j] = Fr::mul(p.c[j], pw);
        pw = Fr::mul(pw, dn.group_gen_inv);
    }
    return r;
}

// build_quotient_polynomial (quotient_poly.rs:15-100) for round
// polynomials of the lengths prove() produces (wires n + 2, z n + 3) whose numerator is known to vanish on the domain.
// The numerator N has degree at most 5n + 6, so t = N / Z_H has at most 4n + 7 coefficients.  The pointwise identity is
// evaluated on the 4n points g w_4n^k -- every second point of the reference's 8n coset, so the same kernel runs with
// the slice parameters (first 0, step 2) and the "next row" four places on -- and the inverse coset transform returns
// t modulo X^4n - g^4n, i.e. r_i = t_i + g^4n t_(4n+i) for i < 7 and t_i elsewhere.  The seven top coefficients are
// those of N, t_(4n+i) = N_(5n+i), and only alpha (z prod_w (w + beta k_w X + gamma) - z(wX) prod_w (w + beta sigma_w +
// gamma)) and the widget terms (a selector times at most four wire factors: degree 5n + 3) reach those degrees: they are
// products of the seven top coefficients of each factor, which the host holds in tops_host_ (wires: coefficients
// n-5 .. n+1, z: n-4 .. n+2, gathered right after blinding), sigma_top_ and sel_top_.  Every value equals the
// reference's: both are the coefficients of the one polynomial t.
uint64_t Prover::quotient_half_coset(const Fr *z, const Fr *const wires[4], const Fr ch[7], const Fr *pi_values, const Fr *pi_poly,
                                     Fr *t_out) {
    const size_t n = n_, L = 4 * n;
    Fr *ev = small_.as<Fr>();
    const uint32_t ready = ev_ready_;
    ev_ready_ = 0;
    if (!(ready & 1u)) coset_fft_4n(z, n + 3, ev);
    if (!(ready & 2u))
        for (int k = 0; k < 4; k++) coset_fft_4n(wires[k], n + 2, ev + (size_t)(k + 1) * L);
    if (pi_values == nullptr) {
        coset_fft_4n(pi_poly, n, ev + 5 * L);
    } else {
        // sum_j pi_j L_j over the resident Lagrange vectors (build_pi_lagrange)
        LinArgs lin;
        lin.n_terms = (int)pi_lagrange_idx_.size();
        for (int j = 0; j < lin.n_terms; j++) {
            lin.t[j].coeffs = pi_lagrange_.as<Fr>() + (size_t)j * L;
            lin.t[j].len = L;
            lin.t[j].scalar = pi_values[j];
        }
        lin.constant = Fr::zero();
        lin.out = ev + 5 * L;
        lin.out_len = L;
        poly_.linear_combination(lin);
    }
    QuotientArgs q;
    q.z = ev;
    q.a = ev + L;
    q.b = ev + 2 * L;
    q.c = ev + 3 * L;
    q.d = ev + 4 * L;
    q.pi = ev + 5 * L;
    const Fr *key = evals4_.as<Fr>();
    for (int k = 0; k < N_SEL; k++) q.sel[k] = key + (size_t)k * L;
    q.linear = key + (size_t)N_SEL * L;
    q.l1 = key + (size_t)(N_SEL + 1) * L;
    q.out = t_out;
    q.alpha = ch[0];
    q.alpha_sq = Fr::sqr(ch[0]);
    q.beta = ch[1];
    q.gamma = ch[2];
    q.range_ch = ch[3];
    q.logic_ch = ch[4];
    q.fixed_ch = ch[5];
    q.var_ch = ch[6];
    for (int i = 0; i < 8; i++) q.vh_inv[i] = vh_inv_[i];
    q.n8 = (uint32_t)L;
    q.world = 2;
    q.first = 0;
    q.rot = 4;
    q.widgets = widgets_;
    poly_.quotient(q);
    ntt_.transform(NTT_COSET_IFFT, t_out, L, t_out, log_n_ + 2);

    // the seven top coefficients, as power series in Y = 1/X truncated at Y^7: entry j belongs to the factor's degree
    // bound minus j
    const Fr &alpha = ch[0], &beta = ch[1];
    const Fr *win = tops_host_;
    Fr id[7], cp[7], f[7];
    const Domain dn = Domain::make(n);
    {
        // z(wX) has the coefficients z_j w^j, and w^n = 1: coefficient n+2-j carries w^(2-j)
        Fr pw = Fr::sqr(dn.group_gen);
        for (int j = 0; j < 7; j++) {
            id[j] = win[29 + 6 - j];
            cp[j] = Fr::mul(id[j], pw);
            pw = Fr::mul(pw, dn.group_gen_inv);
        }
    }
    for (int k = 0; k < 4; k++) {
        for (int j = 0; j < 7; j++) f[j] = win[7 * k + 6 - j];   // beta k_w X + gamma lies below degree n - 5
        series_mul7(id, f);
        for (int j = 0; j < 7; j++) f[j] = Fr::add(f[j], Fr::mul(beta, sigma_top_[k][j]));
        series_mul7(cp, f);
    }
    // numerator coefficients 5n+6-j, j < 7
    Fr ntop[7];
    for (int j = 0; j < 7; j++) ntop[j] = Fr::mul(alpha, Fr::sub(id[j], cp[j]));
    if (widgets_) {
        typedef TopRing R;
        Top7 w[4], sel[N_SEL_TOPS];
        for (int k = 0; k < 4; k++) {
            w[k].deg = (int64_t)n + 1;
            for (int j = 0; j < 7; j++) w[k].c[j] = win[7 * k + 6 - j];
        }
        for (int k = 0; k < N_SEL_TOPS; k++) {
            sel[k].deg = (int64_t)n - 1;
            for (int j = 0; j < 7; j++) sel[k].c[j] = sel_top_[k][j];
        }
        const Top7 &a = w[0], &b = w[1], &c = w[2], &d = w[3];
        const Top7 a_w = top_next_row(a, dn), b_w = top_next_row(b, dn), d_w = top_next_row(d, dn);
        const Top7 &q_l = sel[4], &q_r = sel[5], &q_c = sel[6];   // SEL_TOPS: range logic fixed var q_l q_r q_c
        Top7 p[5], term[4];
        bool have[4] = {false, false, false, false};
        if (widgets_ & 1u) {
            w_range_parts<R>(a, b, c, d, d_w, p);
            term[0] = w_combine<R>(p, W_RANGE_PARTS, sep_powers<InlineMul>(ch[3]));
            have[0] = true;
        }
        if (widgets_ & 2u) {
            w_logic_parts<R>(a, a_w, b, b_w, c, d, d_w, q_c, p);
            term[1] = w_combine<R>(p, W_LOGIC_PARTS, sep_powers<InlineMul>(ch[4]));
            have[1] = true;
        }
        if (widgets_ & 4u) {
            w_fixed_base_parts<R>(a, a_w, b, b_w, c, d, d_w, q_l, q_r, q_c, p);
            term[2] = w_combine<R>(p, W_FIXED_PARTS, sep_powers<InlineMul>(ch[5]));
            have[2] = true;
        }
        if (widgets_ & 8u) {
            w_curve_add_parts<R>(a, a_w, b, b_w, c, d, d_w, p);
            term[3] = w_combine<R>(p, W_CURVE_PARTS, sep_powers<InlineMul>(ch[6]));
            have[3] = true;
        }
        for (int k = 0; k < 4; k++) {
            if (!have[k]) continue;
            const Top7 t = R::scale(R::mul(term[k], sel[k]), ch[3 + k]);   // widget `compute_quotient_i`
            const int64_t shift = (int64_t)(5 * n + 6) - t.deg;
            ZK_REQUIRE(shift >= 0, ZKGPU_ERR_INTERNAL, "a widget term above the numerator's degree bound");
            for (int64_t j = shift; j < 7; j++) ntop[j] = Fr::add(ntop[j], t.c[j - shift]);
        }
    }
    const Domain d8 = Domain::make(8 * n);
    const Fr g4n = Fr::pow_u64(Fr::inv(d8.generator_inv), L);
    uint64_t idx[7];
    Fr hi[7], lo[7];
    bool sub[7], set[7];
    for (int i = 0; i < 7; i++) {
        hi[i] = ntop[6 - i];   // t_(4n+i) = N_(5n+i): entry j = 6 - i
        lo[i] = Fr::mul(g4n, hi[i]);
        idx[i] = (uint64_t)i;
        sub[i] = true;
        set[i] = false;
    }
    poly_.patch(t_out, idx, lo, sub, 7);            // t_i = r_i - g^4n t_(4n+i)
    for (int i = 0; i < 7; i++) idx[i] = L + i;
    poly_.patch(t_out, idx, hi, set, 7);
    // Polynomial::from_coefficients_vec drops leading zeros: the top coefficients are known here, so the length
    // needs the device only if all seven vanish
    for (int i = 6; i >= 0; i--)
        if (!hi[i].is_zero()) return L + i + 1;
    return poly_.length_without_leading_zeros(t_out, L);
}

// proof_system/proof.rs:511-561
static Fr barycentric_eval(const std::vector<Fr> &evaluations, const Fr &point, const Domain &domain) {
    Fr numerator = Fr::mul(Fr::sub(Fr::pow_u64(point, domain.size), Fr::one()), domain.size_inv);
    std::vector<size_t> nz;
    for (size_t i = 0; i < evaluations.size(); i++)
        if (!evaluations[i].is_zero()) nz.push_back(i);
    std::vector<Fr> den(nz.size());
    for (size_t k = 0; k < nz.size(); k++)
        den[k] = Fr::sub(Fr::mul(Fr::pow_u64(domain.group_gen_inv, nz[k]), point), Fr::one());
    host_batch_inverse(den);
    Fr result = Fr::zero();
    for (size_t k = 0; k < nz.size(); k++) result = Fr::add(result, Fr::mul(den[k], evaluations[nz[k]]));
    return Fr::mul(result, numerator);
}

// evaluate_all_lagrange_coefficients(tau)[0] (fft/domain.rs:200-246)
static Fr first_lagrange_coefficient(const Domain &d, const Fr &tau) {
    Fr t_size = Fr::pow_u64(tau, d.size);
    if (t_size == Fr::one()) return tau == Fr::one() ? Fr::one() : Fr::zero();
    Fr l = Fr::mul(Fr::sub(t_size, Fr::one()), d.size_inv);
    return Fr::mul(l, Fr::inv(Fr::sub(tau, Fr::one())));
}

// build_linearization_polynomial (linearization_poly.rs:146-272): scalars on the host from the 15 evaluations, one fused
// pass over the polynomials on the device.  ch7 = alpha beta gamma range logic fixed var; e in ProofEvaluations order.
// Returns the number of coefficients written to r_out (the longest term; leading zeros are not dropped here).
uint64_t Prover::linearization(const Fr ch7[7], const Fr &z_ch, const Fr *e, const Fr *p_z, uint64_t z_len, const Fr *const t_parts[4],
                               const uint64_t t_lens[4], const std::vector<Fr> &public_inputs, Fr *r_out, uint64_t r_cap,
                               uint64_t z_len_without_leading_zeros) {
    ZK_REQUIRE(loaded(), ZKGPU_ERR_NOT_LOADED, "linearization: no prover key loaded");
    enum { A, B, C, D, AW, BW, DW, QARITH, QC, QL, QR, SG1, SG2, SG3, ZE };
    typedef InlineMul M;
    const uint64_t n = n_;
    const Fr &alpha = ch7[0], &beta = ch7[1], &gamma = ch7[2], &range_ch = ch7[3], &logic_ch = ch7[4], &fixed_ch = ch7[5],
             &var_ch = ch7[6];
    const Domain domain = Domain::make(constraints_);
    LinArgs L;
    int k = 0;
    uint64_t r_len = 0;
    auto term = [&](const Fr *p, uint64_t len, const Fr &s) {
        L.t[k].coeffs = p;
        L.t[k].len = len;
        L.t[k].scalar = s;
        k++;
        if (len > r_len) r_len = len;
    };
    // arithmetic/proverkey.rs:75-107
    const Fr qa = e[QARITH];
    term(key_poly(Q_M), n, Fr::mul(Fr::mul(e[A], e[B]), qa));
    term(key_poly(Q_L), n, Fr::mul(e[A], qa));
    term(key_poly(Q_R), n, Fr::mul(e[B], qa));
    term(key_poly(Q_O), n, Fr::mul(e[C], qa));
    term(key_poly(Q_F), n, Fr::mul(e[D], qa));
    term(key_poly(Q_C), n, qa);
    // range / logic / fixed-base / curve-addition compute_linearization
    term(key_poly(Q_RANGE), n, Fr::mul(w_range_identity<M>(sep_powers<M>(range_ch), e[A], e[B], e[C], e[D], e[DW]), range_ch));
    term(key_poly(Q_LOGIC), n,
         Fr::mul(w_logic_identity<M>(sep_powers<M>(logic_ch), e[A], e[AW], e[B], e[BW], e[C], e[D], e[DW], e[QC]), logic_ch));
    term(key_poly(Q_FIXED), n,
         Fr::mul(w_fixed_base_identity<M>(sep_powers<M>(fixed_ch), e[A], e[AW], e[B], e[BW], e[C], e[D], e[DW], e[QL], e[QR], e[QC]),
                 fixed_ch));
    term(key_poly(Q_VAR), n, Fr::mul(w_curve_add_identity<M>(sep_powers<M>(var_ch), e[A], e[AW], e[B], e[BW], e[C], e[D], e[DW]), var_ch));
    // permutation/proverkey.rs:121-245
    const Fr beta_z = Fr::mul(beta, z_ch);
    Fr acc = Fr::add(Fr::add(e[A], beta_z), gamma);
    acc = Fr::mul(acc, Fr::add(Fr::add(e[B], mul_small<7>(beta_z)), gamma));
    acc = Fr::mul(acc, Fr::add(Fr::add(e[C], mul_small<13>(beta_z)), gamma));
    acc = Fr::mul(acc, Fr::add(Fr::add(e[D], mul_small<17>(beta_z)), gamma));
    acc = Fr::mul(acc, alpha);
    Fr cacc = Fr::add(Fr::add(e[A], Fr::mul(beta, e[SG1])), gamma);
    cacc = Fr::mul(cacc, Fr::add(Fr::add(e[B], Fr::mul(beta, e[SG2])), gamma));
    cacc = Fr::mul(cacc, Fr::add(Fr::add(e[C], Fr::mul(beta, e[SG3])), gamma));
    cacc = Fr::mul(cacc, Fr::mul(beta, e[ZE]));
    cacc = Fr::mul(cacc, alpha);
    // EvaluationDomain::new(z_poly.degree() - 2) (permutation/proverkey.rs:156)
    // (the caller may know it: prove() has written z's top coefficients itself)
    const uint64_t z_eff = z_len_without_leading_zeros ? z_len_without_leading_zeros : poly_.length_without_leading_zeros(p_z, z_len);
    const uint64_t z_deg = z_eff ? z_eff - 1 : 0;
    const Domain dz = Domain::make(z_deg >= 2 ? z_deg - 2 : 0);
    const Fr l1_z = first_lagrange_coefficient(dz, z_ch);
    term(p_z, z_len, Fr::add(acc, Fr::mul(l1_z, Fr::sqr(alpha))));
    term(key_poly(S4), n, Fr::neg(cacc));
    // linearization_poly.rs:211-228
    const Fr z_n = Fr::pow_u64(z_ch, n), z_2n = Fr::pow_u64(z_ch, 2 * n), z_3n = Fr::pow_u64(z_ch, 3 * n);
    const Fr zh = Fr::neg(Fr::sub(z_n, Fr::one()));
    term(t_parts[0], t_lens[0], zh);
    term(t_parts[1], t_lens[1], Fr::mul(z_n, zh));
    term(t_parts[2], t_lens[2], Fr::mul(z_2n, zh));
    term(t_parts[3], t_lens[3], Fr::mul(z_3n, zh));
    L.n_terms = k;
    // linearization_poly.rs:187-190 — the reference passes the SPARSE public-input list here
    L.constant = barycentric_eval(public_inputs, z_ch, domain);
    if (r_len == 0) r_len = 1;   // the constant term
    ZK_REQUIRE(r_len <= r_cap, ZKGPU_ERR_DEGREE_LARGE, "PolynomialDegreeTooLarge");
    L.out = r_out;
    L.out_len = r_len;
    poly_.linear_combination(L);
    return r_len;
}

// CommitKey::compute_aggregate_witness (kzg10/key.rs:131-146): sum_j v^j p_j, divided by (X - point) with the remainder
// dropped (Polynomial::ruffini, fft/polynomial.rs:343-357).  `sum_tmp` holds the longest polynomial; returns the
// witness's coefficient count (one less).
uint64_t Prover::aggregate_witness(const Fr *const *polys, const uint64_t *lens, int count, const Fr &point, const Fr &v, Fr *sum_tmp,
                                   Fr *out) {
    ZK_REQUIRE(count >= 1 && count <= MAX_LIN_TERMS, ZKGPU_ERR_INVALID_ARG, "aggregate_witness: 1 to 18 polynomials");
    LinArgs L;
    Fr pw = Fr::one();
    uint64_t len = 0;
    for (int k = 0; k < count; k++) {
        L.t[k].coeffs = polys[k];
        L.t[k].len = lens[k];
        L.t[k].scalar = pw;
        pw = Fr::mul(pw, v);
        if (lens[k] > len) len = lens[k];
    }
    L.n_terms = count;
    L.constant = Fr::zero();
    L.out = sum_tmp;
    L.out_len = len;
    if (len == 0) return 0;
    poly_.linear_combination(L);
    poly_.ruffini(sum_tmp, len, point, out);
    return len - 1;
}

int Prover::prove(const uint64_t *wires_host, const uint64_t *public_inputs_host, size_t n_pi, const uint64_t *blinders14,
                  ProofOut &out, ProveTimings *timings) {
    ZK_REQUIRE(loaded(), ZKGPU_ERR_NOT_LOADED, "prove: no prover key loaded");
    ZK_REQUIRE(wires_host != nullptr && blinders14 != nullptr, ZKGPU_ERR_INVALID_ARG, "prove: null input");
    typedef std::chrono::steady_clock clk;
    auto t_start = clk::now();
    auto lap = [&](int slot, clk::time_point &since) {
        if (!timings) return;
        ZK_HIP(hipStreamSynchronize(stream_));
        auto now = clk::now();
        timings->ms[slot] = std::chrono::duration<float, std::milli>(now - since).count();
        since = now;
    };
    auto t_phase = t_start;

    const size_t n = n_, N8 = 8 * n, cap = n + 8;
    const Domain domain = Domain::make(constraints_);  // prover.rs:224
    ZK_REQUIRE(domain.size == n, ZKGPU_ERR_INVALID_ARG, "prove: constraints do not match the loaded key size");
    Fr bl[14];
    for (int i = 0; i < 14; i++) bl[i] = fr_from_words(blinders14 + 4 * i);
    std::vector<Fr> public_inputs(n_pi);
    for (size_t i = 0; i < n_pi; i++) public_inputs[i] = fr_from_words(public_inputs_host + 4 * i);

    // per-proof device memory
    //   wires 4n | a b c d z polys 5*cap | z evals n | pi n | t 8n | t_low t_mid t_high 3*cap | r, agg, witness 3*(5n) | evals 16
    const size_t lin_cap = 5 * n + 16;
    work_.reserve((4 * n + 5 * cap + 2 * n + N8 + 3 * cap + 3 * lin_cap + 64) * sizeof(Fr));
    Fr *w = work_.as<Fr>();
    Fr *wires = w;
    Fr *p_a = wires + 4 * n, *p_b = p_a + cap, *p_c = p_b + cap, *p_d = p_c + cap, *p_z = p_d + cap;
    Fr *z_evals = p_z + cap;
    Fr *p_pi = z_evals + n;
    Fr *t = p_pi + n;
    Fr *t_low = t + N8, *t_mid = t_low + cap, *t_high = t_mid + cap;
    Fr *r_poly = t_high + cap, *agg = r_poly + lin_cap, *wit = agg + lin_cap;
    Fr *ev_dev = wit + lin_cap;

    // transcript.rs:51-65 + widget.rs:199-228 (seed_transcript; "s_sigma_4" carries s_sigma_1, as in the reference).
    // The seeded transcript depends on the key only: it is built once and every proof starts from a copy
    // (prover.rs:226: self.transcript.clone()).
    if (!seed_tr_) {
        PlonkTranscript seed(label_);
        seed.circuit_domain_sep(constraints_);
        static const struct {
            const char *label;
            int sel;
        } order[] = {{"q_m", Q_M},         {"q_l", Q_L},         {"q_r", Q_R},     {"q_o", Q_O},
                     {"q_c", Q_C},         {"q_f", Q_F},         {"q_arith", Q_ARITH}, {"q_range", Q_RANGE},
                     {"q_logic", Q_LOGIC}, {"q_variable_group_add", Q_VAR}, {"q_fixed_group_add", Q_FIXED},
                     {"s_sigma_1", S1},    {"s_sigma_2", S2},    {"s_sigma_3", S3}, {"s_sigma_4", S1}};
        for (auto &o : order) seed.append_commitment(o.label, sel_comms_[o.sel]);
        seed.circuit_domain_sep(vk_n_);   // widget.rs:227: the verifier key's own n
        seed_tr_.reset(new MerlinTranscript(seed.t));
    }
    PlonkTranscript tr(*seed_tr_);
    for (auto &pi : public_inputs) tr.append_scalar("pi", pi);  // prover.rs:236-238

    // composer.rs:1001-1014: the dense public inputs, built on the device from the few (index, value) pairs; a later
    // pair with the same index overrides an earlier one, as the reference's sequential writes do.
    std::map<uint64_t, Fr> pi_last;
    for (size_t i = 0; i < pi_indexes_.size() && i < n_pi; i++)
        if (pi_indexes_[i] < n) pi_last[pi_indexes_[i]] = public_inputs[i];
    {
        std::vector<uint64_t> idx;
        std::vector<Fr> vals;
        for (auto &kv : pi_last) {
            idx.push_back(kv.first);
            vals.push_back(kv.second);
        }
        poly_.zero_and_scatter(p_pi, n, idx.data(), vals.data(), (uint32_t)idx.size());
    }

    // protocol phase 1: wire polynomials (prover.rs:240-269)
    // hipMemcpyDefault: the caller's pointer may be host memory or already resident in HBM
    // (a witness that already lies in this GPU's memory is read where it is: nothing below writes to it)
    int wit_slot = -1;   // the prefetched buffer this proof reads, if any
    for (int s = 0; s < 2; s++)
        if (wit_src_[s] == wires_host && wit_buf_[s].cap >= 4 * n * sizeof(Fr)) wit_slot = s;
    if (wit_slot >= 0) {
        // uploaded by prefetch_witness: the copy ran (or runs) on the copy stream beside the previous proof
        ZK_HIP(hipStreamWaitEvent(stream_, wit_ready_[wit_slot], 0));
        wires = wit_buf_[wit_slot].as<Fr>();
        wit_src_[wit_slot] = nullptr;
    } else {
        hipPointerAttribute_t attr;
        int current = -1;
        const bool resident = hipPointerGetAttributes(&attr, wires_host) == hipSuccess && attr.type == hipMemoryTypeDevice &&
                              hipGetDevice(&current) == hipSuccess && attr.device == current;
        if (resident) {
            wires = const_cast<Fr *>(reinterpret_cast<const Fr *>(wires_host));
        } else {
            (void)hipGetLastError();   // a pageable host pointer is unknown to the runtime: not an error here
            ZK_HIP(hipMemcpyAsync(wires, wires_host, 4 * n * sizeof(Fr), hipMemcpyDefault, stream_));
        }
    }
    struct MarkConsumed {   // on every way out: the next upload into the buffer waits for this proof's reads of it
        hipEvent_t ev;
        hipStream_t st;
        ~MarkConsumed() {
            if (ev) (void)hipEventRecord(ev, st);
        }
    } mark_consumed{wit_slot >= 0 ? wit_consumed_[wit_slot] : nullptr, stream_};
    Fr *wp[4] = {p_a, p_b, p_c, p_d};
    for (int k = 0; k < 4; k++) blind_poly(wires + k * n, 1, bl + 2 * k, wp[k]);
    int rc;
    // A key that proves over the 4n coset: the quotient may take that route only if its numerator vanishes on the
    // domain, i.e. every row satisfies its gate (checked here, the flag arrives with the wire commitments) and the
    // permutation accumulator closes (known when it is built).  Otherwise the proof runs over the 8n coset like the
    // reference, garbage quotient and degree error included.
    const uint32_t FLAG_AT = 28, Z_AT = 29, TOPS = 36;   // tops_host_: 4 wire windows, the flag, z's window
    if (half_coset_) {
        tops_dev_.reserve(TOPS * sizeof(Fr));
        if (!tops_host_) ZK_HIP(hipHostMalloc(reinterpret_cast<void **>(&tops_host_), TOPS * sizeof(Fr), hipHostMallocDefault));
        Fr *td = tops_dev_.as<Fr>();
        ZK_HIP(hipMemsetAsync(td + FLAG_AT, 0, sizeof(Fr), stream_));
        poly_.gate_check(wires, sel_h_.as<Fr>(), p_pi, (uint32_t)n, reinterpret_cast<uint32_t *>(td + FLAG_AT));
        poly_.widget_check(wires, sel_h_.as<Fr>(), (uint32_t)n, widgets_, reinterpret_cast<uint32_t *>(td + FLAG_AT));
        const Fr *src[4] = {p_a, p_b, p_c, p_d};
        const uint64_t first[4] = {n - 5, n - 5, n - 5, n - 5};
        poly_.gather_windows(4, src, first, 7, td);
        ZK_HIP(hipMemcpyAsync(tops_host_, td, (FLAG_AT + 1) * sizeof(Fr), hipMemcpyDeviceToHost, stream_));
    }
    last_half_ = false;
    // The coset evaluations the quotient needs of a polynomial depend on no challenge: they are launched right behind
    // the polynomial's commitment and run while the host adds up the commitment and feeds the transcript.
    ev_ready_ = 0;
    struct ClearOnExit {
        uint32_t &flags;
        ~ClearOnExit() { flags = 0; }   // an error return must not leave the marks for a later quotient() call
    } clear_ready{ev_ready_};
    small_.reserve(coset_buffer_bytes());
    Fr *const coset_ev = small_.as<Fr>();
    {
        const Fr *cp[4] = {p_a, p_b, p_c, p_d};
        const uint64_t cl[4] = {n + 2, n + 2, n + 2, n + 2};
        rc = commit_many(4, cp, cl, &out.comms[0], [&] {
            // (over the 4n coset on the expectation that the checks pass)
            for (int k = 0; k < 4; k++) {
                if (half_coset_)
                    coset_fft_4n(wp[k], n + 2, coset_ev + (size_t)(k + 1) * 4 * n);
                else
                    coset_fft_8n(wp[k], n + 2, coset_ev + (size_t)(k + 1) * ev_len_);
            }
            ev_ready_ |= 2u;
        });
        if (rc) return rc;
    }
    bool half = half_coset_ && *reinterpret_cast<const volatile uint32_t *>(tops_host_ + FLAG_AT) == 0;
    tr.append_commitment("a_comm", out.comms[0]);
    tr.append_commitment("b_comm", out.comms[1]);
    tr.append_commitment("c_comm", out.comms[2]);
    tr.append_commitment("d_comm", out.comms[3]);
    lap(0, t_phase);

    // protocol phase 2: permutation accumulator (prover.rs:271-293)
    const Fr beta = tr.challenge_scalar("beta");
    tr.append_scalar("beta", beta);
    const Fr gamma = tr.challenge_scalar("gamma");
    {
        const Fr *wv[4] = {wires, wires + n, wires + 2 * n, wires + 3 * n};
        const Fr *sv[4] = {sigma_.as<Fr>(), sigma_.as<Fr>() + n, sigma_.as<Fr>() + 2 * n, sigma_.as<Fr>() + 3 * n};
        bool closed = false;
        poly_.permutation_z(wv, sv, beta, gamma, ntt_.roots_half(log_n_), (uint32_t)n, z_evals, &closed);
        half = half && closed;
    }
    blind_poly(z_evals, 2, bl + 8, p_z);
    if (half) {
        Fr *td = tops_dev_.as<Fr>();
        const Fr *src[1] = {p_z};
        const uint64_t first[1] = {n - 4};
        poly_.gather_windows(1, src, first, 7, td + Z_AT);
        ZK_HIP(hipMemcpyAsync(tops_host_ + Z_AT, td + Z_AT, 7 * sizeof(Fr), hipMemcpyDeviceToHost, stream_));
    } else if (half_coset_) {
        ev_ready_ = 0;   // the wires' evaluations over the 4n coset are of no use to the 8n quotient
    }
    {
        const Fr *cp[1] = {p_z};
        const uint64_t cl[1] = {n + 3};
        rc = commit_many(1, cp, cl, &out.comms[4], [&] {
            if (half) {
                coset_fft_4n(p_z, n + 3, coset_ev);
                ev_ready_ |= 1u;
            } else if (!half_coset_) {
                coset_fft_8n(p_z, n + 3, coset_ev);
                ev_ready_ |= 1u;
            }
        });
        if (rc) return rc;
    }
    tr.append_commitment("z_comm", out.comms[4]);
    lap(1, t_phase);

    // protocol phase 3: quotient (prover.rs:295-361)
    const Fr alpha = tr.challenge_scalar("alpha");
    const Fr range_ch = tr.challenge_scalar("range separation challenge");
    const Fr logic_ch = tr.challenge_scalar("logic separation challenge");
    const Fr fixed_ch = tr.challenge_scalar("fixed base separation challenge");
    const Fr var_ch = tr.challenge_scalar("variable base separation challenge");
    std::vector<Fr> pi_resident;
    // the resident Lagrange vectors lie over the coset the key proves on: a proof that fell back to the 8n coset
    // transforms its public-input polynomial
    if (!pi_lagrange_idx_.empty() && half == half_coset_) {
        // the values in the order of the resident Lagrange vectors; positions this proof leaves out are zero
        pi_resident.assign(pi_lagrange_idx_.size(), Fr::zero());
        for (size_t j = 0; j < pi_lagrange_idx_.size(); j++) {
            auto it = pi_last.find(pi_lagrange_idx_[j]);
            if (it != pi_last.end()) pi_resident[j] = it->second;
        }
    } else {
        ntt_.transform(NTT_IFFT, p_pi, n, p_pi, log_n_);   // prover.rs:305-306
    }
    const Fr ch7[7] = {alpha, beta, gamma, range_ch, logic_ch, fixed_ch, var_ch};
    const uint64_t wl[4] = {n + 2, n + 2, n + 2, n + 2};
    const uint64_t t_len = half ? quotient_half_coset(p_z, wp, ch7, pi_resident.empty() ? nullptr : pi_resident.data(), p_pi, t)
                           : pi_resident.empty() ? quotient(p_z, n + 3, wp, wl, p_pi, n, ch7, t)
                                                 : quotient(p_z, n + 3, wp, wl, nullptr, 0, ch7, t, pi_resident.data());
    last_half_ = half;
    proved_once_ = true;
    lap(2, t_phase);
    // prover.rs:327-351: the reference slices t[3n..] and indexes t_fourth[0]; both panic when t is shorter
    if (t_len < 3 * n + 1) return ZKGPU_ERR_QUOTIENT_SHAPE;
    const uint64_t t4_len = t_len - 3 * n;
    Fr *t_fourth = t + 3 * n;
    {
        // t_low.push(b_12); t_mid[0] -= b_12, push(b_13); t_high[0] -= b_13, push(b_14); t_fourth[0] -= b_14
        const Fr b[3] = {bl[11], bl[12], bl[13]};
        poly_.split_quotient(t, n, t_low, t_mid, t_high, b);
    }
    {
        const Fr *cp[4] = {t_low, t_mid, t_high, t_fourth};
        const uint64_t cl[4] = {n + 1, n + 1, n + 1, t4_len};
        if ((rc = commit_many(4, cp, cl, &out.comms[5]))) return rc;
    }
    tr.append_commitment("t_low_comm", out.comms[5]);
    tr.append_commitment("t_mid_comm", out.comms[6]);
    tr.append_commitment("t_high_comm", out.comms[7]);
    tr.append_commitment("t_fourth_comm", out.comms[8]);
    lap(3, t_phase);

    // protocol phase 4: evaluations (prover.rs:363-437)
    const Fr z_ch = tr.challenge_scalar("z_challenge");
    const Fr z_w = Fr::mul(z_ch, domain.group_gen);
    enum { A, B, C, D, AW, BW, DW, QARITH, QC, QL, QR, SG1, SG2, SG3, ZE };
    {
        EvalJob jobs[15];
        auto job = [&](int slot, const Fr *p, uint64_t len, const Fr &x) {
            jobs[slot].coeffs = p;
            jobs[slot].len = len;
            jobs[slot].x = x;
        };
        job(A, p_a, n + 2, z_ch);
        job(B, p_b, n + 2, z_ch);
        job(C, p_c, n + 2, z_ch);
        job(D, p_d, n + 2, z_ch);
        job(AW, p_a, n + 2, z_w);
        job(BW, p_b, n + 2, z_w);
        job(DW, p_d, n + 2, z_w);
        job(QARITH, key_poly(Q_ARITH), n, z_ch);
        job(QC, key_poly(Q_C), n, z_ch);
        job(QL, key_poly(Q_L), n, z_ch);
        job(QR, key_poly(Q_R), n, z_ch);
        job(SG1, key_poly(S1), n, z_ch);
        job(SG2, key_poly(S2), n, z_ch);
        job(SG3, key_poly(S3), n, z_ch);
        job(ZE, p_z, n + 3, z_w);
        poly_.evaluate(jobs, 15, ev_dev);
        ZK_HIP(hipMemcpyAsync(out.evals, ev_dev, 15 * sizeof(Fr), hipMemcpyDeviceToHost, stream_));
        ZK_HIP(hipStreamSynchronize(stream_));
    }
    const Fr *e = out.evals;
    tr.append_scalar("a_eval", e[A]);
    tr.append_scalar("b_eval", e[B]);
    tr.append_scalar("c_eval", e[C]);
    tr.append_scalar("d_eval", e[D]);
    tr.append_scalar("s_sigma_1_eval", e[SG1]);
    tr.append_scalar("s_sigma_2_eval", e[SG2]);
    tr.append_scalar("s_sigma_3_eval", e[SG3]);
    tr.append_scalar("z_eval", e[ZE]);
    tr.append_scalar("a_w_eval", e[AW]);
    tr.append_scalar("b_w_eval", e[BW]);
    tr.append_scalar("d_w_eval", e[DW]);
    tr.append_scalar("q_arith_eval", e[QARITH]);
    tr.append_scalar("q_c_eval", e[QC]);
    tr.append_scalar("q_l_eval", e[QL]);
    tr.append_scalar("q_r_eval", e[QR]);
    lap(4, t_phase);

    // protocol phase 5: linearization (linearization_poly.rs:146-272) and openings (prover.rs:439-487)
    const Fr v_ch = tr.challenge_scalar("v_challenge");
    const Fr *t_parts[4] = {t_low, t_mid, t_high, t_fourth};
    const uint64_t t_lens[4] = {n + 1, n + 1, n + 1, t4_len};
    // z's coefficients n, n + 1, n + 2 are its blinders (prover.rs:64-83): its length is known unless all three vanish
    uint64_t z_eff = 0;
    for (int i = 2; i >= 0 && !z_eff; i--)
        if (!bl[8 + i].is_zero()) z_eff = n + i + 1;
    const uint64_t r_len = linearization(ch7, z_ch, e, p_z, n + 3, t_parts, t_lens, public_inputs, r_poly, lin_cap, z_eff);
    lap(5, t_phase);
    // Both opening witnesses are built first and committed together: the prover's transcript does not absorb
    // w_z_chall_comm (prover.rs:439-479), so v_w does not depend on it.  t[0..3n) is free again and holds the second.
    Fr *wit2 = t;
    uint64_t wit_len;
    {
        // compute_aggregate_witness (kzg10/key.rs:131-146) at z with powers of v
        const Fr *ps[8] = {r_poly, p_a, p_b, p_c, p_d, key_poly(S1), key_poly(S2), key_poly(S3)};
        const uint64_t ls[8] = {r_len, n + 2, n + 2, n + 2, n + 2, n, n, n};
        wit_len = aggregate_witness(ps, ls, 8, z_ch, v_ch, agg, wit);
    }
    const Fr v_w_ch = tr.challenge_scalar("v_w_challenge");
    {
        const Fr *ps[4] = {p_z, p_a, p_b, p_d};
        const uint64_t ls[4] = {n + 3, n + 2, n + 2, n + 2};
        aggregate_witness(ps, ls, 4, z_w, v_w_ch, agg, wit2);
    }
    {
        const Fr *cp[2] = {wit, wit2};
        const uint64_t cl[2] = {wit_len, n + 2};
        if ((rc = commit_many(2, cp, cl, &out.comms[9]))) return rc;
    }
    lap(6, t_phase);
    const Fr chs[10] = {beta, gamma, alpha, range_ch, logic_ch, fixed_ch, var_ch, z_ch, v_ch, v_w_ch};
    for (int i = 0; i < 10; i++) out.challenges[i] = chs[i];
    if (timings) timings->ms[7] = std::chrono::duration<float, std::milli>(clk::now() - t_start).count();
    return 0;
}

}  // namespace zk
