// extern "C" boundary (include/zkgpu.h): context management and argument checking around the engines.
#include "../../include/zkgpu.h"
#include "common.hpp"
#include "msm.hpp"
#include "fq28.hpp"
#include "ntt.hpp"
#include "poly.hpp"
#include "prover.hpp"
#include "transcript.hpp"
#include <cstring>
#include <memory>
#include <vector>
#include <mutex>

using namespace zk;

struct zkgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::unique_ptr<NttEngine> ntt;
    std::unique_ptr<MsmEngine> msm;
    std::unique_ptr<PolyEngine> poly;
    std::unique_ptr<Prover> prover;
    DevBuf stage_a, stage_b;  // staging for the host-pointer entry points
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string last_error;
    Collective coll;                   // as set by zkgpu_set_collective
    void *noalloc_scratch = nullptr;   // this device's copy of zkgpu_noalloc_scratch (device address)
    // Calls on one context are serialised here: the reference's Prover::prove takes &self and may be called from
    // several threads (plonk/src/compiler/prover.rs:21,210), while a context owns one stream and one set of buffers.
    std::recursive_mutex lock;
};

// ---- scratch of the allocation-free entry point (zkgpu_ntt_dev_noalloc) ---------------------------------------------
// The buffer between the passes of a transform of up to 2^ZKGPU_NOALLOC_MAX_LOG_N elements.  Every GPU the library
// runs on holds its own copy of this array, at its own address, until the process ends.  In host code the array's
// name is only a handle for hipGetSymbolAddress: the device address is looked up once per device in zkgpu_init and
// kept in g_noalloc_scratch; that pointer, never the name, is what reaches the kernels.
static __device__ __align__(256) uint4 zkgpu_noalloc_scratch[((size_t)1 << ZKGPU_NOALLOC_MAX_LOG_N) * 2];   // 32 bytes per Fr
static const int MAX_DEVICES = 64;
static void *g_noalloc_scratch[MAX_DEVICES] = {nullptr};
// Contexts on one GPU share that GPU's array while each runs on a stream of its own.  Their uses are put in order by
// one event per device: a call makes its stream wait for the event, launches, and records the event again behind its
// last pass, all under g_noalloc_lock, so that the next call (from whichever context) waits for exactly the previous
// one.  Waiting for and recording an existing event allocates nothing.
static hipEvent_t g_noalloc_done[MAX_DEVICES] = {nullptr};
static std::mutex g_noalloc_lock;

// With `device` current: the address of this device's copy of the scratch array (looked up, and the device's ordering
// event created, on the first call for that device).
static void *noalloc_scratch_for(int device) {
    ZK_REQUIRE(device >= 0 && device < MAX_DEVICES, ZKGPU_ERR_INVALID_ARG, "device id beyond the library's table");
    std::lock_guard<std::mutex> hold(g_noalloc_lock);
    if (!g_noalloc_scratch[device]) {
        void *ptr = nullptr;
        ZK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(zkgpu_noalloc_scratch)));
        ZK_REQUIRE(ptr != nullptr, ZKGPU_ERR_HIP, "hipGetSymbolAddress returned a null scratch address");
        hipEvent_t ev = nullptr;
        ZK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g_noalloc_done[device] = ev;
        g_noalloc_scratch[device] = ptr;
    }
    return g_noalloc_scratch[device];
}

static thread_local std::string g_init_error;
static thread_local std::string t_error;                 // message of this thread's last failing call on t_error_ctx
static thread_local const zkgpu_ctx *t_error_ctx = nullptr;

// Runs `body`, mapping exceptions to error codes and leaving the message in `err`; never lets one cross the C
// boundary.
template <class F>
static int no_throw(std::string &err, F body) {
    try {
        body();
        return ZKGPU_OK;
    } catch (const ZkError &e) {
        err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        err = "host allocation failed";
        return ZKGPU_ERR_OOM;
    } catch (const std::exception &e) {
        err = e.what();
        return ZKGPU_ERR_INTERNAL;
    } catch (...) {
        err = "unknown error";
        return ZKGPU_ERR_INTERNAL;
    }
}
// The same for a call on a context, holding the context's lock for the whole call.
template <class F>
static int guarded(zkgpu_ctx *ctx, F body) {
    if (!ctx) return ZKGPU_ERR_INVALID_ARG;
    std::lock_guard<std::recursive_mutex> hold(ctx->lock);
    const int rc = no_throw(ctx->last_error, [&] {
        ZK_HIP(hipSetDevice(ctx->device));
        body();
    });
    if (rc != ZKGPU_OK) {   // the calling thread's own copy: another thread's failure cannot overwrite it
        t_error = ctx->last_error;
        t_error_ctx = ctx;
    }
    return rc;
}

static void fr_to_u64(const Fr &a, uint64_t out[4]) { memcpy(out, a.v, 32); }

static void affine_out(const G1Proj &p, uint64_t out_xy[12], uint8_t *out_infinity) {
    Fq x, y;
    bool inf;
    g1_to_affine(p, x, y, inf);
    memcpy(out_xy, x.v, 48);
    memcpy(out_xy + 6, y.v, 48);
    if (out_infinity) *out_infinity = inf ? 1 : 0;
}

extern "C" {

int zkgpu_init(zkgpu_ctx **out, int device_id) {
    if (!out) return ZKGPU_ERR_INVALID_ARG;
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_init_error = "no HIP device available (this library has no CPU fallback)";
        return ZKGPU_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) {
        g_init_error = "device id out of range";
        return ZKGPU_ERR_INVALID_ARG;
    }
    zkgpu_ctx *ctx = new (std::nothrow) zkgpu_ctx();
    if (!ctx) {
        g_init_error = "host allocation failed";
        return ZKGPU_ERR_OOM;
    }
    ctx->device = device_id;
    const int rc = no_throw(g_init_error, [&] {
        ZK_HIP(hipSetDevice(device_id));
        ZK_HIP(hipStreamCreate(&ctx->stream));
        ZK_HIP(hipEventCreate(&ctx->ev0));
        ZK_HIP(hipEventCreate(&ctx->ev1));
        ctx->noalloc_scratch = noalloc_scratch_for(device_id);
        ctx->ntt.reset(new NttEngine(ctx->stream));
        ctx->msm.reset(new MsmEngine(ctx->stream));
        ctx->poly.reset(new PolyEngine(ctx->stream));
        ctx->prover.reset(new Prover(ctx->stream, *ctx->ntt, *ctx->msm, *ctx->poly));
    });
    if (rc != ZKGPU_OK) {
        zkgpu_destroy(ctx);   // releases whatever was created: engines, events, stream
        return rc;
    }
    *out = ctx;
    return ZKGPU_OK;
}

void zkgpu_destroy(zkgpu_ctx *ctx) {
    if (!ctx) return;
    if (t_error_ctx == ctx) t_error_ctx = nullptr;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->prover.reset();
    ctx->poly.reset();
    ctx->ntt.reset();
    ctx->msm.reset();
    ctx->stage_a.release();
    ctx->stage_b.release();
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int zkgpu_ctx_clone(zkgpu_ctx *src, zkgpu_ctx **out) {
    if (!src || !out) return ZKGPU_ERR_INVALID_ARG;
    *out = nullptr;
    zkgpu_ctx *ctx = nullptr;
    int rc = zkgpu_init(&ctx, src->device);
    if (rc != ZKGPU_OK) return rc;
    rc = guarded(src, [&] {
        ZK_HIP(hipStreamSynchronize(src->stream));   // whatever built the table and the key has finished
        if (src->msm->n_points() > 0) ctx->msm->share_table(*src->msm);
        ctx->prover->force_full_coset(src->prover->force_full());
        if (src->prover->loaded()) ctx->prover->share_key(*src->prover);
    });
    if (rc != ZKGPU_OK) {
        zkgpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return ZKGPU_OK;
}

size_t zkgpu_device_bytes_shared(const zkgpu_ctx *ctx) {
    if (!ctx) return 0;
    std::lock_guard<std::recursive_mutex> hold(const_cast<zkgpu_ctx *>(ctx)->lock);
    return ctx->msm->shared_table_bytes() + ctx->prover->shared_key_bytes();
}

const char *zkgpu_last_error(const zkgpu_ctx *ctx) {
    if (!ctx) return g_init_error.c_str();
    if (ctx == t_error_ctx) return t_error.c_str();
    return ctx->last_error.c_str();
}

int zkgpu_sync(zkgpu_ctx *ctx) {
    return guarded(ctx, [&] { ZK_HIP(hipStreamSynchronize(ctx->stream)); });
}

size_t zkgpu_device_bytes(const zkgpu_ctx *ctx) {
    if (!ctx) return 0;
    std::lock_guard<std::recursive_mutex> hold(const_cast<zkgpu_ctx *>(ctx)->lock);
    return ctx->ntt->device_bytes() + ctx->msm->device_bytes() + ctx->poly->device_bytes() + ctx->prover->device_bytes() +
           ctx->stage_a.cap + ctx->stage_b.cap;
}

int zkgpu_timer_start(zkgpu_ctx *ctx) {
    return guarded(ctx, [&] { ZK_HIP(hipEventRecord(ctx->ev0, ctx->stream)); });
}
int zkgpu_timer_stop(zkgpu_ctx *ctx, float *elapsed_ms) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(elapsed_ms != nullptr, ZKGPU_ERR_INVALID_ARG, "timer_stop: null output");
        ZK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        ZK_HIP(hipEventSynchronize(ctx->ev1));
        ZK_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    });
}

int zkgpu_domain_new(uint64_t num_coeffs, zkgpu_domain *out) {
    if (!out) return ZKGPU_ERR_INVALID_ARG;
    return no_throw(g_init_error, [&] {
        Domain d = Domain::make(num_coeffs);
        out->size = d.size;
        out->log_size_of_group = d.log_size;
        fr_to_u64(d.size_as_field_element, out->size_as_field_element);
        fr_to_u64(d.size_inv, out->size_inv);
        fr_to_u64(d.group_gen, out->group_gen);
        fr_to_u64(d.group_gen_inv, out->group_gen_inv);
        fr_to_u64(d.generator_inv, out->generator_inv);
    });
}

int zkgpu_ntt_dev(zkgpu_ctx *ctx, const void *in_dev, size_t len_in, void *out_dev, uint32_t log_n, int mode) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(mode >= 0 && mode <= 3, ZKGPU_ERR_INVALID_ARG, "ntt: mode must be 0..3");
        ctx->ntt->transform((NttMode)mode, static_cast<const Fr *>(in_dev), len_in, static_cast<Fr *>(out_dev), log_n);
    });
}

int zkgpu_ntt_dev_noalloc(zkgpu_ctx *ctx, const void *in_dev, size_t len_in, void *out_dev, uint32_t log_n, int mode) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(mode >= 0 && mode <= 3, ZKGPU_ERR_INVALID_ARG, "ntt: mode must be 0..3");
        ZK_REQUIRE(log_n <= ZKGPU_NOALLOC_MAX_LOG_N, ZKGPU_ERR_INVALID_ARG,
                   "ntt_dev_noalloc: the transform exceeds the 2^20-element scratch of the allocation-free entry point");
        // one user of the device's scratch array at a time, in the order the calls arrive (see g_noalloc_done)
        std::lock_guard<std::mutex> hold(g_noalloc_lock);
        hipEvent_t done = g_noalloc_done[ctx->device];
        ZK_HIP(hipStreamWaitEvent(ctx->stream, done, 0));
        struct RecordOnExit {   // also behind a call that failed half-way: its launched passes still use the array
            hipEvent_t ev;
            hipStream_t st;
            ~RecordOnExit() { (void)hipEventRecord(ev, st); }
        } record{done, ctx->stream};
        ctx->ntt->transform_no_alloc((NttMode)mode, static_cast<const Fr *>(in_dev), len_in, static_cast<Fr *>(out_dev), log_n,
                                     static_cast<Fr *>(ctx->noalloc_scratch));
    });
}

int zkgpu_ntt(zkgpu_ctx *ctx, uint64_t *data, size_t len_in, uint32_t log_n, int mode) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(mode >= 0 && mode <= 3, ZKGPU_ERR_INVALID_ARG, "ntt: mode must be 0..3");
        ZK_REQUIRE(data != nullptr, ZKGPU_ERR_INVALID_ARG, "ntt: null data");
        ZK_REQUIRE(log_n < 32, ZKGPU_ERR_DOMAIN_SIZE, "InvalidEvalDomainSize: log2(size) must be below 32");
        const size_t n = (size_t)1 << log_n;
        if (len_in > n) len_in = n;
        ctx->stage_a.reserve(n * sizeof(Fr));
        if (len_in)
            ZK_HIP(hipMemcpyAsync(ctx->stage_a.p, data, len_in * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        ctx->ntt->transform((NttMode)mode, ctx->stage_a.as<Fr>(), len_in, ctx->stage_a.as<Fr>(), log_n);
        ZK_HIP(hipMemcpyAsync(data, ctx->stage_a.p, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int zkgpu_srs_load(zkgpu_ctx *ctx, const uint64_t *xy, const uint8_t *infinity, size_t n_points) {
    return guarded(ctx, [&] { ctx->msm->load_points(xy, infinity, n_points); });
}
int zkgpu_srs_load_dev(zkgpu_ctx *ctx, const void *xy_dev, size_t n_points) {
    return guarded(ctx, [&] { ctx->msm->load_points_device(xy_dev, n_points); });
}

static void msm_checked(zkgpu_ctx *ctx, const Fr *scalars_dev, size_t n, uint64_t out_xy[12], uint8_t *out_infinity) {
    ZK_REQUIRE(out_xy != nullptr, ZKGPU_ERR_INVALID_ARG, "msm: null output");
    ZK_REQUIRE(ctx->msm->n_points() > 0, ZKGPU_ERR_NOT_LOADED, "msm: no SRS loaded");
    if (n > ctx->msm->n_points()) n = ctx->msm->n_points();  // the reference's zip stops at the shorter slice
    G1Proj sum = n ? ctx->msm->msm(scalars_dev, n, 0) : G1Proj::identity();
    affine_out(sum, out_xy, out_infinity);
}

int zkgpu_msm_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n, uint64_t out_xy[12], uint8_t *out_infinity) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(scalars_dev != nullptr || n == 0, ZKGPU_ERR_INVALID_ARG, "msm: null scalars");
        msm_checked(ctx, static_cast<const Fr *>(scalars_dev), n, out_xy, out_infinity);
    });
}

int zkgpu_ntt_set_params(zkgpu_ctx *ctx, uint32_t tile_log, uint32_t elems_per_thread_log) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(tile_log == 0 || (tile_log >= 6 && tile_log <= 12), ZKGPU_ERR_INVALID_ARG,
                   "ntt_set_params: tile_log must be 0 or within 6..12");
        ZK_REQUIRE(elems_per_thread_log <= 3, ZKGPU_ERR_INVALID_ARG, "ntt_set_params: elems_per_thread_log must be within 0..3");
        ctx->ntt->set_tile(tile_log ? tile_log : NTT_DEFAULT_TILE_LOG, elems_per_thread_log ? elems_per_thread_log : NTT_DEFAULT_EPT_LOG);
    });
}

int zkgpu_srs_setup(zkgpu_ctx *ctx, const uint64_t tau[4], const uint64_t g_xy[12], size_t n_points) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(tau && g_xy, ZKGPU_ERR_INVALID_ARG, "srs_setup: null argument");
        Fr t;
        G1Affine g;
        memcpy(&t, tau, sizeof t);
        memcpy(&g, g_xy, sizeof g);
        ctx->msm->setup_powers(t, g, n_points);
    });
}

int zkgpu_srs_export(zkgpu_ctx *ctx, uint64_t *out_xy, uint8_t *out_infinity, size_t n) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(out_xy || n == 0, ZKGPU_ERR_INVALID_ARG, "srs_export: null output");
        ctx->msm->export_points(out_xy, out_infinity, n);
    });
}

int zkgpu_srs_load_raw(zkgpu_ctx *ctx, const uint8_t *bytes, size_t len) {
    return guarded(ctx, [&] { ctx->msm->load_points_raw(bytes, len); });
}

int zkgpu_srs_export_raw(zkgpu_ctx *ctx, uint8_t *out, size_t cap, size_t *written) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(written != nullptr, ZKGPU_ERR_INVALID_ARG, "srs_export_raw: null size output");
        *written = ctx->msm->raw_size();
        if (out == nullptr || cap < *written) return;
        ctx->msm->export_points_raw(out);
    });
}

int zkgpu_msm_set_params(zkgpu_ctx *ctx, uint32_t window_bits, uint32_t segment_bits) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(window_bits == 0 || (window_bits >= 4 && window_bits <= 22), ZKGPU_ERR_INVALID_ARG,
                   "msm_set_params: window_bits must be 0 or within 4..22");
        ZK_REQUIRE(segment_bits <= 6, ZKGPU_ERR_INVALID_ARG, "msm_set_params: segment_bits must be 0 or within 1..6");
        ctx->msm->set_window_override(window_bits);
        ctx->msm->set_segment_bits(segment_bits);
    });
}

int zkgpu_msm_get_params(zkgpu_ctx *ctx, uint32_t *window_bits, uint32_t *n_windows) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(window_bits && n_windows, ZKGPU_ERR_INVALID_ARG, "msm_get_params: null output");
        ZK_REQUIRE(ctx->msm->n_points() > 0, ZKGPU_ERR_NOT_LOADED, "no SRS loaded");
        *window_bits = ctx->msm->window_bits();
        *n_windows = ctx->msm->n_windows();
    });
}

int zkgpu_msm(zkgpu_ctx *ctx, const uint64_t *scalars, size_t n, uint64_t out_xy[12], uint8_t *out_infinity) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(scalars != nullptr || n == 0, ZKGPU_ERR_INVALID_ARG, "msm: null scalars");
        size_t m = n;
        if (m > ctx->msm->n_points()) m = ctx->msm->n_points();
        ctx->stage_b.reserve((m ? m : 1) * sizeof(Fr));
        if (m) ZK_HIP(hipMemcpyAsync(ctx->stage_b.p, scalars, m * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        msm_checked(ctx, ctx->stage_b.as<Fr>(), m, out_xy, out_infinity);
    });
}

int zkgpu_msm_partial_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n, size_t base, uint64_t out_xyz[18]) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(out_xyz != nullptr, ZKGPU_ERR_INVALID_ARG, "msm: null output");
        ZK_REQUIRE(scalars_dev != nullptr || n == 0, ZKGPU_ERR_INVALID_ARG, "msm: null scalars");
        // written so that a huge base cannot wrap the sum
        ZK_REQUIRE(n <= ctx->msm->n_points() && base <= ctx->msm->n_points() - n, ZKGPU_ERR_INVALID_ARG,
                   "msm: point range outside the loaded SRS");
        G1Proj sum = n ? ctx->msm->msm(static_cast<const Fr *>(scalars_dev), n, base) : G1Proj::identity();
        memcpy(out_xyz, sum.x.v, 48);
        memcpy(out_xyz + 6, sum.y.v, 48);
        memcpy(out_xyz + 12, sum.z.v, 48);
    });
}

int zkgpu_msm_kernels_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(scalars_dev != nullptr && n > 0, ZKGPU_ERR_INVALID_ARG, "msm: null scalars");
        uint32_t K, c;
        ctx->msm->msm_device(static_cast<const Fr *>(scalars_dev), n, 0, K, c);
    });
}

int zkgpu_msm_profile_dev(zkgpu_ctx *ctx, const void *scalars_dev, size_t n, float phase_ms[ZKGPU_MSM_PHASES]) {
    static_assert(ZKGPU_MSM_PHASES == MsmEngine::N_PHASES, "phase count of the header and the engine differ");
    return guarded(ctx, [&] {
        ZK_REQUIRE(scalars_dev != nullptr && n > 0 && phase_ms, ZKGPU_ERR_INVALID_ARG, "msm_profile: null argument");
        hipEvent_t ev[ZKGPU_MSM_PHASES + 1];
        for (auto &e : ev) e = nullptr;
        auto drop = [&] {
            ctx->msm->set_profile_events(nullptr);
            for (auto &e : ev)
                if (e) (void)hipEventDestroy(e);
        };
        try {
            for (auto &e : ev) ZK_HIP(hipEventCreate(&e));
            ctx->msm->set_profile_events(ev);
            uint32_t groups, log_seg;
            ctx->msm->msm_device(static_cast<const Fr *>(scalars_dev), n, 0, groups, log_seg);
            ZK_HIP(hipEventSynchronize(ev[ZKGPU_MSM_PHASES]));
            for (int i = 0; i < ZKGPU_MSM_PHASES; i++) ZK_HIP(hipEventElapsedTime(&phase_ms[i], ev[i], ev[i + 1]));
        } catch (...) {
            drop();
            throw;
        }
        drop();
    });
}

int zkgpu_g1_sum_to_affine(const uint64_t *xyz, size_t count, uint64_t out_xy[12], uint8_t *out_infinity) {
    if ((!xyz && count) || !out_xy) return ZKGPU_ERR_INVALID_ARG;
    G1Proj total = G1Proj::identity();
    for (size_t i = 0; i < count; i++) {
        G1Proj p;
        memcpy(p.x.v, xyz + 18 * i, 48);
        memcpy(p.y.v, xyz + 18 * i + 6, 48);
        memcpy(p.z.v, xyz + 18 * i + 12, 48);
        total = g1_add(total, p);
    }
    affine_out(total, out_xy, out_infinity);
    return ZKGPU_OK;
}

int zkgpu_commit(zkgpu_ctx *ctx, const uint64_t *coeffs, size_t len, uint64_t out_xy[12], uint8_t *out_infinity) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(coeffs != nullptr || len == 0, ZKGPU_ERR_INVALID_ARG, "commit: null polynomial");
        ZK_REQUIRE(ctx->msm->n_points() > 0, ZKGPU_ERR_NOT_LOADED, "commit: no SRS loaded");
        // Polynomial::degree (plonk/src/fft/polynomial.rs:79-93): index of the highest non-zero coefficient, 0 if none
        size_t degree = 0;
        for (size_t i = len; i-- > 0;) {
            const uint64_t *c = coeffs + 4 * i;
            if (c[0] | c[1] | c[2] | c[3]) {
                degree = i;
                break;
            }
        }
        // check_commit_degree_is_within_bounds (key.rs:108-117); max_degree = powers_of_g.len() - 1
        ZK_REQUIRE(degree != 0, ZKGPU_ERR_DEGREE_ZERO, "PolynomialDegreeIsZero");
        ZK_REQUIRE(degree <= ctx->msm->n_points() - 1, ZKGPU_ERR_DEGREE_LARGE, "PolynomialDegreeTooLarge");
        size_t m = len < ctx->msm->n_points() ? len : ctx->msm->n_points();
        ctx->stage_b.reserve(m * sizeof(Fr));
        ZK_HIP(hipMemcpyAsync(ctx->stage_b.p, coeffs, m * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        msm_checked(ctx, ctx->stage_b.as<Fr>(), m, out_xy, out_infinity);
    });
}

int zkgpu_commit_many(zkgpu_ctx *ctx, uint32_t count, const uint64_t *const *coeffs, const size_t *lens, uint64_t *out_xy,
                      uint8_t *out_infinity) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(count >= 1 && count <= MsmEngine::MAX_BATCH, ZKGPU_ERR_INVALID_ARG, "commit_many: 1 to 4 polynomials");
        ZK_REQUIRE(coeffs != nullptr && lens != nullptr && out_xy != nullptr, ZKGPU_ERR_INVALID_ARG, "commit_many: null argument");
        ZK_REQUIRE(ctx->msm->n_points() > 0, ZKGPU_ERR_NOT_LOADED, "commit_many: no SRS loaded");
        const size_t n_points = ctx->msm->n_points();
        size_t take[MsmEngine::MAX_BATCH], total = 0;
        for (uint32_t j = 0; j < count; j++) {
            ZK_REQUIRE(coeffs[j] != nullptr || lens[j] == 0, ZKGPU_ERR_INVALID_ARG, "commit_many: null polynomial");
            size_t degree = 0;   // Polynomial::degree (plonk/src/fft/polynomial.rs:79-93)
            for (size_t i = lens[j]; i-- > 0;) {
                const uint64_t *c = coeffs[j] + 4 * i;
                if (c[0] | c[1] | c[2] | c[3]) {
                    degree = i;
                    break;
                }
            }
            ZK_REQUIRE(degree != 0, ZKGPU_ERR_DEGREE_ZERO, "PolynomialDegreeIsZero");
            ZK_REQUIRE(degree <= n_points - 1, ZKGPU_ERR_DEGREE_LARGE, "PolynomialDegreeTooLarge");
            take[j] = degree + 1;   // the coefficients above the degree are zero
            total += take[j];
        }
        ctx->stage_b.reserve(total * sizeof(Fr));
        const Fr *ptr[MsmEngine::MAX_BATCH];
        size_t at = 0;
        for (uint32_t j = 0; j < count; j++) {
            Fr *dst = ctx->stage_b.as<Fr>() + at;
            ZK_HIP(hipMemcpyAsync(dst, coeffs[j], take[j] * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
            ptr[j] = dst;
            at += take[j];
        }
        G1Proj sums[MsmEngine::MAX_BATCH];
        ctx->msm->msm_batch(count, ptr, take, 0, sums);
        Fq xs[MsmEngine::MAX_BATCH], ys[MsmEngine::MAX_BATCH];
        bool inf[MsmEngine::MAX_BATCH];
        g1_to_affine_many(sums, count, xs, ys, inf);   // one inversion for the batch
        for (uint32_t j = 0; j < count; j++) {
            memcpy(out_xy + 12 * j, xs[j].v, 48);
            memcpy(out_xy + 12 * j + 6, ys[j].v, 48);
            if (out_infinity) out_infinity[j] = inf[j] ? 1 : 0;
        }
    });
}

int zkgpu_field_mul_selfcheck(int field, const uint64_t *a, const uint64_t *b, size_t count, uint64_t *out) {
    if ((field != 0 && field != 1) || ((!a || !b || !out) && count)) return ZKGPU_ERR_INVALID_ARG;
    for (size_t i = 0; i < count; i++) {
        if (field == 0) {
            Fr x, y;
            memcpy(&x, a + 4 * i, sizeof x);
            memcpy(&y, b + 4 * i, sizeof y);
            Fr z = Fr::mul(x, y);
            memcpy(out + 4 * i, &z, sizeof z);
        } else {
            Fq x, y;
            memcpy(&x, a + 6 * i, sizeof x);
            memcpy(&y, b + 6 * i, sizeof y);
            Fq z = Fq::mul(x, y);
            memcpy(out + 6 * i, &z, sizeof z);
        }
    }
    return ZKGPU_OK;
}

int zkgpu_field_inv_selfcheck(int field, const uint64_t *a, size_t count, uint64_t *out) {
    if ((field != 0 && field != 1) || ((!a || !out) && count)) return ZKGPU_ERR_INVALID_ARG;
    for (size_t i = 0; i < count; i++) {
        if (field == 0) {
            Fr x;
            memcpy(&x, a + 4 * i, sizeof x);
            const Fr z = Fr::inv(x);
            memcpy(out + 4 * i, &z, sizeof z);
        } else {
            Fq x;
            memcpy(&x, a + 6 * i, sizeof x);
            const Fq z = Fq::inv(x);
            memcpy(out + 6 * i, &z, sizeof z);
        }
    }
    return ZKGPU_OK;
}

int zkgpu_g1_bucket_selfcheck(const uint64_t *xy, const uint8_t *infinity, const uint8_t *negate, size_t n,
                              uint64_t out_xyz[18], uint64_t *violations) {
    if ((!xy && n) || !out_xyz) return ZKGPU_ERR_INVALID_ARG;
    const long before = fq28_violations();
    G1Xyzz acc = G1Xyzz::identity();
    for (size_t i = 0; i < n; i++) {
        G1Affine p;
        memcpy(&p, xy + 12 * i, sizeof p);
        if (infinity && infinity[i]) {
            p.x = Fq::zero();
            p.y = Fq::zero();
        }
        acc = xyzz_add_mixed(acc, affine_to_28(p), negate && negate[i]);
    }
    G1Proj r = xyzz_to_proj(acc);
    memcpy(out_xyz, &r, sizeof r);
    if (violations) *violations = (uint64_t)(fq28_violations() - before);
    return ZKGPU_OK;
}

int zkgpu_g1_bucket_selfcheck_tree(const uint64_t *xy, const uint8_t *infinity, const uint8_t *negate, size_t n,
                                   uint64_t out_xyz[18], uint64_t *violations) {
    if ((!xy && n) || !out_xyz) return ZKGPU_ERR_INVALID_ARG;
    return no_throw(g_init_error, [&] {
    const long before = fq28_violations();
    std::vector<G1Xyzz> level(n ? n : 1, G1Xyzz::identity());
    for (size_t i = 0; i < n; i++) {
        G1Affine p;
        memcpy(&p, xy + 12 * i, sizeof p);
        if (infinity && infinity[i]) {
            p.x = Fq::zero();
            p.y = Fq::zero();
        }
        level[i] = xyzz_add_mixed(G1Xyzz::identity(), affine_to_28(p), negate && negate[i]);
    }
    for (size_t len = level.size(); len > 1; len = (len + 1) / 2) {
        for (size_t i = 0; i < len / 2; i++) level[i] = xyzz_add(level[2 * i], level[2 * i + 1]);
        if (len & 1) level[len / 2] = level[len - 1];
    }
    G1Proj r = xyzz_to_proj(level[0]);
    memcpy(out_xyz, &r, sizeof r);
    if (violations) *violations = (uint64_t)(fq28_violations() - before);
    });
}

int zkgpu_g1_compress(const uint64_t xy[12], uint8_t infinity, uint8_t out[48]) {
    if (!xy || !out) return ZKGPU_ERR_INVALID_ARG;
    Fq x, y;
    memcpy(x.v, xy, 48);
    memcpy(y.v, xy + 6, 48);
    if (infinity) x = Fq::zero();
    Fq xc = Fq::from_mont(x), yc = Fq::from_mont(y);
    for (int i = 0; i < 12; i++) {  // big-endian bytes of the canonical integer (fp.rs:213-228)
        uint32_t limb = xc.v[11 - i];
        out[4 * i] = (uint8_t)(limb >> 24);
        out[4 * i + 1] = (uint8_t)(limb >> 16);
        out[4 * i + 2] = (uint8_t)(limb >> 8);
        out[4 * i + 3] = (uint8_t)limb;
    }
    out[0] |= 0x80;
    if (infinity) {
        out[0] |= 0x40;
    } else {
        // lexicographically_largest (fp.rs:264-278): y > (p - 1) / 2
        static const uint32_t HALF[12] = {0xffffd556u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                          0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
        uint64_t borrow = 0;
        for (int i = 0; i < 12; i++) {
            uint64_t t = (uint64_t)yc.v[i] - HALF[i] - borrow;
            borrow = (t >> 32) & 1;
        }
        if (!borrow) out[0] |= 0x20;
    }
    return ZKGPU_OK;
}

}  // extern "C"

// ---- ProverKey / Prover -----------------------------------------------------------------------------------------
static void affine_words(const AffineOut &a, uint64_t *xy, uint8_t *inf) {
    memcpy(xy, a.x.v, 48);
    memcpy(xy + 6, a.y.v, 48);
    *inf = a.infinity ? 1 : 0;
}

extern "C" {

int zkgpu_prover_key_load(zkgpu_ctx *ctx, uint32_t n, uint64_t constraints, const uint64_t *selector_evals,
                          const char *label, size_t label_len, const uint64_t *pi_indexes, size_t n_pi) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(label != nullptr || label_len == 0, ZKGPU_ERR_INVALID_ARG, "prover_key_load: null label");
        ZK_REQUIRE(pi_indexes != nullptr || n_pi == 0, ZKGPU_ERR_INVALID_ARG, "prover_key_load: null public input indexes");
        ctx->prover->load_key(n, constraints, selector_evals, std::string(label ? label : "", label_len), pi_indexes, n_pi);
    });
}

int zkgpu_prover_load_bytes(zkgpu_ctx *ctx, const uint8_t *bytes, size_t len, const uint64_t *pi_indexes, size_t n_pi,
                            uint32_t *n_out) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(pi_indexes != nullptr || n_pi == 0, ZKGPU_ERR_INVALID_ARG, "prover_load_bytes: null public input indexes");
        ctx->prover->load_key_bytes(bytes, len, pi_indexes, n_pi);
        if (n_out) *n_out = ctx->prover->n();
    });
}

int zkgpu_prover_to_bytes(zkgpu_ctx *ctx, uint8_t *out, size_t cap, size_t *written) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(written != nullptr, ZKGPU_ERR_INVALID_ARG, "prover_to_bytes: null size output");
        *written = ctx->prover->to_bytes(out, cap);
    });
}

int zkgpu_prover_set_sharding(zkgpu_ctx *ctx, uint64_t shard_base, uint64_t total_points, uint32_t world,
                              zkgpu_exchange_fn exchange, void *user) {
    return guarded(ctx, [&] { ctx->prover->set_sharding(shard_base, total_points, world, exchange, user); });
}

int zkgpu_set_collective(zkgpu_ctx *ctx, uint32_t rank, uint32_t world, zkgpu_collective_fn fn, void *user) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(fn == nullptr || (world >= 1 && rank < world), ZKGPU_ERR_INVALID_ARG, "set_collective: rank outside the world");
        ctx->coll.fn = fn;
        ctx->coll.user = user;
        ctx->coll.rank = fn ? rank : 0;
        ctx->coll.world = fn ? world : 1;
        ctx->prover->set_collective(rank, world, fn, user);
    });
}

int zkgpu_ntt_slice_dev(zkgpu_ctx *ctx, const void *in_dev, size_t len_in, void *out_dev, uint32_t log_n, int mode,
                        uint32_t rank, uint32_t world) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(mode == ZKGPU_FFT || mode == ZKGPU_COSET_FFT, ZKGPU_ERR_INVALID_ARG, "ntt_slice: mode must be fft or coset_fft");
        ZK_REQUIRE(out_dev != nullptr && (in_dev != nullptr || len_in == 0), ZKGPU_ERR_INVALID_ARG, "ntt_slice: null buffer");
        ctx->ntt->slice_forward(mode == ZKGPU_COSET_FFT, static_cast<const Fr *>(in_dev), len_in, static_cast<Fr *>(out_dev), log_n,
                                rank, world);
    });
}

int zkgpu_intt_sharded_dev(zkgpu_ctx *ctx, const void *slice_dev, void *out_dev, uint32_t log_n, int mode) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(mode == ZKGPU_IFFT || mode == ZKGPU_COSET_IFFT, ZKGPU_ERR_INVALID_ARG, "intt_sharded: mode must be ifft or coset_ifft");
        ZK_REQUIRE(slice_dev != nullptr && out_dev != nullptr, ZKGPU_ERR_INVALID_ARG, "intt_sharded: null buffer");
        ZK_REQUIRE(log_n < 32, ZKGPU_ERR_DOMAIN_SIZE, "InvalidEvalDomainSize: log2(size) must be below 32");
        ZK_REQUIRE(ctx->coll.fn != nullptr, ZKGPU_ERR_NOT_LOADED, "intt_sharded: call zkgpu_set_collective first");
        ctx->ntt->inverse_sharded(mode == ZKGPU_COSET_IFFT, static_cast<const Fr *>(slice_dev), static_cast<Fr *>(out_dev), log_n,
                                  ctx->coll);
    });
}

int zkgpu_prover_key_commitments(zkgpu_ctx *ctx, uint64_t *out_xy, uint8_t *out_infinity) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(ctx->prover->loaded(), ZKGPU_ERR_NOT_LOADED, "no prover key loaded");
        ZK_REQUIRE(out_xy && out_infinity, ZKGPU_ERR_INVALID_ARG, "null output");
        for (int k = 0; k < N_SEL; k++) affine_words(ctx->prover->selector_comms()[k], out_xy + 12 * k, out_infinity + k);
    });
}

int zkgpu_prover_key_poly(zkgpu_ctx *ctx, int which, uint64_t *out) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(ctx->prover->loaded(), ZKGPU_ERR_NOT_LOADED, "no prover key loaded");
        ZK_REQUIRE(which >= 0 && which < N_SEL && out, ZKGPU_ERR_INVALID_ARG, "prover_key_poly: bad argument");
        ZK_HIP(hipMemcpyAsync(out, ctx->prover->key_poly(which), (size_t)ctx->prover->n() * sizeof(Fr), hipMemcpyDeviceToHost,
                              ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

size_t zkgpu_prover_key_eval_len(const zkgpu_ctx *ctx) {
    if (!ctx) return 0;
    // a key load on another thread rewrites what is read here: same lock as every other entry point
    std::lock_guard<std::recursive_mutex> hold(const_cast<zkgpu_ctx *>(ctx)->lock);
    return ctx->prover->loaded() ? ctx->prover->eval_len() : 0;
}

int zkgpu_prover_key_eval8n(zkgpu_ctx *ctx, int which, uint64_t *out) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(ctx->prover->loaded(), ZKGPU_ERR_NOT_LOADED, "no prover key loaded");
        ZK_REQUIRE(which >= 0 && which < N_SEL + 2 && out, ZKGPU_ERR_INVALID_ARG, "prover_key_eval8n: bad argument");
        ZK_HIP(hipMemcpyAsync(out, ctx->prover->key_eval8n(which), ctx->prover->eval_len() * sizeof(Fr), hipMemcpyDeviceToHost,
                              ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int zkgpu_g1_decompress(const uint8_t in[48], uint64_t out_xy[12], uint8_t *out_infinity) {
    if (!in || !out_xy || !out_infinity) return ZKGPU_ERR_INVALID_ARG;
    std::string err;
    bool ok = false;
    const int rc = no_throw(err, [&] {
        AffineOut p;
        ok = g1_decompress(in, p);
        if (ok) {
            memcpy(out_xy, p.x.v, 48);
            memcpy(out_xy + 6, p.y.v, 48);
            *out_infinity = p.infinity ? 1 : 0;
        }
    });
    if (rc != ZKGPU_OK) return rc;
    return ok ? ZKGPU_OK : ZKGPU_ERR_INVALID_ARG;
}

// Proof::to_bytes (plonk/src/proof_system/proof.rs:95-123): eleven compressed commitments, then the fifteen
// evaluations as canonical little-endian scalars (ProofEvaluations::to_bytes, linearization_poly.rs:76-101)
int zkgpu_proof_to_bytes(const uint64_t *comms_xy, const uint8_t *comms_infinity, const uint64_t *evals, uint8_t out[ZKGPU_PROOF_BYTES]) {
    if (!comms_xy || !comms_infinity || !evals || !out) return ZKGPU_ERR_INVALID_ARG;
    for (int k = 0; k < 11; k++) {
        const int rc = zkgpu_g1_compress(comms_xy + 12 * k, comms_infinity[k], out + 48 * k);
        if (rc != ZKGPU_OK) return rc;
    }
    for (int k = 0; k < 15; k++) {
        Fr e;
        memcpy(e.v, evals + 4 * k, 32);
        if (!e.below_modulus()) return ZKGPU_ERR_INVALID_ARG;
        const Fr c = Fr::from_mont(e);
        memcpy(out + 11 * 48 + 32 * k, c.v, 32);
    }
    return ZKGPU_OK;
}

// Proof::from_bytes (proof.rs:125-157, linearization_poly.rs:103-137): ZKGPU_ERR_INVALID_ARG where the reference
// returns an Err (an invalid point encoding, a point outside the subgroup, a scalar not below the modulus)
int zkgpu_proof_from_bytes(const uint8_t in[ZKGPU_PROOF_BYTES], uint64_t *comms_xy, uint8_t *comms_infinity, uint64_t *evals) {
    if (!in || !comms_xy || !comms_infinity || !evals) return ZKGPU_ERR_INVALID_ARG;
    for (int k = 0; k < 11; k++) {
        const int rc = zkgpu_g1_decompress(in + 48 * k, comms_xy + 12 * k, comms_infinity + k);
        if (rc != ZKGPU_OK) return rc;
    }
    for (int k = 0; k < 15; k++) {
        Fr c;
        memcpy(c.v, in + 11 * 48 + 32 * k, 32);
        if (!c.below_modulus()) return ZKGPU_ERR_INVALID_ARG;   // BlsScalar::from_bytes (scalar.rs:256-265)
        const Fr e = Fr::to_mont(c);
        memcpy(evals + 4 * k, e.v, 32);
    }
    return ZKGPU_OK;
}

int zkgpu_prover_force_full_coset(zkgpu_ctx *ctx, int on) {
    return guarded(ctx, [&] { ctx->prover->force_full_coset(on != 0); });
}

int zkgpu_prover_last_proof_coset(const zkgpu_ctx *ctx) {
    if (!ctx) return 0;
    std::lock_guard<std::recursive_mutex> hold(const_cast<zkgpu_ctx *>(ctx)->lock);
    if (!ctx->prover->proved_once()) return 0;
    return ctx->prover->last_proof_half_coset() ? 4 : 8;
}

int zkgpu_quotient(zkgpu_ctx *ctx, const uint64_t *polys, const size_t lens[6], const uint64_t *challenges7,
                   uint64_t *t_out, size_t *t_len) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(ctx->prover->loaded(), ZKGPU_ERR_NOT_LOADED, "no prover key loaded");
        ZK_REQUIRE(polys && lens && challenges7 && t_out && t_len, ZKGPU_ERR_INVALID_ARG, "quotient: null argument");
        const size_t n = ctx->prover->n(), N8 = 8 * n;
        size_t total = 0;
        for (int i = 0; i < 6; i++) total += lens[i];
        ctx->stage_a.reserve((total + 1) * sizeof(Fr));
        ctx->stage_b.reserve(N8 * sizeof(Fr));
        if (total) ZK_HIP(hipMemcpyAsync(ctx->stage_a.p, polys, total * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        const Fr *base = ctx->stage_a.as<Fr>();
        const Fr *ptr[6];
        size_t off = 0;
        for (int i = 0; i < 6; i++) {
            ptr[i] = base + off;
            off += lens[i];
        }
        const Fr *wires[4] = {ptr[1], ptr[2], ptr[3], ptr[4]};
        const uint64_t wl[4] = {lens[1], lens[2], lens[3], lens[4]};
        Fr ch[7];
        memcpy(ch, challenges7, sizeof ch);
        uint64_t len = ctx->prover->quotient(ptr[0], lens[0], wires, wl, ptr[5], lens[5], ch, ctx->stage_b.as<Fr>());
        ZK_HIP(hipMemcpyAsync(t_out, ctx->stage_b.p, N8 * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        *t_len = (size_t)len;
    });
}

int zkgpu_quotient_kernel_ms(zkgpu_ctx *ctx, float *ms) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(ms != nullptr, ZKGPU_ERR_INVALID_ARG, "quotient_kernel_ms: null output");
        *ms = ctx->poly->last_quotient_ms();
    });
}

int zkgpu_perm_z(zkgpu_ctx *ctx, uint32_t n, const uint64_t *wires, const uint64_t *sigma_evals, const uint64_t beta[4],
                 const uint64_t gamma[4], uint64_t *z_out) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(wires && sigma_evals && beta && gamma && z_out, ZKGPU_ERR_INVALID_ARG, "perm_z: null argument");
        ZK_REQUIRE(n >= 1 && (n & (n - 1)) == 0, ZKGPU_ERR_INVALID_ARG, "perm_z: n must be a power of two");
        uint32_t log_n = 0;
        while ((1u << log_n) < n) log_n++;
        ctx->stage_a.reserve((size_t)9 * n * sizeof(Fr));
        Fr *d = ctx->stage_a.as<Fr>();
        ZK_HIP(hipMemcpyAsync(d, wires, (size_t)4 * n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(d + (size_t)4 * n, sigma_evals, (size_t)4 * n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        const Fr *w[4] = {d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n};
        const Fr *sg[4] = {d + 4 * (size_t)n, d + 5 * (size_t)n, d + 6 * (size_t)n, d + 7 * (size_t)n};
        Fr b, g;
        memcpy(b.v, beta, 32);
        memcpy(g.v, gamma, 32);
        ctx->poly->permutation_z(w, sg, b, g, ctx->ntt->roots_half(log_n), n, d + 8 * (size_t)n);
        ZK_HIP(hipMemcpyAsync(z_out, d + 8 * (size_t)n, (size_t)n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int zkgpu_poly_eval(zkgpu_ctx *ctx, const uint64_t *coeffs, size_t len, const uint64_t x[4], uint64_t out[4]) {
    return guarded(ctx, [&] {
        ZK_REQUIRE((coeffs || len == 0) && x && out, ZKGPU_ERR_INVALID_ARG, "poly_eval: null argument");
        ctx->stage_a.reserve((len + 2) * sizeof(Fr));
        Fr *d = ctx->stage_a.as<Fr>();
        if (len) ZK_HIP(hipMemcpyAsync(d + 1, coeffs, len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        EvalJob job;
        job.coeffs = d + 1;
        job.len = len;
        memcpy(job.x.v, x, 32);
        ctx->poly->evaluate(&job, 1, d);
        ZK_HIP(hipMemcpyAsync(out, d, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int zkgpu_poly_ruffini(zkgpu_ctx *ctx, const uint64_t *coeffs, size_t len, const uint64_t root[4], uint64_t *out) {
    return guarded(ctx, [&] {
        ZK_REQUIRE((coeffs || len == 0) && root && (out || len <= 1), ZKGPU_ERR_INVALID_ARG, "poly_ruffini: null argument");
        if (len <= 1) return;
        ctx->stage_a.reserve(2 * len * sizeof(Fr));
        Fr *d = ctx->stage_a.as<Fr>();
        ZK_HIP(hipMemcpyAsync(d, coeffs, len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        Fr r;
        memcpy(r.v, root, 32);
        ctx->poly->ruffini(d, len, r, d + len);
        ZK_HIP(hipMemcpyAsync(out, d + len, (len - 1) * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    });
}

int zkgpu_linearization(zkgpu_ctx *ctx, const uint64_t *polys, const size_t lens[5], const uint64_t *challenges7,
                        const uint64_t z_challenge[4], const uint64_t *evals15, const uint64_t *public_inputs, size_t n_pi,
                        uint64_t *r_out, size_t r_cap, size_t *r_len) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(ctx->prover->loaded(), ZKGPU_ERR_NOT_LOADED, "no prover key loaded");
        ZK_REQUIRE(polys && lens && challenges7 && z_challenge && evals15 && r_out && r_len && (public_inputs || n_pi == 0),
                   ZKGPU_ERR_INVALID_ARG, "linearization: null argument");
        const size_t n = ctx->prover->n();
        size_t total = 0, longest = n;
        for (int i = 0; i < 5; i++) {
            total += lens[i];
            if (lens[i] > longest) longest = lens[i];
        }
        ZK_REQUIRE(r_cap >= longest, ZKGPU_ERR_INVALID_ARG, "linearization: the output holds fewer coefficients than the longest term");
        ctx->stage_a.reserve((total + 1) * sizeof(Fr));
        ctx->stage_b.reserve(longest * sizeof(Fr));
        if (total) ZK_HIP(hipMemcpyAsync(ctx->stage_a.p, polys, total * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        const Fr *base = ctx->stage_a.as<Fr>();
        const Fr *ptr[5];
        size_t off = 0;
        for (int i = 0; i < 5; i++) {
            ptr[i] = base + off;
            off += lens[i];
        }
        Fr ch[7], z, e[15];
        memcpy(ch, challenges7, sizeof ch);
        memcpy(&z, z_challenge, sizeof z);
        memcpy(e, evals15, sizeof e);
        std::vector<Fr> pis(n_pi);
        if (n_pi) memcpy(pis.data(), public_inputs, n_pi * sizeof(Fr));
        const Fr *t_parts[4] = {ptr[1], ptr[2], ptr[3], ptr[4]};
        const uint64_t t_lens[4] = {lens[1], lens[2], lens[3], lens[4]};
        const uint64_t len = ctx->prover->linearization(ch, z, e, ptr[0], lens[0], t_parts, t_lens, pis, ctx->stage_b.as<Fr>(), longest);
        // Polynomial arithmetic truncates leading zeros (fft/polynomial.rs:66-77)
        const uint64_t eff = ctx->prover->poly().length_without_leading_zeros(ctx->stage_b.as<Fr>(), len);
        if (eff) ZK_HIP(hipMemcpyAsync(r_out, ctx->stage_b.p, eff * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        *r_len = (size_t)eff;
    });
}

int zkgpu_aggregate_witness(zkgpu_ctx *ctx, const uint64_t *polys, const size_t *lens, size_t n_polys, const uint64_t point[4],
                            const uint64_t v[4], uint64_t *out, size_t *out_len) {
    return guarded(ctx, [&] {
        ZK_REQUIRE(polys && lens && point && v && out_len && n_polys >= 1 && n_polys <= (size_t)MAX_LIN_TERMS, ZKGPU_ERR_INVALID_ARG,
                   "aggregate_witness: null argument or more than 18 polynomials");
        size_t total = 0, longest = 0;
        for (size_t i = 0; i < n_polys; i++) {
            total += lens[i];
            if (lens[i] > longest) longest = lens[i];
        }
        ZK_REQUIRE(out || longest <= 1, ZKGPU_ERR_INVALID_ARG, "aggregate_witness: null output");
        ctx->stage_a.reserve((total + 1) * sizeof(Fr));
        ctx->stage_b.reserve((2 * longest + 2) * sizeof(Fr));
        if (total) ZK_HIP(hipMemcpyAsync(ctx->stage_a.p, polys, total * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        const Fr *ptr[MAX_LIN_TERMS];
        uint64_t ls[MAX_LIN_TERMS];
        size_t off = 0;
        for (size_t i = 0; i < n_polys; i++) {
            ptr[i] = ctx->stage_a.as<Fr>() + off;
            ls[i] = lens[i];
            off += lens[i];
        }
        Fr z, vv;
        memcpy(&z, point, sizeof z);
        memcpy(&vv, v, sizeof vv);
        Fr *sum = ctx->stage_b.as<Fr>(), *wit = sum + longest + 1;
        const uint64_t len = ctx->prover->aggregate_witness(ptr, ls, (int)n_polys, z, vv, sum, wit);
        const uint64_t eff = len ? ctx->prover->poly().length_without_leading_zeros(wit, len) : 0;
        if (eff) ZK_HIP(hipMemcpyAsync(out, wit, eff * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        *out_len = (size_t)eff;
    });
}

int zkgpu_prove(zkgpu_ctx *ctx, const uint64_t *wires, const uint64_t *public_inputs, size_t n_pi,
                const uint64_t *blinders14, uint64_t *out_comms_xy, uint8_t *out_comms_infinity, uint64_t *out_evals15,
                uint64_t *out_challenges10, float *timings8) {
    int proof_rc = 0;
    int rc = guarded(ctx, [&] {
        ZK_REQUIRE(out_comms_xy && out_comms_infinity && out_evals15, ZKGPU_ERR_INVALID_ARG, "prove: null output");
        ZK_REQUIRE(public_inputs != nullptr || n_pi == 0, ZKGPU_ERR_INVALID_ARG, "prove: null public inputs");
        ProofOut proof;
        ProveTimings tm;
        memset(&tm, 0, sizeof tm);
        proof_rc = ctx->prover->prove(wires, public_inputs, n_pi, blinders14, proof, timings8 ? &tm : nullptr);
        if (proof_rc != 0) {
            ctx->last_error = proof_rc == ZKGPU_ERR_DEGREE_ZERO    ? "PolynomialDegreeIsZero"
                              : proof_rc == ZKGPU_ERR_DEGREE_LARGE ? "PolynomialDegreeTooLarge"
                                                                   : "quotient polynomial shorter than 3n + 1";
            t_error = ctx->last_error;   // this thread's copy, taken while the context is still locked
            t_error_ctx = ctx;
            return;
        }
        for (int i = 0; i < 11; i++) affine_words(proof.comms[i], out_comms_xy + 12 * i, out_comms_infinity + i);
        memcpy(out_evals15, proof.evals, sizeof proof.evals);
        if (out_challenges10) memcpy(out_challenges10, proof.challenges, sizeof proof.challenges);
        if (timings8) memcpy(timings8, tm.ms, sizeof tm.ms);
    });
    return rc != 0 ? rc : proof_rc;
}

int zkgpu_witness_prefetch(zkgpu_ctx *ctx, const uint64_t *wires_host) {
    return guarded(ctx, [&] { ctx->prover->prefetch_witness(wires_host); });
}

void *zkgpu_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void zkgpu_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int zkgpu_transcript_challenge(const char *label, const char *msg_label, const uint8_t *msg, size_t msg_len,
                               const char *ch_label, uint8_t *out, size_t out_len) {
    if (!label || !msg_label || !ch_label || (!msg && msg_len) || !out) return ZKGPU_ERR_INVALID_ARG;
    MerlinTranscript t(label);
    t.append_message(msg_label, msg, msg_len);
    t.challenge_bytes(ch_label, out, out_len);
    return ZKGPU_OK;
}

}  // extern "C"
