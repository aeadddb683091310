// Fr number-theoretic transform over the reference's EvaluationDomain
// (plonk/src/fft/domain.rs:106-196): fft / ifft / coset_fft / coset_ifft, natural order in and out.
#pragma once
#include "common.hpp"
#include "ff.hpp"
#include <map>
#include <vector>

namespace zk {

// Mirror of EvaluationDomain (domain.rs:23-44), computed on the host.
struct Domain {
    uint64_t size = 0;
    uint32_t log_size = 0;
    Fr size_as_field_element, size_inv, group_gen, group_gen_inv, generator_inv;
    // domain.rs:106-133; throws ZKGPU_ERR_DOMAIN_SIZE when log2(size) >= 32 (TWO_ADACITY)
    static Domain make(uint64_t num_coeffs);
    static Domain compute(uint32_t log_size);   // what make() keeps one copy of per size
};

enum NttMode { NTT_FFT = 0, NTT_IFFT = 1, NTT_COSET_FFT = 2, NTT_COSET_IFFT = 3 };

// The exchange steps of a transform shared by several GPUs (one context per GPU) go through the caller: `fn` runs
// one collective over device buffers of this process and returns 0 when the data has arrived.
//   op 0, all-to-all: send holds `world` blocks of block_bytes, block r for rank r; recv receives block r from rank r
//   op 1, all-gather: send holds one block; recv receives the blocks of all ranks in rank order
struct Collective {
    typedef int (*Fn)(void *user, int op, const void *send_dev, void *recv_dev, size_t block_bytes);
    Fn fn = nullptr;
    void *user = nullptr;
    uint32_t rank = 0, world = 1;
    uint32_t log_world() const {
        uint32_t l = 0;
        while ((1u << l) < world) l++;
        return l;
    }
    // the sharded transforms need a power-of-two number of ranks that divides the 8 coset rotations of the quotient
    bool usable() const { return fn != nullptr && (world == 2 || world == 4 || world == 8) && rank < world; }
};

// 2^9-element tiles (18 KiB of LDS, 128 threads, each carrying four elements through two stages): eight workgroups
// per CU overlap one tile's loads and stores with the butterflies of the others; measured best on MI355X (2^12 tiles
// leave one workgroup per CU and run 25% slower).
static const uint32_t NTT_DEFAULT_TILE_LOG = 9, NTT_DEFAULT_EPT_LOG = 2;

class NttEngine {
   public:
    explicit NttEngine(hipStream_t s) : stream_(s) {}
    ~NttEngine();
    // Transforms `len_in` elements at `in` (zero-padded or truncated to 2^log_n, domain.rs:146) into the
    // 2^log_n elements at `out`; both are device pointers and may alias.  Asynchronous on the engine's stream.
    void transform(NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n);
    // The same transform for callers that must not allocate: `work` (2^log_n elements of device memory, distinct from
    // `in` and `out`) takes the place of the engine's own inter-pass buffer, and the twiddle tables of this mode and
    // size must already be resident (any earlier transform of the same mode at this or, except for coset_ifft, a
    // larger size leaves them behind): otherwise ZKGPU_ERR_NOT_LOADED.  Calls no allocator, device or host.
    void transform_no_alloc(NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n, Fr *work);
    // True when transform(mode, .., log_n) would find every table it needs.
    bool tables_resident(NttMode mode, uint32_t log_n) const;

    // ---- one transform over `world` GPUs (four-step with the small dimension across the GPUs) ---------------------
    // The 2^log_n evaluation points w^k (times the coset generator g when `coset`) are dealt out cyclically: rank c
    // owns the points k = c + world * k'.  Those are the coset (g w^c) * <w^world>, so a rank obtains its 2^log_n / world
    // evaluations of a polynomial it holds in full with ONE local transform of that size over its own coset shift and
    // no exchange at all (inputs longer than the local size are folded modulo X^M - shift^M first).
    // out receives 2^log_n / world elements: out[k'] = p((g) w^(rank + world k')).
    void slice_forward(bool coset, const Fr *in, size_t len_in, Fr *out, uint32_t log_n, uint32_t rank, uint32_t world);
    // Inverse: `slice` holds this rank's evaluations (as slice_forward leaves them) of a polynomial of degree below
    // 2^log_n; every rank receives all 2^log_n coefficients in natural order.  Local inverse transform over the
    // rank's coset (coefficients modulo X^M - shift^M), all-to-all transpose of M / world elements per peer, a
    // world-point inverse DFT across the ranks' residues, then an all-gather.  Synchronises the stream around the
    // caller's collectives.  `slice` and `out_full` must not overlap.
    void inverse_sharded(bool coset, const Fr *slice, Fr *out_full, uint32_t log_n, const Collective &coll);
    // w^i for i < n/2 of the 2^log_n domain (device pointer into the forward stage tables); log_n >= 1
    const Fr *roots_half(uint32_t log_n);
    // Largest transform the twiddle tables currently cover (grown on demand).
    uint32_t table_log() const { return table_log_; }
    size_t device_bytes() const;
    // Tuning: log2 of the LDS tile (elements) and log2 of the elements per thread.
    void set_tile(uint32_t tile_log, uint32_t ept_log) {
        tile_log_ = tile_log;
        ept_log_ = ept_log;
    }

   private:
    struct Pass {
        uint32_t b0, log_r, log_t;
    };
    struct Plan {   // at most ceil(31 / 6) passes
        Pass pass[8];
        uint32_t count = 0;
    };
    // Tables of one coset shift s: s^i for the input scaling of coset_fft, and per size the stage tables
    // s^(-n/2h) * w_{2h}^(-i) of coset_ifft.
    struct ShiftTables {
        Fr shift, shift_inv;
        DevBuf lo, hi;                      // s^i = hi[i >> 12] * lo[i & 4095]
        uint32_t pow_log = 0;
        std::map<uint32_t, DevBuf> cinv;    // per log_n
        void release();
        size_t bytes() const;
    };
    void ensure_tables(uint32_t log_n);
    const void *coset_inv_table(ShiftTables &st, uint32_t log_n);
    void ensure_coset_powers(ShiftTables &st, uint32_t log_n);
    void set_alt_shift(const Fr &s);
    void transform_with(ShiftTables &st, NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n);
    Plan plan(uint32_t log_n) const;
    void run_passes(NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n, const void *tw, Fr *work,
                    const ShiftTables &st);

    hipStream_t stream_;
    uint32_t table_log_ = 0;
    DevBuf tw_plain_;                   // unscaled forward table of one size, for roots_half()
    uint32_t plain_log_ = 0;
    DevBuf tw_fwd_, tw_inv_;            // stage tables: entry h-1+i = w_{2h}^{+-i}, h = 1,2,4,...
    ShiftTables g_;                     // the coset generator g = 7 of coset_fft / coset_ifft (domain.rs:101,131)
    bool g_set_ = false;
    ShiftTables alt_;                   // this rank's shift (g) w^rank of a transform shared by several GPUs
    bool alt_set_ = false;
    DevBuf fold_, xchg_a_, xchg_b_;     // folded input; exchange buffers of inverse_sharded
    DevBuf work_;                       // ping buffer for multi-pass transforms
    DevBuf stage_consts_;
    bool lds_attr_set_ = false;
    uint32_t tile_log_ = NTT_DEFAULT_TILE_LOG, ept_log_ = NTT_DEFAULT_EPT_LOG;
};

}  // namespace zk
