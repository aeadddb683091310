// G1 multi-scalar multiplication over a device-resident point table: the MSM behind KZG10
// CommitKey::commit (plonk/src/commitment_scheme/kzg10/key.rs:119-129 ->
// coset-bls12_381/src/coset/multiscalar_mul.rs:143-220).
#pragma once
#include "common.hpp"
#include "g1.hpp"
#include <memory>
#include <vector>

namespace zk {

class MsmEngine {
   public:
    explicit MsmEngine(hipStream_t s) : stream_(s) {}
    ~MsmEngine();

    // Uploads n affine points (x | y, 12 x u64 Montgomery limbs each = the first 96 bytes of the reference's
    // raw point encoding, g1/coset.rs:8-24) and extends them on the device to the table of window multiples
    // 2^(c w) * P_i.  `infinity` may be null.  Replaces any previous table.
    void load_points(const uint64_t *xy_host, const uint8_t *infinity_host, size_t n);
    // Same, from a device buffer of n x 96 bytes (copied; infinity = all-zero x and y).
    void load_points_device(const void *xy_dev, size_t n);
    // Generates the table instead of uploading it: points[i] = tau^i * g (PublicParameters::setup, kzg10/srs.rs:40-71).
    void setup_powers(const Fr &tau, const G1Affine &g, size_t n);
    // Copies the first n loaded points back in the reference form (identity = (0, 1) + flag).
    void export_points(uint64_t *xy_host, uint8_t *infinity_host, size_t n);
    // CommitKey::to_raw_var_bytes / from_slice_unchecked (kzg10/key.rs:38-66): a little-endian u64 count, then 97
    // bytes per point (x | y as Montgomery limbs, infinity byte).  raw_size() = 8 + 97 * n_points().
    void load_points_raw(const uint8_t *bytes, size_t len);
    size_t raw_size() const { return 8 + 97 * n_points_; }
    void export_points_raw(uint8_t *out);
    size_t n_points() const { return n_points_; }
    // Makes this engine use the point table `src` holds, by reference: the contexts cloned from one another on a GPU
    // read one copy of the SRS and its window multiples (the reference's Prover is Clone and its clones share nothing
    // mutable, plonk/src/compiler/prover.rs:20-30).  The table lives as long as any engine refers to it; a later load
    // on either engine replaces only that engine's reference.  Both engines must be on the same device.
    void share_table(const MsmEngine &src);
    // bytes of the table when other engines refer to it too (they count in device_bytes() of every one of them)
    size_t shared_table_bytes() const { return table_ && table_.use_count() > 1 ? points_.cap + table28_.cap : 0; }

    // sum_i scalars[i] * points[base + i], i < n.  `scalars_dev` holds n Fr elements in Montgomery form
    // (device memory).  Synchronous: returns the projective sum after the device work has finished.
    G1Proj msm(const Fr *scalars_dev, size_t n, size_t base = 0);

    // Device phase only: leaves the group sums of the bucket reduction in an internal buffer and returns their
    // count and the segment width; finish() then combines them on the host.  Used to time the kernels alone.
    void msm_device(const Fr *scalars_dev, size_t n, size_t base, uint32_t &n_groups, uint32_t &log_seg);
    G1Proj finish(uint32_t n_groups, uint32_t log_seg);

    // Up to MAX_BATCH sums over the same points in one pass: out[j] = sum_i scalars[j][i] * points[base + i],
    // i < lens[j].  The sums share the accumulation launch and the latency-bound reduction launches.
    static const uint32_t MAX_BATCH = 4;
    void msm_batch(uint32_t m, const Fr *const *scalars_dev, const size_t *lens, size_t base, G1Proj *out);
    void msm_batch_device(uint32_t m, const Fr *const *scalars_dev, const size_t *lens, size_t base, uint32_t &n_groups,
                          uint32_t &log_seg);
    void finish_batch(uint32_t m, uint32_t n_groups, uint32_t log_seg, G1Proj *out);
    // finish_batch in two steps, for a caller that has other device work to launch while the host adds the group
    // sums: enqueue_result_copy() puts the copy of the m * n_groups group sums into pinned host memory on the stream;
    // once the stream (or an event recorded behind the copy) has been waited for, finish_host() combines them.
    void enqueue_result_copy(uint32_t m, uint32_t n_groups);
    void finish_host(uint32_t m, uint32_t n_groups, uint32_t log_seg, G1Proj *out);
    // 64 bytes of pinned host memory for a caller's own small read-back that rides on the same wait
    void *pinned_scratch();

    // Five HIP events recorded on the stream at the phase boundaries of the next device phases: start, before and
    // after msm_accumulate_kernel, before the reduction, end.  nullptr switches the recording off.
    // Phase boundaries of one device MSM, for timing with HIP events: event k is recorded after phase k - 1 ends.
    //   digits | range partition | count pass | offsets (totals, scan, cursors) | scatter pass | schedule |
    //   accumulate | oversized buckets | bucket reduction
    static const int N_PHASES = 9;
    void set_profile_events(hipEvent_t *events /* N_PHASES + 1 */) { profile_ = events; }
    static uint32_t choose_window(size_t n);
    size_t device_bytes() const;
    // Test hooks: window width used by the next load_points*, segment width of the bucket reduction.
    void set_window_override(uint32_t c) { window_override_ = c; }
    // 0 = automatic: 2-bucket segments up to 2^12 buckets, 4-bucket ones up to 2^15 (bound by the depth of the chains), above that 8-bucket
    // segments, 16-bucket ones for batches of three or four sums (whose reduction is bound by work)
    void set_segment_bits(uint32_t b) { seg_bits_ = b; }
    uint32_t window_bits() const { return c_; }
    uint32_t n_windows() const { return K_; }

   private:
    hipStream_t stream_;
    // second stream for the sorts of a batch's later sums, which run beside the first sum's accumulation
    hipStream_t aux_stream_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    void build_table(size_t n);

    // The resident points: owned by `table_`, which engines on one GPU may share; points_ and table28_ are this engine's
    // views of its buffers (same pointers).
    struct PointTable {
        DevBuf points, table28;
        ~PointTable() {
            points.release();
            table28.release();
        }
    };
    std::shared_ptr<PointTable> table_;
    void new_table();   // drops the reference to the current table and starts an empty one of this engine's own
    DevBuf points_;   // the loaded points in the reference's affine form (what export_points returns)
    DevBuf table28_;  // table[w * n_points + i] = 2^(c w) * P_i, affine, unsaturated limbs (112 bytes), K levels
    size_t n_points_ = 0;
    uint32_t c_ = 0, K_ = 0, seg_bits_ = 0;
    DevBuf counts_, offsets_, cursor_, sorted_, buckets_, block_sums_, tasks_, partials_, misc_, seg_, group_sums_;
    DevBuf order_, digits_, range_cnt_, mid_, slice_tasks_;
    uint32_t window_override_ = 0;
    hipEvent_t *profile_ = nullptr;
    bool lds_attr_set_ = false;
    void *host_sums_ = nullptr;   // pinned: 64 scratch bytes, then MAX_BATCH * MAX_GROUPS group sums
    static const uint32_t MAX_GROUPS = 40;
    void ensure_pinned();
    struct TailPool;              // host threads that add up the group sums of a batch side by side (finish_host)
    TailPool *tail_pool_ = nullptr;
    bool tail_pool_failed_ = false;
};

}  // namespace zk
