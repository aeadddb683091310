// Pippenger bucket MSM for gfx950 over a resident point table with precomputed window multiples.
//
// The SRS is loaded once and reused by every commitment, and HBM is plentiful, so loading it also stores
// 2^(c w) * P_i for every window w (K = floor(255/c) + 1 levels, affine).  A scalar's K signed c-bit digits then
// all select entries of ONE bucket set: sum_i s_i P_i = sum_b (b+1) * sum_{(i,w): |digit_w(s_i)| = b+1} +-T[w][i].
// There is a single bucket reduction per MSM instead of one per window, and no doubling chain at the end.
//
//   1. digits    every scalar leaves Montgomery form and is cut into K signed digits (|digit| <= 2^(c-1)); from 2^17
//                points up the partition passes of the sort do this themselves, in registers, and no digit array exists;
//   2. sort      counting sort of the n*K (entry, bucket) pairs with per-workgroup histograms in LDS (no global
//                atomics): count pass, exclusive scan = bucket start offsets, scatter pass writing table indices
//                (with the digit's sign in bit 31) bucket by bucket.  More than 2^15 buckets are first partitioned
//                into ranges of 2^13, and the workgroups of a range are kept on one XCD while they scatter;
//   3. schedule  buckets are ordered by decreasing size so that the lanes of a wave carry equal work; buckets far
//                above the expected size are cut into fixed-size tasks for whole workgroups;
//   4. accumulate one thread per bucket adds its entries with the extended-Jacobian mixed addition on 28-bit limbs
//                (fq28.hpp); oversized buckets are summed by workgroups (strided partial sums + LDS tree);
//   5. reduce    sum_b (b+1) * bucket_b: running sums over segments of L buckets give per-segment totals T_s and
//                local weighted sums W_s; sum_s s*T_s is then taken bit by bit (Q_k = sum of the T_s whose index
//                has bit k set), every sum a wide tree;
//   6. the few group sums return to the host: total = sum W + L * sum_k 2^k Q_k + sum T, then to affine.
// Up to MAX_BATCH sums over the same table run as one batch and share the sort and reduction launches.
// The reference's unsigned digits, window size and special cases for 0 and 1
// (multiscalar_mul.rs:150-195) change only the order of group additions; the sum, and so the affine
// commitment compared by the tests, is the same group element.
#include "msm.hpp"
#include "fq28.hpp"
#include <cmath>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace zk {

static const uint32_t BIG_BUCKET = 256;   // number of size classes of the schedule (count >> shift, shift grows with the
                                          // mean bucket size); also the floor of the model's oversize allowance
static const uint32_t BIG_CHUNK = 4096;   // entries per workgroup task
static const uint32_t TOP_FEW_BITS = 11;  // a top window this short piles its digits on few enough buckets for the task path
static const uint32_t SCAN_ITEMS = 1024;  // histogram entries scanned per block

struct MsmTask {
    uint32_t key, start, len, first;  // first = number of tasks of this bucket when this is its first task, else 0
};

__device__ __forceinline__ Fr gload_fr(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ Fq gload_fq(const uint4 *q) {
    uint4 a = q[0], b = q[1], c = q[2];
    Fq r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; r.v[11] = c.w;
    return r;
}
__device__ __forceinline__ void gstore_fq(uint4 *q, const Fq &r) {
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
    q[2] = make_uint4(r.v[8], r.v[9], r.v[10], r.v[11]);
}
__device__ __forceinline__ G1Affine gload_affine(const G1Affine *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    G1Affine r;
    r.x = gload_fq(q);
    r.y = gload_fq(q + 3);
    return r;
}
// Table entries sit on a 128-byte pitch (112 bytes of limbs + 16 of padding) so that the random read of one entry
// touches a single 128-byte line instead of two.
static const uint32_t TABLE_PITCH = 128;
struct alignas(128) G1Table28 {
    uint4 q[8];
};
static_assert(sizeof(G1Table28) == TABLE_PITCH, "table pitch");
__device__ __forceinline__ G1Affine28 gload_affine28(const G1Table28 *p) {
    const uint4 *q = p->q;
    G1Affine28 r;
    uint32_t w[28];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        uint4 t = q[k];
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < 14; k++) {
        r.x.v[k] = w[k];
        r.y.v[k] = w[14 + k];
    }
    return r;
}
__device__ __forceinline__ void gstore_affine28(G1Table28 *p, const G1Affine28 &a) {
    uint4 *q = p->q;
    uint32_t w[28];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        w[k] = a.x.v[k];
        w[14 + k] = a.y.v[k];
    }
#pragma unroll
    for (int k = 0; k < 7; k++) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ G1Xyzz gload_xyzz(const G1Xyzz *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint32_t w[56];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint4 t = q[k];
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
    G1Xyzz r;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        r.x.v[k] = w[k];
        r.y.v[k] = w[14 + k];
        r.zz.v[k] = w[28 + k];
        r.zzz.v[k] = w[42 + k];
    }
    return r;
}
__device__ __forceinline__ void gstore_xyzz(G1Xyzz *p, const G1Xyzz &a) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    uint32_t w[56];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        w[k] = a.x.v[k];
        w[14 + k] = a.y.v[k];
        w[28 + k] = a.zz.v[k];
        w[42 + k] = a.zzz.v[k];
    }
#pragma unroll
    for (int k = 0; k < 14; k++) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ G1Proj gload_proj(const G1Proj *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    G1Proj r;
    r.x = gload_fq(q);
    r.y = gload_fq(q + 3);
    r.z = gload_fq(q + 6);
    return r;
}
__device__ __forceinline__ void gstore_proj(G1Proj *p, const G1Proj &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    gstore_fq(q, r.x);
    gstore_fq(q + 3, r.y);
    gstore_fq(q + 6, r.z);
}

// Geometry of one sum's counting sort: slice = ceil(n_items / S)
struct SortGeom {
    uint32_t n_items, n, stride, base, B, log_range, S, slice;
    // dyn != 0: S is only the most slices a range may have; a range takes as many as its item count asks for
    // (range_slices), so that a range which repeated scalars have filled is shared by several workgroups
    uint32_t dyn;
};
// Slices of a range with `count` items when at most `cap` are allowed: two staged tiles (2^15 items) per slice
__device__ __forceinline__ uint32_t range_slices(uint32_t count, uint32_t cap) {
    const uint32_t s = (count + 32767u) >> 15;
    return s < 1u ? 1u : (s > cap ? cap : s);
}
// Per-sum arguments of a batch, passed by value: the small kernels below serve sum first_sum + blockIdx.y (or .z),
// so one launch covers either a single sum or the whole batch.
struct SumArgs {
    const Fr *scalars[MsmEngine::MAX_BATCH];
    uint32_t *dig[MsmEngine::MAX_BATCH];
    SortGeom g[MsmEngine::MAX_BATCH];
    // two-level sort: start of every (range, block) run in the partitioned list, n_blocks entries per range + the total
    const uint32_t *range_start[MsmEngine::MAX_BATCH];
    uint32_t n_blocks[MsmEngine::MAX_BATCH];
};
// Slices that hold entries of bucket b of sum `sum`
__device__ __forceinline__ uint32_t slices_of_bucket(const SumArgs &a, uint32_t sum, uint32_t b) {
    const SortGeom &g = a.g[sum];
    if (!g.dyn) return g.S;
    const uint32_t r = b >> g.log_range, nbk = a.n_blocks[sum];
    const uint32_t *rs = a.range_start[sum];
    return range_slices(rs[(size_t)(r + 1) * nbk] - rs[(size_t)r * nbk], g.S);
}

// dig[w * n + i] = |digit| | sign << 31 of scalar i in window w (0 when the digit is zero)
__global__ void __launch_bounds__(256) msm_digits_kernel(SumArgs a, uint32_t first_sum, uint32_t c, uint32_t K) {
    const uint32_t sum = first_sum + blockIdx.y;
    const Fr *scalars = a.scalars[sum];
    uint32_t *dig = a.dig[sum];
    const uint32_t n = a.g[sum].n;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = Fr::from_mont(gload_fr(scalars + i));
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t carry = 0;
    for (uint32_t w = 0; w < K; w++) {
        uint32_t raw = (s.v[0] & mask) + carry;
#pragma unroll
        for (int k = 0; k < 7; k++) s.v[k] = (s.v[k] >> c) | (s.v[k + 1] << (32 - c));
        s.v[7] >>= c;
        uint32_t mag = raw, neg = 0;
        carry = 0;
        if (raw > half) {
            mag = (1u << c) - raw;
            neg = 1;
            carry = 1;
        }
        dig[(size_t)w * n + i] = mag ? (mag | (neg << 31)) : 0u;
    }
}

// Counting sort without global atomics over the flat list of n_items = n * K digits (item j = w * n + i).
// Workgroup (s, r) owns slice s of the items, bucket range r, and a private histogram of that range in LDS.
//   pass 0 (count):   part[s * B + bucket] = number of slice entries in the bucket
//   pass 1 (scatter): part[...] holds the first output position for (s, bucket); the table index
//                     w * stride + base + i of every entry is written there
// Sum first_sum + blockIdx.z; its counters start at part + sum * part_stride.
template <int PASS>
__global__ void __launch_bounds__(1024) msm_sort_kernel(SumArgs a, uint32_t first_sum, uint32_t *part, size_t part_stride,
                                                        uint32_t *sorted) {
    extern __shared__ uint32_t hist[];
    const uint32_t sum = first_sum + blockIdx.z;
    const SortGeom g = a.g[sum];
    const uint32_t *dig = a.dig[sum];
    part += (size_t)sum * part_stride;
    const uint32_t range = 1u << g.log_range;
    const uint32_t s = blockIdx.x, r = blockIdx.y;
    if (s >= g.S) return;   // the grid is sized for the sum with the most slices
    uint32_t *mine = part + (size_t)s * g.B + ((size_t)r << g.log_range);
    for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) hist[t] = PASS ? mine[t] : 0u;
    __syncthreads();
    const uint32_t lo = s * g.slice;
    uint32_t hi = lo + g.slice;
    if (hi > g.n_items || hi < lo) hi = g.n_items;
    // four independent loads in flight per thread: the scan is bound by L2 latency, not by bandwidth
    for (uint32_t j0 = lo + threadIdx.x; j0 < hi; j0 += 4 * blockDim.x) {
        uint32_t vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = j0 + u * blockDim.x;
            vv[u] = j < hi ? dig[j] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = j0 + u * blockDim.x, v = vv[u];
            uint32_t mag = v & 0x7fffffffu;
            if (!mag) continue;
            uint32_t b = mag - 1;
            if ((b >> g.log_range) != r) continue;
            uint32_t pos = atomicAdd(&hist[b & (range - 1)], 1u);
            if (PASS) {
                uint32_t w = j / g.n, i = j - w * g.n;
                sorted[pos] = (w * g.stride + g.base + i) | (v & 0x80000000u);
            }
        }
    }
    if (!PASS) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) mine[t] = hist[t];
    }
}

// ---- two-level variant for more than one bucket range ----------------------------------------------------------
// With B > 2^15 buckets the one-level sort above makes every bucket range scan the whole digit list.  Here the
// items are first partitioned by bucket range (a counting sort with n_ranges bins, items of one workgroup counted
// and placed with LDS atomics), then every range sorts only its own items with the LDS histogram.
static const uint32_t RANGE_PER_THREAD = 16;   // items per thread in the range partition
static const uint32_t MAX_RANGES = 1024;
// pass 0: blockcnt[r * n_blocks + blk] = items of block blk in range r;  pass 1: place them
//   mid_key = bucket index inside the range | sign << 31,  mid_val = table index w * stride + base + i
template <int PASS, uint32_t THREADS>
__global__ void __launch_bounds__(THREADS) msm_range_kernel(const uint32_t *dig, SortGeom g, uint32_t n_ranges, uint32_t n_blocks,
                                                            uint32_t *blockcnt, uint32_t *mid_key, uint32_t *mid_val) {
    __shared__ uint32_t bins[MAX_RANGES];
    const uint32_t blk = blockIdx.x;
    for (uint32_t t = threadIdx.x; t < n_ranges; t += THREADS) bins[t] = PASS ? blockcnt[(size_t)t * n_blocks + blk] : 0u;
    __syncthreads();
    const uint32_t lo = blk * (RANGE_PER_THREAD * THREADS);
    uint32_t vv[RANGE_PER_THREAD];
#pragma unroll
    for (uint32_t u = 0; u < RANGE_PER_THREAD; u++) {
        const uint32_t j = lo + u * THREADS + threadIdx.x;
        vv[u] = j < g.n_items ? dig[j] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < RANGE_PER_THREAD; u++) {
        const uint32_t j = lo + u * THREADS + threadIdx.x, v = vv[u];
        const uint32_t mag = v & 0x7fffffffu;
        if (!mag) continue;
        const uint32_t b = mag - 1, r = b >> g.log_range;
        const uint32_t pos = atomicAdd(&bins[r], 1u);
        if (PASS) {
            const uint32_t w = j / g.n, i = j - w * g.n;
            mid_key[pos] = (b & ((1u << g.log_range) - 1u)) | (v & 0x80000000u);
            mid_val[pos] = w * g.stride + g.base + i;
        }
    }
    if (!PASS) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n_ranges; t += THREADS) blockcnt[(size_t)t * n_blocks + blk] = bins[t];
    }
}
// adds the scanned block sums: offsets[i] += block_sums[i / SCAN_ITEMS]
__global__ void __launch_bounds__(256) scan_add_kernel(uint32_t *offsets, const uint32_t *block_sums, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) offsets[i] += block_sums[i / SCAN_ITEMS];
}
// Workgroups b and b + 8 of a launch share an XCD and with it an L2.  The scatter pass writes 4-byte entries all over
// its range's output region; when the slices of one range run on one XCD those stores combine into full lines in
// that L2 before they leave it (measured without: 8.7 bytes written per byte of output).  S and n_ranges are powers
// of two; ranges are taken up eight at a time, one per XCD, so an L2 holds one output region at a time.
__device__ __forceinline__ void sort2_block(uint32_t lin, uint32_t S, uint32_t n_ranges, uint32_t &s, uint32_t &r) {
    if (((S * n_ranges) & 7u) != 0) {
        s = lin % S;
        r = lin / S;
        return;
    }
    const uint32_t x = lin & 7u, q = lin >> 3;
    if (n_ranges >= 8) {
        r = x + 8 * (q / S);
        s = q % S;
    } else {
        const uint32_t per = 8 / n_ranges;   // XCDs per range
        r = x % n_ranges;
        s = q * per + x / n_ranges;
    }
}
// ---- staged variants: stores leave the workgroup as contiguous runs ---------------------------------------------
// A lane-scattered 4-byte store costs one memory transaction per lane whatever the L2 later makes of it, and that
// transaction rate, not bandwidth, bounds the two placing passes above.  Here a workgroup first orders its 2^14
// items by destination in LDS (rank by LDS atomics, exclusive scan of the counts) and then writes them out slot by
// slot: consecutive lanes store consecutive addresses within a destination's run, so a wavefront's store touches a
// few lines instead of 64.
static const uint32_t STAGE_THREADS = 1024;
static const uint32_t STAGE_MAX_BINS = 1024;
// dynamic LDS for tiles of TILE items: three tables of STAGE_MAX_BINS words, TILE words of payload, TILE bin numbers (16 bit)
static constexpr uint32_t stage_lds_bytes(uint32_t tile) { return 3 * STAGE_MAX_BINS * 4 + tile * 4 + tile * 2; }

// out[t] = sum of in[0..t) for t < count <= 1024, by the 1024 threads of the workgroup; returns the total.
// wtot: 17 words of LDS.  Ends with a barrier; `in` and `out` may be the same array.
__device__ __forceinline__ uint32_t stage_exclusive_scan(const uint32_t *in, uint32_t *out, uint32_t count, uint32_t *wtot) {
    const uint32_t tid = threadIdx.x;
    const uint32_t v = tid < count ? in[tid] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((tid & 63u) >= (uint32_t)d) incl += up;
    }
    if ((tid & 63u) == 63u) wtot[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        const uint32_t t = wtot[w];
        if (w < (tid >> 6)) base += t;
        total += t;
    }
    if (tid < count) out[tid] = base + incl - v;
    __syncthreads();
    return total;
}

// The (range, slice) pairs of a sum whose ranges take range_slices() slices each: tasks[0] = their number, then
// range << 8 | slice in range order.  One workgroup of 1024 threads; n_ranges <= 1024.  The workgroups of the count
// pass and of the placing pass take one task each, so a launch needs n_ranges + items / 2^15 workgroups at most.
__global__ void __launch_bounds__(STAGE_THREADS) msm_slice_tasks_kernel(const uint32_t *range_start, uint32_t n_blocks,
                                                                        uint32_t n_ranges, uint32_t cap, uint32_t *tasks) {
    __shared__ uint32_t wtot[17], sl[STAGE_MAX_BINS], off[STAGE_MAX_BINS];
    const uint32_t tid = threadIdx.x;
    uint32_t mine = 0;
    if (tid < n_ranges) mine = range_slices(range_start[(size_t)(tid + 1) * n_blocks] - range_start[(size_t)tid * n_blocks], cap);
    sl[tid] = mine;
    __syncthreads();
    const uint32_t total = stage_exclusive_scan(sl, off, n_ranges, wtot);
    if (tid < n_ranges)
        for (uint32_t k = 0; k < mine; k++) tasks[1 + off[tid] + k] = (tid << 8) | k;
    if (tid == 0) tasks[0] = total;
}

// ---- the range partition straight from the scalars ----------------------------------------------------------------
// The two partition passes above read a digit array that a kernel of its own has written (4 bytes per (scalar, window)
// pair out, 8 in).  Here a thread takes ONE scalar, leaves Montgomery form, cuts its K <= 16 signed digits and keeps
// them in registers: the count pass and the placing pass each read the scalars (32 bytes per scalar) and repeat that
// arithmetic (one field product per scalar), and no digit array exists.  A workgroup owns 1024 consecutive scalars, all
// their windows: K * 1024 <= 2^14 items, one staged tile.  Item order inside a bucket changes with this (scalar-major
// per workgroup); the bucket sums do not depend on it.
static const uint32_t SCALAR_K_MAX = 16;
__device__ __forceinline__ void scalar_digits(const Fr *scalar, uint32_t c, uint32_t K, uint32_t *dg) {
    Fr s = Fr::from_mont(gload_fr(scalar));
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t w = 0; w < SCALAR_K_MAX; w++) {
        const uint32_t raw = (s.v[0] & mask) + carry;
#pragma unroll
        for (int k = 0; k < 7; k++) s.v[k] = (s.v[k] >> c) | (s.v[k + 1] << (32 - c));
        s.v[7] >>= c;
        uint32_t mag = raw, neg = 0;
        carry = 0;
        if (raw > half) {
            mag = (1u << c) - raw;
            neg = 1;
            carry = 1;
        }
        dg[w] = (w < K && mag) ? (mag | (neg << 31)) : 0u;
    }
}
// count pass: blockcnt[r * n_blocks + blk] = items of workgroup blk's scalars in range r
__global__ void __launch_bounds__(STAGE_THREADS) msm_range_count_scalars_kernel(const Fr *scalars, SortGeom g, uint32_t c, uint32_t K,
                                                                                uint32_t n_ranges, uint32_t n_blocks,
                                                                                uint32_t *blockcnt) {
    __shared__ uint32_t bins[MAX_RANGES];
    const uint32_t blk = blockIdx.x, tid = threadIdx.x;
    if (tid < n_ranges) bins[tid] = 0u;
    __syncthreads();
    const uint32_t i = blk * STAGE_THREADS + tid;
    if (i < g.n) {
        uint32_t dg[SCALAR_K_MAX];
        scalar_digits(scalars + i, c, K, dg);
#pragma unroll
        for (uint32_t w = 0; w < SCALAR_K_MAX; w++) {
            const uint32_t mag = dg[w] & 0x7fffffffu;
            if (mag) atomicAdd(&bins[(mag - 1) >> g.log_range], 1u);
        }
    }
    __syncthreads();
    if (tid < n_ranges) blockcnt[(size_t)tid * n_blocks + blk] = bins[tid];
}
// placing pass with staged stores (the workgroup's items are ordered by range in LDS, then stored run by run): mid_key = bucket inside the range | sign << 31,
// mid_val = table index w * stride + base + i.  (Finding a staged position's range by bisection over the ranges' start
// offsets instead of the per-position table would let two workgroups share a CU - 76 KiB of LDS each - but measured
// slower: 2.1 against 1.9 ms for the partition of 2^24 scalars.)
__global__ void __launch_bounds__(STAGE_THREADS) msm_range_place_scalars_kernel(const Fr *scalars, SortGeom g, uint32_t c, uint32_t K,
                                                                                uint32_t n_ranges, uint32_t n_blocks,
                                                                                const uint32_t *blockcnt, uint32_t *mid_key,
                                                                                uint32_t *mid_val) {
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t wtot[17];
    uint32_t *cnt = lds, *loc = lds + STAGE_MAX_BINS, *gpos = lds + 2 * STAGE_MAX_BINS, *stage = lds + 3 * STAGE_MAX_BINS;
    uint16_t *sbin = reinterpret_cast<uint16_t *>(stage + SCALAR_K_MAX * STAGE_THREADS);
    const uint32_t blk = blockIdx.x, tid = threadIdx.x;
    if (tid < n_ranges) {
        cnt[tid] = 0u;
        gpos[tid] = blockcnt[(size_t)tid * n_blocks + blk];   // where this workgroup's items of the range go
    }
    __syncthreads();
    const uint32_t mask = (1u << g.log_range) - 1u;
    const uint32_t i = blk * STAGE_THREADS + tid;
    uint32_t dg[SCALAR_K_MAX], slot[SCALAR_K_MAX];
    if (i < g.n) {
        scalar_digits(scalars + i, c, K, dg);
    } else {
#pragma unroll
        for (uint32_t w = 0; w < SCALAR_K_MAX; w++) dg[w] = 0u;
    }
#pragma unroll
    for (uint32_t w = 0; w < SCALAR_K_MAX; w++) {
        const uint32_t mag = dg[w] & 0x7fffffffu;
        slot[w] = mag ? atomicAdd(&cnt[(mag - 1) >> g.log_range], 1u) : 0u;
    }
    __syncthreads();
    const uint32_t total = stage_exclusive_scan(cnt, loc, n_ranges, wtot);
#pragma unroll
    for (uint32_t w = 0; w < SCALAR_K_MAX; w++) {
        const uint32_t mag = dg[w] & 0x7fffffffu;
        if (!mag) continue;
        const uint32_t b = mag - 1, r = b >> g.log_range;
        slot[w] += loc[r];
        stage[slot[w]] = (b & mask) | (dg[w] & 0x80000000u);
        sbin[slot[w]] = (uint16_t)r;
    }
    __syncthreads();
    for (uint32_t k = tid; k < total; k += STAGE_THREADS) {
        const uint32_t r = sbin[k];
        mid_key[gpos[r] + (k - loc[r])] = stage[k];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t w = 0; w < SCALAR_K_MAX; w++)
        if (dg[w] & 0x7fffffffu) stage[slot[w]] = w * g.stride + g.base + i;
    __syncthreads();
    for (uint32_t k = tid; k < total; k += STAGE_THREADS) {
        const uint32_t r = sbin[k];
        mid_val[gpos[r] + (k - loc[r])] = stage[k];
    }
}

// msm_sort2_kernel<1> with staged stores, for ranges of at most STAGE_MAX_BINS buckets: the slice is taken tile by
// tile; a tile's entries are ordered by bucket in LDS and appended to their buckets' runs.  A range that fits one
// tile (2^20 points) leaves the workgroup as one contiguous block.
template <uint32_t TILE>
__global__ void __launch_bounds__(STAGE_THREADS) msm_sort3_kernel(const uint32_t *mid_key, const uint32_t *mid_val,
                                                                  const uint32_t *range_start, uint32_t n_blocks, SortGeom g,
                                                                  const uint32_t *tasks, const uint32_t *part, uint32_t *sorted) {
    constexpr uint32_t PER_THREAD = TILE / STAGE_THREADS;
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t wtot[17];
    uint32_t *cursor = lds, *tcnt = lds + STAGE_MAX_BINS, *toff = lds + 2 * STAGE_MAX_BINS, *stage = lds + 3 * STAGE_MAX_BINS;
    uint16_t *sbkt = reinterpret_cast<uint16_t *>(stage + TILE);
    const uint32_t range = 1u << g.log_range, tid = threadIdx.x;
    if (blockIdx.x >= tasks[0]) return;
    const uint32_t task = tasks[1 + blockIdx.x], s = task & 0xffu, r = task >> 8;
    const uint32_t *mine = part + (size_t)s * g.B + ((size_t)r << g.log_range);
    if (tid < range) {
        cursor[tid] = mine[tid];
        tcnt[tid] = 0u;
    }
    __syncthreads();
    const uint32_t first = range_start[(size_t)r * n_blocks], last = range_start[(size_t)(r + 1) * n_blocks];
    const uint32_t n_slices = range_slices(last - first, g.S);
    const uint32_t slice = (last - first + n_slices - 1) / n_slices;
    const uint32_t lo = first + s * slice;
    uint32_t hi = lo + slice;
    if (hi > last) hi = last;
    for (uint32_t t0 = lo; t0 < hi; t0 += TILE) {
        const uint32_t tile_n = hi - t0 < TILE ? hi - t0 : TILE;
        uint32_t kk[PER_THREAD], vv[PER_THREAD], rk[PER_THREAD];
#pragma unroll
        for (uint32_t u = 0; u < PER_THREAD; u++) {
            const uint32_t k = u * STAGE_THREADS + tid;
            kk[u] = k < tile_n ? __builtin_nontemporal_load(mid_key + t0 + k) : 0xffffffffu;
            vv[u] = k < tile_n ? __builtin_nontemporal_load(mid_val + t0 + k) : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < PER_THREAD; u++)
            rk[u] = (u * STAGE_THREADS + tid) < tile_n ? atomicAdd(&tcnt[kk[u] & 0x7fffffffu], 1u) : 0u;
        __syncthreads();
        stage_exclusive_scan(tcnt, toff, range, wtot);
#pragma unroll
        for (uint32_t u = 0; u < PER_THREAD; u++) {
            if ((u * STAGE_THREADS + tid) >= tile_n) continue;
            const uint32_t b = kk[u] & 0x7fffffffu, at = toff[b] + rk[u];
            stage[at] = vv[u] | (kk[u] & 0x80000000u);
            sbkt[at] = (uint16_t)b;
        }
        __syncthreads();
        for (uint32_t k = tid; k < tile_n; k += STAGE_THREADS) {
            const uint32_t b = sbkt[k];
            sorted[cursor[b] + (k - toff[b])] = stage[k];
        }
        __syncthreads();
        if (tid < range) {
            cursor[tid] += tcnt[tid];
            tcnt[tid] = 0u;
        }
        __syncthreads();
    }
}

// Workgroup (s, r) owns slice s of range r's items [range_start[r * n_blocks], range_start[(r + 1) * n_blocks]).
template <int PASS>
__global__ void __launch_bounds__(1024) msm_sort2_kernel(const uint32_t *mid_key, const uint32_t *mid_val,
                                                         const uint32_t *range_start, uint32_t n_blocks, SortGeom g,
                                                         const uint32_t *tasks, uint32_t *part, uint32_t *sorted) {
    extern __shared__ uint32_t hist[];
    const uint32_t range = 1u << g.log_range;
    uint32_t s, r;
    if (g.dyn) {   // one (range, slice) task per workgroup
        if (blockIdx.x >= tasks[0]) return;
        const uint32_t task = tasks[1 + blockIdx.x];
        s = task & 0xffu;
        r = task >> 8;
    } else {
        sort2_block(blockIdx.x, g.S, g.B >> g.log_range, s, r);
    }
    uint32_t *mine = part + (size_t)s * g.B + ((size_t)r << g.log_range);
    for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) hist[t] = PASS ? mine[t] : 0u;
    __syncthreads();
    const uint32_t first = range_start[(size_t)r * n_blocks], last = range_start[(size_t)(r + 1) * n_blocks];
    const uint32_t n_slices = g.dyn ? range_slices(last - first, g.S) : g.S;   // counters of unused slices are never read
    const uint32_t slice = (last - first + n_slices - 1) / n_slices;
    const uint32_t lo = first + s * slice;
    uint32_t hi = lo + slice;
    if (hi > last) hi = last;
    for (uint32_t j0 = lo + threadIdx.x; j0 < hi; j0 += 4 * blockDim.x) {
        uint32_t kk[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = j0 + u * blockDim.x;
            // read once: streaming loads keep these lists from evicting the half-filled output lines of the scatter
            kk[u] = j < hi ? __builtin_nontemporal_load(mid_key + j) : 0xffffffffu;
            if (PASS) vv[u] = j < hi ? __builtin_nontemporal_load(mid_val + j) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (j0 + u * blockDim.x >= hi) continue;
            const uint32_t pos = atomicAdd(&hist[kk[u] & 0x7fffffffu], 1u);
            if (PASS) sorted[pos] = vv[u] | (kk[u] & 0x80000000u);
        }
    }
    if (!PASS) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) mine[t] = hist[t];
    }
}

// counts[b] = sum over slices
__global__ void __launch_bounds__(256) msm_totals_kernel(SumArgs a, uint32_t first_sum, const uint32_t *part, size_t part_stride,
                                                         uint32_t *counts) {
    const uint32_t sum = first_sum + blockIdx.y;
    const uint32_t B = a.g[sum].B;
    part += (size_t)sum * part_stride;
    counts += (size_t)sum * B;
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t S = slices_of_bucket(a, sum, b);
    uint32_t tot = 0;
    for (uint32_t s = 0; s < S; s++) tot += part[(size_t)s * B + b];
    counts[b] = tot;
}
// part[...] <- first output position of every (slice, bucket)
__global__ void __launch_bounds__(256) msm_cursors_kernel(SumArgs a, uint32_t first_sum, uint32_t *part, size_t part_stride,
                                                          const uint32_t *offsets) {
    const uint32_t sum = first_sum + blockIdx.y;
    const uint32_t B = a.g[sum].B;
    part += (size_t)sum * part_stride;
    offsets += (size_t)sum * B;
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t S = slices_of_bucket(a, sum, b);
    uint32_t run = offsets[b];
    for (uint32_t s = 0; s < S; s++) {
        size_t idx = (size_t)s * B + b;
        uint32_t t = part[idx];
        part[idx] = run;
        run += t;
    }
}

// ---- exclusive scan of the histogram (three small kernels) ------------------------------------------------
__global__ void __launch_bounds__(256) scan_block_kernel(const uint32_t *counts, uint32_t *offsets, uint32_t *block_sums,
                                                         uint32_t nb) {
    __shared__ uint32_t wave_tot[4];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * SCAN_ITEMS + tid * 4;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = (base + k < nb) ? counts[base + k] : 0u;
        sum += v[k];
    }
    uint32_t incl = sum;  // inclusive scan of `sum` across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t up = __shfl_up(incl, d);
        if ((tid & 63) >= (uint32_t)d) incl += up;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) wave_base += wave_tot[w];
    uint32_t run = wave_base + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < nb) offsets[base + k] = run;
        run += v[k];
    }
    if (tid == 255) block_sums[blockIdx.x] = run;
}

__global__ void __launch_bounds__(1024) scan_sums_kernel(uint32_t *block_sums, uint32_t n_blocks) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t start = 0; start < n_blocks; start += 1024) {
        uint32_t i = start + tid;
        uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t up = __shfl_up(incl, d);
            if ((tid & 63) >= (uint32_t)d) incl += up;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        uint32_t wave_base = 0;
        for (uint32_t w = 0; w < (tid >> 6); w++) wave_base += wave_tot[w];
        uint32_t carry = carry_s;
        if (i < n_blocks) block_sums[i] = carry + wave_base + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wave_base + incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) scan_finish_kernel(const uint32_t *counts, uint32_t *offsets,
                                                          const uint32_t *block_sums, uint32_t nb, uint32_t big, MsmTask *tasks,
                                                          uint32_t *task_count, uint32_t max_tasks, uint32_t sorted_base,
                                                          uint32_t key_base) {
    // counts / offsets: the nb buckets scanned together, which are the buckets key_base ... of the batch; their runs in
    // `sorted` start at sorted_base
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    uint32_t off = offsets[i] + block_sums[i / SCAN_ITEMS] + sorted_base;
    offsets[i] = off;
    uint32_t cnt = counts[i];
    if (cnt > big) {
        uint32_t nt = (cnt + BIG_CHUNK - 1) / BIG_CHUNK;
        uint32_t t0 = atomicAdd(task_count, nt);
        for (uint32_t k = 0; k < nt && t0 + k < max_tasks; k++) {
            MsmTask t;
            t.key = key_base + i;
            t.start = off + k * BIG_CHUNK;
            t.len = (cnt - k * BIG_CHUNK) < BIG_CHUNK ? (cnt - k * BIG_CHUNK) : BIG_CHUNK;
            t.first = k == 0 ? nt : 0;
            tasks[t0 + k] = t;
        }
    }
}

// ---- bucket schedule --------------------------------------------------------------------------------------
// Buckets are visited in order of decreasing size so that the 64 lanes of a wave own buckets of (nearly) equal
// size.  Size classes by count >> shift: class 0 = the largest ... class BIG_BUCKET = the smallest; oversized
// buckets are left out (they go through msm_big_*).  Two passes over the counts: class histogram, then placement.
static const uint32_t N_CLASSES = BIG_BUCKET + 1;

template <int PASS>
__global__ void __launch_bounds__(1024) msm_schedule_kernel(const uint32_t *counts, uint32_t nb, uint32_t shift,
                                                            uint32_t big, uint32_t *class_cursor, uint32_t *order,
                                                            uint32_t first_sum) {
    __shared__ uint32_t hist[N_CLASSES];
    __shared__ uint32_t base[N_CLASSES];
    const uint32_t sum = first_sum + blockIdx.y;   // its nb buckets, class cursors and order follow the previous sum's
    counts += (size_t)sum * nb;
    order += (size_t)sum * nb;
    class_cursor += (size_t)sum * (N_CLASSES + 1);
    for (uint32_t t = threadIdx.x; t < N_CLASSES; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cls = 0xffffffffu, rank = 0;
    if (i < nb) {
        uint32_t cnt = counts[i];
        if (cnt <= big) {
            cls = BIG_BUCKET - (cnt >> shift);
            rank = atomicAdd(&hist[cls], 1u);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < N_CLASSES; t += blockDim.x)
        if (hist[t]) base[t] = atomicAdd(&class_cursor[t], hist[t]);
    if (!PASS) return;
    __syncthreads();
    if (cls != 0xffffffffu) order[base[cls] + rank] = i;
}
// exclusive scan of the class histogram (in place), total scheduled buckets in class_cursor[N_CLASSES]
__global__ void msm_schedule_scan_kernel(uint32_t *class_cursor, uint32_t first_sum) {
    if (threadIdx.x) return;
    class_cursor += (size_t)(first_sum + blockIdx.x) * (N_CLASSES + 1);
    uint32_t run = 0;
    for (uint32_t t = 0; t < N_CLASSES; t++) {
        uint32_t v = class_cursor[t];
        class_cursor[t] = run;
        run += v;
    }
    class_cursor[N_CLASSES] = run;
}

// The three schedule launches in one, for bucket sets small enough that one workgroup per sum walks all of them
// (nb <= 2^13): class histogram in LDS, its scan, and the placement.  Leaves `order` and the sum's total in
// class_cursor[N_CLASSES] (all the accumulation reads of the cursors); needs no cleared cursors.
__global__ void __launch_bounds__(1024) msm_schedule_small_kernel(const uint32_t *counts, uint32_t nb, uint32_t shift, uint32_t big,
                                                                  uint32_t *class_cursor, uint32_t *order) {
    __shared__ uint32_t hist[N_CLASSES + 1];
    const uint32_t sum = blockIdx.x;
    counts += (size_t)sum * nb;
    order += (size_t)sum * nb;
    class_cursor += (size_t)sum * (N_CLASSES + 1);
    for (uint32_t t = threadIdx.x; t <= N_CLASSES; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) {
        const uint32_t cnt = counts[i];
        if (cnt <= big) atomicAdd(&hist[BIG_BUCKET - (cnt >> shift)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // 257 entries
        uint32_t run = 0;
        for (uint32_t t = 0; t < N_CLASSES; t++) {
            const uint32_t v = hist[t];
            hist[t] = run;
            run += v;
        }
        class_cursor[N_CLASSES] = run;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) {
        const uint32_t cnt = counts[i];
        if (cnt <= big) order[atomicAdd(&hist[BIG_BUCKET - (cnt >> shift)], 1u)] = i;
    }
}

// ---- bucket accumulation ----------------------------------------------------------------------------------
// 2^LOG_P neighbouring lanes share a bucket: lane `part` adds entries part, part + P, ... and the partial sums
// meet through lane shuffles.  With fewer buckets than the chip has lanes the launch is bound by the longest chain
// of dependent additions on one lane, which the split shortens; LOG_P = 0 is the plain one-lane-per-bucket form.
__device__ __forceinline__ G1Xyzz shfl_down_xyzz(const G1Xyzz &p, uint32_t delta) {
    G1Xyzz r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.x.v[i] = __shfl_down(p.x.v[i], delta);
        r.y.v[i] = __shfl_down(p.y.v[i], delta);
        r.zz.v[i] = __shfl_down(p.zz.v[i], delta);
        r.zzz.v[i] = __shfl_down(p.zzz.v[i], delta);
    }
    return r;
}
// Sum j = first_sum + blockIdx.y of a batch owns buckets [j B, (j + 1) B) and the class cursors
// [j (N_CLASSES + 1), ...), whose last entry is its number of scheduled buckets.
template <int LOG_P>
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const G1Table28 *points, const uint32_t *sorted,
                                                             const uint32_t *offsets, const uint32_t *counts,
                                                             const uint32_t *order, const uint32_t *class_cursor,
                                                             G1Xyzz *buckets, uint32_t B, uint32_t first_sum) {
    constexpr uint32_t P = 1u << LOG_P;
    const uint32_t j = first_sum + blockIdx.y;
    offsets += (size_t)j * B;
    counts += (size_t)j * B;
    order += (size_t)j * B;
    buckets += (size_t)j * B;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slot = gid >> LOG_P, part = gid & (P - 1);
    if (slot >= class_cursor[(size_t)j * (N_CLASSES + 1) + N_CLASSES]) return;   // the same for all P lanes of a bucket
    const uint32_t b = order[slot];
    const uint32_t cnt = counts[b], start = offsets[b];
    G1Xyzz acc = G1Xyzz::identity();
    for (uint32_t k = part; k < cnt; k += P) {
        uint32_t v = sorted[start + k];
        G1Affine28 p = gload_affine28(points + (v & 0x7fffffffu));
        acc = xyzz_add_mixed(acc, p, (v >> 31) != 0);
    }
    if (LOG_P) {
        for (uint32_t s = P >> 1; s >= 1; s >>= 1) {
            const G1Xyzz other = shfl_down_xyzz(acc, s);
            if (part < s) acc = xyzz_add(acc, other);
        }
        if (part) return;
    }
    gstore_xyzz(buckets + b, acc);
}

static const uint32_t TREE_LDS_WORDS = 128 * 56;   // 128 accumulators of 56 words
__device__ __forceinline__ void lds_store_xyzz(uint32_t *sh, uint32_t slot, const G1Xyzz &p) {
    uint32_t *d = sh + slot * 56;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        d[i] = p.x.v[i];
        d[14 + i] = p.y.v[i];
        d[28 + i] = p.zz.v[i];
        d[42 + i] = p.zzz.v[i];
    }
}
__device__ __forceinline__ G1Xyzz lds_load_xyzz(const uint32_t *sh, uint32_t slot) {
    const uint32_t *d = sh + slot * 56;
    G1Xyzz p;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        p.x.v[i] = d[i];
        p.y.v[i] = d[14 + i];
        p.zz.v[i] = d[28 + i];
        p.zzz.v[i] = d[42 + i];
    }
    return p;
}
// ---- one addition on four lanes ---------------------------------------------------------------------------------
// The sums at the end of an MSM are trees a few dozen additions deep with ever fewer additions per level: what bounds
// them is the time of ONE addition on one lane (fourteen field products one after the other), while most lanes idle.
// The fourteen products of an addition fall into four groups of at most four that do not depend on each other, so four
// neighbouring lanes (a quad) holding the same two operands take one product each, four times over, and pass the
// results round with the quad-broadcast data-parallel-primitive move: a third of the dependent arithmetic per addition.
template <int K>
__device__ __forceinline__ Fq28 quad_broadcast(const Fq28 &x) {   // lane K of every quad to its four lanes
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.v[i], K * 0x55, 0xf, 0xf, false);
    return r;
}
__device__ __forceinline__ Fq28 quad_pick(uint32_t role, const Fq28 &x0, const Fq28 &x1, const Fq28 &x2, const Fq28 &x3) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t lo = (role & 1u) ? x1.v[i] : x0.v[i];
        const uint32_t hi = (role & 1u) ? x3.v[i] : x2.v[i];
        r.v[i] = (role & 2u) ? hi : lo;
    }
    return r;
}
// a + b, both held by all four lanes of a quad (role = lane within the quad); the sum comes back in all four.  Same
// formulas and operand bounds as xyzz_add; Y3 = R (Q - X3) + (4p - S1) PPP is formed from two reduced products (below
// 4p, normalised) instead of one double product.
__device__ G1Xyzz xyzz_add_quad(const G1Xyzz &a, const G1Xyzz &b, uint32_t role) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    Fq28 prod = Fq28::mul(quad_pick(role, a.x, b.x, a.y, b.y), quad_pick(role, b.zz, a.zz, b.zzz, a.zzz));
    const Fq28 u1 = quad_broadcast<0>(prod), u2 = quad_broadcast<1>(prod);
    const Fq28 s1 = quad_broadcast<2>(prod), s2 = quad_broadcast<3>(prod);
    const Fq28 pd = Fq28::sub4(u2, u1);          // < 6p, limbs < 1.5 * 2^29
    const Fq28 rd = Fq28::sub4(s2, s1);
    prod = Fq28::mul(quad_pick(role, pd, rd, a.zz, a.zzz), quad_pick(role, pd, rd, b.zz, b.zzz));
    const Fq28 pp = quad_broadcast<0>(prod), rr = quad_broadcast<1>(prod);
    const Fq28 zz12 = quad_broadcast<2>(prod), zzz12 = quad_broadcast<3>(prod);
    prod = Fq28::mul(quad_pick(role, pd, u1, zz12, rd), quad_pick(role, pp, pp, pp, Fq28::one()));
    const Fq28 ppp = quad_broadcast<0>(prod), q = quad_broadcast<1>(prod);
    const Fq28 zz3 = quad_broadcast<2>(prod);
    if (zz3.is_zero_mod_p()) {                   // the same point, or its negative
        if (quad_broadcast<3>(prod).is_zero_mod_p()) return xyzz_double(a);
        return G1Xyzz::identity();
    }
    G1Xyzz r;
    const Fq28 nq = Fq28::sub4(Fq28::zero(), q);
    r.x = Fq28::normalize(Fq28::add(Fq28::sub4(rr, ppp), Fq28::add(nq, nq)));   // < 14p
    const Fq28 t = Fq28::sub16(q, r.x);
    const Fq28 ns1 = Fq28::normalize(Fq28::sub4(Fq28::zero(), s1));            // 4p - s1, limbs < 2^28
    prod = Fq28::mul(quad_pick(role, rd, ns1, zzz12, zzz12), quad_pick(role, t, ppp, ppp, ppp));
    r.y = Fq28::normalize(Fq28::add(quad_broadcast<0>(prod), quad_broadcast<1>(prod)));   // < 4p
    r.zz = zz3;
    r.zzz = quad_broadcast<2>(prod);
    return r;
}

// Sums the per-thread accumulators of the first 2 * first_stride threads of a 256-thread workgroup (the rest must hold
// the identity or be beyond the data); the total is returned in thread 0.  A level of 128 additions has no lanes to
// spare and runs one addition per lane; from 64 additions down the accumulators live in LDS and every addition is
// shared by a quad.
__device__ __forceinline__ G1Xyzz block_tree_sum(uint32_t *sh, G1Xyzz acc, uint32_t first_stride = 128) {
    const uint32_t tid = threadIdx.x;
    uint32_t stride = first_stride;
    if (stride == 128) {
        if (tid >= 128) lds_store_xyzz(sh, tid - 128, acc);
        __syncthreads();
        if (tid < 128) acc = xyzz_add(acc, lds_load_xyzz(sh, tid));
        __syncthreads();
        stride = 64;
    }
    if (tid < 2 * stride) lds_store_xyzz(sh, tid, acc);
    __syncthreads();
    const uint32_t quad = tid >> 2, role = tid & 3u;
    for (; stride >= 1; stride >>= 1) {
        G1Xyzz sum;
        if (quad < stride) sum = xyzz_add_quad(lds_load_xyzz(sh, quad), lds_load_xyzz(sh, quad + stride), role);
        __syncthreads();
        if (quad < stride && role == 0) lds_store_xyzz(sh, quad, sum);
        __syncthreads();
    }
    return lds_load_xyzz(sh, 0);
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msm_big_partial_kernel(const G1Table28 *points, const uint32_t *sorted,
                                                              const MsmTask *tasks, const uint32_t *task_count,
                                                              uint32_t max_tasks, G1Xyzz *partials) {
    __shared__ uint32_t sh[TREE_LDS_WORDS];
    uint32_t n_tasks = *task_count;
    if (n_tasks > max_tasks) n_tasks = max_tasks;
    for (uint32_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
        MsmTask task = tasks[t];
        G1Xyzz part = G1Xyzz::identity();
        for (uint32_t k = threadIdx.x; k < task.len; k += 256) {
            uint32_t v = sorted[task.start + k];
            G1Affine28 p = gload_affine28(points + (v & 0x7fffffffu));
            part = xyzz_add_mixed(part, p, (v >> 31) != 0);
        }
        part = block_tree_sum(sh, part);
        if (threadIdx.x == 0) gstore_xyzz(partials + t, part);
    }
}

// One workgroup per oversized bucket: strided sums of its task partials, then the LDS tree.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msm_big_combine_kernel(const MsmTask *tasks, const uint32_t *task_count,
                                                              uint32_t max_tasks, const G1Xyzz *partials, G1Xyzz *buckets) {
    __shared__ uint32_t sh[TREE_LDS_WORDS];
    uint32_t n_tasks = *task_count;
    if (n_tasks > max_tasks) n_tasks = max_tasks;
    for (uint32_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
        MsmTask task = tasks[t];
        if (!task.first) continue;  // uniform across the workgroup
        uint32_t cnt = task.first;
        if (t + cnt > n_tasks) cnt = n_tasks - t;
        G1Xyzz acc = G1Xyzz::identity();
        for (uint32_t k = threadIdx.x; k < cnt; k += 256) acc = xyzz_add(acc, gload_xyzz(partials + t + k));
        acc = block_tree_sum(sh, acc);
        if (threadIdx.x == 0) gstore_xyzz(buckets + task.key, acc);
    }
}

// ---- bucket reduction -------------------------------------------------------------------------------------
// Level 1: one thread per segment s of L = 2^log_seg buckets: T_s = sum_j bucket[sL + j] and
// W_s = sum_j j * bucket[sL + j] by running sums from the top.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msm_segment_kernel(const G1Xyzz *buckets, uint32_t log_seg, uint32_t n_segments,
                                                          G1Xyzz *t_out, G1Xyzz *w_out) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_segments) return;
    const uint32_t seg = 1u << log_seg;
    const G1Xyzz *bp = buckets + ((size_t)gid << log_seg);
    G1Xyzz run = G1Xyzz::identity(), acc = G1Xyzz::identity();
    for (int r = (int)seg - 1; r >= 1; r--) {
        run = xyzz_add(run, gload_xyzz(bp + r));
        acc = xyzz_add(acc, run);
    }
    run = xyzz_add(run, gload_xyzz(bp));
    gstore_xyzz(t_out + gid, run);
    gstore_xyzz(w_out + gid, acc);
}

// Level 2: equally sized groups of n_segments / 2 elements each, so that every workgroup carries the same work:
//   g < n_bits       the T_s with bit g of s set            (Q_g)
//   g = n_bits       the T_s with the top bit of s clear    (with Q_top: the sum of all T_s)
//   g = n_bits + 1/2 the lower / upper half of the W_s
// With a single segment (n_bits = 0) there are two groups of one element: T_0 and W_0.
// grid = (blocks per group, groups); each block leaves one partial sum.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msm_group_partial_kernel(const G1Xyzz *t_in, const G1Xyzz *w_in, uint32_t n_segments,
                                                                uint32_t n_bits, G1Xyzz *partials) {
    __shared__ uint32_t sh[TREE_LDS_WORDS];
    const uint32_t g = blockIdx.y, sum = blockIdx.z;   // n_segments = segments per sum
    t_in += (size_t)sum * n_segments;
    w_in += (size_t)sum * n_segments;
    const uint32_t count = n_bits ? (n_segments >> 1) : 1u;
    const G1Xyzz *src = t_in;
    uint32_t k = g, set = 1u, offset = 0;
    if (n_bits == 0) {
        if (g == 1) src = w_in;
        k = 0xffffffffu;
    } else if (g == n_bits) {
        k = n_bits - 1;
        set = 0;
    } else if (g > n_bits) {
        src = w_in;
        k = 0xffffffffu;
        offset = (g - n_bits - 1) * count;
    }
    G1Xyzz acc = G1Xyzz::identity();
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < count; j += gridDim.x * 256) {
        uint32_t e = j + offset;
        if (k != 0xffffffffu) e = ((j >> k) << (k + 1)) | (set << k) | (j & ((1u << k) - 1u));   // j-th index with bit k set/clear
        acc = xyzz_add(acc, gload_xyzz(src + e));
    }
    acc = block_tree_sum(sh, acc);
    if (threadIdx.x == 0) gstore_xyzz(partials + ((size_t)sum * gridDim.y + g) * gridDim.x + blockIdx.x, acc);
}

// Level 3: one workgroup per group adds that group's partial sums.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msm_group_final_kernel(const G1Xyzz *partials, uint32_t per_group, G1Proj *out) {
    __shared__ uint32_t sh[TREE_LDS_WORDS];
    const uint32_t g = blockIdx.x;
    G1Xyzz acc = G1Xyzz::identity();
    for (uint32_t k = threadIdx.x; k < per_group; k += 256) acc = xyzz_add(acc, gload_xyzz(partials + (size_t)g * per_group + k));
    uint32_t first = 128;
    while (first > 1 && first >= per_group) first >>= 1;   // threads >= per_group hold the identity
    acc = block_tree_sum(sh, acc, first);
    if (threadIdx.x == 0) gstore_proj(out + g, xyzz_to_proj(acc));
}

// ---- table of window multiples ----------------------------------------------------------------------------
__device__ __forceinline__ void gstore_affine(G1Affine *p, const Fq &x, const Fq &y) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    gstore_fq(q, x);
    gstore_fq(q + 3, y);
}
// Thread t extends point first + t: level w holds 2^(c w) * P in affine form.  The projective multiples are
// written unnormalised first; their Z coordinates and running products go to scratch so that one inversion per
// point converts all levels (Montgomery's trick along the levels).
// `upper` holds the unnormalised multiples of this chunk between the two halves: level w of thread t at (w - 1) * count + t.
__global__ void __launch_bounds__(256) msm_table_kernel(const G1Affine *points, G1Affine *upper, G1Table28 *table28, uint32_t stride,
                                                        uint32_t first, uint32_t count, uint32_t c, uint32_t K, Fq *z_scratch,
                                                        Fq *prefix_scratch) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t i = first + t;
    const G1Affine p = gload_affine(points + i);
    gstore_affine28(table28 + i, affine_to_28(p));
    if (p.is_identity()) {
        G1Affine28 inf;
        inf.x = Fq28::zero();
        inf.y = Fq28::zero();
        for (uint32_t w = 1; w < K; w++) gstore_affine28(table28 + (size_t)w * stride + i, inf);
        return;
    }
    G1Proj q;
    q.x = p.x;
    q.y = p.y;
    q.z = Fq::one();
    Fq pre = Fq::one();
    for (uint32_t w = 1; w < K; w++) {
        for (uint32_t k = 0; k < c; k++) q = g1_double(q);
        gstore_affine(upper + (size_t)(w - 1) * count + t, q.x, q.y);
        Fq z = q.z.is_zero() ? Fq::one() : q.z;   // cannot happen for points of prime order; keeps the product invertible
        pre = Fq::mul(pre, z);
        uint4 *zq = reinterpret_cast<uint4 *>(z_scratch + (size_t)(w - 1) * count + t);
        uint4 *pq = reinterpret_cast<uint4 *>(prefix_scratch + (size_t)(w - 1) * count + t);
        gstore_fq(zq, q.z);
        gstore_fq(pq, pre);
    }
    Fq inv = Fq::inv(pre);
    for (uint32_t w = K - 1; w >= 1; w--) {
        Fq z = gload_fq(reinterpret_cast<const uint4 *>(z_scratch + (size_t)(w - 1) * count + t));
        const G1Affine *slot = upper + (size_t)(w - 1) * count + t;
        if (z.is_zero()) {
            G1Affine28 inf;
            inf.x = Fq28::zero();
            inf.y = Fq28::zero();
            gstore_affine28(table28 + (size_t)w * stride + i, inf);
            continue;
        }
        Fq zinv = inv;
        if (w > 1) zinv = Fq::mul(inv, gload_fq(reinterpret_cast<const uint4 *>(prefix_scratch + (size_t)(w - 2) * count + t)));
        inv = Fq::mul(inv, z);
        G1Affine u = gload_affine(slot);
        u.x = Fq::mul(u.x, zinv);
        u.y = Fq::mul(u.y, zinv);
        gstore_affine28(table28 + (size_t)w * stride + i, affine_to_28(u));
    }
}

// (0,0)-encode points flagged as infinity
__global__ void msm_mark_infinity_kernel(G1Affine *points, const uint8_t *inf, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !inf[i]) return;
    uint4 *q = reinterpret_cast<uint4 *>(points + i);
#pragma unroll
    for (int k = 0; k < 6; k++) q[k] = make_uint4(0, 0, 0, 0);
}

// ---- SRS generation ----------------------------------------------------------------------------------------------
// out[i] = tau^i * g, affine, reference form: PublicParameters::setup (plonk/src/commitment_scheme/kzg10/srs.rs:40-71)
// computes the same points as one full scalar multiplication per power followed by batch_normalize.  Thread i
// raises tau to i (32 squarings), leaves Montgomery form and runs double-and-add on the extended Jacobian
// accumulator; the affine conversion spends one inversion per point.
__global__ void __launch_bounds__(256) msm_powers_kernel(Fr tau, G1Affine g, uint32_t n, G1Affine *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr e = Fr::one(), b = tau;
    for (uint32_t k = i; k; k >>= 1) {
        if (k & 1) e = Fr::mul(e, b);
        b = Fr::sqr(b);
    }
    e = Fr::from_mont(e);
    const G1Affine28 g28 = affine_to_28(g);
    G1Xyzz acc = G1Xyzz::identity();
    for (int bit = 254; bit >= 0; bit--) {
        if (!acc.is_identity()) acc = xyzz_double(acc);
        if ((e.v[bit >> 5] >> (bit & 31)) & 1u) acc = xyzz_add_mixed(acc, g28, false);
    }
    G1Affine r;
    if (acc.is_identity()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
    } else {
        const G1Proj p = xyzz_to_proj(acc);
        const Fq zinv = Fq::inv(p.z);
        r.x = Fq::mul(p.x, zinv);
        r.y = Fq::mul(p.y, zinv);
    }
    uint4 *q = reinterpret_cast<uint4 *>(out + i);
    gstore_fq(q, r.x);
    gstore_fq(q + 3, r.y);
}

// ---------------------------------------------------------------------------------------------------------
// MAX_BATCH - 1 host threads that sleep between batches.  run(m, f) has f(1) .. f(m - 1) done by them while the caller
// does f(0), and returns when all are done.  One caller at a time (the engine's owner serialises its calls).
struct MsmEngine::TailPool {
    std::mutex mu;
    std::condition_variable cv;
    const std::function<void(uint32_t)> *job = nullptr;
    uint64_t generation = 0;
    uint32_t count = 0;
    std::atomic<uint32_t> done{0};
    bool stop = false;
    std::thread workers[MAX_BATCH - 1];
    uint32_t n_workers = 0;

    TailPool() {
        try {
            for (; n_workers < MAX_BATCH - 1; n_workers++) workers[n_workers] = std::thread([this, id = n_workers + 1] { loop(id); });
        } catch (const std::system_error &) {
            shut_down();
            throw;
        }
    }
    ~TailPool() { shut_down(); }
    void shut_down() {
        {
            std::lock_guard<std::mutex> hold(mu);
            stop = true;
        }
        cv.notify_all();
        for (uint32_t i = 0; i < n_workers; i++) workers[i].join();
        n_workers = 0;
    }
    void loop(uint32_t id) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(uint32_t)> *f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                if (id >= count) continue;   // a batch smaller than the pool
                f = job;
            }
            (*f)(id);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void run(uint32_t m, const std::function<void(uint32_t)> &f) {
        {
            std::lock_guard<std::mutex> hold(mu);
            job = &f;
            count = m;
            done.store(0, std::memory_order_relaxed);
            generation++;
        }
        cv.notify_all();
        f(0);
        while (done.load(std::memory_order_acquire) < m - 1) std::this_thread::yield();   // tens of microseconds at most
    }
};

MsmEngine::~MsmEngine() {
    delete tail_pool_;
    if (aux_stream_) {
        (void)hipStreamSynchronize(aux_stream_);
        (void)hipStreamDestroy(aux_stream_);
    }
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    DevBuf *all[] = {&counts_, &offsets_, &cursor_, &sorted_, &buckets_, &block_sums_, &tasks_, &partials_,
                     &misc_,   &seg_,    &group_sums_, &order_, &digits_, &range_cnt_, &mid_, &slice_tasks_};
    for (DevBuf *b : all) b->release();
    // points_ and table28_ are views of table_'s buffers, which go with the last engine that refers to them
    if (host_sums_) (void)hipHostFree(host_sums_);
}

size_t MsmEngine::device_bytes() const {
    return points_.cap + counts_.cap + offsets_.cap + cursor_.cap + sorted_.cap + buckets_.cap + block_sums_.cap +
           tasks_.cap + partials_.cap + misc_.cap + seg_.cap + group_sums_.cap + order_.cap + digits_.cap + table28_.cap + range_cnt_.cap + mid_.cap + slice_tasks_.cap;
}

// Window width for a table of n points, from a time model (milliseconds) fitted to MI355X measurements
// (tools/msm_window_sweep.py).  K = floor(255 / c) + 1 signed digits always cover a 255-bit scalar.
//   accumulate: the larger of the throughput term (10 products per addition at the achieved ~61 G products/s) and
//               the depth term: the deepest bucket's additions at ~16 us each on one lane (up to four lanes share a
//               bucket when there are at most 2^15 of them) — few buckets mean long dependent chains on a mostly
//               idle chip;
//   reduce:     0.2-0.3 ms of latency plus the work of 2 additions per bucket;
//   sort:       ~80 ps per (scalar, digit) pair in one level, ~21 ps through the range partition;
//   a short top window (255 - (K-1) c bits) piles its n digits onto 2^top buckets, all levels sharing one bucket
//   set: they either deepen the deepest bucket or go through the oversized-bucket path.
uint32_t MsmEngine::choose_window(size_t n) {
    uint32_t best = 13;
    double best_ms = 1e300;
    // 20 bits at most: beyond 2^19 buckets the sort's range count and the reduction grow faster than the model says
    // (measured at 2^24 points: 39 ms with 20 bits, 47 ms with 22)
    // Up to about 2^13 points 13 bits at most: the model is fitted to single sums, and in the batches of a proof the
    // reduction over 2^15 buckets per sum costs more than it says (whole proofs: 2^12 gates 3.9 ms with 13 bits, 4.4
    // with 16; 2^13: 4.3 against 4.6; 2^14: 5.6 against 4.8).
    const uint32_t c_max = n < ((size_t)3 << 12) ? 13u : 20u;   // an SRS for 2^13 gates holds 2^13 + 7 points
    for (uint32_t c = 4; c <= c_max; c++) {
        const uint32_t K = 255 / c + 1, top_bits = 255 - (K - 1) * c;
        const double B = (double)(1u << (c - 1)), items = (double)n * K, mean = items / B;
        const double log_b = (double)(c - 1);
        // ~0.2 ms of tree depth up to 2^12 buckets, +0.04 per doubling up to 2^15; above that 8- or 16-bucket segments
        double reduce = 0.20;
        if (log_b > 12) reduce += 0.04 * ((log_b < 15 ? log_b : 15) - 12);
        if (log_b > 15) reduce += 0.09;
        if (B > 65536.0) reduce += (B - 65536.0) * 0.5e-6;
        // one-level sort (LDS counters for all buckets) up to 2^15 buckets, range partition + sort above
        const double sort = log_b <= 15 ? 0.03 + items * 80e-9 : 0.08 + items * 21e-9;
        double oversized = 0, deepest = mean;
        if (top_bits < c - 1) {
            const double extra = (double)n / (double)(1u << top_bits);
            // few buckets take the top window's digits: they go to the task path once they pass the oversize threshold
            if (top_bits <= TOP_FEW_BITS && extra > 8.0 * std::sqrt(mean) + 16.0) oversized = 0.35 + (double)n * 10 / 30e9 * 1e3;
            else deepest = mean + extra;
        }
        // up to 2^15 buckets, up to four lanes share a bucket (two combining additions at the end)
        double depth = deepest * 0.016;
        if (log_b <= 15) {
            uint32_t log_p = 0;
            while (log_p < 2 && deepest / (double)(1u << log_p) >= 6.0) log_p++;
            depth = deepest / (double)(1u << log_p) * 0.016 + log_p * 0.02;
        }
        // the first entry of a bucket is a copy, not an addition
        const double additions = items - B * (1.0 - std::exp(-mean));
        const double throughput = additions * 10 / 61e9 * 1e3;
        const double ms = (depth > throughput ? depth : throughput) + 0.05 + reduce + sort + oversized;
        if (ms < best_ms) {
            best_ms = ms;
            best = c;
        }
    }
    return best;
}

void MsmEngine::new_table() {
    n_points_ = 0;
    points_ = DevBuf();
    table28_ = DevBuf();
    table_ = std::make_shared<PointTable>();   // the previous one goes with its last engine
}

void MsmEngine::share_table(const MsmEngine &src) {
    table_ = src.table_;
    points_ = src.points_;
    table28_ = src.table28_;
    n_points_ = src.n_points_;
    c_ = src.c_;
    K_ = src.K_;
}

void MsmEngine::build_table(size_t n) {
    c_ = window_override_ ? window_override_ : choose_window(n);
    K_ = 255 / c_ + 1;
    ZK_REQUIRE((uint64_t)n * K_ < (1ull << 31), ZKGPU_ERR_INVALID_ARG, "srs_load: table of window multiples exceeds 2^31 entries");
    n_points_ = n;
    if (n == 0) return;
    table_->table28.reserve((size_t)K_ * n * sizeof(G1Table28));
    table28_ = table_->table28;
    // per chunk of points: the Z coordinates and running products of the K - 1 upper levels (48 bytes each) and the
    // unnormalised multiples themselves (96 bytes), all scratch of the build
    const uint32_t chunk = n < (1u << 18) ? (uint32_t)n : (1u << 18);
    misc_.reserve((size_t)(K_ - 1) * chunk * (2 * sizeof(Fq) + sizeof(G1Affine)));
    Fq *z_scratch = misc_.as<Fq>(), *prefix_scratch = z_scratch + (size_t)(K_ - 1) * chunk;
    G1Affine *upper = reinterpret_cast<G1Affine *>(prefix_scratch + (size_t)(K_ - 1) * chunk);
    for (size_t first = 0; first < n; first += chunk) {
        uint32_t count = (uint32_t)(n - first < chunk ? n - first : chunk);
        hipLaunchKernelGGL(msm_table_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, stream_, points_.as<G1Affine>(), upper,
                           table28_.as<G1Table28>(), (uint32_t)n, (uint32_t)first, count, c_, K_, z_scratch, prefix_scratch);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream_));
    misc_.release();   // up to 0.6 GiB of build scratch: nothing else uses it
}

void MsmEngine::load_points(const uint64_t *xy_host, const uint8_t *infinity_host, size_t n) {
    ZK_REQUIRE(xy_host != nullptr || n == 0, ZKGPU_ERR_INVALID_ARG, "srs_load: null points");
    ZK_REQUIRE(n < (1ull << 31), ZKGPU_ERR_INVALID_ARG, "srs_load: more than 2^31 - 1 points");
    new_table();
    table_->points.reserve(n * sizeof(G1Affine) + 16);
    points_ = table_->points;
    if (n) ZK_HIP(hipMemcpyAsync(points_.p, xy_host, n * sizeof(G1Affine), hipMemcpyHostToDevice, stream_));
    if (infinity_host && n) {
        bool any = false;
        for (size_t i = 0; i < n && !any; i++) any = infinity_host[i] != 0;
        if (any) {
            misc_.reserve(n);
            ZK_HIP(hipMemcpyAsync(misc_.p, infinity_host, n, hipMemcpyHostToDevice, stream_));
            hipLaunchKernelGGL(msm_mark_infinity_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream_,
                               points_.as<G1Affine>(), misc_.as<uint8_t>(), (uint32_t)n);
            ZK_HIP(hipGetLastError());
        }
    }
    ZK_HIP(hipStreamSynchronize(stream_));
    build_table(n);
}

void MsmEngine::setup_powers(const Fr &tau, const G1Affine &g, size_t n) {
    ZK_REQUIRE(n >= 1 && n < (1ull << 31), ZKGPU_ERR_INVALID_ARG, "srs_setup: bad point count");
    new_table();
    table_->points.reserve(n * sizeof(G1Affine) + 16);
    points_ = table_->points;
    hipLaunchKernelGGL(msm_powers_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream_, tau, g, (uint32_t)n,
                       points_.as<G1Affine>());
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream_));
    build_table(n);
}

void MsmEngine::export_points(uint64_t *xy_host, uint8_t *infinity_host, size_t n) {
    ZK_REQUIRE(n <= n_points_, ZKGPU_ERR_INVALID_ARG, "srs_export: more points than loaded");
    if (!n) return;
    ZK_HIP(hipMemcpyAsync(xy_host, points_.p, n * sizeof(G1Affine), hipMemcpyDeviceToHost, stream_));
    ZK_HIP(hipStreamSynchronize(stream_));
    for (size_t i = 0; i < n; i++) {
        bool inf = true;
        for (int k = 0; k < 12; k++) inf = inf && xy_host[12 * i + k] == 0;
        if (infinity_host) infinity_host[i] = inf ? 1 : 0;
        if (inf) {   // the reference's affine identity is (0, 1, infinity)
            Fq one = Fq::one();
            memcpy(xy_host + 12 * i + 6, one.v, 48);
        }
    }
}

void MsmEngine::load_points_raw(const uint8_t *bytes, size_t len) {
    ZK_REQUIRE(bytes != nullptr && len >= 8, ZKGPU_ERR_INVALID_ARG, "srs_load_raw: need at least the 8-byte count");
    uint64_t count = 0;
    for (int k = 0; k < 8; k++) count |= (uint64_t)bytes[k] << (8 * k);
    const uint64_t present = (len - 8) / 97;
    if (count > present) count = present;   // chunks_exact(..).zip(0..len) stops at the shorter of the two
    std::vector<uint64_t> xy((size_t)count * 12);
    std::vector<uint8_t> inf((size_t)count);
    for (uint64_t i = 0; i < count; i++) {
        const uint8_t *p = bytes + 8 + 97 * i;
        for (int w = 0; w < 12; w++) {
            uint64_t v = 0;
            for (int k = 0; k < 8; k++) v |= (uint64_t)p[8 * w + k] << (8 * k);
            xy[12 * i + w] = v;
        }
        inf[i] = p[96] ? 1 : 0;
    }
    load_points(xy.data(), inf.data(), (size_t)count);
}

void MsmEngine::export_points_raw(uint8_t *out) {
    const size_t n = n_points_;
    std::vector<uint64_t> xy(n * 12);
    std::vector<uint8_t> inf(n);
    export_points(xy.data(), inf.data(), n);
    for (int k = 0; k < 8; k++) out[k] = (uint8_t)((uint64_t)n >> (8 * k));
    for (size_t i = 0; i < n; i++) {
        uint8_t *p = out + 8 + 97 * i;
        for (int w = 0; w < 12; w++)
            for (int k = 0; k < 8; k++) p[8 * w + k] = (uint8_t)(xy[12 * i + w] >> (8 * k));
        p[96] = inf[i];
    }
}

void MsmEngine::load_points_device(const void *xy_dev, size_t n) {
    ZK_REQUIRE(xy_dev != nullptr || n == 0, ZKGPU_ERR_INVALID_ARG, "srs_load: null points");
    ZK_REQUIRE(n < (1ull << 31), ZKGPU_ERR_INVALID_ARG, "srs_load: more than 2^31 - 1 points");
    new_table();
    table_->points.reserve(n * sizeof(G1Affine) + 16);
    points_ = table_->points;
    if (n) ZK_HIP(hipMemcpyAsync(points_.p, xy_dev, n * sizeof(G1Affine), hipMemcpyDeviceToDevice, stream_));
    ZK_HIP(hipStreamSynchronize(stream_));
    build_table(n);
}

void MsmEngine::msm_batch_device(uint32_t m, const Fr *const *scalars_dev, const size_t *lens, size_t base,
                                 uint32_t &n_groups, uint32_t &log_seg_out) {
    ZK_REQUIRE(m >= 1 && m <= MAX_BATCH, ZKGPU_ERR_INVALID_ARG, "msm: batch size out of range");
    const uint32_t c = c_, K = K_;
    const uint32_t log_B = c - 1, B = 1u << log_B;
    const uint32_t nb = m * B;   // bucket j * B + b belongs to sum j
    uint64_t total_items = 0, max_items = 0;
    for (uint32_t j = 0; j < m; j++) {
        ZK_REQUIRE(lens[j] <= n_points_ && base <= n_points_ - lens[j], ZKGPU_ERR_INVALID_ARG,
                   "msm: more scalars than loaded points");
        ZK_REQUIRE(lens[j] >= 1 && lens[j] < (1ull << 31), ZKGPU_ERR_INVALID_ARG, "msm: bad length");
        total_items += (uint64_t)lens[j] * K;
        if ((uint64_t)lens[j] * K > max_items) max_items = (uint64_t)lens[j] * K;
    }
    ZK_REQUIRE(total_items < (1ull << 32) - 2048, ZKGPU_ERR_INVALID_ARG, "msm: too many (point, window) pairs for 32-bit offsets");
    // A bucket is summed by one thread unless it is oversized: beyond expect + 8 sqrt(expect) + 16 entries, where
    // expect is the size of the fullest ordinary buckets (mean + the share of a short top window); random digits
    // stay below that.  The chain of a longer bucket on one thread would be the tail of the accumulate launch.
    const uint32_t top_bits = 255 - (K - 1) * c;
    uint64_t expect = max_items >> log_B;
    if (top_bits < log_B) {
        // when thousands of buckets take a share of the top window it is part of their ordinary size; when only a few
        // do (a top window of at most TOP_FEW_BITS bits) their long chains go to workgroups, as choose_window assumes
        const uint64_t extra = (max_items / K) >> top_bits;
        if (top_bits > TOP_FEW_BITS) expect += extra;
    }
    uint64_t big64 = expect + 16 + (uint64_t)(8.0 * std::sqrt((double)expect));
    uint32_t class_shift = 0;   // size classes are count >> shift, shift the smallest with big <= 256 << shift
    while (((uint64_t)BIG_BUCKET << class_shift) < big64 && class_shift < 12) class_shift++;
    if (big64 > ((uint64_t)BIG_BUCKET << class_shift)) big64 = (uint64_t)BIG_BUCKET << class_shift;
    const uint32_t big = (uint32_t)big64;
    // every oversized bucket gives ceil(count / BIG_CHUNK) tasks: at most one per `big` entries plus one per chunk
    const uint32_t max_tasks = (uint32_t)(total_items / big + total_items / BIG_CHUNK + m);
    const uint32_t n_scan_blocks = ceil_div(nb, SCAN_ITEMS);
    // segments of the bucket reduction: few buckets leave the chip idle, so short segments (shallow chains) win there
    const uint32_t seg_bits = seg_bits_ ? seg_bits_ : (log_B <= 12 ? 1u : log_B <= 15 ? 2u : (m >= 3 ? 4u : 3u));
    const uint32_t log_seg = log_B < seg_bits ? log_B : seg_bits;
    const uint32_t segs_per_sum = 1u << (log_B - log_seg);
    const uint32_t n_bits = log_B - log_seg;
    const uint32_t groups = n_bits ? n_bits + 3 : 2;
    // two resident workgroups per CU run these latency-bound sums in a single round: ~512 workgroups in total
    uint32_t blocks_per_group = 512 / (groups * m);
    const uint32_t max_useful = ceil_div(n_bits ? segs_per_sum / 2 : 1, 256);
    if (blocks_per_group > max_useful) blocks_per_group = max_useful;
    if (blocks_per_group < 1) blocks_per_group = 1;

    // sort geometry: up to 2^15 buckets sort in one level (128 KiB of LDS counters per workgroup); above that, ranges
    // of 2^13 buckets, whose output regions (a few MB at 2^20..2^22 points) sit in one XCD's L2 while they are written
    // (measured best among 2^9..2^14 at 2^20 and at 2^24 points); slices so that ~1024 workgroups exist
    // Between 2^16 and 2^20 buckets the two placing passes stage their stores through LDS (msm_range_place_scalars_kernel,
    // msm_sort3_kernel): 1024 ranges of at most 2^10 buckets.  Beyond that the ranges would outgrow the staged
    // kernels' tables: ranges of 2^13 buckets and the lane-scattered kernels.
    const bool staged = log_B > 15 && log_B <= 20;
    const uint32_t log_range = log_B <= 15 ? log_B : staged ? (log_B > 19 ? log_B - 10 : 9) : 13;
    // the staged partition takes the scalars themselves, 1024 per workgroup with all their windows (K <= 16 there: the
    // window is at least 17 bits); the other one reads the digit array in blocks of 4096 items
    const uint32_t RANGE_ITEMS = staged ? STAGE_THREADS * K : RANGE_PER_THREAD * 256;
    const uint32_t n_ranges = 1u << (log_B - log_range);
    uint32_t S_max = 1024 / n_ranges;
    if (S_max < 1) S_max = 1;
    if (staged) S_max = 16;   // the most a range may take (SortGeom::dyn); an evenly filled range of 2^20 points takes one
    const size_t lds_bytes = (size_t)4 << log_range;
    if (!lds_attr_set_) {
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_sort_kernel<0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_sort_kernel<1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_sort2_kernel<0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_sort2_kernel<1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_range_place_scalars_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, stage_lds_bytes(16384)));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_sort3_kernel<16384>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, stage_lds_bytes(16384)));
        lds_attr_set_ = true;
    }

    size_t range_cnt_stride = 0;
    if (n_ranges > 1) {
        const size_t n_cnt_max = (size_t)n_ranges * ceil_div(max_items, RANGE_ITEMS) + 1;
        range_cnt_stride = n_cnt_max + ceil_div(n_cnt_max, SCAN_ITEMS) + 8;
        range_cnt_.reserve((size_t)m * range_cnt_stride * 4);
        mid_.reserve((size_t)m * 2 * max_items * 4);
    }
    const size_t slice_tasks_stride = staged ? (size_t)n_ranges + (size_t)(max_items >> 15) + 2 : 0;
    if (staged) slice_tasks_.reserve((size_t)m * slice_tasks_stride * 4);
    counts_.reserve((size_t)nb * 4);
    offsets_.reserve((size_t)nb * 4);
    cursor_.reserve((size_t)nb * S_max * 4);            // per-(sum, slice, bucket) counts, then cursors
    order_.reserve((size_t)nb * 4 + (size_t)m * (N_CLASSES + 1) * 4);
    if (!staged) digits_.reserve(total_items * 4);
    block_sums_.reserve(2 * ((size_t)n_scan_blocks * 4 + 4));   // two scans at a time when the sorts of a batch overlap
    sorted_.reserve(total_items * 4);
    buckets_.reserve((size_t)nb * sizeof(G1Xyzz));
    tasks_.reserve((size_t)max_tasks * sizeof(MsmTask) + 16);
    const size_t group_partials = (size_t)m * groups * blocks_per_group;
    partials_.reserve((max_tasks > group_partials ? max_tasks : group_partials) * sizeof(G1Xyzz));
    seg_.reserve((size_t)2 * m * segs_per_sum * sizeof(G1Xyzz));
    group_sums_.reserve((size_t)m * groups * sizeof(G1Proj));
    uint32_t *task_count = reinterpret_cast<uint32_t *>(tasks_.as<char>() + (size_t)max_tasks * sizeof(MsmTask));
    uint32_t *order = order_.as<uint32_t>();
    uint32_t *class_cursor = order + nb;
    const uint32_t n_segments = m * segs_per_sum;
    G1Xyzz *t_seg = seg_.as<G1Xyzz>(), *w_seg = t_seg + n_segments;

    const G1Table28 *table = table28_.as<G1Table28>();
    ZK_HIP(hipMemsetAsync(task_count, 0, 4, stream_));
    // up to 2^13 buckets one workgroup per sum schedules faster than three launches (11 against 19 us; at 2^15 buckets
    // the atomics of a single workgroup on the few busy classes make it slower: 28 us)
    const bool small_schedule = log_B <= 13;
    if (!small_schedule) ZK_HIP(hipMemsetAsync(class_cursor, 0, (size_t)m * (N_CLASSES + 1) * 4, stream_));

    auto mark = [&](int k) {
        if (profile_) ZK_HIP(hipEventRecord(profile_[k], stream_));
    };
    mark(0);
    // per sum: digits and the count pass of its own items into its own B counters
    SumArgs sums;
    SortGeom *geom = sums.g;
    uint32_t **dig = sums.dig;
    uint64_t item_base = 0;
    size_t max_len = 0;
    for (uint32_t j = 0; j < MAX_BATCH; j++) {
        const uint32_t k = j < m ? j : 0;   // unused entries repeat sum 0
        const uint64_t n_items = (uint64_t)lens[k] * K;
        uint32_t S = S_max;
        if (!staged) {
            // keep slices long enough to amortise the histogram flush and the walks over the slices' counters: 2^14 items,
            // 2^15 from 2^20 items up (2^16- and 2^17-gate proofs 1.2 % and 1.6 % faster than with 2^14)
            const uint64_t slice_min = n_items >= (1u << 20) ? 32768 : 16384;
            while (S > 1 && n_items / S < slice_min) S >>= 1;
        }
        geom[j].n_items = (uint32_t)n_items;
        geom[j].n = (uint32_t)lens[k];
        geom[j].stride = (uint32_t)n_points_;
        geom[j].base = (uint32_t)base;
        geom[j].B = B;
        geom[j].log_range = log_range;
        geom[j].S = S;
        geom[j].slice = (uint32_t)((n_items + S - 1) / S);
        geom[j].dyn = staged ? 1u : 0u;
        sums.range_start[j] = nullptr;
        sums.n_blocks[j] = 0;
        sums.scalars[j] = scalars_dev[k];
        if (j < m) {
            dig[j] = digits_.as<uint32_t>() + item_base;
            item_base += n_items;
            if (lens[j] > max_len) max_len = lens[j];
        } else {
            dig[j] = dig[0];
        }
    }
    const size_t part_stride = (size_t)B * S_max;
    uint32_t *const part_all = cursor_.as<uint32_t>();
    // The sort of the sums [first, first + count) of the batch, enqueued on `st`: partition / count pass, the buckets'
    // offsets (one scan over the count * B buckets, their runs in `sorted` starting at sorted_base), the placing pass and
    // the schedule.  The small kernels (digits, totals, cursors, schedule, and the one-level sort) cover those sums in one
    // launch each; the range partition and its sort run per sum.  `bs`: n_scan_blocks + 1 words for the scan.
    for (uint32_t j = 0; j < m; j++) {
        if (n_ranges == 1) break;
        const uint32_t n_blocks = staged ? ceil_div(geom[j].n, STAGE_THREADS) : ceil_div(geom[j].n_items, RANGE_ITEMS);
        sums.range_start[j] = range_cnt_.as<uint32_t>() + (size_t)j * range_cnt_stride;
        sums.n_blocks[j] = n_blocks;
    }
    auto sort_sums = [&](uint32_t first, uint32_t count, hipStream_t st, uint32_t *bs, uint32_t sorted_base, bool marks) {
        auto mark_if = [&](int k) {
            if (marks) mark(k);
        };
        uint32_t s_used = 1;
        for (uint32_t j = first; j < first + count; j++)
            if (geom[j].S > s_used) s_used = geom[j].S;
        if (!staged) hipLaunchKernelGGL(msm_digits_kernel, dim3(ceil_div(max_len, 256), count), dim3(256), 0, st, sums, first, c, K);
        mark_if(1);
        if (n_ranges == 1) {
            mark_if(2);
            hipLaunchKernelGGL(msm_sort_kernel<0>, dim3(s_used, 1, count), dim3(1024), lds_bytes, st, sums, first, part_all, part_stride,
                               sorted_.as<uint32_t>());
        } else {
            for (uint32_t j = first; j < first + count; j++) {
                uint32_t *part = part_all + (size_t)j * part_stride;
                const uint64_t n_items = geom[j].n_items;
                // range partition: count, scan (range-major, one extra entry = the total), place
                const uint32_t n_blocks = sums.n_blocks[j];
                const uint32_t n_cnt = n_ranges * n_blocks + 1, n_cnt_blocks = ceil_div(n_cnt, SCAN_ITEMS);
                uint32_t *blockcnt = range_cnt_.as<uint32_t>() + (size_t)j * range_cnt_stride;
                uint32_t *range_sums = blockcnt + n_cnt;
                uint32_t *mkey = mid_.as<uint32_t>() + (size_t)j * 2 * max_items, *mval = mkey + max_items;
                ZK_HIP(hipMemsetAsync(blockcnt + n_cnt - 1, 0, 4, st));
                if (staged)
                    hipLaunchKernelGGL(msm_range_count_scalars_kernel, dim3(n_blocks), dim3(STAGE_THREADS), 0, st, sums.scalars[j],
                                       geom[j], c, K, n_ranges, n_blocks, blockcnt);
                else
                    hipLaunchKernelGGL((msm_range_kernel<0, 256>), dim3(n_blocks), dim3(256), 0, st, dig[j], geom[j], n_ranges, n_blocks,
                                       blockcnt, mkey, mval);
                hipLaunchKernelGGL(scan_block_kernel, dim3(n_cnt_blocks), dim3(256), 0, st, blockcnt, blockcnt, range_sums, n_cnt);
                hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, range_sums, n_cnt_blocks);
                hipLaunchKernelGGL(scan_add_kernel, dim3(ceil_div(n_cnt, 256)), dim3(256), 0, st, blockcnt, range_sums, n_cnt);
                if (staged)
                    hipLaunchKernelGGL(msm_range_place_scalars_kernel, dim3(n_blocks), dim3(STAGE_THREADS), stage_lds_bytes(16384), st,
                                       sums.scalars[j], geom[j], c, K, n_ranges, n_blocks, blockcnt, mkey, mval);
                else
                    hipLaunchKernelGGL((msm_range_kernel<1, 256>), dim3(n_blocks), dim3(256), 0, st, dig[j], geom[j], n_ranges, n_blocks,
                                       blockcnt, mkey, mval);
                uint32_t *tasks = nullptr, sort_grid = geom[j].S * n_ranges;
                if (staged) {
                    sort_grid = n_ranges + (uint32_t)(n_items >> 15) + 1;
                    tasks = slice_tasks_.as<uint32_t>() + (size_t)j * slice_tasks_stride;
                    hipLaunchKernelGGL(msm_slice_tasks_kernel, dim3(1), dim3(STAGE_THREADS), 0, st, blockcnt, n_blocks, n_ranges,
                                       geom[j].S, tasks);
                }
                if (m == 1) mark_if(2);   // with several sums the partitions and count passes alternate: all of it counts as partition
                // the placing pass advanced nothing in global memory: blockcnt still holds the start positions
                hipLaunchKernelGGL(msm_sort2_kernel<0>, dim3(sort_grid), dim3(1024), lds_bytes, st, mkey, mval, blockcnt, n_blocks,
                                   geom[j], tasks, part, sorted_.as<uint32_t>());
            }
            if (m > 1) mark_if(2);
        }
        mark_if(3);
        const uint32_t nb_sub = count * B, n_sub_blocks = ceil_div(nb_sub, SCAN_ITEMS);
        uint32_t *counts = counts_.as<uint32_t>() + (size_t)first * B, *offsets = offsets_.as<uint32_t>() + (size_t)first * B;
        hipLaunchKernelGGL(msm_totals_kernel, dim3(ceil_div(B, 256), count), dim3(256), 0, st, sums, first, part_all, part_stride,
                           counts_.as<uint32_t>());
        // one scan over these sums' buckets: every bucket's run in `sorted`
        hipLaunchKernelGGL(scan_block_kernel, dim3(n_sub_blocks), dim3(256), 0, st, counts, offsets, bs, nb_sub);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, bs, n_sub_blocks);
        hipLaunchKernelGGL(scan_finish_kernel, dim3(ceil_div(nb_sub, 256)), dim3(256), 0, st, counts, offsets, bs, nb_sub, big,
                           tasks_.as<MsmTask>(), task_count, max_tasks, sorted_base, first * B);
        hipLaunchKernelGGL(msm_cursors_kernel, dim3(ceil_div(B, 256), count), dim3(256), 0, st, sums, first, part_all, part_stride,
                           offsets_.as<uint32_t>());
        mark_if(4);
        if (n_ranges == 1) {
            hipLaunchKernelGGL(msm_sort_kernel<1>, dim3(s_used, 1, count), dim3(1024), lds_bytes, st, sums, first, part_all, part_stride,
                               sorted_.as<uint32_t>());
        } else {
            for (uint32_t j = first; j < first + count; j++) {
                uint32_t *part = part_all + (size_t)j * part_stride;
                const uint32_t n_blocks = sums.n_blocks[j];
                const uint32_t *blockcnt = range_cnt_.as<uint32_t>() + (size_t)j * range_cnt_stride;
                const uint32_t *mkey = mid_.as<uint32_t>() + (size_t)j * 2 * max_items, *mval = mkey + max_items;
                if (staged)
                    hipLaunchKernelGGL(msm_sort3_kernel<16384>, dim3(n_ranges + (geom[j].n_items >> 15) + 1), dim3(STAGE_THREADS),
                                       stage_lds_bytes(16384), st, mkey, mval, blockcnt, n_blocks, geom[j],
                                       slice_tasks_.as<uint32_t>() + (size_t)j * slice_tasks_stride, part, sorted_.as<uint32_t>());
                else
                    hipLaunchKernelGGL(msm_sort2_kernel<1>, dim3(geom[j].S * n_ranges), dim3(1024), lds_bytes, st, mkey, mval, blockcnt,
                                       n_blocks, geom[j], nullptr, part, sorted_.as<uint32_t>());
            }
        }
        mark_if(5);
        // schedule: class histogram, scan, placement
        if (small_schedule) {
            hipLaunchKernelGGL(msm_schedule_small_kernel, dim3(count), dim3(1024), 0, st, counts, B, class_shift, big,
                               class_cursor + (size_t)first * (N_CLASSES + 1), order + (size_t)first * B);
        } else {
            hipLaunchKernelGGL(msm_schedule_kernel<0>, dim3(ceil_div(B, 1024), count), dim3(1024), 0, st, counts_.as<uint32_t>(), B,
                               class_shift, big, class_cursor, order, first);
            hipLaunchKernelGGL(msm_schedule_scan_kernel, dim3(count), dim3(64), 0, st, class_cursor, first);
            hipLaunchKernelGGL(msm_schedule_kernel<1>, dim3(ceil_div(B, 1024), count), dim3(1024), 0, st, counts_.as<uint32_t>(), B,
                               class_shift, big, class_cursor, order, first);
        }
        mark_if(6);
    };
    auto accumulate_sums = [&](uint32_t first, uint32_t count, uint32_t log_p, bool fused) {
        const dim3 grid(ceil_div((uint64_t)B << log_p, 256), fused ? count : 1);
        for (uint32_t j = first; j < (fused ? first + 1 : first + count); j++) {
            auto launch = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream_, table, sorted_.as<uint32_t>(), offsets_.as<uint32_t>(),
                                   counts_.as<uint32_t>(), order, class_cursor, buckets_.as<G1Xyzz>(), B, j);
            };
            if (log_p == 0) launch(msm_accumulate_kernel<0>);
            else if (log_p == 1) launch(msm_accumulate_kernel<1>);
            else if (log_p == 2) launch(msm_accumulate_kernel<2>);
            else if (log_p == 3) launch(msm_accumulate_kernel<3>);
            else launch(msm_accumulate_kernel<4>);
        }
    };
    // Up to 2^15 buckets per sum the launch is bound by the longest chain of additions on one lane, not by throughput:
    // the sums of a batch then share one launch, and two or more lanes share each bucket while the chains are worth
    // cutting (measured on 2^15-gate proofs: 8.8 ms with one lane per bucket and one launch per sum, 6.1 ms this way).
    // Larger bucket sets run one launch per sum (a single launch over all m * B buckets runs measurably slower there)
    // with one lane per bucket.
    const bool fused = log_B <= 15 && m > 1;
    uint32_t log_p = 0;
    uint64_t chain = max_items >> log_B;   // the fullest ordinary buckets: the mean plus their share of a short top window
    if (top_bits < log_B) chain += (max_items / K) >> top_bits;
    // four lanes at most unless the buckets are so few that even eight or sixteen lanes each leave the chip unfilled
    while (log_B <= 15 && (chain >> log_p) >= 6 && (log_p < 2 || (log_p < 4 && ((uint64_t)m * B << log_p) < 65536))) log_p++;
    // ... and no more lanes in total than fill the chip 4/3 times over (4096 waves): the joining additions at the end
    // (a full addition per halving, executed by the whole wave) are then dearer than the chains they shorten (2^15-gate
    // proof, four sums per launch: 3.88 ms with four lanes per bucket, 3.77 with two, 4.12 with one)
    while (log_p > 0 && ((uint64_t)m * B << log_p) > 262144) log_p--;
    // With several sums over 2^16 buckets or more, the accumulation of the first sum (bound by the integer pipeline) runs
    // beside the sort of the others (bound by memory and LDS latency) on a second stream; only the first sum's sort is
    // then exposed.  Both streams are joined again before the remaining accumulations.  Phase timing (profile_) keeps
    // everything on one stream.
    static const bool overlap_off = getenv("ZKGPU_MSM_NO_OVERLAP") != nullptr;   // for A/B measurements
    const bool overlap = staged && m >= 2 && !fused && !profile_ && !overlap_off;
    if (overlap) {
        if (!aux_stream_) {
            ZK_HIP(hipStreamCreateWithFlags(&aux_stream_, hipStreamNonBlocking));   // (its priority made no difference)
            ZK_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
            ZK_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
        }
        uint32_t *bs = block_sums_.as<uint32_t>();
        ZK_HIP(hipEventRecord(ev_fork_, stream_));            // behind the clears above and whatever used the buffers before
        ZK_HIP(hipStreamWaitEvent(aux_stream_, ev_fork_, 0));
        sort_sums(1, m - 1, aux_stream_, bs + n_scan_blocks + 1, geom[0].n_items, false);
        ZK_HIP(hipEventRecord(ev_join_, aux_stream_));
        sort_sums(0, 1, stream_, bs, 0, false);
        accumulate_sums(0, 1, log_p, false);
        ZK_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));
        accumulate_sums(1, m - 1, log_p, false);
    } else {
        sort_sums(0, m, stream_, block_sums_.as<uint32_t>(), 0, true);
        accumulate_sums(0, m, log_p, fused);
    }
    mark(7);
    const uint32_t big_grid = max_tasks < 2048 ? max_tasks : 2048;
    hipLaunchKernelGGL(msm_big_partial_kernel, dim3(big_grid), dim3(256), 0, stream_, table, sorted_.as<uint32_t>(),
                       tasks_.as<MsmTask>(), task_count, max_tasks, partials_.as<G1Xyzz>());
    hipLaunchKernelGGL(msm_big_combine_kernel, dim3(big_grid), dim3(256), 0, stream_, tasks_.as<MsmTask>(), task_count,
                       max_tasks, partials_.as<G1Xyzz>(), buckets_.as<G1Xyzz>());
    mark(8);
    // reduction of all sums together (partials_ is free again: the oversized-bucket kernels above have consumed it in
    // stream order); the segments of sum j are [j * segs_per_sum, (j + 1) * segs_per_sum)
    hipLaunchKernelGGL(msm_segment_kernel, dim3(ceil_div(n_segments, 256)), dim3(256), 0, stream_, buckets_.as<G1Xyzz>(), log_seg,
                       n_segments, t_seg, w_seg);
    hipLaunchKernelGGL(msm_group_partial_kernel, dim3(blocks_per_group, groups, m), dim3(256), 0, stream_, t_seg, w_seg,
                       segs_per_sum, n_bits, partials_.as<G1Xyzz>());
    hipLaunchKernelGGL(msm_group_final_kernel, dim3(groups * m), dim3(256), 0, stream_, partials_.as<G1Xyzz>(), blocks_per_group,
                       group_sums_.as<G1Proj>());
    mark(9);
    ZK_HIP(hipGetLastError());
    n_groups = groups;
    log_seg_out = log_seg;
}

void MsmEngine::msm_device(const Fr *scalars_dev, size_t n, size_t base, uint32_t &n_groups, uint32_t &log_seg_out) {
    msm_batch_device(1, &scalars_dev, &n, base, n_groups, log_seg_out);
}

void MsmEngine::ensure_pinned() {
    if (!host_sums_) ZK_HIP(hipHostMalloc(&host_sums_, 64 + (size_t)MAX_BATCH * MAX_GROUPS * sizeof(G1Proj), hipHostMallocDefault));
}
void *MsmEngine::pinned_scratch() {
    ensure_pinned();
    return host_sums_;
}
void MsmEngine::enqueue_result_copy(uint32_t m, uint32_t groups) {
    ZK_REQUIRE(m <= MAX_BATCH && groups <= MAX_GROUPS, ZKGPU_ERR_INTERNAL, "msm: group sums beyond the pinned buffer");
    ensure_pinned();
    ZK_HIP(hipMemcpyAsync(static_cast<char *>(host_sums_) + 64, group_sums_.p, (size_t)m * groups * sizeof(G1Proj),
                          hipMemcpyDeviceToHost, stream_));
}
void MsmEngine::finish_host(uint32_t m, uint32_t groups, uint32_t log_seg, G1Proj *out) {
    const G1Proj *all = reinterpret_cast<const G1Proj *>(static_cast<const char *>(host_sums_) + 64);
    auto one = [&](uint32_t j) {
        const G1Proj *gs = all + (size_t)j * groups;
        // sum_b (b+1) bucket_b = sum_s W_s + L * sum_k 2^k Q_k + sum_s T_s (group layout: msm_group_partial_kernel)
        if (groups == 2) {
            out[j] = g1_add(gs[0], gs[1]);
            return;
        }
        const uint32_t n_bits = groups - 3;
        G1Proj total = G1Proj::identity();
        for (int k = (int)n_bits - 1; k >= 0; k--) {
            total = g1_double(total);
            total = g1_add(total, gs[k]);
        }
        for (uint32_t k = 0; k < log_seg; k++) total = g1_double(total);
        total = g1_add(total, g1_add(gs[n_bits - 1], gs[n_bits]));
        total = g1_add(total, g1_add(gs[n_bits + 1], gs[n_bits + 2]));
        out[j] = total;
    };
    // A sum's chain of some forty dependent curve operations is twenty microseconds of host work while the GPU waits:
    // the sums of a batch are added up side by side by threads that sleep between batches (starting a thread per sum
    // would cost as much as the chain itself).
    if (m > 1 && groups > 2 && !tail_pool_ && !tail_pool_failed_) {
        try {
            tail_pool_ = new TailPool();
        } catch (const std::exception &) {   // no threads to be had: the sums are added one after the other
            tail_pool_failed_ = true;
        }
    }
    if (m > 1 && groups > 2 && tail_pool_) {
        const std::function<void(uint32_t)> f = one;
        tail_pool_->run(m, f);
    } else {
        for (uint32_t j = 0; j < m; j++) one(j);
    }
}
void MsmEngine::finish_batch(uint32_t m, uint32_t groups, uint32_t log_seg, G1Proj *out) {
    enqueue_result_copy(m, groups);
    ZK_HIP(hipStreamSynchronize(stream_));
    finish_host(m, groups, log_seg, out);
}

G1Proj MsmEngine::finish(uint32_t groups, uint32_t log_seg) {
    G1Proj r;
    finish_batch(1, groups, log_seg, &r);
    return r;
}

G1Proj MsmEngine::msm(const Fr *scalars_dev, size_t n, size_t base) {
    uint32_t groups, log_seg;
    msm_device(scalars_dev, n, base, groups, log_seg);
    return finish(groups, log_seg);
}

void MsmEngine::msm_batch(uint32_t m, const Fr *const *scalars_dev, const size_t *lens, size_t base, G1Proj *out) {
    uint32_t groups, log_seg;
    msm_batch_device(m, scalars_dev, lens, base, groups, log_seg);
    finish_batch(m, groups, log_seg, out);
}

}  // namespace zk
