// Fr NTT for gfx950.
//
// Algorithm: one radix-2 decimation-in-frequency network over the whole 2^k-point vector (natural order in,
// bit-reversed order out), cut into passes by groups of index bits.  A pass keeps a tile of 2^log_r rows x
// 2^log_t columns in LDS, runs its log_r butterfly stages there, and writes the tile back; the last pass
// writes each element to its bit-reversed position, so the result is in natural order like the reference's
// serial_fft (plonk/src/fft/domain.rs:300-340).  Every butterfly takes its twiddle from a per-stage table
// (entry h-1+i of the table is w_{2h}^i for the stage with half-size h), so a transform costs exactly one
// field multiplication per butterfly and none for index bookkeeping:
//   * ifft       = the same network with w^-1 tables, outputs multiplied by 2^-k (domain.rs:150-166) with a
//                  k-bit Montgomery-style reduction instead of a full multiplication;
//   * coset_ifft = the network with tables g^(-n/2h) * w_{2h}^-i, which is the transpose of the
//                  decimation-in-time evaluation network at the points g^-1 w^-i and therefore yields
//                  g^-j * (iDFT)_j directly (domain.rs:187-196);
//   * coset_fft  = inputs j < len_in multiplied by g^j while they are loaded (domain.rs:176-185: the powers
//                  are distributed over the un-padded coefficients), then the forward network.
// HBM traffic per pass is one read and one write of the vector plus the pass's share of the twiddle table.
// Inside a pass the tile is held as 9 limbs of 29 bits with lazy reduction (see "tile arithmetic" below), two
// butterfly stages run per LDS round trip (a thread carries four rows in registers), and the tables hold the
// twiddles in the same limb form, ready to multiply by; the kernel is bound by integer VALU issue, not by memory.
#include "ntt.hpp"
#include <mutex>
#include "fr9.hpp"

namespace zk {

static Fr fr_from_u64x4(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    Fr r;
    const uint64_t w[4] = {a, b, c, d};
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)w[i];
        r.v[2 * i + 1] = (uint32_t)(w[i] >> 32);
    }
    return r;
}
// coset-bls12_381/src/scalar.rs:191-196 and :101-106 (Montgomery form)
static Fr root_of_unity_2_32() {
    return fr_from_u64x4(0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL, 0x5bf3adda19e9b27bULL);
}
static Fr coset_generator() {
    return fr_from_u64x4(0x0000000efffffff1ULL, 0x17e363d300189c0fULL, 0xff9c57876f8457b0ULL, 0x351332208fc5a8c4ULL);
}
static const uint32_t TWO_ADACITY = 32;  // scalar.rs:187

Domain Domain::make(uint64_t num_coeffs) {
    uint64_t s = 1;
    uint32_t lg = 0;
    while (s < num_coeffs) {
        s <<= 1;
        lg++;
    }
    ZK_REQUIRE(lg < TWO_ADACITY, ZKGPU_ERR_DOMAIN_SIZE, "InvalidEvalDomainSize: log2(size) must be below 32");
    // three field inversions stand behind a domain; a proof asks for the same two or three domains several times
    static Domain cache[TWO_ADACITY];
    static std::once_flag built[TWO_ADACITY];
    std::call_once(built[lg], [&] { cache[lg] = compute(lg); });
    return cache[lg];
}

Domain Domain::compute(uint32_t lg) {
    Domain d;
    const uint64_t s = 1ull << lg;
    d.size = s;
    d.log_size = lg;
    d.group_gen = root_of_unity_2_32();
    for (uint32_t i = lg; i < TWO_ADACITY; i++) d.group_gen = Fr::sqr(d.group_gen);
    Fr raw = Fr::zero();
    raw.v[0] = (uint32_t)s;
    raw.v[1] = (uint32_t)(s >> 32);
    d.size_as_field_element = Fr::to_mont(raw);
    d.size_inv = Fr::inv(d.size_as_field_element);
    d.group_gen_inv = Fr::inv(d.group_gen);
    d.generator_inv = Fr::inv(coset_generator());
    return d;
}

// ---------------------------------------------------------------------------------------------------------
// Entry of a transform's stage table: the twiddle as the nine 29-bit limbs the tile arithmetic multiplies by, so that a
// butterfly loads its factor ready for use (36 bytes instead of the 32 of the packed form, and no unpacking).
struct Tw9 {
    uint32_t v[9];
};
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void tw_load(const Tw9 *p, uint32_t *w) {
    const u32x4_unaligned a = *reinterpret_cast<const u32x4_unaligned *>(p->v);
    const u32x4_unaligned b = *reinterpret_cast<const u32x4_unaligned *>(p->v + 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    w[8] = p->v[8];
}

struct NttPassArgs {
    const Fr *in;
    Fr *out;
    const Tw9 *tw;  // stage tables
    const Fr *gl;  // g^i, i < 4096      (coset_fft input scaling)
    const Fr *gh;  // g^(4096 i)
    uint64_t len_in;
    uint32_t log_n, b0, log_r, log_t;
    uint32_t last, coset_in, inv_shift;
    uint32_t unit_last;  // 1 when the table entry of the h = 1 stage is 1 (plain tables): skip that product
};

__device__ __forceinline__ Fr gload(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void gstore(Fr *p, const Fr &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// x * 2^-k mod r for 0 <= k < 32: add the multiple of r that clears the low k bits, then shift.
// r = 1 mod 2^32, so that multiple is (-x) mod 2^k; the quotient is already below r.
__device__ __forceinline__ Fr fr_div_pow2(const Fr &x, uint32_t k) {
    if (k == 0) return x;
    uint32_t m = (0u - x.v[0]) & ((1u << k) - 1u);
    uint32_t t[9];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t cur = (uint64_t)m * FrParams::mod(i) + x.v[i] + c;
        t[i] = (uint32_t)cur;
        c = cur >> 32;
    }
    t[8] = (uint32_t)c;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (t[i] >> k) | (t[i + 1] << (32 - k));
    return r;
}

__device__ __forceinline__ void ntt_load_bias(uint32_t km, uint32_t *bias) {
    // literals selected by uniform branches (a table indexed by km would be fetched from memory)
    if (km == 0) ntt_bias<0>(bias);
    else if (km == 1) ntt_bias<1>(bias);
    else if (km == 2) ntt_bias<2>(bias);
    else ntt_bias<3>(bias);
}
__device__ __forceinline__ void ntt_load_bias_wide(uint32_t km, uint32_t *bias) {
    if (km == 0) ntt_bias_wide<0>(bias);
    else if (km == 1) ntt_bias_wide<1>(bias);
    else if (km == 2) ntt_bias_wide<2>(bias);
    else ntt_bias_wide<3>(bias);
}
// One decimation-in-frequency butterfly on tile limbs: x <- x + y (normalised when NORM_SUM), y <- (x - y + M r) * w
// with the stage's bias M r; tw == nullptr stands for w = 1 (the last stage of a plain table).  Without the carry sweep
// the sum's limbs reach 2^30 - 2: the next stage then subtracts it under the wide bias (the product's column sums
// still fit: 9 * 2.5 * 2^30 * 2^29 + 9 * 2^58 < 2^64) and sweeps its own sum.
template <bool NORM_SUM = true>
__device__ __forceinline__ void u9_butterfly(uint32_t *x, uint32_t *y, const uint32_t *bias, const Tw9 *tw) {
    uint32_t d[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        d[k] = x[k] + (bias[k] - y[k]);
        x[k] += y[k];
    }
    if (NORM_SUM) u9_normalize(x);
    if (tw) {
        uint32_t w[9];
        tw_load(tw, w);
        Fr::mul_limbs(d, w, y);
    } else {
        u9_normalize(d);
#pragma unroll
        for (int k = 0; k < 9; k++) y[k] = d[k];
    }
}

// LOG_E != 0 fixes the tile at 2^LOG_E elements at compile time, so that the nine limb planes sit at constant LDS
// offsets; LOG_E == 0 takes any tile geometry (the tuning knob and small transforms).
template <uint32_t LOG_E, uint32_t MAX_THREADS>
__global__ void __launch_bounds__(MAX_THREADS) ntt_pass_kernel(NttPassArgs a) {
    extern __shared__ uint32_t lds[];
    const uint32_t R = 1u << a.log_r, T = 1u << a.log_t, E = LOG_E ? (1u << LOG_E) : R * T;
    const uint32_t tid = threadIdx.x, nt = blockDim.x, tile = blockIdx.x;
    const uint32_t pb = a.log_n - a.log_r - a.b0;  // prefix bits above the row field

    uint32_t base, col0;
    if (!a.last) {
        const uint32_t tpp = a.b0 - a.log_t;  // log2(tiles per prefix)
        const uint32_t prefix = tile >> tpp;
        col0 = (tile & ((1u << tpp) - 1u)) << a.log_t;
        base = (prefix << (a.b0 + a.log_r)) | col0;
    } else {
        base = tile;  // low part q of the prefix; column u supplies the top log_t prefix bits
        col0 = 0;
    }

    // ---- load ----
    for (uint32_t e = tid; e < E; e += nt) {
        uint32_t j, t, gi;
        if (!a.last) {
            j = e >> a.log_t;
            t = e & (T - 1);
            gi = base | (j << a.b0) | t;
        } else {
            t = e >> a.log_r;
            j = e & (R - 1);
            gi = ((((t << (pb - a.log_t)) | base)) << a.log_r) | j;
        }
        Fr x = Fr::zero();
        if ((uint64_t)gi < a.len_in) {
            x = gload(a.in + gi);
            if (a.coset_in) {
                Fr g = gload(a.gl + (gi & 4095u));
                if (gi >= 4096u) g = g * gload(a.gh + (gi >> 12));
                x = x * g;
            }
        }
        uint32_t u[9];
        Fr::unpack(x, u, 0);
        u9_store(lds, E, j * T + t, u);
    }

    // ---- butterflies ----
    // A zero-padded input (the 8n coset evaluation of an n-coefficient polynomial) leaves most rows of a first-pass
    // tile zero: `live` counts the leading rows of every butterfly block that can be non-zero.  While live <= hj the
    // upper operand of every butterfly is zero, so the lower one stays as it is and the upper one becomes x * w;
    // butterflies whose lower operand is zero as well are skipped.  After such a stage every half block again has
    // `live` leading rows; the first stage with live > hj fills the blocks and the rest run in full.
    uint32_t live = R;
    if (!a.last && pb == 0 && a.len_in < (1ull << a.log_n)) {
        const uint64_t rows = a.len_in > col0 ? ((a.len_in - col0 - 1) >> a.b0) + 1 : 0;
        live = rows < R ? (uint32_t)rows : R;
    }
    const uint32_t col_term = a.last ? 0u : col0;   // twiddle index = (row offset << b0) + column
    uint32_t stage = 0;
    while (stage < a.log_r && live <= (R >> (stage + 1))) {
        const uint32_t hj = R >> (stage + 1);
        __syncthreads();
        const uint32_t h = hj << a.b0;  // half-size of this stage in the whole network
        const uint32_t lhj = 31 - __clz(hj);
        for (uint32_t bi = tid; bi < (E >> 1); bi += nt) {
            const uint32_t t = bi & (T - 1);
            const uint32_t jb = bi >> a.log_t;
            const uint32_t off = jb & (hj - 1);
            if (off >= live) continue;
            const uint32_t j = ((jb >> lhj) << (lhj + 1)) | off;
            const uint32_t e0 = j * T + t, e1 = e0 + hj * T;
            uint32_t x[9], y[9];
            u9_load(lds, E, e0, x);      // normalised and below 2r: it has only been through products so far
            if (h > 1 || !a.unit_last) {
                const uint32_t widx = (off << a.b0) + col0 + t;
                uint32_t w[9];
                tw_load(a.tw + (h - 1) + widx, w);
                Fr::mul_limbs(x, w, y);
                u9_store(lds, E, e1, y);
            } else {
                u9_store(lds, E, e1, x);
            }
        }
        stage++;
    }
    // The full stages: one single stage when their number is odd, then two stages per LDS round trip.  A thread
    // holds the four rows j0 + m * hq (m = 0..3) of one column: the stage with half-size 2 hq pairs (0,2) and (1,3), the
    // stage with half-size hq pairs (0,1) and (2,3), whose twiddles coincide.  Operand bounds are those of the
    // stage-by-stage form: inputs of stage k below 2^(k mod 4) * 2r, a quarter-reduction before every fourth stage.
    if ((a.log_r - stage) & 1u) {
        const uint32_t hj = R >> (stage + 1);
        __syncthreads();
        const uint32_t h = hj << a.b0;
        const uint32_t lhj = 31 - __clz(hj);
        const uint32_t km = stage & 3u;
        const bool reduce_in = km == 0 && stage > 0;
        const bool has_w = h > 1 || !a.unit_last;
        uint32_t bias[9];
        ntt_load_bias(km, bias);
        for (uint32_t bi = tid; bi < (E >> 1); bi += nt) {
            const uint32_t t = bi & (T - 1);
            const uint32_t jb = bi >> a.log_t;
            const uint32_t off = jb & (hj - 1);
            const uint32_t j = ((jb >> lhj) << (lhj + 1)) | off;
            const uint32_t e0 = j * T + t, e1 = e0 + hj * T;
            uint32_t x[9], y[9];
            u9_load(lds, E, e0, x);
            u9_load(lds, E, e1, y);
            if (reduce_in) {
                u9_qreduce(x);
                u9_qreduce(y);
            }
            u9_butterfly(x, y, bias, has_w ? a.tw + (h - 1) + (off << a.b0) + (a.last ? 0u : t) + col_term : nullptr);
            u9_store(lds, E, e1, y);
            u9_store(lds, E, e0, x);
        }
        stage++;
    }
    for (; stage + 1 < a.log_r; stage += 2) {
        const uint32_t hq = R >> (stage + 2);          // the second stage's half-size in rows; the first one's is 2 hq
        __syncthreads();
        const uint32_t h_a = (2 * hq) << a.b0, h_b = hq << a.b0;
        const uint32_t lhq = 31 - __clz(hq);
        const uint32_t km_a = stage & 3u, km_b = (stage + 1) & 3u;
        const bool reduce_a = km_a == 0 && stage > 0, reduce_b = km_b == 0;
        const bool has_w_b = h_b > 1 || !a.unit_last;
        // the first of the two stages leaves its sums without a carry sweep unless the second one starts with a
        // reduction (which wants normalised limbs)
        const bool lazy = !reduce_b;
        uint32_t bias_a[9], bias_b[9];
        ntt_load_bias(km_a, bias_a);
        if (lazy) ntt_load_bias_wide(km_b, bias_b);
        else ntt_load_bias(km_b, bias_b);
        for (uint32_t g = tid; g < (E >> 2); g += nt) {
            const uint32_t t = g & (T - 1);
            const uint32_t jg = g >> a.log_t;
            const uint32_t off = jg & (hq - 1);
            const uint32_t j0 = ((jg >> lhq) << (lhq + 2)) | off;
            const uint32_t e0 = j0 * T + t, step = hq * T;
            uint32_t x0[9], x1[9], x2[9], x3[9];
            u9_load(lds, E, e0, x0);
            u9_load(lds, E, e0 + step, x1);
            u9_load(lds, E, e0 + 2 * step, x2);
            u9_load(lds, E, e0 + 3 * step, x3);
            if (reduce_a) {
                u9_qreduce(x0);
                u9_qreduce(x1);
                u9_qreduce(x2);
                u9_qreduce(x3);
            }
            const uint32_t col = (a.last ? 0u : t) + col_term;
            const Tw9 *tw_a = a.tw + (h_a - 1) + (off << a.b0) + col;         // row offsets off and off + hq
            if (lazy) {
                u9_butterfly<false>(x0, x2, bias_a, tw_a);
                u9_butterfly<false>(x1, x3, bias_a, tw_a + ((size_t)hq << a.b0));
            } else {
                u9_butterfly(x0, x2, bias_a, tw_a);
                u9_butterfly(x1, x3, bias_a, tw_a + ((size_t)hq << a.b0));
            }
            if (reduce_b) {
                u9_qreduce(x0);
                u9_qreduce(x1);
                u9_qreduce(x2);
                u9_qreduce(x3);
            }
            const Tw9 *tw_b = has_w_b ? a.tw + (h_b - 1) + (off << a.b0) + col : nullptr;
            u9_butterfly(x0, x1, bias_b, tw_b);
            u9_butterfly(x2, x3, bias_b, tw_b);
            u9_store(lds, E, e0, x0);
            u9_store(lds, E, e0 + step, x1);
            u9_store(lds, E, e0 + 2 * step, x2);
            u9_store(lds, E, e0 + 3 * step, x3);
        }
    }
    __syncthreads();

    // ---- store: back to the canonical 32-bit-word form ----
    for (uint32_t e = tid; e < E; e += nt) {
        uint32_t le, gi;
        if (!a.last) {
            const uint32_t j = e >> a.log_t, t = e & (T - 1);
            gi = base | (j << a.b0) | t;
            le = e;
        } else {
            // consecutive lanes -> consecutive output positions: the column index is bit-reversed
            const uint32_t tr = e & (T - 1), j = e >> a.log_t;
            const uint32_t t = a.log_t ? (__brev(tr) >> (32 - a.log_t)) : 0u;
            const uint32_t src = ((((t << (pb - a.log_t)) | base)) << a.log_r) | j;
            gi = a.log_n ? (__brev(src) >> (32 - a.log_n)) : 0u;
            le = j * T + t;
        }
        uint32_t u[9];
        u9_load(lds, E, le, u);
        u9_qreduce(u);
        Fr::cond_sub_limbs(u);
        Fr x = Fr::pack(u);
        gstore(a.out + gi, a.last ? fr_div_pow2(x, a.inv_shift) : x);
    }
}

// out[gid] = c_q * base_q^idx for gid = 2^q - 1 + idx, idx < 2^q ; consts = {base_0, c_0, base_1, c_1, ...}
// LIMBS: entries in the tile arithmetic's limb form (the transforms' tables); otherwise packed field elements
template <bool LIMBS>
__global__ void gen_stage_tables_kernel(void *out, uint64_t count, const Fr *consts) {
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t q = 63 - __clzll(gid + 1);
    uint64_t idx = gid + 1 - (1ull << q);
    Fr b = gload(consts + 2 * q), c = gload(consts + 2 * q + 1);
    Fr r = c;
    for (uint32_t bit = 0; bit < q; bit++) {
        if ((idx >> bit) & 1) r = r * b;
        b = Fr::sqr(b);
    }
    if (LIMBS) {
        uint32_t u[9];
        Fr::unpack(r, u, 0);
        Tw9 *dst = static_cast<Tw9 *>(out) + gid;
#pragma unroll
        for (int k = 0; k < 9; k++) dst->v[k] = u[k];
    } else {
        gstore(static_cast<Fr *>(out) + gid, r);
    }
}

// out[i] = base^i
__global__ void gen_powers_kernel(Fr *out, uint32_t count, Fr base) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    Fr r = Fr::one(), b = base;
    for (uint32_t e = gid; e; e >>= 1) {
        if (e & 1) r = r * b;
        b = Fr::sqr(b);
    }
    gstore(out + gid, r);
}

// ---------------------------------------------------------------------------------------------------------
NttEngine::~NttEngine() {
    tw_fwd_.release();
    tw_plain_.release();
    tw_inv_.release();
    g_.release();
    alt_.release();
    fold_.release();
    xchg_a_.release();
    xchg_b_.release();
    work_.release();
    stage_consts_.release();
}

void NttEngine::ShiftTables::release() {
    lo.release();
    hi.release();
    for (auto &kv : cinv) kv.second.release();
    cinv.clear();
    pow_log = 0;
}
size_t NttEngine::ShiftTables::bytes() const {
    size_t b = lo.cap + hi.cap;
    for (auto &kv : cinv) b += kv.second.cap;
    return b;
}

size_t NttEngine::device_bytes() const {
    return tw_fwd_.cap + tw_inv_.cap + tw_plain_.cap + g_.bytes() + alt_.bytes() + fold_.cap + xchg_a_.cap + xchg_b_.cap +
           work_.cap + stage_consts_.cap;
}

// times_32: a transform's table (limb entries, see Tw9); otherwise the packed table behind roots_half()
static void fill_stage_tables(hipStream_t stream, DevBuf &consts_buf, void *table, uint32_t log_n, const Fr &w_n,
                              const Fr *coset_base /* g^-1 or null */, bool times_32) {
    // stage q (half-size h = 2^q) uses w_{2h} = w_n^(n/2h) and, for the coset table, c_q = g^(-n/2h)
    std::vector<Fr> consts(2 * 32, Fr::one());
    Fr w = w_n;  // w_{2h} for q = log_n - 1
    Fr c = coset_base ? *coset_base : Fr::one();
    // the transform's tables carry a factor 32 (see the tile arithmetic above); the plain table feeds roots_half()
    const Fr scale = times_32 ? Fr::from_u32(32) : Fr::one();
    for (int q = (int)log_n - 1; q >= 0; q--) {
        consts[2 * q] = w;
        consts[2 * q + 1] = Fr::mul(coset_base ? c : Fr::one(), scale);
        w = Fr::sqr(w);
        if (coset_base) c = Fr::sqr(c);
    }
    consts_buf.reserve(consts.size() * sizeof(Fr));
    // the previous table generation (if any) reads consts_buf on this stream: order the overwrite after it
    ZK_HIP(hipStreamSynchronize(stream));
    ZK_HIP(hipMemcpyAsync(consts_buf.p, consts.data(), consts.size() * sizeof(Fr), hipMemcpyHostToDevice, stream));
    uint64_t count = (1ull << log_n) - 1;
    if (count == 0) return;
    if (times_32)
        hipLaunchKernelGGL(gen_stage_tables_kernel<true>, dim3(ceil_div(count, 256)), dim3(256), 0, stream, table, count,
                           consts_buf.as<Fr>());
    else
        hipLaunchKernelGGL(gen_stage_tables_kernel<false>, dim3(ceil_div(count, 256)), dim3(256), 0, stream, table, count,
                           consts_buf.as<Fr>());
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream));  // consts is a host temporary
}

void NttEngine::ensure_tables(uint32_t log_n) {
    if (log_n <= table_log_) return;
    Domain d = Domain::make(1ull << log_n);
    size_t bytes = ((1ull << log_n)) * sizeof(Tw9);
    tw_fwd_.reserve(bytes);
    tw_inv_.reserve(bytes);
    fill_stage_tables(stream_, stage_consts_, tw_fwd_.p, log_n, d.group_gen, nullptr, true);
    fill_stage_tables(stream_, stage_consts_, tw_inv_.p, log_n, d.group_gen_inv, nullptr, true);
    table_log_ = log_n;
}

const Fr *NttEngine::roots_half(uint32_t log_n) {
    if (log_n == 0) log_n = 1;
    if (plain_log_ != log_n) {
        Domain d = Domain::make(1ull << log_n);
        tw_plain_.reserve(((size_t)1 << log_n) * sizeof(Fr));
        fill_stage_tables(stream_, stage_consts_, tw_plain_.p, log_n, d.group_gen, nullptr, false);
        plain_log_ = log_n;
    }
    // the stage with half-size h = n/2 holds w_n^i at entries h-1 .. 2h-2
    return tw_plain_.as<Fr>() + ((size_t)1 << (log_n - 1)) - 1;
}

const void *NttEngine::coset_inv_table(ShiftTables &st, uint32_t log_n) {
    auto it = st.cinv.find(log_n);
    if (it != st.cinv.end()) return it->second.p;
    Domain d = Domain::make(1ull << log_n);
    DevBuf &buf = st.cinv[log_n];
    buf.reserve((1ull << log_n) * sizeof(Tw9));
    // c_q = (s^-1)^(n/2h): for the top stage (h = n/2) that is s^-1 itself
    fill_stage_tables(stream_, stage_consts_, buf.p, log_n, d.group_gen_inv, &st.shift_inv, true);
    return buf.p;
}

void NttEngine::ensure_coset_powers(ShiftTables &st, uint32_t log_n) {
    uint32_t need = log_n > 12 ? log_n : 12;
    if (st.pow_log >= need) return;
    st.lo.reserve(4096 * sizeof(Fr));
    uint32_t n_hi = 1u << (need - 12);
    st.hi.reserve((size_t)n_hi * sizeof(Fr));
    hipLaunchKernelGGL(gen_powers_kernel, dim3(16), dim3(256), 0, stream_, st.lo.as<Fr>(), 4096u, st.shift);
    Fr s4096 = Fr::pow_u64(st.shift, 4096);
    hipLaunchKernelGGL(gen_powers_kernel, dim3(ceil_div(n_hi, 256)), dim3(256), 0, stream_, st.hi.as<Fr>(), n_hi, s4096);
    ZK_HIP(hipGetLastError());
    st.pow_log = need;
}

void NttEngine::set_alt_shift(const Fr &s) {
    if (alt_set_ && alt_.shift == s) return;
    ZK_HIP(hipStreamSynchronize(stream_));   // kernels in flight may still read the old tables
    alt_.release();
    alt_.shift = s;
    alt_.shift_inv = Fr::inv(s);
    alt_set_ = true;
}

// Split the k index bits into passes of at most 10 row bits; tiles hold 2^tile_log_ elements (32 bytes each in LDS).
NttEngine::Plan NttEngine::plan(uint32_t k) const {
    Plan p;
    const uint32_t TILE_LOG = tile_log_, MAX_R = TILE_LOG < 10 ? TILE_LOG : 10;
    if (k <= TILE_LOG) {
        p.pass[p.count++] = Pass{0, k, 0};
        return p;
    }
    uint32_t L = (k + MAX_R - 1) / MAX_R;
    uint32_t hi = k;
    for (uint32_t i = 0; i < L; i++) {
        uint32_t r = k / L + (i < k % L ? 1 : 0);
        uint32_t b0 = hi - r;
        uint32_t lt = TILE_LOG - r;
        if (i + 1 < L) {
            if (lt > b0) lt = b0;
        } else {
            uint32_t pb = k - r;  // last pass: columns are prefixes
            if (lt > pb) lt = pb;
        }
        p.pass[p.count++] = Pass{b0, r, lt};
        hi = b0;
    }
    return p;
}

bool NttEngine::tables_resident(NttMode mode, uint32_t log_n) const {
    if (mode == NTT_COSET_IFFT) return log_n == 0 || g_.cinv.find(log_n) != g_.cinv.end();
    if (log_n > table_log_) return false;
    if (mode == NTT_COSET_FFT && g_.pow_log < (log_n > 12 ? log_n : 12)) return false;
    return true;
}

void NttEngine::transform(NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n) {
    if (!g_set_) {
        g_.shift = coset_generator();
        g_.shift_inv = Fr::inv(g_.shift);
        g_set_ = true;
    }
    transform_with(g_, mode, in, len_in, out, log_n);
}

void NttEngine::transform_with(ShiftTables &st, NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n) {
    ZK_REQUIRE(log_n < TWO_ADACITY, ZKGPU_ERR_DOMAIN_SIZE, "InvalidEvalDomainSize: log2(size) must be below 32");
    ZK_REQUIRE(in != nullptr || len_in == 0, ZKGPU_ERR_INVALID_ARG, "ntt: null input");
    ZK_REQUIRE(out != nullptr, ZKGPU_ERR_INVALID_ARG, "ntt: null output");
    const bool inverse = (mode == NTT_IFFT || mode == NTT_COSET_IFFT);
    const void *tw;
    if (mode == NTT_COSET_IFFT) {
        tw = log_n ? coset_inv_table(st, log_n) : nullptr;
    } else {
        ensure_tables(log_n);
        tw = inverse ? tw_inv_.p : tw_fwd_.p;
    }
    if (mode == NTT_COSET_FFT) ensure_coset_powers(st, log_n);
    if (plan(log_n).count > 1) work_.reserve(((size_t)1 << log_n) * sizeof(Fr));
    run_passes(mode, in, len_in, out, log_n, tw, work_.as<Fr>(), st);
}

void NttEngine::transform_no_alloc(NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n, Fr *work) {
    ZK_REQUIRE(log_n < TWO_ADACITY, ZKGPU_ERR_DOMAIN_SIZE, "InvalidEvalDomainSize: log2(size) must be below 32");
    ZK_REQUIRE(in != nullptr || len_in == 0, ZKGPU_ERR_INVALID_ARG, "ntt: null input");
    ZK_REQUIRE(out != nullptr && work != nullptr, ZKGPU_ERR_INVALID_ARG, "ntt: null output");
    ZK_REQUIRE(tables_resident(mode, log_n), ZKGPU_ERR_NOT_LOADED,
               "ntt: the twiddle tables of this mode and size are not resident; run the allocating transform once first");
    const void *tw;
    if (mode == NTT_COSET_IFFT)
        tw = log_n ? g_.cinv.find(log_n)->second.p : nullptr;
    else
        tw = (mode == NTT_IFFT) ? tw_inv_.p : tw_fwd_.p;
    run_passes(mode, in, len_in, out, log_n, tw, work, g_);
}

void NttEngine::run_passes(NttMode mode, const Fr *in, size_t len_in, Fr *out, uint32_t log_n, const void *tw, Fr *work,
                           const ShiftTables &st) {
    const uint64_t n = 1ull << log_n;
    if (len_in > n) len_in = n;  // Vec::resize truncates (domain.rs:146)
    const bool inverse = (mode == NTT_IFFT || mode == NTT_COSET_IFFT);
    const Plan passes = plan(log_n);
    const Fr *src = in;
    for (uint32_t i = 0; i < passes.count; i++) {
        const bool last = (i + 1 == passes.count);
        NttPassArgs a;
        a.in = src;
        a.out = last ? out : work;
        a.tw = static_cast<const Tw9 *>(tw);
        a.gl = st.lo.as<Fr>();
        a.gh = st.hi.as<Fr>();
        a.len_in = (i == 0) ? (uint64_t)len_in : n;
        a.log_n = log_n;
        a.b0 = passes.pass[i].b0;
        a.log_r = passes.pass[i].log_r;
        a.log_t = passes.pass[i].log_t;
        a.last = last ? 1 : 0;
        a.coset_in = (i == 0 && mode == NTT_COSET_FFT) ? 1 : 0;
        a.inv_shift = (last && inverse) ? log_n : 0;
        a.unit_last = (mode == NTT_COSET_IFFT) ? 0 : 1;
        const uint32_t tile_log = a.log_r + a.log_t;
        const uint64_t E = 1ull << tile_log;
        uint32_t threads = (uint32_t)(E >> ept_log_);
        if (threads > 1024) threads = 1024;
        if (threads < 64) threads = 64;
        const uint32_t grid = (uint32_t)(n >> tile_log);
        const size_t lds_bytes = E * 9 * sizeof(uint32_t);
        if (!lds_attr_set_) {
            ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_pass_kernel<0, 1024>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            lds_attr_set_ = true;
        }
        if (tile_log == 9 && threads <= 256)
            hipLaunchKernelGGL((ntt_pass_kernel<9, 256>), dim3(grid), dim3(threads), lds_bytes, stream_, a);
        else
            hipLaunchKernelGGL((ntt_pass_kernel<0, 1024>), dim3(grid), dim3(threads), lds_bytes, stream_, a);
        ZK_HIP(hipGetLastError());
        src = work;
    }
}

// ---- one transform over several GPUs --------------------------------------------------------------------------------
// out[m] = sum_q in[m + q M] * sM^q for m < min(len, M): the input modulo X^M - s^M
__global__ void __launch_bounds__(256) ntt_fold_kernel(const Fr *in, uint64_t len, Fr *out, uint64_t M, Fr sM) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M || m >= len) return;
    Fr acc = gload(in + m), pw = sM;
    for (uint64_t j = m + M; j < len; j += M) {
        acc = acc + gload(in + j) * pw;
        pw = pw * sM;
    }
    gstore(out + m, acc);
}

// The world-point inverse DFT across the ranks' residues: recv holds block c (from rank c) at recv + c B, B elements
// each: the coefficients of this rank's index range modulo X^M - s_c^M.  out[q B + m'] = sum_c mat[q world + c] * recv[c B + m'].
struct CombineArgs {
    const Fr *recv;
    Fr *out;
    const Fr *mat;   // world x world constants
    uint64_t B;
    uint32_t world;
};
__global__ void __launch_bounds__(256) ntt_combine_kernel(CombineArgs a) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.B) return;
    Fr u[8];
    for (uint32_t c = 0; c < a.world; c++) u[c] = gload(a.recv + c * a.B + m);
    for (uint32_t q = 0; q < a.world; q++) {
        Fr acc = u[0] * gload(a.mat + q * a.world);
        for (uint32_t c = 1; c < a.world; c++) acc = acc + u[c] * gload(a.mat + q * a.world + c);
        gstore(a.out + q * a.B + m, acc);
    }
}
// gathered[r][q][m'] -> full[q M + r B + m']
__global__ void __launch_bounds__(256) ntt_unshuffle_kernel(const Fr *gathered, Fr *full, uint64_t B, uint32_t log_world) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // index into gathered
    const uint64_t M = B << log_world;
    if (i >= (M << log_world)) return;
    const uint64_t r = i / M, rest = i - r * M, q = rest / B, m = rest - q * B;
    gstore(full + q * M + r * B + m, gload(gathered + i));
}

void NttEngine::slice_forward(bool coset, const Fr *in, size_t len_in, Fr *out, uint32_t log_n, uint32_t rank, uint32_t world) {
    ZK_REQUIRE(log_n < TWO_ADACITY, ZKGPU_ERR_DOMAIN_SIZE, "InvalidEvalDomainSize: log2(size) must be below 32");
    uint32_t lw = 0;
    while ((1u << lw) < world) lw++;
    ZK_REQUIRE((1u << lw) == world && lw <= log_n && rank < world, ZKGPU_ERR_INVALID_ARG,
               "ntt slice: the number of ranks must be a power of two not above the domain size");
    const uint64_t n = 1ull << log_n, M = n >> lw;
    if (len_in > n) len_in = n;   // Vec::resize truncates (domain.rs:146)
    // this rank's points: (g) w^rank * <w^world>
    const Domain d = Domain::make(n);
    Fr s = Fr::pow_u64(d.group_gen, rank);
    if (coset) s = s * coset_generator();
    set_alt_shift(s);
    const Fr *src = in;
    if (len_in > M) {
        fold_.reserve(M * sizeof(Fr));
        hipLaunchKernelGGL(ntt_fold_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, stream_, in, (uint64_t)len_in, fold_.as<Fr>(), M,
                           Fr::pow_u64(s, M));
        ZK_HIP(hipGetLastError());
        src = fold_.as<Fr>();
        len_in = M;
    }
    transform_with(alt_, NTT_COSET_FFT, src, len_in, out, log_n - lw);
}

void NttEngine::inverse_sharded(bool coset, const Fr *slice, Fr *out_full, uint32_t log_n, const Collective &coll) {
    ZK_REQUIRE(coll.fn != nullptr, ZKGPU_ERR_INVALID_ARG, "sharded inverse transform: no collective function set");
    const uint32_t world = coll.world, lw = coll.log_world();
    ZK_REQUIRE((1u << lw) == world && world <= 8 && 2 * lw <= log_n && coll.rank < world, ZKGPU_ERR_INVALID_ARG,
               "sharded inverse transform: 2, 4 or 8 ranks and at least world^2 points");
    const uint64_t n = 1ull << log_n, M = n >> lw, B = M >> lw;
    const Domain d = Domain::make(n);
    const Fr g = coset ? coset_generator() : Fr::one();
    set_alt_shift(Fr::pow_u64(d.group_gen, coll.rank) * g);
    xchg_a_.reserve(n * sizeof(Fr) + (size_t)world * world * sizeof(Fr));
    xchg_b_.reserve(M * sizeof(Fr));
    Fr *a = xchg_a_.as<Fr>(), *b = xchg_b_.as<Fr>();
    Fr *mat_dev = a + n;
    // u = the polynomial modulo X^M - s^M, natural order: block r of B elements goes to rank r
    transform_with(alt_, NTT_COSET_IFFT, slice, M, b, log_n - lw);
    // mat[q][c] = g^(-q M) * world^-1 * w_world^(-c q)
    std::vector<Fr> mat((size_t)world * world);
    {
        const Fr w_inv = Fr::pow_u64(d.group_gen_inv, M);   // w_world^-1
        const Fr gM_inv = Fr::inv(Fr::pow_u64(g, M));
        const Fr world_inv = Fr::inv(Fr::from_u32(world));
        Fr gq = world_inv;
        for (uint32_t q = 0; q < world; q++) {
            const Fr wq = Fr::pow_u64(w_inv, q);
            Fr cur = gq;
            for (uint32_t c = 0; c < world; c++) {
                mat[(size_t)q * world + c] = cur;
                cur = cur * wq;
            }
            gq = gq * gM_inv;
        }
    }
    ZK_HIP(hipMemcpyAsync(mat_dev, mat.data(), mat.size() * sizeof(Fr), hipMemcpyHostToDevice, stream_));
    ZK_HIP(hipStreamSynchronize(stream_));
    ZK_REQUIRE(coll.fn(coll.user, 0, b, a, (size_t)B * sizeof(Fr)) == 0, ZKGPU_ERR_INTERNAL, "the all-to-all exchange failed");
    CombineArgs ca;
    ca.recv = a;
    ca.out = b;
    ca.mat = mat_dev;
    ca.B = B;
    ca.world = world;
    hipLaunchKernelGGL(ntt_combine_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, stream_, ca);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream_));
    ZK_REQUIRE(coll.fn(coll.user, 1, b, a, (size_t)M * sizeof(Fr)) == 0, ZKGPU_ERR_INTERNAL, "the all-gather exchange failed");
    hipLaunchKernelGGL(ntt_unshuffle_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream_, a, out_full, B, lw);
    ZK_HIP(hipGetLastError());
}

}  // namespace zk
