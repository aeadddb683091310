// A/B of two ways to sum the buckets of a large MSM, on real curve points and with the memory footprint of the real
// thing: (A) the production form, one lane per bucket adding its entries one after the other in extended Jacobian
// coordinates (8 products + 2 squarings per addition, one 128-byte table line read per addition); (B) batch-affine
// tree rounds: inside every bucket neighbouring entries are added pairwise in affine coordinates, a lane's pairs of a
// round sharing ONE field inversion through running products that are spilled to device memory (5 products + 1
// squaring per addition, plus the inversion's ~490 over the batch), three rounds (an eighth of the points left), and the
// production form on what remains.  Both end in the same bucket sums, which the tool checks.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkvm_amd/csrc tools/batch_affine_ab.hip -o tools/batch_affine_ab
//   tools/batch_affine_ab [log2 points = 22] [entries per bucket = points * 13 / 2^19] [lanes per CU for B = 512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fq28.hpp"

using namespace zk;

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

struct alignas(128) Entry {   // x | y in 14 + 14 limbs of 28 bits, on the production table's 128-byte pitch
    uint4 q[8];
};
struct alignas(64) Pre {      // one running product, padded to 64 bytes
    uint4 q[4];
};

__device__ __forceinline__ Fq28 load_x(const Entry *p) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 t = p->q[k];
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
    Fq28 r;
#pragma unroll
    for (int k = 0; k < 14; k++) r.v[k] = w[k];
    return r;
}
__device__ __forceinline__ G1Affine28 load_xy(const Entry *p) {
    uint32_t w[28];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const uint4 t = p->q[k];
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
    G1Affine28 r;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        r.x.v[k] = w[k];
        r.y.v[k] = w[14 + k];
    }
    return r;
}
__device__ __forceinline__ void store_xy(Entry *p, const G1Affine28 &a) {
    uint32_t w[28];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        w[k] = a.x.v[k];
        w[14 + k] = a.y.v[k];
    }
#pragma unroll
    for (int k = 0; k < 7; k++) p->q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ Fq28 load_pre(const Pre *p) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 t = p->q[k];
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
    Fq28 r;
#pragma unroll
    for (int k = 0; k < 14; k++) r.v[k] = w[k];
    return r;
}
__device__ __forceinline__ void store_pre(Pre *p, const Fq28 &a) {
    uint32_t w[16] = {0};
#pragma unroll
    for (int k = 0; k < 14; k++) w[k] = a.v[k];
#pragma unroll
    for (int k = 0; k < 4; k++) p->q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ void store_xyzz(G1Xyzz *p, const G1Xyzz &a) {
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
__device__ __forceinline__ G1Xyzz load_xyzz(const G1Xyzz *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint32_t w[56];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const uint4 t = q[k];
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

// a^(p - 2) by square-and-multiply over the 381 bits of p - 2: 380 squarings and 200-odd products
__device__ __noinline__ Fq28 fq28_inv(const Fq28 &a) {
    // p - 2, 28-bit limbs (p's lowest limb is ...aaab)
    Fq28 r = a;
    bool started = false;
    for (int i = 13; i >= 0; i--) {
        uint32_t limb = Fq28::P(i);
        if (i == 0) limb -= 2;
        for (int b = 27; b >= 0; b--) {
            const uint32_t bit = (limb >> b) & 1u;
            if (!started) {
                if (bit) started = true;   // r = a
                continue;
            }
            r = Fq28::sqr(r);
            if (bit) r = Fq28::mul(r, a);
        }
    }
    return r;
}

// entries[e] = (e + 1) * G in affine form: double-and-add on the extended Jacobian accumulator, one inversion
__global__ void __launch_bounds__(256) make_points_kernel(G1Affine28 g, uint32_t count, Entry *entries) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const uint32_t k = e + 1;
    G1Xyzz acc = G1Xyzz::identity();
    for (int bit = 31; bit >= 0; bit--) {
        if (!acc.is_identity()) acc = xyzz_double(acc);
        if ((k >> bit) & 1u) acc = xyzz_add_mixed(acc, g, false);
    }
    // x = X / ZZ, y = Y / ZZZ
    const Fq28 zinv = fq28_inv(Fq28::mul(acc.zz, acc.zzz));      // 1 / (ZZ ZZZ)
    G1Affine28 p;
    p.x = Fq28::mul(acc.x, Fq28::mul(zinv, acc.zzz));
    p.y = Fq28::mul(acc.y, Fq28::mul(zinv, acc.zz));
    store_xy(entries + e, p);
}

// The sorted list of the production sort, synthetic: position pos of bucket-major order holds entry
// (pos * A) mod n_entries (A coprime to n_entries: every entry exactly once), sign from a hash bit.
__global__ void __launch_bounds__(256) make_sorted_kernel(uint32_t n_entries, uint32_t mult, uint32_t *sorted) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n_entries) return;
    const uint32_t e = (uint32_t)(((uint64_t)pos * mult) % n_entries);
    const uint32_t h = (pos * 2654435761u) >> 13;
    sorted[pos] = e | ((h & 1u) << 31);
}

// ---- (A) production form ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) accumulate_xyzz_kernel(const Entry *table, const uint32_t *sorted, uint32_t per_bucket,
                                                              uint32_t n_buckets, G1Xyzz *buckets) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    G1Xyzz acc = G1Xyzz::identity();
    const uint32_t start = b * per_bucket;
    for (uint32_t k = 0; k < per_bucket; k++) {
        const uint32_t v = sorted[start + k];
        const G1Affine28 p = load_xy(table + (v & 0x7fffffffu));
        acc = xyzz_add_mixed(acc, p, (v >> 31) != 0);
    }
    store_xyzz(buckets + b, acc);
}
// the same over points that already lie bucket by bucket (what the tree rounds leave); LAZY_X: their x coordinates come
// lazily reduced (below 130p) and are brought below 2p first, as the accumulator's invariants want
template <bool LAZY_X>
__global__ void __launch_bounds__(256) accumulate_rest_kernel(const Entry *points, uint32_t per_bucket, uint32_t n_buckets,
                                                              G1Xyzz *buckets) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    G1Xyzz acc = G1Xyzz::identity();
    const Entry *p = points + (size_t)b * per_bucket;
    for (uint32_t k = 0; k < per_bucket; k++) {
        G1Affine28 q = load_xy(p + k);
        if (LAZY_X) q.x = Fq28::mul(q.x, Fq28::one());
        acc = xyzz_add_mixed(acc, q, false);
    }
    store_xyzz(buckets + b, acc);
}

// ---- (B) one batch-affine tree round ----------------------------------------------------------------------------
// Pair q adds inputs 2q and 2q + 1 (neighbours inside one bucket: buckets hold an even number of inputs) into output q.
// Lane t takes the pairs t, t + T, t + 2T, ... (T = all lanes of the launch): consecutive lanes read consecutive pairs.
// Pass 1 multiplies the pairs' denominators x2 - x1 up and spills every running product; one inversion; pass 2 walks
// back, peels each denominator's inverse off the running inverse and finishes the additions.
// GATHER: the inputs are table entries named by the sorted list (sign in bit 31); otherwise `in` holds them in order.
// Equal x coordinates (the same point twice, or a point and its negative) take the tangent's denominator 2y or cancel
// to the point at infinity (all-zero entry); an all-zero input leaves the other input unchanged.
struct Operand {
    Fq28 x, y;
    bool inf;
};
template <bool GATHER>
__device__ __forceinline__ const Entry *operand_at(const Entry *in, const uint32_t *sorted, size_t idx, bool &neg) {
    if (GATHER) {
        const uint32_t v = sorted[idx];
        neg = (v >> 31) != 0;
        return in + (v & 0x7fffffffu);
    }
    neg = false;
    return in + idx;
}
// the denominator this pair contributes to the shared inversion, and which case it is:
//   0 ordinary (x2 - x1), 1 doubling (2 y1), 2 no division (an input at infinity, or P + (-P))
__device__ __forceinline__ int pair_denominator(const G1Affine28 &p1, bool neg1, const G1Affine28 &p2, bool neg2, Fq28 &den) {
    if (p1.is_identity() || p2.is_identity()) {
        den = Fq28::one();
        return 2;
    }
    const Fq28 dx = Fq28::normalize(Fq28::sub4(p2.x, p1.x));     // x entries are below 2p, normalised
    if (!Fq28::mul(dx, Fq28::one()).is_zero_mod_p()) {
        den = dx;
        return 0;
    }
    const Fq28 y1 = neg1 ? Fq28::normalize(Fq28::sub4(Fq28::zero(), p1.y)) : p1.y;
    const Fq28 y2 = neg2 ? Fq28::normalize(Fq28::sub4(Fq28::zero(), p2.y)) : p2.y;
    if (Fq28::mul(Fq28::normalize(Fq28::sub4(y2, y1)), Fq28::one()).is_zero_mod_p()) {
        den = Fq28::normalize(Fq28::add(y1, y1));
        return 1;
    }
    den = Fq28::one();
    return 2;
}

template <bool GATHER>
__global__ void __launch_bounds__(256) affine_round_kernel(const Entry *in, const uint32_t *sorted, uint64_t n_pairs, Pre *spill,
                                                           Entry *out) {
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x, t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs) return;
    const uint64_t mine = (n_pairs - t + T - 1) / T;
    // pass 1: running products of the denominators (x coordinates only in the ordinary case)
    Fq28 run = Fq28::one();
    for (uint64_t j = 0; j < mine; j++) {
        const uint64_t q = t + j * T;
        bool n1, n2;
        const Entry *e1 = operand_at<GATHER>(in, sorted, 2 * q, n1), *e2 = operand_at<GATHER>(in, sorted, 2 * q + 1, n2);
        const Fq28 x1 = load_x(e1), x2 = load_x(e2);
        Fq28 den = Fq28::normalize(Fq28::sub4(x2, x1));
        // rare cases need the whole points: equal x (also two points at infinity, whose x are both zero), or an x of zero
        if (Fq28::mul(den, Fq28::one()).is_zero_mod_p() || x1.limbs_zero() || x2.limbs_zero())
            (void)pair_denominator(load_xy(e1), n1, load_xy(e2), n2, den);
        store_pre(spill + q, run);          // the product of the denominators BEFORE this pair
        run = Fq28::mul(run, den);
    }
    Fq28 inv = fq28_inv(run);
    // pass 2, backwards: 1 / den_j = inv * (product before j); inv <- inv * den_j
    for (uint64_t j = mine; j-- > 0;) {
        const uint64_t q = t + j * T;
        bool n1, n2;
        const Entry *e1 = operand_at<GATHER>(in, sorted, 2 * q, n1), *e2 = operand_at<GATHER>(in, sorted, 2 * q + 1, n2);
        const G1Affine28 p1 = load_xy(e1), p2 = load_xy(e2);
        Fq28 den;
        const int kind = pair_denominator(p1, n1, p2, n2, den);
        const Fq28 dinv = Fq28::mul(inv, load_pre(spill + q));
        inv = Fq28::mul(inv, den);
        const Fq28 y1 = n1 ? Fq28::normalize(Fq28::sub4(Fq28::zero(), p1.y)) : p1.y;
        const Fq28 y2 = n2 ? Fq28::normalize(Fq28::sub4(Fq28::zero(), p2.y)) : p2.y;
        G1Affine28 r;
        if (kind == 2) {
            if (p1.is_identity()) {
                r.x = p2.x;
                r.y = y2;
            } else if (p2.is_identity()) {
                r.x = p1.x;
                r.y = y1;
            } else {
                r.x = Fq28::zero();
                r.y = Fq28::zero();
            }
        } else {
            Fq28 num;
            if (kind == 0) {
                num = Fq28::normalize(Fq28::sub4(y2, y1));
            } else {
                const Fq28 xx = Fq28::sqr(p1.x);
                num = Fq28::normalize(Fq28::add(Fq28::add(xx, xx), xx));     // 3 x^2 (the curve has a = 0)
            }
            const Fq28 lambda = Fq28::mul(num, dinv);
            const Fq28 ll = Fq28::sqr(lambda);
            // x3 = lambda^2 - x1 - x2 ; y3 = lambda (x1 - x3) - y1
            r.x = Fq28::normalize(Fq28::sub4(Fq28::normalize(Fq28::sub4(ll, p1.x)), p2.x));
            r.x = Fq28::mul(r.x, Fq28::one());                              // back below 2p for the next subtraction
            const Fq28 dx = Fq28::normalize(Fq28::sub4(p1.x, r.x));
            const Fq28 ly = Fq28::mul(lambda, dx);
            r.y = Fq28::mul(Fq28::normalize(Fq28::sub4(ly, y1)), Fq28::one());
        }
        store_xy(out + q, r);
    }
}


// ---- the same round without the rare cases on its critical path -------------------------------------------------
// Values stay lazily reduced between rounds (no product spent on bringing a coordinate back below 2p): a round's inputs
// are below `in_bound` multiples of p with normalised limbs, every subtraction adds a multiple of p that covers its
// subtrahend (bias tables made on the host: k p with the limbs below the top raised by 2^28), x leaves a round below
// 2 + 2 k_x multiples of p and y below 2p (its last step is a double product under one Montgomery reduction).  A pair
// with equal x coordinates, or an input at infinity, would put a zero into the running product: the lane then only
// raises `*flag`, and the host repeats all rounds with the checked kernel above.
struct RoundBias {
    Fq28 kx;    // >= the inputs' x bound: den = x2 - x1 + kx, x3 = lambda^2 - x1 - x2 + 2 kx
    Fq28 kdx;   // >= the outputs' x bound: dx = x1 - x3 + kdx
    Fq28 ky;    // >= the inputs' y bound (4p: table entries are below 2p, their negatives below 4p)
};
__device__ __forceinline__ Fq28 sub_bias(const Fq28 &a, const Fq28 &b, const Fq28 &bias) {   // a - b + bias, b normalised
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.v[i] = a.v[i] + (bias.v[i] - b.v[i]);
    return r;
}
template <bool GATHER>
__global__ void __launch_bounds__(256) affine_round_fast_kernel(const Entry *in, const uint32_t *sorted, uint64_t n_pairs, Pre *spill,
                                                                Entry *out, RoundBias bias, uint32_t *flag) {
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x, t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs) return;
    const uint64_t mine = (n_pairs - t + T - 1) / T;
    Fq28 run = Fq28::one();
    for (uint64_t j = 0; j < mine; j++) {
        const uint64_t q = t + j * T;
        bool n1, n2;
        const Entry *e1 = operand_at<GATHER>(in, sorted, 2 * q, n1), *e2 = operand_at<GATHER>(in, sorted, 2 * q + 1, n2);
        const Fq28 den = sub_bias(load_x(e2), load_x(e1), bias.kx);
        store_pre(spill + q, run);
        run = Fq28::mul(run, den);
    }
    if (Fq28::mul(run, Fq28::one()).is_zero_mod_p()) {
        *flag = 1u;
        return;
    }
    Fq28 inv = fq28_inv(run);
    for (uint64_t j = mine; j-- > 0;) {
        const uint64_t q = t + j * T;
        bool n1, n2;
        const Entry *e1 = operand_at<GATHER>(in, sorted, 2 * q, n1), *e2 = operand_at<GATHER>(in, sorted, 2 * q + 1, n2);
        const G1Affine28 p1 = load_xy(e1), p2 = load_xy(e2);
        const Fq28 den = sub_bias(p2.x, p1.x, bias.kx);
        const Fq28 dinv = Fq28::mul(inv, load_pre(spill + q));
        inv = Fq28::mul(inv, den);
        // numerator (+-y2) - (+-y1), below 2 ky + in_bound
        Fq28 num;
        if (n1 == n2) {
            num = n1 ? sub_bias(p1.y, p2.y, bias.ky) : sub_bias(p2.y, p1.y, bias.ky);
        } else if (n1) {
            num = Fq28::add(p2.y, p1.y);
        } else {
            num = sub_bias(sub_bias(Fq28::zero(), p2.y, bias.ky), p1.y, bias.ky);
        }
        const Fq28 lambda = Fq28::mul(num, dinv);
        const Fq28 ll = Fq28::sqr(lambda);
        G1Affine28 r;
        r.x = Fq28::normalize(sub_bias(sub_bias(ll, p1.x, bias.kx), p2.x, bias.kx));
        const Fq28 dx = sub_bias(p1.x, r.x, bias.kdx);
        // y3 = lambda dx - (+-y1): the second product is (-+y1) * 1 under the same reduction
        const Fq28 my1 = n1 ? p1.y : Fq28::normalize(sub_bias(Fq28::zero(), p1.y, bias.ky));
        r.y = Fq28::mul_add(lambda, dx, my1, Fq28::one());
        store_xy(out + q, r);
    }
}

// buckets equal as points? X1 ZZ2 = X2 ZZ1 and Y1 ZZZ2 = Y2 ZZZ1 (both nonzero, or both the identity)
__global__ void __launch_bounds__(256) compare_kernel(const G1Xyzz *a, const G1Xyzz *b, uint32_t n, uint32_t *mismatches) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Xyzz p = load_xyzz(a + i), q = load_xyzz(b + i);
    bool same;
    if (p.is_identity() || q.is_identity()) {
        same = p.is_identity() && q.is_identity();
    } else {
        const Fq28 l1 = Fq28::mul(p.x, q.zz), r1 = Fq28::mul(q.x, p.zz);
        const Fq28 l2 = Fq28::mul(p.y, q.zzz), r2 = Fq28::mul(q.y, p.zzz);
        same = Fq28::mul(Fq28::normalize(Fq28::sub4(l1, r1)), Fq28::one()).is_zero_mod_p() &&
               Fq28::mul(Fq28::normalize(Fq28::sub4(l2, r2)), Fq28::one()).is_zero_mod_p();
    }
    if (!same) atomicAdd(mismatches, 1u);
}

static G1Affine28 generator28() {
    // the G1 generator (coset-bls12_381/src/g1.rs:218-238) as x * 2^384 mod p, y * 2^384 mod p in 32-bit words
    const uint64_t gx[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                            0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
    const uint64_t gy[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                            0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
    G1Affine g;
    for (int i = 0; i < 6; i++) {
        g.x.v[2 * i] = (uint32_t)gx[i];
        g.x.v[2 * i + 1] = (uint32_t)(gx[i] >> 32);
        g.y.v[2 * i] = (uint32_t)gy[i];
        g.y.v[2 * i + 1] = (uint32_t)(gy[i] >> 32);
    }
    return affine_to_28(g);
}

// k p with every limb below the top raised by 2^28 (borrowed from the limb above): limb-wise "bias - b" stays
// non-negative for a normalised b below k p - 2^364
static Fq28 bias_multiple(uint32_t k) {
    uint64_t limb[14];
    uint64_t carry = 0;
    for (int i = 0; i < 14; i++) {
        const uint64_t t = (uint64_t)Fq28::P(i) * k + carry;
        limb[i] = t & Fq28::MASK;
        carry = t >> 28;
    }
    limb[13] += carry << 28;
    for (int i = 0; i < 13; i++) {
        limb[i] += 1ull << 28;
        limb[i + 1] -= 1;
    }
    Fq28 r;
    for (int i = 0; i < 14; i++) r.v[i] = (uint32_t)limb[i];
    return r;
}

static float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main(int argc, char **argv) {
    const uint32_t log_n = argc > 1 ? (uint32_t)atoi(argv[1]) : 22;
    const uint32_t K = 13, log_buckets = 19, n_buckets = 1u << log_buckets;
    const uint64_t n_entries64 = (uint64_t)K << log_n;
    uint32_t per_bucket = argc > 2 ? (uint32_t)atoi(argv[2]) : (uint32_t)(n_entries64 >> log_buckets);
    const uint32_t lanes_per_cu = argc > 3 ? (uint32_t)atoi(argv[3]) : 512;
    const uint32_t ROUNDS = 3;
    if (per_bucket % (1u << ROUNDS) != 0 || per_bucket == 0 || (uint64_t)per_bucket * n_buckets > n_entries64 ||
        n_entries64 >= (1ull << 31)) {
        printf("entries per bucket must be a positive multiple of %u and fit the table\n", 1u << ROUNDS);
        return 1;
    }
    const uint32_t n_entries = (uint32_t)n_entries64;
    const uint64_t used = (uint64_t)per_bucket * n_buckets;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    printf("2^%u points x %u table levels = %u entries of 128 bytes (%.1f GiB), 2^%u buckets of %u entries, %u CUs\n", log_n, K,
           n_entries, n_entries * 128.0 / (1 << 30), log_buckets, per_bucket, cus);

    Entry *table, *round_out[3];
    uint32_t *sorted, *mismatches;
    Pre *spill;
    G1Xyzz *buckets_a, *buckets_b;
    CHECK(hipMalloc(&table, (size_t)n_entries * sizeof(Entry)));
    CHECK(hipMalloc(&sorted, (size_t)n_entries * 4));
    CHECK(hipMalloc(&spill, (size_t)(used / 2) * sizeof(Pre)));
    for (uint32_t r = 0; r < ROUNDS; r++) CHECK(hipMalloc(&round_out[r], (size_t)(used >> (r + 1)) * sizeof(Entry)));
    CHECK(hipMalloc(&buckets_a, (size_t)n_buckets * sizeof(G1Xyzz)));
    CHECK(hipMalloc(&buckets_b, (size_t)n_buckets * sizeof(G1Xyzz)));
    CHECK(hipMalloc(&mismatches, 4));
    CHECK(hipMemset(mismatches, 0, 4));

    hipLaunchKernelGGL(make_points_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, 0, generator28(), n_entries, table);
    uint32_t mult = 2654435761u % n_entries;
    while (mult % 2 == 0 || mult % 13 == 0) mult++;   // coprime to 13 * 2^log_n
    hipLaunchKernelGGL(make_sorted_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, 0, n_entries, mult, sorted);
    CHECK(hipDeviceSynchronize());

    hipEvent_t ev[8];
    for (int i = 0; i < 8; i++) CHECK(hipEventCreate(&ev[i]));
    const int REPS = 2;
    float a_ms = 0, b_ms[ROUNDS + 1] = {0};
    for (int rep = 0; rep < REPS; rep++) {
        CHECK(hipEventRecord(ev[0], 0));
        hipLaunchKernelGGL(accumulate_xyzz_kernel, dim3(n_buckets / 256), dim3(256), 0, 0, table, sorted, per_bucket, n_buckets, buckets_a);
        CHECK(hipEventRecord(ev[1], 0));
        const Entry *in = table;
        for (uint32_t r = 0; r < ROUNDS; r++) {
            const uint64_t n_pairs = used >> (r + 1);
            const uint32_t grid = cus * lanes_per_cu / 256;
            if (r == 0)
                hipLaunchKernelGGL(affine_round_kernel<true>, dim3(grid), dim3(256), 0, 0, in, sorted, n_pairs, spill, round_out[r]);
            else
                hipLaunchKernelGGL(affine_round_kernel<false>, dim3(grid), dim3(256), 0, 0, in, sorted, n_pairs, spill, round_out[r]);
            in = round_out[r];
            CHECK(hipEventRecord(ev[2 + r], 0));
        }
        hipLaunchKernelGGL(accumulate_rest_kernel<false>, dim3(n_buckets / 256), dim3(256), 0, 0, in, per_bucket >> ROUNDS, n_buckets, buckets_b);
        CHECK(hipEventRecord(ev[2 + ROUNDS], 0));
        CHECK(hipDeviceSynchronize());
        CHECK(hipGetLastError());
        a_ms = elapsed(ev[0], ev[1]);
        for (uint32_t r = 0; r <= ROUNDS; r++) b_ms[r] = elapsed(ev[1 + r], ev[2 + r]);
    }
    hipLaunchKernelGGL(compare_kernel, dim3(n_buckets / 256), dim3(256), 0, 0, buckets_a, buckets_b, n_buckets, mismatches);
    uint32_t bad = 0;
    CHECK(hipMemcpy(&bad, mismatches, 4, hipMemcpyDeviceToHost));

    // (B') the same rounds without the rare cases in the loop and without reductions between the rounds
    float f_ms[ROUNDS + 1] = {0};
    uint32_t *flag;
    CHECK(hipMalloc(&flag, 4));
    CHECK(hipMemset(flag, 0, 4));
    CHECK(hipMemset(buckets_b, 0, (size_t)n_buckets * sizeof(G1Xyzz)));
    for (int rep = 0; rep < REPS; rep++) {
        const Entry *in = table;
        uint32_t in_bound = 2;   // multiples of p the x coordinates of a round's inputs stay below
        CHECK(hipEventRecord(ev[1], 0));
        for (uint32_t r = 0; r < ROUNDS; r++) {
            const uint64_t n_pairs = used >> (r + 1);
            const uint32_t grid = cus * lanes_per_cu / 256;
            uint32_t kx = 4;
            while (kx <= in_bound) kx *= 4;
            const uint32_t out_bound = 2 + 2 * kx;
            uint32_t kdx = 4;
            while (kdx <= out_bound) kdx *= 4;
            RoundBias rb;
            rb.kx = bias_multiple(kx);
            rb.kdx = bias_multiple(kdx);
            rb.ky = bias_multiple(4);
            if (r == 0)
                hipLaunchKernelGGL(affine_round_fast_kernel<true>, dim3(grid), dim3(256), 0, 0, in, sorted, n_pairs, spill, round_out[r], rb, flag);
            else
                hipLaunchKernelGGL(affine_round_fast_kernel<false>, dim3(grid), dim3(256), 0, 0, in, sorted, n_pairs, spill, round_out[r], rb, flag);
            in = round_out[r];
            in_bound = out_bound;
            CHECK(hipEventRecord(ev[2 + r], 0));
        }
        hipLaunchKernelGGL(accumulate_rest_kernel<true>, dim3(n_buckets / 256), dim3(256), 0, 0, in, per_bucket >> ROUNDS, n_buckets, buckets_b);
        CHECK(hipEventRecord(ev[2 + ROUNDS], 0));
        CHECK(hipDeviceSynchronize());
        CHECK(hipGetLastError());
        for (uint32_t r = 0; r <= ROUNDS; r++) f_ms[r] = elapsed(ev[1 + r], ev[2 + r]);
    }
    uint32_t flagged = 0, bad_fast = 0;
    CHECK(hipMemcpy(&flagged, flag, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemset(mismatches, 0, 4));
    hipLaunchKernelGGL(compare_kernel, dim3(n_buckets / 256), dim3(256), 0, 0, buckets_a, buckets_b, n_buckets, mismatches);
    CHECK(hipMemcpy(&bad_fast, mismatches, 4, hipMemcpyDeviceToHost));

    const double adds_a = (double)used - n_buckets;
    double total_b = 0;
    for (uint32_t r = 0; r <= ROUNDS; r++) total_b += b_ms[r];
    printf("(A) extended Jacobian, one lane per bucket:         %8.3f ms  (%.1f M additions, %.3f ns each)\n", a_ms, adds_a / 1e6,
           a_ms * 1e6 / adds_a);
    for (uint32_t r = 0; r < ROUNDS; r++) {
        const double pairs = (double)(used >> (r + 1));
        // per addition: two x reads (pass 1), two whole entries (pass 2), the spilled product written and read, the sum written
        const double bytes = pairs * ((r == 0 ? 2 * 64 + 2 * 128 : 2 * 64 + 2 * 128) + 2 * 64 + 128 + (r == 0 ? 16 : 0));
        printf("(B) affine round %u (%s): %8.3f ms  (%.1f M additions, %.3f ns each, %.0f per lane and inversion, %.2f TB/s of "
               "requested bytes)\n", r, r == 0 ? "table gather" : "in order    ", b_ms[r], pairs / 1e6, b_ms[r] * 1e6 / pairs,
               pairs / (cus * lanes_per_cu), bytes / (b_ms[r] * 1e-3) / 1e12);
    }
    printf("(B) extended Jacobian over the remaining %u per bucket: %8.3f ms\n", per_bucket >> ROUNDS, b_ms[ROUNDS]);
    printf("(B) total %8.3f ms  vs (A) %8.3f ms  -> B / A = %.3f ; bucket sums %s (%u of %u differ)\n", total_b, a_ms, total_b / a_ms,
           bad ? "DIFFER" : "equal", bad, n_buckets);
    double total_f = 0;
    for (uint32_t r = 0; r <= ROUNDS; r++) total_f += f_ms[r];
    for (uint32_t r = 0; r < ROUNDS; r++) {
        const double pairs = (double)(used >> (r + 1));
        printf("(B') lazy affine round %u: %8.3f ms  (%.3f ns per addition)\n", r, f_ms[r], f_ms[r] * 1e6 / pairs);
    }
    printf("(B') extended Jacobian over the remaining %u per bucket: %8.3f ms\n", per_bucket >> ROUNDS, f_ms[ROUNDS]);
    printf("(B') total %8.3f ms  vs (A) %8.3f ms  -> B' / A = %.3f ; rare-case flag %u ; bucket sums %s (%u of %u differ)\n", total_f,
           a_ms, total_f / a_ms, flagged, bad_fast ? "DIFFER" : "equal", bad_fast, n_buckets);
    return (bad || bad_fast) ? 2 : 0;
}
