// Prover-side polynomial kernels for gfx950 (see poly.hpp).  All field results are canonical, so any order of
// exact field operations gives the reference's limbs; the kernels are free to replace the reference's serial
// recurrences (per-gate inversion + running product, Horner sums, synthetic division) by scans and reductions.
#include "poly.hpp"
#include "fr9.hpp"
#include "widgets.hpp"
#include <vector>

namespace zk {

__device__ __forceinline__ Fr ld(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void st(Fr *p, const Fr &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// ---- inclusive scans over Fr --------------------------------------------------------------------------------
struct OpMul {
    __device__ static Fr id() { return Fr::one(); }
    __device__ static Fr op(const Fr &a, const Fr &b) { return Fr::mul(a, b); }
};
struct OpAdd {
    __device__ static Fr id() { return Fr::zero(); }
    __device__ static Fr op(const Fr &a, const Fr &b) { return Fr::add(a, b); }
};

static const uint32_t SCAN_T = 256, SCAN_E = 4, SCAN_B = SCAN_T * SCAN_E;

__device__ __forceinline__ void lds_st(uint32_t *sh, uint32_t slot, const Fr &r) {
#pragma unroll
    for (int i = 0; i < 8; i++) sh[slot * 8 + i] = r.v[i];
}
__device__ __forceinline__ Fr lds_ld(const uint32_t *sh, uint32_t slot) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = sh[slot * 8 + i];
    return r;
}

// Hillis-Steele inclusive scan of one value per thread across a 256-thread block; returns the exclusive prefix of
// the calling thread and leaves the block total in *total (valid in every thread).
template <class Op>
__device__ __forceinline__ Fr block_exclusive(uint32_t *sh /* 2 * 256 * 8 words */, const Fr &mine, Fr *total) {
    const uint32_t tid = threadIdx.x;
    uint32_t *cur = sh, *nxt = sh + SCAN_T * 8;
    lds_st(cur, tid, mine);
    __syncthreads();
    for (uint32_t d = 1; d < SCAN_T; d <<= 1) {
        Fr v = lds_ld(cur, tid);
        if (tid >= d) v = Op::op(lds_ld(cur, tid - d), v);
        lds_st(nxt, tid, v);
        __syncthreads();
        uint32_t *t = cur;
        cur = nxt;
        nxt = t;
    }
    *total = lds_ld(cur, SCAN_T - 1);
    Fr ex = tid ? lds_ld(cur, tid - 1) : Op::id();
    __syncthreads();
    return ex;
}

// Logical element j lives at physical index j (forward) or n-1-j (reverse).
template <class Op, bool REV>
__global__ void __launch_bounds__(256) scan_local_kernel(const Fr *in, Fr *out, Fr *totals, uint64_t n) {
    __shared__ uint32_t sh[2 * SCAN_T * 8];
    const uint64_t j0 = (uint64_t)blockIdx.x * SCAN_B + (uint64_t)threadIdx.x * SCAN_E;
    Fr v[SCAN_E];
    Fr run = Op::id();
#pragma unroll
    for (uint32_t k = 0; k < SCAN_E; k++) {
        uint64_t j = j0 + k;
        Fr x = Op::id();
        if (j < n) x = ld(in + (REV ? n - 1 - j : j));
        run = Op::op(run, x);
        v[k] = run;
    }
    Fr total;
    Fr ex = block_exclusive<Op>(sh, run, &total);
#pragma unroll
    for (uint32_t k = 0; k < SCAN_E; k++) {
        uint64_t j = j0 + k;
        if (j < n) st(out + (REV ? n - 1 - j : j), Op::op(ex, v[k]));
    }
    if (threadIdx.x == 0) st(totals + blockIdx.x, total);
}

// exclusive scan of the block totals, in place (single workgroup, chunks of 256 with a running carry)
template <class Op>
__global__ void __launch_bounds__(256) scan_totals_kernel(Fr *totals, uint32_t n_blocks) {
    __shared__ uint32_t sh[2 * SCAN_T * 8];
    Fr carry = Op::id();
    for (uint32_t start = 0; start < n_blocks; start += SCAN_T) {
        uint32_t i = start + threadIdx.x;
        Fr v = i < n_blocks ? ld(totals + i) : Op::id();
        Fr total;
        Fr ex = block_exclusive<Op>(sh, v, &total);
        if (i < n_blocks) st(totals + i, Op::op(carry, ex));
        carry = Op::op(carry, total);
    }
}

template <class Op, bool REV>
__global__ void __launch_bounds__(256) scan_apply_kernel(Fr *out, const Fr *totals, uint64_t n) {
    if (blockIdx.x == 0) return;  // prefix of block 0 is the identity
    const Fr pre = ld(totals + blockIdx.x);
    const uint64_t j0 = (uint64_t)blockIdx.x * SCAN_B + (uint64_t)threadIdx.x * SCAN_E;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_E; k++) {
        uint64_t j = j0 + k;
        if (j < n) {
            Fr *p = out + (REV ? n - 1 - j : j);
            st(p, Op::op(pre, ld(p)));
        }
    }
}

template <class Op, bool REV>
static void inclusive_scan(hipStream_t stream, const Fr *in, Fr *out, uint64_t n, DevBuf &totals) {
    if (n == 0) return;
    const uint32_t n_blocks = (uint32_t)((n + SCAN_B - 1) / SCAN_B);
    totals.reserve((size_t)n_blocks * sizeof(Fr));
    hipLaunchKernelGGL((scan_local_kernel<Op, REV>), dim3(n_blocks), dim3(SCAN_T), 0, stream, in, out, totals.as<Fr>(), n);
    if (n_blocks > 1) {
        hipLaunchKernelGGL((scan_totals_kernel<Op>), dim3(1), dim3(SCAN_T), 0, stream, totals.as<Fr>(), n_blocks);
        hipLaunchKernelGGL((scan_apply_kernel<Op, REV>), dim3(n_blocks), dim3(SCAN_T), 0, stream, out, totals.as<Fr>(), n);
    }
    ZK_HIP(hipGetLastError());
}

// ---- permutation accumulator --------------------------------------------------------------------------------
struct PermArgs {
    const Fr *w[4];
    const Fr *s[4];
    const Fr *roots_half;
    Fr beta, gamma;
    uint32_t n;
};

__global__ void __launch_bounds__(256) perm_numden_kernel(PermArgs a, Fr *num, Fr *den) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t half = a.n >> 1;
    Fr root;
    if (a.n == 1) {
        root = Fr::one();
    } else {
        root = ld(a.roots_half + (i & (half - 1)));
        if (i >= half) root = Fr::neg(root);  // w^(i) = -w^(i - n/2)
    }
    const Fr br = Fr::mul(a.beta, root);
    Fr w0 = ld(a.w[0] + i), w1 = ld(a.w[1] + i), w2 = ld(a.w[2] + i), w3 = ld(a.w[3] + i);
    // K1, K2, K3 = 7, 13, 17 (plonk/src/composer/permutation/constants.rs:7-9)
    Fr n0 = Fr::add(Fr::add(w0, br), a.gamma);
    Fr n1 = Fr::add(Fr::add(w1, mul_small<7>(br)), a.gamma);
    Fr n2 = Fr::add(Fr::add(w2, mul_small<13>(br)), a.gamma);
    Fr n3 = Fr::add(Fr::add(w3, mul_small<17>(br)), a.gamma);
    st(num + i, Fr::mul(Fr::mul(n0, n1), Fr::mul(n2, n3)));
    Fr d0 = Fr::add(Fr::add(w0, Fr::mul(a.beta, ld(a.s[0] + i))), a.gamma);
    Fr d1 = Fr::add(Fr::add(w1, Fr::mul(a.beta, ld(a.s[1] + i))), a.gamma);
    Fr d2 = Fr::add(Fr::add(w2, Fr::mul(a.beta, ld(a.s[2] + i))), a.gamma);
    Fr d3 = Fr::add(Fr::add(w3, Fr::mul(a.beta, ld(a.s[3] + i))), a.gamma);
    st(den + i, Fr::mul(Fr::mul(d0, d1), Fr::mul(d2, d3)));
}

// z[i] = (prod_{j<i} num_j) * (prod_{j>=i} den_j) / (prod_j den_j)
__global__ void __launch_bounds__(256) perm_finish_kernel(const Fr *pnum_incl, const Fr *sden_incl, Fr inv_total, uint32_t n,
                                                          Fr *z) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = Fr::mul(ld(sden_incl + i), inv_total);
    if (i) v = Fr::mul(v, ld(pnum_incl + i - 1));
    st(z + i, v);
}

void PolyEngine::permutation_z(const Fr *const wires[4], const Fr *const sigma_evals[4], const Fr &beta, const Fr &gamma,
                               const Fr *roots_half, uint32_t n, Fr *z_out, bool *closed) {
    ZK_REQUIRE(n >= 1, ZKGPU_ERR_INVALID_ARG, "permutation_z: empty domain");
    s0_.reserve((size_t)n * sizeof(Fr));
    s1_.reserve((size_t)n * sizeof(Fr));
    PermArgs a;
    for (int k = 0; k < 4; k++) {
        a.w[k] = wires[k];
        a.s[k] = sigma_evals[k];
    }
    a.roots_half = roots_half;
    a.beta = beta;
    a.gamma = gamma;
    a.n = n;
    Fr *num = s0_.as<Fr>(), *den = s1_.as<Fr>();
    hipLaunchKernelGGL(perm_numden_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream_, a, num, den);
    inclusive_scan<OpMul, false>(stream_, num, num, n, s2_);
    inclusive_scan<OpMul, true>(stream_, den, den, n, s2_);
    // the one inversion of the whole accumulator: a single field element, so it is cheaper on the host than as a
    // ~100k-instruction dependent chain on one GPU lane
    Fr total, total_num;
    ZK_HIP(hipMemcpyAsync(&total, den, sizeof(Fr), hipMemcpyDeviceToHost, stream_));
    if (closed) ZK_HIP(hipMemcpyAsync(&total_num, num + (n - 1), sizeof(Fr), hipMemcpyDeviceToHost, stream_));
    ZK_HIP(hipStreamSynchronize(stream_));
    if (closed) *closed = !total.is_zero() && total_num == total;
    hipLaunchKernelGGL(perm_finish_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream_, num, den, Fr::inv(total), n, z_out);
    ZK_HIP(hipGetLastError());
}

// ---- arithmetic gate, row by row ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gate_check_kernel(const Fr *wires, const Fr *sel_h, const Fr *pi, uint32_t n, uint32_t *flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr a = ld(wires + i), b = ld(wires + n + i), c = ld(wires + 2 * (size_t)n + i), d = ld(wires + 3 * (size_t)n + i);
    Fr t = Fr::mul(Fr::mul(a, b), ld(sel_h + i));
    t = Fr::add(t, Fr::mul(a, ld(sel_h + (size_t)n + i)));
    t = Fr::add(t, Fr::mul(b, ld(sel_h + 2 * (size_t)n + i)));
    t = Fr::add(t, Fr::mul(c, ld(sel_h + 3 * (size_t)n + i)));
    t = Fr::add(t, Fr::mul(d, ld(sel_h + 4 * (size_t)n + i)));
    t = Fr::add(t, ld(sel_h + 5 * (size_t)n + i));
    t = Fr::add(Fr::mul(t, ld(sel_h + 6 * (size_t)n + i)), ld(pi + i));
    if (!t.is_zero()) *flag = 1u;   // every writer stores the same value
}
void PolyEngine::gate_check(const Fr *wires, const Fr *sel_h, const Fr *pi, uint32_t n, uint32_t *flag_dev) {
    if (!n) return;
    hipLaunchKernelGGL(gate_check_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream_, wires, sel_h, pi, n, flag_dev);
    ZK_HIP(hipGetLastError());
}

// ---- widget gates, row by row -----------------------------------------------------------------------------------
// For every row whose range, logic, fixed-base or curve-addition selector is not zero: does each part of that widget's
// identity vanish on the witness?  (The prover weights the parts with powers of a challenge; all of them zero is what
// makes the weighted sum zero whatever the challenge.)  sel_h: the selector values on the domain in the order of Sel,
// Q_M ... Q_VAR.  The next row of the last row is the first (the domain is cyclic).
static __device__ __noinline__ Fr fr_mul_outlined(Fr a, Fr b) { return Fr::mul(a, b); }
struct OutlinedMul {
    __device__ __forceinline__ static Fr mul(const Fr &a, const Fr &b) { return fr_mul_outlined(a, b); }
};
__global__ void __launch_bounds__(256) widget_check_kernel(const Fr *wires, const Fr *sel_h, uint32_t n, uint32_t widgets, uint32_t *flag) {
    typedef FrRing<OutlinedMul> R;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = n;
    bool on[4];
    bool any = false;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        on[w] = ((widgets >> w) & 1u) && !ld(sel_h + (Q_RANGE + w) * N + i).is_zero();
        any = any || on[w];
    }
    if (!any) return;
    const uint32_t iw = (i + 1) & (n - 1);
    const Fr a = ld(wires + i), b = ld(wires + N + i), c = ld(wires + 2 * N + i), d = ld(wires + 3 * N + i);
    const Fr a_w = ld(wires + iw), b_w = ld(wires + N + iw), d_w = ld(wires + 3 * N + iw);
    Fr p[5];
    bool bad = false;
    if (on[0]) {
        w_range_parts<R>(a, b, c, d, d_w, p);
        for (int j = 0; j < W_RANGE_PARTS; j++) bad = bad || !p[j].is_zero();
    }
    if (on[1]) {
        w_logic_parts<R>(a, a_w, b, b_w, c, d, d_w, ld(sel_h + Q_C * N + i), p);
        for (int j = 0; j < W_LOGIC_PARTS; j++) bad = bad || !p[j].is_zero();
    }
    if (on[2]) {
        w_fixed_base_parts<R>(a, a_w, b, b_w, c, d, d_w, ld(sel_h + Q_L * N + i), ld(sel_h + Q_R * N + i), ld(sel_h + Q_C * N + i), p);
        for (int j = 0; j < W_FIXED_PARTS; j++) bad = bad || !p[j].is_zero();
    }
    if (on[3]) {
        w_curve_add_parts<R>(a, a_w, b, b_w, c, d, d_w, p);
        for (int j = 0; j < W_CURVE_PARTS; j++) bad = bad || !p[j].is_zero();
    }
    if (bad) *flag = 1u;   // every writer stores the same value
}
void PolyEngine::widget_check(const Fr *wires, const Fr *sel_h, uint32_t n, uint32_t widgets, uint32_t *flag_dev) {
    if (!n || !widgets) return;
    ZK_REQUIRE((n & (n - 1)) == 0, ZKGPU_ERR_INVALID_ARG, "widget_check: n must be a power of two");
    hipLaunchKernelGGL(widget_check_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream_, wires, sel_h, n, widgets, flag_dev);
    ZK_HIP(hipGetLastError());
}

struct WindowArgs {
    const Fr *src[8];
    uint64_t first[8];
    uint32_t m, count;
};
__global__ void gather_windows_kernel(WindowArgs a, Fr *dst) {
    const uint32_t j = threadIdx.x / 8, i = threadIdx.x % 8;
    if (j < a.m && i < a.count) st(dst + j * a.count + i, ld(a.src[j] + a.first[j] + i));
}
void PolyEngine::gather_windows(uint32_t m, const Fr *const *src, const uint64_t *first, uint32_t count, Fr *dst) {
    ZK_REQUIRE(m <= 8 && count <= 8, ZKGPU_ERR_INVALID_ARG, "gather_windows: at most 8 windows of 8 elements");
    if (!m || !count) return;
    WindowArgs a;
    a.m = m;
    a.count = count;
    for (uint32_t j = 0; j < m; j++) {
        a.src[j] = src[j];
        a.first[j] = first[j];
    }
    hipLaunchKernelGGL(gather_windows_kernel, dim3(1), dim3(64), 0, stream_, a, dst);
    ZK_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) take_even_kernel(const Fr *src, uint64_t count, Fr *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) st(dst + i, ld(src + 2 * i));
}
void PolyEngine::take_even(const Fr *src, uint64_t count, Fr *dst) {
    if (!count) return;
    hipLaunchKernelGGL(take_even_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, stream_, src, count, dst);
    ZK_HIP(hipGetLastError());
}

// ---- quotient -----------------------------------------------------------------------------------------------
// ---- the gate identities on unsaturated limbs ----------------------------------------------------------------------
// Every operand is unpacked once into nine 29-bit limbs and stays there: sums are plain limb additions without a
// modular correction, differences add a multiple of r first, and only the one stored result is brought back to
// canonical form.  A product x * (32 y) / 2^261 = x y / 2^256 keeps the Montgomery scaling when its right operand
// carries a factor 32: a packed operand gets it for free while it is unpacked (shift 5), a computed one by using
// (32 beta) and (32 gamma) in place of beta and gamma.  With rho = 2^261 / r = 76.5 a product of values below x r and
// y r is below (x y / rho + 1) r; the comments give these bounds in units of r.  They stay below 2^261 / r throughout,
// so every normalised value has its top limb below 2^29 and every column sum of a product fits 64 bits.
struct L9 {
    uint32_t v[9];
};
static __device__ __forceinline__ L9 l9_mul(L9 x, L9 y) {
    L9 r;
    Fr::mul_limbs(x.v, y.v, r.v);
    return r;
}
static __device__ __forceinline__ L9 LM(const L9 &x, const L9 &y) { return l9_mul(x, y); }
__device__ __forceinline__ L9 l9(const Fr &x) {      // the integer x
    L9 r;
    Fr::unpack(x, r.v, 0);
    return r;
}
__device__ __forceinline__ L9 l9x32(const Fr &x) {   // the integer 32 x
    L9 r;
    Fr::unpack(x, r.v, 5);
    return r;
}
__device__ __forceinline__ L9 l9_sum3(const L9 &x, const L9 &y, const L9 &z) {   // normalised x + y + z
    L9 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = x.v[k] + y.v[k] + z.v[k];
    u9_normalize(r.v);
    return r;
}
__device__ __forceinline__ L9 l9_small(const L9 &x, uint32_t k) {
    L9 r;
    u9_small_mul(x.v, k, r.v);
    return r;
}

// ---- widget terms ------------------------------------------------------------------------------------------------
// The range, logic, fixed-base and curve-addition terms (widget `compute_quotient_i`, each multiplied by its selector
// and its separation challenge) are evaluated by a launch of their own per widget whose selector polynomial is not
// identically zero, and summed per point in `sum`; quotient_limb_kernel adds that sum to the arithmetic and permutation
// terms.  Field addition is exact, so neither the order of the terms nor the way a term is factored shows in the
// (canonical) result.
//   W = 0 range/proverkey.rs:31-59, 1 logic/proverkey.rs:34-78, 2 ecc/scalar_mul/fixed_base/proverkey.rs:38-101,
//   3 ecc/curve_addition/proverkey.rs:32-83
// These kernels work on "scaled" values: x is carried as an integer congruent to 32 x.  The limb product of two
// scaled values, (32 x)(32 y) / 2^261 = 32 (x y / 2^256), is the scaled Montgomery product, so chains of products
// need no factor of their own; a wire value is scaled for free while it is unpacked (shift 5), constants arrive scaled
// and canonical (below r) from the host, and the last product, by the unscaled selector, returns to the plain form.
// Bounds in the comments are in units of r, rho = 2^261 / r = 76.5 as above; l9_red brings any value below 2^263 to
// below 1.01 (the quotient estimate of u9_qreduce leaves less than r + (q + 1) 2^232).

// M r with every limb below the top raised by RAISE * 2^29, borrowed from the limb above: x + (bias - y) is x - y + M r
// limb by limb, without a borrow, for y with limbs below RAISE * 2^29 and a value below (M - 1) r.  x's limbs must
// leave room: below 2^32 - (RAISE + 1) 2^29.
template <uint32_t M, uint32_t RAISE>
struct U9Bias {
    uint32_t v[9];
    constexpr U9Bias() : v{} {
        uint64_t c = 0;
        for (int k = 0; k < 9; k++) {
            c += (uint64_t)M * FrParams::umod(k);
            v[k] = k < 8 ? (uint32_t)c & U9_MASK : (uint32_t)c;
            c >>= 29;
        }
        for (int k = 0; k < 8; k++) {
            v[k] += RAISE << 29;
            v[k + 1] -= RAISE;
        }
    }
};
template <uint32_t M, uint32_t RAISE>
static __device__ __forceinline__ L9 l9_sub(const L9 &x, const L9 &y) {   // x - y + M r, not normalised
    constexpr U9Bias<M, RAISE> B;
    L9 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = x.v[k] + (B.v[k] - y.v[k]);
    return r;
}
static __device__ __forceinline__ L9 l9_add(const L9 &x, const L9 &y) {
    L9 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = x.v[k] + y.v[k];
    return r;
}
static __device__ __forceinline__ L9 l9_shl(const L9 &x, int s) {         // 2^s x, limb by limb
    L9 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = x.v[k] << s;
    return r;
}
static __device__ __forceinline__ L9 l9_norm(L9 x) {
    u9_normalize(x.v);
    return x;
}
static __device__ __forceinline__ L9 l9_red(L9 x) {                        // any limbs, below 2^263 -> normalised, below 1.01
    u9_normalize(x.v);
    u9_qreduce(x.v);
    return x;
}
// scaled x - 4 y of two wire values (range/proverkey.rs:44-56, logic/proverkey.rs:57-63): below 32 + 256 -> below 1.01
static __device__ __forceinline__ L9 w_quad_diff(const Fr &x, const Fr &y) {
    return l9_red(l9_sub<256, 4>(l9x32(x), l9_shl(l9x32(y), 2)));
}
// scaled f (f - 1)(f - 2)(f - 3) = u (u + 2) with u = f^2 - 3 f (range/proverkey.rs:92-97, logic/proverkey.rs:124-129),
// f scaled, normalised and below 1.01: below 2.19
static __device__ __forceinline__ L9 w_delta_scaled(const L9 &f, const L9 &two32) {
    const L9 u = l9_norm(l9_sub<8, 1>(LM(f, f), l9_small(f, 3)));          // 1.02 - 3.03 + 8: below 9.02
    return LM(u, l9_add(u, two32));                                        // 9.02 * 10.02 / rho + 1
}
// the widget's term, scaled and below 82 with any limbs -> times the selector and the scaled challenge, plus the terms
// summed so far, canonical
static __device__ __forceinline__ void w_finish(const L9 &s, const Fr &selector, const Fr &ch32, Fr *sum, uint32_t i, bool first) {
    L9 t = LM(LM(l9_norm(s), l9(selector)), l9(ch32));                     // below 2.08 -> below 1.03, plain form
    if (!first) u9_add(t.v, l9(ld(sum + i)).v);                            // below 2.03
    t = l9_red(t);
    Fr::cond_sub_limbs(t.v);
    st(sum + i, Fr::pack(t.v));
}

template <int W>
__global__ void __launch_bounds__(256) quotient_widget_kernel(QuotientArgs q, Fr *sum, bool first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.n8) return;
    const uint32_t iw = (i + q.rot) & (q.n8 - 1);  // "next row" on the 8n coset (quotient_poly.rs:53-59,134-136)
    const QuotientArgs::WidgetConsts &wc = q.wc[W];
    if constexpr (W == 0) {
        // delta(c - 4d) + kappa delta(b - 4c) + kappa^2 delta(a - 4b) + kappa^3 delta(d_w - 4a), by Horner's rule
        const Fr a = ld(q.a + i), b = ld(q.b + i), c = ld(q.c + i), d = ld(q.d + i);
        const L9 two32 = l9(q.two32), k = l9(wc.k32[0]);
        L9 s = w_delta_scaled(w_quad_diff(ld(q.d + iw), a), two32);        // below 2.19
        s = l9_add(LM(s, k), w_delta_scaled(w_quad_diff(a, b), two32));    // 1.03 + 2.19, limbs below 2^30
        s = l9_add(LM(s, k), w_delta_scaled(w_quad_diff(b, c), two32));    // 1.05 + 2.19
        s = l9_add(LM(s, k), w_delta_scaled(w_quad_diff(c, d), two32));    // below 3.24
        w_finish(s, ld(q.sel[Q_RANGE] + i), wc.ch32, sum, i, first);
    } else if constexpr (W == 1) {
        const L9 two32 = l9(q.two32);
        const L9 ai = w_quad_diff(ld(q.a + iw), ld(q.a + i)), bi = w_quad_diff(ld(q.b + iw), ld(q.b + i));
        const L9 di = w_quad_diff(ld(q.d + iw), ld(q.d + i));              // each below 1.01
        const L9 w = l9x32(ld(q.c + i));                                    // below 32
        L9 s = w_delta_scaled(ai, two32);                                   // c_0: below 2.19
        u9_add(s.v, LM(w_delta_scaled(bi, two32), l9(wc.k32[0])).v);        // c_1: below 1.03
        u9_add(s.v, LM(w_delta_scaled(di, two32), l9(wc.k32[1])).v);        // c_2: below 1.03
        u9_add(s.v, LM(l9_sub<4, 1>(w, LM(ai, bi)), l9(wc.k32[2])).v);      // c_3 = kappa^3 (c - a b): 36 / rho + 1 = 1.47
        // logic/proverkey.rs:132-155 with a, b, w, c = ai, bi, c, di
        const L9 ab = l9_add(ai, bi);                                       // below 2.02, limbs below 2^30
        const L9 ab_n = l9_norm(ab);
        // 4 w - 18 (a + b) + 81: below 128 + 64 + 1
        const L9 inner = l9_red(l9_add(l9_sub<64, 1>(l9_shl(w, 2), l9_small(ab, 18)), l9(q.c81_32)));
        const L9 sq = l9_small(l9_add(LM(ai, ai), LM(bi, bi)), 18);         // 18 (a^2 + b^2): below 36.8, normalised
        // w inner + 18 (a^2 + b^2) - 81 (a + b) + 83: below 1.42 + 36.8 + 256 + 1
        const L9 f1 = l9_red(l9_add(l9_add(LM(w, inner), sq), l9_sub<256, 1>(l9(q.c83_32), l9_small(ab_n, 81))));
        const L9 f2 = LM(w, f1);                                            // F: below 32 * 1.01 / rho + 1 = 1.43
        // E = 3 (a + b + c) - 2 F: below 9.1 + 8
        const L9 e = l9_sub<8, 2>(l9_small(l9_add(ab, di), 3), l9_shl(f2, 1));
        // B = q_c (9 c - 3 (a + b)): (9.1 + 8) * 32 / rho + 1 = 8.2
        const L9 bq = LM(l9_sub<8, 1>(l9_small(di, 9), l9_small(ab_n, 3)), l9x32(ld(q.sel[Q_C] + i)));
        u9_add(s.v, LM(l9_norm(l9_add(bq, e)), l9(wc.k32[3])).v);           // c_4: 25.3 / rho + 1; s below 7.1
        w_finish(s, ld(q.sel[Q_LOGIC] + i), wc.ch32, sum, i, first);
    } else if constexpr (W == 2) {
        const L9 one32 = l9(q.one32);
        const L9 a32 = l9x32(ld(q.a + i)), b32 = l9x32(ld(q.b + i));
        const Fr c = ld(q.c + i);
        // bit = d_w - 2 d (fixed_base/proverkey.rs:160-162): below 32 + 128 -> 1.01
        const L9 bit = l9_red(l9_sub<128, 2>(l9x32(ld(q.d + iw)), l9_shl(l9x32(ld(q.d + i)), 1)));
        const L9 bb = LM(bit, bit);                                         // below 1.02
        L9 s = LM(bit, l9_sub<4, 1>(bb, one32));                            // bit (bit - 1)(bit + 1) = bit (bit^2 - 1): 1.07
        // y_alpha = bit^2 (q_r - 1) + 1: 1.02 * 36 / rho + 2 = 2.48, limbs below 2^30
        const L9 ya = l9_add(LM(bb, l9_sub<4, 1>(l9x32(ld(q.sel[Q_R] + i)), one32)), one32);
        const L9 xa = LM(bit, l9x32(ld(q.sel[Q_L] + i)));                   // x_alpha = bit q_l: below 1.43
        // kappa (bit q_c - c): the difference in plain form (1.02 + 4 - c), back to the scaled one through 1024 kappa
        u9_add(s.v, LM(l9_sub<4, 1>(LM(bit, l9(ld(q.sel[Q_C] + i))), l9(c)), l9(wc.k1024)).v);   // below 1.07
        const L9 dxy = LM(LM(LM(l9x32(c), a32), b32), l9(q.edwards_d32));   // d c a b: 14.4 -> 7.03 -> 1.1
        const L9 x_lhs = LM(l9x32(ld(q.a + iw)), l9_add(dxy, one32));       // a_w (1 + d x y): 32 * 2.1 / rho + 1 = 1.88
        const L9 x_rhs = l9_add(LM(a32, ya), LM(b32, xa));                  // 2.04 + 1.6, limbs below 2^30
        u9_add(s.v, LM(l9_sub<5, 2>(x_lhs, x_rhs), l9(wc.k32[1])).v);       // kappa^2 (lhs - rhs): 6.9 / rho + 1
        const L9 y_lhs = LM(l9x32(ld(q.b + iw)), l9_sub<4, 1>(one32, dxy)); // b_w (1 - d x y): 32 * 5 / rho + 1 = 3.1
        const L9 y_rhs = l9_add(LM(b32, ya), LM(a32, xa));
        u9_add(s.v, LM(l9_sub<5, 2>(y_lhs, y_rhs), l9(wc.k32[2])).v);       // kappa^3 (lhs - rhs): 8.1 / rho + 1; s below 4.4
        w_finish(s, ld(q.sel[Q_FIXED] + i), wc.ch32, sum, i, first);
    } else {
        // x1 = a, y1 = b, x2 = c, y2 = d, x3 = a_w, y3 = b_w, x1 y2 = d_w
        const L9 one32 = l9(q.one32);
        const L9 a32 = l9x32(ld(q.a + i)), b32 = l9x32(ld(q.b + i)), c32 = l9x32(ld(q.c + i)), d32 = l9x32(ld(q.d + i));
        const L9 dw32 = l9x32(ld(q.d + iw));
        const L9 y1x2 = LM(b32, c32);                                       // products of two wire values: below 14.4
        const L9 dd = LM(LM(l9(q.edwards_d32), dw32), y1x2);                // d x1 y2 y1 x2: 1.42 -> 1.27
        L9 s = l9_sub<64, 1>(LM(a32, d32), dw32);                           // x1 y2 - d_w: below 14.4 + 64
        // kappa (d_w + y1 x2 - x3 (1 + dd)): 32 + 14.4 + 4 = 50.4 -> 1.66
        const L9 x3 = LM(l9x32(ld(q.a + iw)), l9_add(dd, one32));           // 32 * 2.27 / rho + 1 = 1.95
        u9_add(s.v, LM(l9_norm(l9_sub<4, 1>(l9_add(dw32, y1x2), x3)), l9(wc.k32[0])).v);
        // kappa^2 (y1 y2 + x1 x2 - y3 (1 - dd)): 14.4 + 14.4 + 4 -> 1.43
        const L9 y3 = LM(l9x32(ld(q.b + iw)), l9_sub<4, 1>(one32, dd));     // 32 * 5 / rho + 1 = 3.1
        u9_add(s.v, LM(l9_sub<4, 1>(l9_add(LM(b32, d32), LM(a32, c32)), y3), l9(wc.k32[1])).v);   // s below 81.5
        w_finish(s, ld(q.sel[Q_VAR] + i), wc.ch32, sum, i, first);
    }
}

__global__ void __launch_bounds__(256) quotient_limb_kernel(QuotientArgs q) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.n8) return;
    const uint32_t iw = (i + q.rot) & (q.n8 - 1);
    const Fr a = ld(q.a + i), b = ld(q.b + i), c = ld(q.c + i), d = ld(q.d + i);
    const L9 la = l9(a);

    // arithmetic/proverkey.rs:43-73: (a b q_m + a q_l + b q_r + c q_o + d q_4 + q_c) q_arith + PI
    L9 t = LM(LM(la, l9x32(b)), l9x32(ld(q.sel[Q_M] + i)));         // each product below 1.42
    u9_add(t.v, LM(la, l9x32(ld(q.sel[Q_L] + i))).v);
    u9_add(t.v, LM(l9(b), l9x32(ld(q.sel[Q_R] + i))).v);
    u9_add(t.v, LM(l9(c), l9x32(ld(q.sel[Q_O] + i))).v);
    u9_add(t.v, LM(l9(d), l9x32(ld(q.sel[Q_F] + i))).v);
    u9_add(t.v, l9(ld(q.sel[Q_C] + i)).v);                          // below 5 * 1.42 + 1 = 8.1
    u9_normalize(t.v);
    t = LM(t, l9x32(ld(q.sel[Q_ARITH] + i)));                       // below 4.4
    u9_add(t.v, l9(ld(q.pi + i)).v);                                // below 5.4, limbs below 2^30

    // permutation/proverkey.rs:39-119
    const L9 beta32 = l9x32(q.beta), beta_x32 = l9x32(q.beta_x32);
    const L9 gamma = l9(q.gamma), gamma_x32 = l9(q.gamma_x32);
    const L9 lx = l9(ld(q.linear + i));
    const L9 bx = LM(lx, beta32);                                   // beta X, below 1.42
    const L9 bx32 = LM(lx, beta_x32);                               // 32 beta X mod r, below 1.42
    L9 id = l9_sum3(la, bx, gamma);                                 // below 3.42
    id = LM(id, l9_sum3(l9x32(b), l9_small(bx32, 7), gamma_x32));   // factor below 32 + 10 + 1 -> 2.93
    id = LM(id, l9_sum3(l9x32(c), l9_small(bx32, 13), gamma_x32));  // factor below 51.5 -> 2.98
    id = LM(id, l9_sum3(l9x32(d), l9_small(bx32, 17), gamma_x32));  // factor below 57.2 -> 3.23
    id = LM(id, l9x32(ld(q.z + i)));                                // below 2.36
    L9 cp = l9_sum3(la, LM(l9(ld(q.sel[S1] + i)), beta32), gamma);  // below 3.42
    cp = LM(cp, l9_sum3(l9x32(b), LM(l9(ld(q.sel[S2] + i)), beta_x32), gamma_x32));   // factor below 34.5 -> 2.55
    cp = LM(cp, l9_sum3(l9x32(c), LM(l9(ld(q.sel[S3] + i)), beta_x32), gamma_x32));   // 2.15
    cp = LM(cp, l9_sum3(l9x32(d), LM(l9(ld(q.sel[S4] + i)), beta_x32), gamma_x32));   // 1.97
    cp = LM(cp, l9x32(ld(q.z + iw)));                               // below 1.83
    uint32_t bias[9];
    ntt_bias<0>(bias);                                              // 4 r, every limb at least that of a value below 2r
    L9 df;
#pragma unroll
    for (int k = 0; k < 9; k++) df.v[k] = id.v[k] + (bias[k] - cp.v[k]);   // id - cp + 4r: below 6.4, limbs below 2^31
    u9_add(t.v, LM(df, l9x32(q.alpha)).v);                          // + below 3.7
    // (z - 1) L1 alpha^2
    const L9 one = l9(Fr::one());
    L9 zm = l9(ld(q.z + i));
#pragma unroll
    for (int k = 0; k < 9; k++) zm.v[k] += bias[k] - one.v[k];      // z - 1 + 4r: below 5
    u9_add(t.v, LM(LM(zm, l9x32(ld(q.l1 + i))), l9x32(q.alpha_sq)).v);     // 3.1 -> + below 2.3: t below 11.4
    if (q.widget_sum) u9_add(t.v, l9(ld(q.widget_sum + i)).v);             // canonical: t below 12.4
    u9_normalize(t.v);

    // quotient_poly.rs:88-95: numerator / Z_H
    t = LM(t, l9x32(q.vh_inv[(q.first + q.world * i) & 7]));        // below 6.2
    u9_qreduce(t.v);
    Fr::cond_sub_limbs(t.v);
    st(q.out + i, Fr::pack(t.v));
}

void PolyEngine::quotient(const QuotientArgs &args) {
    ZK_REQUIRE(args.n8 >= args.rot && (args.n8 & (args.n8 - 1)) == 0, ZKGPU_ERR_INVALID_ARG, "quotient: 8n must be a power of two");
    ZK_REQUIRE(args.world >= 1 && args.world * args.rot == 8 && args.first < args.world, ZKGPU_ERR_INVALID_ARG,
               "quotient: the GPUs sharing the coset must number 1, 2, 4 or 8");
    if (!q_ev_[0]) {
        ZK_HIP(hipEventCreate(&q_ev_[0]));
        ZK_HIP(hipEventCreate(&q_ev_[1]));
    }
    QuotientArgs q = args;
    q.beta_x32 = mul_small<32>(args.beta);
    q.gamma_x32 = mul_small<32>(args.gamma);
    q.widget_sum = nullptr;
    if (args.widgets) {
        const Fr ch[4] = {args.range_ch, args.logic_ch, args.fixed_ch, args.var_ch};
        for (int w = 0; w < 4; w++) {
            const SepPowers k = sep_powers<InlineMul>(ch[w]);
            const Fr kj[4] = {k.k1, k.k2, k.k3, k.k4};
            for (int j = 0; j < 4; j++) q.wc[w].k32[j] = mul_small<32>(kj[j]);
            q.wc[w].k1024 = mul_small<1024>(k.k1);
            q.wc[w].ch32 = mul_small<32>(ch[w]);
        }
        q.one32 = mul_small<32>(Fr::one());
        q.two32 = mul_small<64>(Fr::one());
        q.c81_32 = mul_small<32 * 81>(Fr::one());
        q.c83_32 = mul_small<32 * 83>(Fr::one());
        q.edwards_d32 = mul_small<32>(edwards_d());
    }
    if (args.widgets) widget_sum_.reserve((size_t)args.n8 * sizeof(Fr));   // before the timed region: may allocate
    ZK_HIP(hipEventRecord(q_ev_[0], stream_));
    if (args.widgets) {
        Fr *sum = widget_sum_.as<Fr>();
        const dim3 grid(ceil_div(args.n8, 256)), block(256);
        bool first = true;
        if (args.widgets & 1u) {
            hipLaunchKernelGGL(quotient_widget_kernel<0>, grid, block, 0, stream_, q, sum, first);
            first = false;
        }
        if (args.widgets & 2u) {
            hipLaunchKernelGGL(quotient_widget_kernel<1>, grid, block, 0, stream_, q, sum, first);
            first = false;
        }
        if (args.widgets & 4u) {
            hipLaunchKernelGGL(quotient_widget_kernel<2>, grid, block, 0, stream_, q, sum, first);
            first = false;
        }
        if (args.widgets & 8u) {
            hipLaunchKernelGGL(quotient_widget_kernel<3>, grid, block, 0, stream_, q, sum, first);
            first = false;
        }
        q.widget_sum = sum;
    }
    hipLaunchKernelGGL(quotient_limb_kernel, dim3(ceil_div(args.n8, 256)), dim3(256), 0, stream_, q);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipEventRecord(q_ev_[1], stream_));
    q_recorded_ = true;
}

float PolyEngine::last_quotient_ms() {
    ZK_REQUIRE(q_recorded_, ZKGPU_ERR_NOT_LOADED, "no quotient has been evaluated on this context yet");
    ZK_HIP(hipEventSynchronize(q_ev_[1]));
    float ms = 0;
    ZK_HIP(hipEventElapsedTime(&ms, q_ev_[0], q_ev_[1]));
    return ms;
}

// ---- evaluation at a point ----------------------------------------------------------------------------------
static const uint32_t EVAL_BLOCKS = 64, EVAL_T = 256, EVAL_STRIDE = EVAL_BLOCKS * EVAL_T;
struct EvalJobDev {
    const Fr *coeffs;
    uint64_t len;
    Fr x, x_stride;  // x^EVAL_STRIDE
};

// Thread t of the job sums c[t + m*STRIDE] * (x^STRIDE)^m by Horner (coalesced loads), scales by x^t, and the
// block adds its 256 values.
// The jobs of a launch travel as a kernel argument: no staging copy, and no wait for one, in front of the launch.
static const int EVAL_JOBS_PER_LAUNCH = 16;
struct EvalJobs {
    EvalJobDev j[EVAL_JOBS_PER_LAUNCH];
};
__global__ void __launch_bounds__(256) eval_partial_kernel(const EvalJobs jobs, Fr *partials) {
    __shared__ uint32_t sh[2 * SCAN_T * 8];
    const EvalJobDev *job = &jobs.j[blockIdx.y];
    const uint64_t len = job->len;
    const uint32_t t = blockIdx.x * EVAL_T + threadIdx.x;
    Fr acc = Fr::zero();
    if (t < len) {
        const Fr xs = job->x_stride;
        uint64_t m = (len - 1 - t) / EVAL_STRIDE;  // highest multiple with t + m*STRIDE < len
        const Fr *p = job->coeffs + t;
        acc = ld(p + m * EVAL_STRIDE);
        while (m > 0) {
            m--;
            acc = Fr::add(Fr::mul(acc, xs), ld(p + m * EVAL_STRIDE));
        }
        if (t) acc = Fr::mul(acc, Fr::pow_u64(job->x, t));
    }
    Fr total;
    (void)block_exclusive<OpAdd>(sh, acc, &total);
    if (threadIdx.x == 0) st(partials + blockIdx.y * EVAL_BLOCKS + blockIdx.x, total);
}
__global__ void __launch_bounds__(64) eval_final_kernel(const Fr *partials, Fr *out) {
    // one wave per job: lane l holds partial l, tree-add through LDS
    __shared__ uint32_t sh[64 * 8];
    lds_st(sh, threadIdx.x, ld(partials + blockIdx.x * EVAL_BLOCKS + threadIdx.x));
    __syncthreads();
    for (uint32_t s = 32; s >= 1; s >>= 1) {
        if (threadIdx.x < s) lds_st(sh, threadIdx.x, Fr::add(lds_ld(sh, threadIdx.x), lds_ld(sh, threadIdx.x + s)));
        __syncthreads();
    }
    if (threadIdx.x == 0) st(out + blockIdx.x, lds_ld(sh, 0));
}

void PolyEngine::evaluate(const EvalJob *jobs, int n_jobs, Fr *out_dev) {
    if (n_jobs <= 0) return;
    jobs_.reserve((size_t)n_jobs * EVAL_BLOCKS * sizeof(Fr));
    Fr *partials = jobs_.as<Fr>();
    for (int first = 0; first < n_jobs; first += EVAL_JOBS_PER_LAUNCH) {
        const int count = n_jobs - first < EVAL_JOBS_PER_LAUNCH ? n_jobs - first : EVAL_JOBS_PER_LAUNCH;
        EvalJobs h;
        for (int j = 0; j < count; j++) {
            h.j[j].coeffs = jobs[first + j].coeffs;
            h.j[j].len = jobs[first + j].len;
            h.j[j].x = jobs[first + j].x;
            h.j[j].x_stride = Fr::pow_u64(jobs[first + j].x, EVAL_STRIDE);
        }
        for (int j = count; j < EVAL_JOBS_PER_LAUNCH; j++) h.j[j] = h.j[0];
        hipLaunchKernelGGL(eval_partial_kernel, dim3(EVAL_BLOCKS, count), dim3(EVAL_T), 0, stream_, h,
                           partials + (size_t)first * EVAL_BLOCKS);
    }
    hipLaunchKernelGGL(eval_final_kernel, dim3(n_jobs), dim3(64), 0, stream_, partials, out_dev);
    ZK_HIP(hipGetLastError());
}

// ---- linear combination -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lincomb_kernel(LinArgs a) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.out_len) return;
    Fr acc = i == 0 ? a.constant : Fr::zero();
    for (int k = 0; k < a.n_terms; k++)
        if (i < a.t[k].len) acc = Fr::add(acc, Fr::mul(a.t[k].scalar, ld(a.t[k].coeffs + i)));
    st(a.out + i, acc);
}
void PolyEngine::linear_combination(const LinArgs &args) {
    ZK_REQUIRE(args.n_terms >= 0 && args.n_terms <= MAX_LIN_TERMS, ZKGPU_ERR_INVALID_ARG, "linear_combination: too many terms");
    if (args.out_len == 0) return;
    hipLaunchKernelGGL(lincomb_kernel, dim3(ceil_div(args.out_len, 256)), dim3(256), 0, stream_, args);
    ZK_HIP(hipGetLastError());
}

// ---- synthetic division -------------------------------------------------------------------------------------
// Synthetic division as a suffix Horner scan: h_i = c_i + z h_(i+1) (h_len = 0) and q_i = h_(i+1).  A workgroup takes
// RUF_B consecutive coefficients: every thread runs Horner over its RUF_E coefficients, the threads' results are
// combined by a doubling scan whose step d uses the constant z^(RUF_E d), and the workgroup leaves its values as if
// nothing lay above it, plus its own h at its lowest index.  One workgroup then runs the same scan over those
// aggregates (step constants z^(RUF_B ...)), and a last pass adds z^(distance to the chunk's top) times the carry that
// enters each chunk.  Two passes over the data instead of the five of scale / scan / unscale, no inverse of z, and
// z = 0 needs no special case.
static const uint32_t RUF_T = 256, RUF_E = 4, RUF_B = RUF_T * RUF_E;
struct RuffiniConsts {
    Fr z;
    Fr low[RUF_E + 1];   // z^k, k <= RUF_E
    Fr step[8];          // z^(RUF_E 2^k): the doubling scan over the 256 threads of a chunk
    Fr run;              // z^RUF_B: the serial part of the scan over the chunks
    Fr top[8];           // z^(RUF_B per_thread 2^k): its doubling part
    uint32_t per_thread; // chunks each thread of that scan takes
};

// v_t <- sum_{u >= t} a_u w^(u - t) over the 256 threads of a workgroup, step[k] = w^(2^k); returns the sum of thread
// t + 1 (what enters thread t from above) and leaves thread 0's sum in *whole
__device__ __forceinline__ Fr block_suffix_horner(uint32_t *sh /* 2 * 256 * 8 words */, const Fr &mine, const Fr *step, Fr *whole) {
    const uint32_t tid = threadIdx.x;
    uint32_t *cur = sh, *nxt = sh + RUF_T * 8;
    lds_st(cur, tid, mine);
    __syncthreads();
#pragma unroll 1
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t d = 1u << k;
        Fr v = lds_ld(cur, tid);
        if (tid + d < RUF_T) v = Fr::add(v, Fr::mul(step[k], lds_ld(cur, tid + d)));
        lds_st(nxt, tid, v);
        __syncthreads();
        uint32_t *t = cur;
        cur = nxt;
        nxt = t;
    }
    *whole = lds_ld(cur, 0);
    Fr above = tid + 1 < RUF_T ? lds_ld(cur, tid + 1) : Fr::zero();
    __syncthreads();
    return above;
}

__global__ void __launch_bounds__(256) ruffini_local_kernel(const Fr *c, uint64_t len, RuffiniConsts k, Fr *out, Fr *aggregates) {
    __shared__ uint32_t sh[2 * RUF_T * 8];
    const uint64_t i0 = (uint64_t)blockIdx.x * RUF_B + (uint64_t)threadIdx.x * RUF_E;
    Fr h[RUF_E];
    Fr run = Fr::zero();
#pragma unroll
    for (int e = RUF_E - 1; e >= 0; e--) {
        const uint64_t i = i0 + e;
        run = Fr::mul(run, k.z);
        if (i < len) run = Fr::add(run, ld(c + i));
        h[e] = run;
    }
    Fr whole;
    const Fr above = block_suffix_horner(sh, run, k.step, &whole);
#pragma unroll
    for (int e = 0; e < (int)RUF_E; e++) {
        const uint64_t i = i0 + e;
        if (i >= 1 && i < len) st(out + (i - 1), Fr::add(h[e], Fr::mul(k.low[RUF_E - e], above)));   // q_(i-1) = h_i
    }
    if (threadIdx.x == 0) st(aggregates + blockIdx.x, whole);
}

// aggregates[b] <- what enters chunk b from above: H_(b+1) with H_b = A_b + z^RUF_B H_(b+1)
__global__ void __launch_bounds__(256) ruffini_carry_kernel(Fr *aggregates, uint32_t n_chunks, RuffiniConsts k) {
    __shared__ uint32_t sh[2 * RUF_T * 8];
    const uint32_t first = threadIdx.x * k.per_thread;
    Fr run = Fr::zero();
    for (int e = (int)k.per_thread - 1; e >= 0; e--) {
        const uint32_t b = first + e;
        run = Fr::mul(run, k.run);
        if (b < n_chunks) run = Fr::add(run, ld(aggregates + b));
    }
    Fr whole;
    Fr carry = block_suffix_horner(sh, run, k.top, &whole);   // enters this thread's topmost chunk
    for (int e = (int)k.per_thread - 1; e >= 0; e--) {
        const uint32_t b = first + e;
        if (b >= n_chunks) continue;
        const Fr mine = ld(aggregates + b);
        st(aggregates + b, carry);
        carry = Fr::add(mine, Fr::mul(k.run, carry));
    }
}

// powers[e] = z^e, e <= RUF_B
__global__ void __launch_bounds__(256) ruffini_powers_kernel(Fr z, Fr *powers) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= RUF_B) st(powers + e, Fr::pow_u64(z, e));
}

__global__ void __launch_bounds__(256) ruffini_apply_kernel(Fr *out, uint64_t len, const Fr *carries, const Fr *powers) {
    const Fr carry = ld(carries + blockIdx.x);
    if (carry.is_zero()) return;   // the topmost chunk, for one
    const uint32_t t0 = threadIdx.x * RUF_E;
    const uint64_t i0 = (uint64_t)blockIdx.x * RUF_B + t0;
#pragma unroll
    for (uint32_t e = 0; e < RUF_E; e++) {
        const uint64_t i = i0 + e;
        if (i >= 1 && i < len) {
            Fr *p = out + (i - 1);
            st(p, Fr::add(ld(p), Fr::mul(ld(powers + (RUF_B - t0 - e)), carry)));   // h_i += z^(chunk top - i) * carry
        }
    }
}

void PolyEngine::ruffini(const Fr *coeffs, uint64_t len, const Fr &root, Fr *out) {
    if (len <= 1) return;  // the quotient is empty
    const uint32_t n_chunks = ceil_div(len, RUF_B);
    RuffiniConsts k;
    k.z = root;
    k.low[0] = Fr::one();
    for (uint32_t e = 1; e <= RUF_E; e++) k.low[e] = Fr::mul(k.low[e - 1], root);
    k.step[0] = k.low[RUF_E];
    for (int j = 1; j < 8; j++) k.step[j] = Fr::sqr(k.step[j - 1]);
    k.run = Fr::sqr(k.step[7]);   // z^(RUF_E * 256)
    k.per_thread = ceil_div(n_chunks, RUF_T);
    k.top[0] = Fr::pow_u64(k.run, k.per_thread);
    for (int j = 1; j < 8; j++) k.top[j] = Fr::sqr(k.top[j - 1]);
    s3_.reserve(((size_t)n_chunks + RUF_B + 1) * sizeof(Fr));
    Fr *aggregates = s3_.as<Fr>(), *powers = aggregates + n_chunks;
    hipLaunchKernelGGL(ruffini_local_kernel, dim3(n_chunks), dim3(RUF_T), 0, stream_, coeffs, len, k, out, aggregates);
    if (n_chunks > 1) {
        hipLaunchKernelGGL(ruffini_powers_kernel, dim3(ceil_div(RUF_B + 1, 256)), dim3(256), 0, stream_, root, powers);
        hipLaunchKernelGGL(ruffini_carry_kernel, dim3(1), dim3(RUF_T), 0, stream_, aggregates, n_chunks, k);
        hipLaunchKernelGGL(ruffini_apply_kernel, dim3(n_chunks), dim3(RUF_T), 0, stream_, out, len, aggregates, powers);
    }
    ZK_HIP(hipGetLastError());
}

// ---- single-coefficient edits -------------------------------------------------------------------------------
struct PatchArgs {
    uint64_t idx[8];
    Fr val[8];
    uint32_t sub[8];
    int count;
};
__global__ void patch_kernel(Fr *p, PatchArgs a) {
    int k = threadIdx.x;
    if (k >= a.count) return;
    Fr v = a.val[k];
    if (a.sub[k]) v = Fr::sub(ld(p + a.idx[k]), v);
    st(p + a.idx[k], v);
}
void PolyEngine::patch(Fr *p, const uint64_t *idx, const Fr *vals, const bool *sub, int count) {
    ZK_REQUIRE(count >= 0 && count <= 8, ZKGPU_ERR_INVALID_ARG, "patch: at most 8 edits");
    if (!count) return;
    PatchArgs a;
    a.count = count;
    for (int k = 0; k < count; k++) {
        a.idx[k] = idx[k];
        a.val[k] = vals[k];
        a.sub[k] = sub[k] ? 1 : 0;
    }
    hipLaunchKernelGGL(patch_kernel, dim3(1), dim3(64), 0, stream_, p, a);
    ZK_HIP(hipGetLastError());
}

// ---- the quotient's four pieces -------------------------------------------------------------------------------
struct SplitArgs {
    Fr *t, *low, *mid, *high;
    uint64_t n;
    Fr b[3];
};
__global__ void __launch_bounds__(256) split_quotient_kernel(SplitArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = a.n;
    if (i < n) {
        st(a.low + i, ld(a.t + i));
    } else if (i < 2 * n) {
        Fr v = ld(a.t + i);
        if (i == n) v = Fr::sub(v, a.b[0]);
        st(a.mid + (i - n), v);
    } else if (i < 3 * n) {
        Fr v = ld(a.t + i);
        if (i == 2 * n) v = Fr::sub(v, a.b[1]);
        st(a.high + (i - 2 * n), v);
    } else if (i == 3 * n) {
        st(a.low + n, a.b[0]);
        st(a.mid + n, a.b[1]);
        st(a.high + n, a.b[2]);
        st(a.t + 3 * n, Fr::sub(ld(a.t + 3 * n), a.b[2]));
    }
}
void PolyEngine::split_quotient(Fr *t, uint64_t n, Fr *low, Fr *mid, Fr *high, const Fr b[3]) {
    SplitArgs a;
    a.t = t;
    a.low = low;
    a.mid = mid;
    a.high = high;
    a.n = n;
    for (int k = 0; k < 3; k++) a.b[k] = b[k];
    hipLaunchKernelGGL(split_quotient_kernel, dim3(ceil_div(3 * n + 1, 256)), dim3(256), 0, stream_, a);
    ZK_HIP(hipGetLastError());
}

// ---- scatter -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) scatter_kernel(Fr *dst, const uint64_t *idx, const Fr *vals, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) st(dst + idx[i], ld(vals + i));
}
void PolyEngine::zero_and_scatter(Fr *dst, uint64_t len, const uint64_t *idx_host, const Fr *vals_host, uint32_t count) {
    ZK_HIP(hipMemsetAsync(dst, 0, len * sizeof(Fr), stream_));
    if (!count) return;
    if (count <= 8) {   // the few values travel with the launch: no copies, no wait
        bool set[8] = {false, false, false, false, false, false, false, false};
        patch(dst, idx_host, vals_host, set, (int)count);
        return;
    }
    s3_.reserve((size_t)count * (sizeof(Fr) + 8));
    Fr *vals = s3_.as<Fr>();
    uint64_t *idx = reinterpret_cast<uint64_t *>(vals + count);
    ZK_HIP(hipMemcpyAsync(vals, vals_host, (size_t)count * sizeof(Fr), hipMemcpyHostToDevice, stream_));
    ZK_HIP(hipMemcpyAsync(idx, idx_host, (size_t)count * 8, hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(scatter_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, stream_, dst, idx, vals, count);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream_));   // the host arrays are the caller's temporaries
}

// ---- canonical bytes <-> Montgomery form ----------------------------------------------------------------------
__global__ void __launch_bounds__(256) from_canonical_kernel(const Fr *raw, uint64_t count, Fr *out, uint64_t pad_to,
                                                             unsigned int *bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pad_to) return;
    if (i >= count) {
        st(out + i, Fr::zero());
        return;
    }
    const Fr a = ld(raw + i);
    if (!a.below_modulus()) atomicOr(bad, 1u);
    st(out + i, Fr::to_mont(a));
}
__global__ void __launch_bounds__(256) to_canonical_kernel(const Fr *in, uint64_t count, Fr *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) st(out + i, Fr::from_mont(ld(in + i)));
}
bool PolyEngine::from_canonical(const Fr *raw, uint64_t count, Fr *out, uint64_t pad_to) {
    if (pad_to < count) pad_to = count;
    if (!pad_to) return true;
    s3_.reserve(64);
    unsigned int *bad = s3_.as<unsigned int>();
    ZK_HIP(hipMemsetAsync(bad, 0, 4, stream_));
    hipLaunchKernelGGL(from_canonical_kernel, dim3(ceil_div(pad_to, 256)), dim3(256), 0, stream_, raw, count, out, pad_to, bad);
    ZK_HIP(hipGetLastError());
    unsigned int flag = 0;
    ZK_HIP(hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, stream_));
    ZK_HIP(hipStreamSynchronize(stream_));
    return flag == 0;
}
void PolyEngine::to_canonical(const Fr *in, uint64_t count, Fr *out_raw) {
    if (!count) return;
    hipLaunchKernelGGL(to_canonical_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, stream_, in, count, out_raw);
    ZK_HIP(hipGetLastError());
}

// ---- degree -------------------------------------------------------------------------------------------------
// out[j] = 1 + index of the highest non-zero coefficient of vector j (0 for the zero polynomial); the vectors of a
// commitment batch share one launch (blockIdx.y = j)
struct DegreeJobs {
    const Fr *coeffs[8];
    uint64_t len[8];
};
__global__ void __launch_bounds__(1024) degree_kernel(const DegreeJobs jobs, unsigned long long *out_all) {
    __shared__ unsigned long long wave_max[16];
    const Fr *coeffs = jobs.coeffs[blockIdx.y];
    const uint64_t len = jobs.len[blockIdx.y];
    unsigned long long *out = out_all + blockIdx.y;
    const uint64_t n_blocks = (len + 4095) / 4096;
    if (blockIdx.x >= n_blocks) return;   // a shorter vector of the batch
    // blocks walk the vector from the top, so the block holding the leading coefficient is among the first to run
    const uint64_t block_from_top = n_blocks - 1 - blockIdx.x;
    const uint64_t base = block_from_top * 4096 + threadIdx.x;
    // a block above has found a coefficient already: nothing down here can raise the result (one thread reads, so
    // that the whole workgroup takes the same way)
    __shared__ unsigned long long seen;
    if (threadIdx.x == 0) seen = *reinterpret_cast<volatile unsigned long long *>(out);
    __syncthreads();
    if (seen > block_from_top * 4096 + 4096) return;
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t i = base + (uint64_t)k * 1024;
        if (i < len && !ld(coeffs + i).is_zero()) mine = i + 1;   // later k = higher index
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(mine, d);
        mine = o > mine ? o : mine;
    }
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < 16; w++) mine = wave_max[w] > mine ? wave_max[w] : mine;
        // a stale read only costs an extra atomic
        if (mine > *reinterpret_cast<volatile unsigned long long *>(out)) atomicMax(out, mine);
    }
}
void PolyEngine::lengths_enqueue(uint32_t m, const Fr *const *coeffs, const uint64_t *lens, unsigned long long *host_out) {
    ZK_REQUIRE(m <= 8, ZKGPU_ERR_INVALID_ARG, "lengths_without_leading_zeros: at most 8 vectors");
    jobs_.reserve(64);
    unsigned long long *d = jobs_.as<unsigned long long>();
    ZK_HIP(hipMemsetAsync(d, 0, 8 * m, stream_));
    DegreeJobs jobs;
    uint64_t longest = 0;
    for (uint32_t j = 0; j < 8; j++) {
        jobs.coeffs[j] = j < m ? coeffs[j] : nullptr;
        jobs.len[j] = j < m ? lens[j] : 0;
        if (jobs.len[j] > longest) longest = jobs.len[j];
    }
    if (longest) hipLaunchKernelGGL(degree_kernel, dim3(ceil_div(longest, 4096), m), dim3(1024), 0, stream_, jobs, d);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(host_out, d, 8 * m, hipMemcpyDeviceToHost, stream_));
}
void PolyEngine::lengths_without_leading_zeros(uint32_t m, const Fr *const *coeffs, const uint64_t *lens, uint64_t *out) {
    unsigned long long h[8];
    lengths_enqueue(m, coeffs, lens, h);
    ZK_HIP(hipStreamSynchronize(stream_));
    for (uint32_t j = 0; j < m; j++) out[j] = (uint64_t)h[j];
}
uint64_t PolyEngine::length_without_leading_zeros(const Fr *coeffs, uint64_t len) {
    uint64_t out = 0;
    lengths_without_leading_zeros(1, &coeffs, &len, &out);
    return out;
}

PolyEngine::~PolyEngine() {
    for (auto &e : q_ev_)
        if (e) (void)hipEventDestroy(e);
    s0_.release();
    s1_.release();
    s2_.release();
    s3_.release();
    jobs_.release();
    widget_sum_.release();
}

}  // namespace zk
