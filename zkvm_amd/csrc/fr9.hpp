// Fr arithmetic on nine unsaturated limbs of 29 bits with lazy reduction: the working form of the transforms' tiles
// and of the quotient kernel.  One value = uint32_t[9]; "normalised" = limbs 0..7 below 2^29, the rest in limb 8.
#pragma once
#include "ff.hpp"

namespace zk {

// ---- tile arithmetic on unsaturated limbs ----------------------------------------------------------------------
// Inside a pass an element is 9 limbs of 29 bits, only loosely reduced; one LDS plane per limb (conflict-free 4-byte
// accesses).  The stage tables hold 32 w (i.e. w 2^261 in Montgomery-256 terms), so that the limb product
// d * (32 w) / 2^261 = d w / 2^256 keeps the data in the reference's Montgomery scaling, and because the table entry
// is reduced the product contracts: below d / 2^6 + r.  The sum path doubles per stage instead, so values are pulled
// back below 2r every fourth stage (on read) with one quotient estimate from the top limb:
//   stage k of a pass: inputs below 2^(k mod 4) * 2r;  s = x + y;  d = (x - y + M r) * w with M = 2^(k mod 4) * 4.
static constexpr uint32_t U9_MASK = 0x1fffffffu;
template <int KM>
__device__ __forceinline__ void ntt_bias(uint32_t *b) {   // M r with M = 4 * 2^KM, limbs raised by 2^29 below the top
    constexpr uint32_t T[4][9] = {
        {0x20000004u, 0x3fffffdfu, 0x3e5bfefeu, 0x2d2017feu, 0x360154eeu, 0x30101342u, 0x3483339cu, 0x2994cebdu, 0x01cfb69cu},
        {0x20000008u, 0x3fffffbfu, 0x3cb7fdfeu, 0x3a402ffeu, 0x2c02a9ddu, 0x20202686u, 0x2906673au, 0x33299d7cu, 0x039f6d39u},
        {0x20000010u, 0x3fffff7fu, 0x396ffbfeu, 0x34805ffeu, 0x380553bcu, 0x20404d0du, 0x320cce75u, 0x26533af9u, 0x073eda74u},
        {0x20000020u, 0x3ffffeffu, 0x32dff7feu, 0x2900bffeu, 0x300aa77au, 0x20809a1cu, 0x24199cebu, 0x2ca675f4u, 0x0e7db4e9u}};
#pragma unroll
    for (int i = 0; i < 9; i++) b[i] = T[KM][i];
}
// The same multiples of r with the limbs below the top raised by 2^30: for a subtrahend that is the limb-wise sum of
// two normalised values and has not been through a carry sweep (limbs up to 2^30 - 2)
template <int KM>
__device__ __forceinline__ void ntt_bias_wide(uint32_t *b) {
    constexpr uint32_t T[4][9] = {
        {0x40000004u, 0x5fffffdeu, 0x5e5bfefdu, 0x4d2017fdu, 0x560154edu, 0x50101341u, 0x5483339bu, 0x4994cebcu, 0x01cfb69bu},
        {0x40000008u, 0x5fffffbeu, 0x5cb7fdfdu, 0x5a402ffdu, 0x4c02a9dcu, 0x40202685u, 0x49066739u, 0x53299d7bu, 0x039f6d38u},
        {0x40000010u, 0x5fffff7eu, 0x596ffbfdu, 0x54805ffdu, 0x580553bbu, 0x40404d0cu, 0x520cce74u, 0x46533af8u, 0x073eda73u},
        {0x40000020u, 0x5ffffefeu, 0x52dff7fdu, 0x4900bffdu, 0x500aa779u, 0x40809a1bu, 0x44199ceau, 0x4ca675f3u, 0x0e7db4e8u}};
#pragma unroll
    for (int i = 0; i < 9; i++) b[i] = T[KM][i];
}
__device__ __forceinline__ void u9_load(const uint32_t *lds, uint32_t E, uint32_t e, uint32_t *x) {
#pragma unroll
    for (int k = 0; k < 9; k++) x[k] = lds[k * E + e];
}
__device__ __forceinline__ void u9_store(uint32_t *lds, uint32_t E, uint32_t e, const uint32_t *x) {
#pragma unroll
    for (int k = 0; k < 9; k++) lds[k * E + e] = x[k];
}
__device__ __forceinline__ void u9_normalize(uint32_t *x) {   // carries into the top limb; value unchanged
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t t = x[k] + c;
        x[k] = t & U9_MASK;
        c = t >> 29;
    }
    x[8] += c;
}
// normalised x below 64r -> normalised, below r + 66 * 2^232 (so below 2r): subtracts q r with
// q = floor(top limb / (floor(r / 2^232) + 1)) <= x / r
__device__ __forceinline__ void u9_qreduce(uint32_t *x) {
    const uint32_t q = x[8] / 7597480u;
    int64_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c += (int64_t)x[k] - (int64_t)((uint64_t)q * FrParams::umod(k));
        x[k] = (uint32_t)c & U9_MASK;
        c >>= 29;
    }
    x[8] = (uint32_t)(c + (int64_t)x[8] - (int64_t)((uint64_t)q * FrParams::umod(8)));
}

// x * k for a small k (k * 2^29 must fit 64 bits with room): normalised in, normalised out
__device__ __forceinline__ void u9_small_mul(const uint32_t *x, uint32_t k, uint32_t *out) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)x[i] * k;
        out[i] = (uint32_t)c & U9_MASK;
        c >>= 29;
    }
    out[8] = x[8] * k + (uint32_t)c;
}
__device__ __forceinline__ void u9_add(uint32_t *x, const uint32_t *y) {
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] += y[i];
}

}  // namespace zk
