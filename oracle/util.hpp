// TEST INFRASTRUCTURE ONLY — CPU restatement of the reference algorithms.
// Nothing under oracle/ is linked into, imported by, or called from the product
// library (libzkgpu.so); only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, as the checker / reported baseline.
//
// Follows coset-bls12_381/src/util.rs:3-20 (adc / sbb / mac on u128).
#pragma once
#include <cstdint>
#include <cstddef>

namespace oracle {

typedef unsigned __int128 u128;

// util.rs:3-7  — a + b + carry -> (lo, carry)
static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t carry, uint64_t &carry_out) {
    u128 ret = (u128)a + (u128)b + (u128)carry;
    carry_out = (uint64_t)(ret >> 64);
    return (uint64_t)ret;
}

// util.rs:9-13 — a - (b + (borrow >> 63)) -> (lo, borrow) ; borrow is 0 or all-ones
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t borrow, uint64_t &borrow_out) {
    u128 ret = (u128)a - ((u128)b + (u128)(borrow >> 63));
    borrow_out = (uint64_t)(ret >> 64);
    return (uint64_t)ret;
}

// util.rs:15-20 — a + b*c + carry -> (lo, carry)
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t carry, uint64_t &carry_out) {
    u128 ret = (u128)a + (u128)b * (u128)c + (u128)carry;
    carry_out = (uint64_t)(ret >> 64);
    return (uint64_t)ret;
}

}  // namespace oracle
