// TEST INFRASTRUCTURE ONLY (see util.hpp).
// Fiat-Shamir transcript of plonk/src/transcript.rs:30-66 over Merlin.
//
// Merlin is a third-party crate that is NOT under /root/reference: `merlin 3.0.0`
// (Cargo.lock:1763-1764) over `keccak 0.1.5` (Cargo.lock:1662-1663). This file restates its
// published construction: STROBE-128/1600 (rate 166 bytes, security 128) driving
// Keccak-f[1600], with Merlin's framing
//   append_message(label, m) = meta-AD(label) || meta-AD(LE32(len m)) ; AD(m)
//   challenge_bytes(label, n) = meta-AD(label) || meta-AD(LE32(n))    ; PRF(n)
// None of the reference's tests pin transcript output, so challenge derivation is
// "parity unpinned" against the reference; tests check this code against the two
// conformance strings of merlin's own test-suite as recalled by the author of this file.
#pragma once
#include "fr.hpp"
#include "g1.hpp"
#include <string>
#include <vector>

namespace oracle {

static inline uint64_t rotl64(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

// Keccak-f[1600], 24 rounds, on 25 little-endian lanes.
static inline void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = PIL[i];
            uint64_t b = st[j];
            st[j] = rotl64(t, ROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}

class Strobe128 {
    static const int R = 166;
    enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };
    uint8_t st[200];
    uint8_t pos, pos_begin, cur_flags;

    void permute() {
        uint64_t lanes[25];
        for (int i = 0; i < 25; i++) {
            lanes[i] = 0;
            for (int b = 0; b < 8; b++) lanes[i] |= (uint64_t)st[8 * i + b] << (8 * b);
        }
        keccak_f1600(lanes);
        for (int i = 0; i < 25; i++)
            for (int b = 0; b < 8; b++) st[8 * i + b] = (uint8_t)(lanes[i] >> (8 * b));
    }
    void run_f() {
        st[pos] ^= pos_begin;
        st[pos + 1] ^= 0x04;
        st[R + 1] ^= 0x80;
        permute();
        pos = 0;
        pos_begin = 0;
    }
    void absorb(const uint8_t *d, size_t n) {
        for (size_t i = 0; i < n; i++) {
            st[pos] ^= d[i];
            if (++pos == R) run_f();
        }
    }
    void overwrite(const uint8_t *d, size_t n) {
        for (size_t i = 0; i < n; i++) {
            st[pos] = d[i];
            if (++pos == R) run_f();
        }
    }
    void squeeze(uint8_t *d, size_t n) {
        for (size_t i = 0; i < n; i++) {
            d[i] = st[pos];
            st[pos] = 0;
            if (++pos == R) run_f();
        }
    }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;  // continuation of the current operation
        uint8_t old_begin = pos_begin;
        pos_begin = pos + 1;
        cur_flags = flags;
        uint8_t hdr[2] = {old_begin, flags};
        absorb(hdr, 2);
        bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
        if (force_f && pos != 0) run_f();
    }

   public:
    explicit Strobe128(const std::string &protocol_label) {
        for (int i = 0; i < 200; i++) st[i] = 0;
        const uint8_t head[6] = {1, R + 2, 1, 0, 1, 96};
        for (int i = 0; i < 6; i++) st[i] = head[i];
        const char *ver = "STROBEv1.0.2";
        for (int i = 0; i < 12; i++) st[6 + i] = (uint8_t)ver[i];
        permute();
        pos = 0;
        pos_begin = 0;
        cur_flags = 0;
        meta_ad((const uint8_t *)protocol_label.data(), protocol_label.size(), false);
    }
    void meta_ad(const uint8_t *d, size_t n, bool more) {
        begin_op(FLAG_M | FLAG_A, more);
        absorb(d, n);
    }
    void ad(const uint8_t *d, size_t n, bool more) {
        begin_op(FLAG_A, more);
        absorb(d, n);
    }
    void prf(uint8_t *d, size_t n, bool more) {
        begin_op(FLAG_I | FLAG_A | FLAG_C, more);
        squeeze(d, n);
    }
    void key(const uint8_t *d, size_t n, bool more) {
        begin_op(FLAG_A | FLAG_C, more);
        overwrite(d, n);
    }
};

class Transcript {
    Strobe128 strobe;

   public:
    explicit Transcript(const std::string &label) : strobe("Merlin v1.0") {
        append_message("dom-sep", (const uint8_t *)label.data(), label.size());
    }
    void append_message(const std::string &label, const uint8_t *m, size_t n) {
        uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
        strobe.meta_ad((const uint8_t *)label.data(), label.size(), false);
        strobe.meta_ad(len, 4, true);
        strobe.ad(m, n, false);
    }
    void append_u64(const std::string &label, uint64_t x) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
        append_message(label, b, 8);
    }
    void challenge_bytes(const std::string &label, uint8_t *out, size_t n) {
        uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
        strobe.meta_ad((const uint8_t *)label.data(), label.size(), false);
        strobe.meta_ad(len, 4, true);
        strobe.prf(out, n, false);
    }
    // plonk/src/transcript.rs:31-49
    void append_commitment(const std::string &label, const G1Affine &c) {
        uint8_t b[48];
        g1a_to_compressed(c, b);
        append_message(label, b, 48);
    }
    void append_scalar(const std::string &label, const Fr &s) {
        uint8_t b[32];
        fr_to_bytes(s, b);
        append_message(label, b, 32);
    }
    Fr challenge_scalar(const std::string &label) {
        uint8_t b[64];
        challenge_bytes(label, b, 64);
        return fr_from_bytes_wide(b);
    }
    void circuit_domain_sep(uint64_t n) {
        append_message("dom-sep", (const uint8_t *)"circuit_size", 12);
        append_u64("n", n);
    }
};

}  // namespace oracle
