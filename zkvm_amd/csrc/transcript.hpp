// Host-side Fiat-Shamir transcript of the prover (plonk/src/transcript.rs:30-66), which the reference builds
// on the third-party `merlin 3.0.0` crate (Cargo.lock:1763-1764; not part of the reference tree).  Merlin is
// STROBE-128 over Keccak-f[1600]: every message is framed as meta-AD(label || LE32(len)) followed by AD(data);
// challenges are PRF output.  The transcript stays on the host: it hashes ~2 KB per proof.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace zk {

class Keccak1600 {
   public:
    uint64_t lane[25];
    Keccak1600() { memset(lane, 0, sizeof lane); }
    uint8_t get(int i) const { return (uint8_t)(lane[i >> 3] >> (8 * (i & 7))); }
    void xor_byte(int i, uint8_t b) { lane[i >> 3] ^= (uint64_t)b << (8 * (i & 7)); }
    void set_byte(int i, uint8_t b) {
        lane[i >> 3] &= ~((uint64_t)0xff << (8 * (i & 7)));
        lane[i >> 3] |= (uint64_t)b << (8 * (i & 7));
    }
    void permute() {
        static const uint64_t RC[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
            0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
            0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
            0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
            0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
            0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        // rotation offsets r[x][y] and the pi permutation expressed on the flat index x + 5y
        static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
        for (int rnd = 0; rnd < 24; rnd++) {
            uint64_t C[5], D[5], B[25];
            for (int x = 0; x < 5; x++) C[x] = lane[x] ^ lane[x + 5] ^ lane[x + 10] ^ lane[x + 15] ^ lane[x + 20];
            for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rot(C[(x + 1) % 5], 1);
            for (int i = 0; i < 25; i++) lane[i] ^= D[i % 5];
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rot(lane[x + 5 * y], RHO[x + 5 * y]);
            for (int y = 0; y < 5; y++)
                for (int x = 0; x < 5; x++) lane[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
            lane[0] ^= RC[rnd];
        }
    }

   private:
    static uint64_t rot(uint64_t v, int s) { return s ? (v << s) | (v >> (64 - s)) : v; }
};

class MerlinTranscript {
   public:
    explicit MerlinTranscript(const std::string &label) {
        static const uint8_t init[18] = {1, RATE + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
        for (int i = 0; i < 18; i++) k_.set_byte(i, init[i]);
        k_.permute();
        meta_ad("Merlin v1.0", 11, false);
        append_message("dom-sep", label.data(), label.size());
    }
    void append_message(const char *label, const void *msg, size_t len) {
        frame(label, len);
        begin(FLAG_A, false);
        absorb(static_cast<const uint8_t *>(msg), len);
    }
    void append_u64(const char *label, uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
        append_message(label, b, 8);
    }
    void challenge_bytes(const char *label, uint8_t *out, size_t len) {
        frame(label, len);
        begin(FLAG_I | FLAG_A | FLAG_C, false);
        for (size_t i = 0; i < len; i++) {
            out[i] = k_.get(pos_);
            k_.set_byte(pos_, 0);
            advance();
        }
    }

   private:
    enum { RATE = 166, FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };
    Keccak1600 k_;
    int pos_ = 0, pos_begin_ = 0;

    void frame(const char *label, size_t len) {
        uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        meta_ad(label, strlen(label), false);
        meta_ad(reinterpret_cast<const char *>(le), 4, true);
    }
    void meta_ad(const char *data, size_t len, bool more) {
        begin(FLAG_M | FLAG_A, more);
        absorb(reinterpret_cast<const uint8_t *>(data), len);
    }
    void run_f() {
        k_.xor_byte(pos_, (uint8_t)pos_begin_);
        k_.xor_byte(pos_ + 1, 0x04);
        k_.xor_byte(RATE + 1, 0x80);
        k_.permute();
        pos_ = 0;
        pos_begin_ = 0;
    }
    void advance() {
        if (++pos_ == RATE) run_f();
    }
    void absorb(const uint8_t *d, size_t len) {
        for (size_t i = 0; i < len; i++) {
            k_.xor_byte(pos_, d[i]);
            advance();
        }
    }
    void begin(uint8_t flags, bool more) {
        if (more) return;
        uint8_t hdr[2] = {(uint8_t)pos_begin_, flags};
        pos_begin_ = pos_ + 1;
        absorb(hdr, 2);
        if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
    }
};

}  // namespace zk
