/* A plain C program against include/zkgpu.h and libzkgpu.so only (no Python, no torch): the way a cgo / Rust FFI
 * caller sees the library.  Checks on a GPU: an NTT round trip, a coset round trip, commit([0, 1]) = tau * G for a
 * generated SRS, the commit error codes, and a serialized SRS round trip.  Exit status 0 = all checks passed. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkgpu.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != 0) {                                                                               \
            fprintf(stderr, "%s failed with %d: %s\n", #call, rc_, ctx ? zkgpu_last_error(ctx) : ""); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

/* BLS12-381 G1 generator, x | y as Montgomery limbs (coset-bls12_381/src/g1.rs:218-238) */
static const uint64_t G1_GEN[12] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull,
                                    0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull,
                                    0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
/* Scalar one and an arbitrary canonical Montgomery residue used as tau */
static const uint64_t FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const uint64_t TAU[4] = {0x0123456789abcdefull, 0xfedcba9876543210ull, 0x0f1e2d3c4b5a6978ull, 0x1122334455667788ull};

int main(void) {
    zkgpu_ctx *ctx = NULL;
    CHECK(zkgpu_init(&ctx, 0));
    const uint32_t log_n = 12;
    const size_t n = (size_t)1 << log_n;
    uint64_t *x = malloc(n * 32), *y = malloc(n * 32);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < 4 * n; i++) {           /* xorshift; top limb kept below the modulus' top limb */
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x[i] = (i % 4 == 3) ? (s >> 2) % 0x73eda753299d7d48ull : s;
    }
    memcpy(y, x, n * 32);
    CHECK(zkgpu_ntt(ctx, y, n, log_n, ZKGPU_FFT));
    if (memcmp(x, y, n * 32) == 0) { fprintf(stderr, "fft left the data unchanged\n"); return 1; }
    CHECK(zkgpu_ntt(ctx, y, n, log_n, ZKGPU_IFFT));
    if (memcmp(x, y, n * 32) != 0) { fprintf(stderr, "fft/ifft round trip differs\n"); return 1; }
    CHECK(zkgpu_ntt(ctx, y, n, log_n, ZKGPU_COSET_FFT));
    CHECK(zkgpu_ntt(ctx, y, n, log_n, ZKGPU_COSET_IFFT));
    if (memcmp(x, y, n * 32) != 0) { fprintf(stderr, "coset round trip differs\n"); return 1; }

    /* SRS = tau^i * G; commit(X) must be point 1 of it, commit(1 + 0 X) is refused (degree 0) */
    CHECK(zkgpu_srs_setup(ctx, TAU, G1_GEN, 64));
    uint64_t srs[64 * 12];
    uint8_t inf[64];
    CHECK(zkgpu_srs_export(ctx, srs, inf, 64));
    if (memcmp(srs, G1_GEN, 96) != 0) { fprintf(stderr, "SRS does not start with the generator\n"); return 1; }
    uint64_t poly[8] = {0};
    memcpy(poly + 4, FR_ONE, 32);                  /* 0 + 1 * X */
    uint64_t c[12];
    uint8_t c_inf = 0;
    CHECK(zkgpu_commit(ctx, poly, 2, c, &c_inf));
    if (c_inf || memcmp(c, srs + 12, 96) != 0) { fprintf(stderr, "commit(X) != tau * G\n"); return 1; }
    memcpy(poly, FR_ONE, 32);
    memset(poly + 4, 0, 32);                       /* the constant 1: PolynomialDegreeIsZero */
    if (zkgpu_commit(ctx, poly, 2, c, &c_inf) != ZKGPU_ERR_DEGREE_ZERO) { fprintf(stderr, "degree-0 commit accepted\n"); return 1; }

    /* raw SRS bytes out and back in */
    size_t need = 0;
    CHECK(zkgpu_srs_export_raw(ctx, NULL, 0, &need));
    if (need != 8 + 97 * 64) { fprintf(stderr, "raw size %zu\n", need); return 1; }
    uint8_t *raw = malloc(need);
    CHECK(zkgpu_srs_export_raw(ctx, raw, need, &need));
    CHECK(zkgpu_srs_load_raw(ctx, raw, need));
    uint64_t srs2[64 * 12];
    CHECK(zkgpu_srs_export(ctx, srs2, inf, 64));
    if (memcmp(srs, srs2, sizeof srs) != 0) { fprintf(stderr, "raw SRS round trip differs\n"); return 1; }

    printf("c client ok: %zu device bytes in use\n", zkgpu_device_bytes(ctx));
    zkgpu_destroy(ctx);
    free(x); free(y); free(raw);
    return 0;
}
