// One rank of the RCCL exchange library (include/zkgpu_rccl.h) without Python in the process: the communicator of a
// single GPU.  Checks the three paths a shared proof takes through RCCL: the partial-sum all-gather behind a
// commitment, and the all-to-all / all-gather on device buffers.  Exit status 0 = all checks passed.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "zkgpu_rccl.h"

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != 0) {                                                                                 \
            fprintf(stderr, "%s failed with %d: %s | %s\n", #call, rc_, ctx ? zkgpu_last_error(ctx) : "", \
                    comm ? zkgpu_rccl_last_error(comm) : "");                                           \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

// BLS12-381 G1 generator, x | y as Montgomery limbs (coset-bls12_381/src/g1.rs:218-238)
static const uint64_t G1_GEN[12] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull,
                                    0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull,
                                    0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
static const uint64_t FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const uint64_t TAU[4] = {0x0123456789abcdefull, 0xfedcba9876543210ull, 0x0f1e2d3c4b5a6978ull, 0x1122334455667788ull};

int main() {
    zkgpu_ctx *ctx = nullptr;
    zkgpu_rccl *comm = nullptr;
    uint8_t id[ZKGPU_RCCL_ID_BYTES];
    CHECK(zkgpu_rccl_unique_id(id));
    CHECK(zkgpu_rccl_init(&comm, id, 0, 1, 0));
    CHECK(zkgpu_init(&ctx, 0));

    // the partial sums of three commitments: with one rank the gathered block is the rank's own
    std::vector<uint64_t> mine(3 * 18), all(3 * 18, 0);
    for (size_t i = 0; i < mine.size(); i++) mine[i] = 0x9e3779b97f4a7c15ull * (i + 1);
    CHECK(zkgpu_rccl_exchange(comm, mine.data(), 3, all.data()));
    if (all != mine) { fprintf(stderr, "gathered partial sums differ\n"); return 1; }

    // device buffers through both collectives
    const size_t block = 1 << 20;
    std::vector<uint8_t> host(block), back(block);
    for (size_t i = 0; i < block; i++) host[i] = (uint8_t)(i * 131 + 7);
    void *send = nullptr, *recv = nullptr;
    if (hipMalloc(&send, block) != hipSuccess || hipMalloc(&recv, block) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    for (int op : {ZKGPU_ALL_TO_ALL, ZKGPU_ALL_GATHER}) {
        if (hipMemcpy(send, host.data(), block, hipMemcpyHostToDevice) != hipSuccess || hipMemset(recv, 0, block) != hipSuccess) return 1;
        CHECK(zkgpu_rccl_collective(comm, op, send, recv, block));
        if (hipMemcpy(back.data(), recv, block, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (back != host) { fprintf(stderr, "collective %d returned other bytes\n", op); return 1; }
    }
    if (zkgpu_rccl_collective(comm, 7, send, recv, block) == 0) { fprintf(stderr, "unknown collective accepted\n"); return 1; }

    // a commitment whose partial sum travels through RCCL: commit(X) = tau * G, the SRS's point 1
    CHECK(zkgpu_srs_setup(ctx, TAU, G1_GEN, 64));
    uint64_t srs[64 * 12];
    uint8_t inf[64];
    CHECK(zkgpu_srs_export(ctx, srs, inf, 64));
    CHECK(zkgpu_rccl_attach(comm, ctx, 0, 64, 1));
    uint64_t poly[8] = {0}, c[12];
    memcpy(poly + 4, FR_ONE, 32);
    uint8_t c_inf = 0;
    CHECK(zkgpu_commit(ctx, poly, 2, c, &c_inf));
    if (c_inf || memcmp(c, srs + 12, 96) != 0) { fprintf(stderr, "shared commit(X) != tau * G\n"); return 1; }
    CHECK(zkgpu_prover_set_sharding(ctx, 0, 0, 1, nullptr, nullptr));
    CHECK(zkgpu_set_collective(ctx, 0, 1, nullptr, nullptr));

    printf("rccl client ok\n");
    (void)hipFree(send);
    (void)hipFree(recv);
    zkgpu_destroy(ctx);
    zkgpu_rccl_destroy(comm);
    return 0;
}
