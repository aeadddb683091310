// The two exchanges of a shared proof over RCCL, for callers without a collective library of their own
// (include/zkgpu_rccl.h).  Host code only: every transfer is an RCCL call on this object's own stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/zkgpu_rccl.h"

struct zkgpu_rccl {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = 0;
    void *stage = nullptr;   // device staging for the partial sums: one send block and world receive blocks
    size_t stage_cap = 0;
    char err[256] = {0};
};

namespace {

int fail(zkgpu_rccl *r, const char *what, const char *detail) {
    if (r) snprintf(r->err, sizeof r->err, "%s: %s", what, detail);
    return 1;
}
#define RC_HIP(r, call)                                                         \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) return fail((r), #call, hipGetErrorString(e_));   \
    } while (0)
#define RC_NCCL(r, call)                                                        \
    do {                                                                        \
        ncclResult_t e_ = (call);                                               \
        if (e_ != ncclSuccess) return fail((r), #call, ncclGetErrorString(e_)); \
    } while (0)

}  // namespace

extern "C" {

int zkgpu_rccl_unique_id(uint8_t id_out[ZKGPU_RCCL_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == ZKGPU_RCCL_ID_BYTES, "identifier size");
    if (!id_out) return 1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int zkgpu_rccl_init(zkgpu_rccl **out, const uint8_t id[ZKGPU_RCCL_ID_BYTES], int rank, int world, int device) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return 1;
    *out = nullptr;
    zkgpu_rccl *r = new (std::nothrow) zkgpu_rccl;
    if (!r) return 1;
    r->rank = rank;
    r->world = world;
    r->device = device;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess ||
        ncclCommInitRank(&r->comm, world, uid, rank) != ncclSuccess) {
        zkgpu_rccl_destroy(r);
        return 1;
    }
    *out = r;
    return 0;
}

void zkgpu_rccl_destroy(zkgpu_rccl *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->comm) ncclCommDestroy(r->comm);
    if (r->stage) (void)hipFree(r->stage);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

const char *zkgpu_rccl_last_error(const zkgpu_rccl *r) { return r ? r->err : "no communicator"; }

int zkgpu_rccl_exchange(void *user, const uint64_t *mine, size_t m, uint64_t *all) {
    zkgpu_rccl *r = static_cast<zkgpu_rccl *>(user);
    if (!r || !mine || !all || m == 0) return fail(r, "zkgpu_rccl_exchange", "bad argument");
    const size_t block = m * 18 * sizeof(uint64_t), need = block * ((size_t)r->world + 1);
    RC_HIP(r, hipSetDevice(r->device));
    if (need > r->stage_cap) {
        if (r->stage) RC_HIP(r, hipFree(r->stage));
        r->stage = nullptr;
        r->stage_cap = 0;
        RC_HIP(r, hipMalloc(&r->stage, need));
        r->stage_cap = need;
    }
    char *send = static_cast<char *>(r->stage), *recv = send + block;
    RC_HIP(r, hipMemcpyAsync(send, mine, block, hipMemcpyHostToDevice, r->stream));
    RC_NCCL(r, ncclAllGather(send, recv, block, ncclUint8, r->comm, r->stream));
    RC_HIP(r, hipMemcpyAsync(all, recv, block * r->world, hipMemcpyDeviceToHost, r->stream));
    RC_HIP(r, hipStreamSynchronize(r->stream));
    return 0;
}

int zkgpu_rccl_collective(void *user, int op, const void *send_dev, void *recv_dev, size_t block_bytes) {
    zkgpu_rccl *r = static_cast<zkgpu_rccl *>(user);
    if (!r || !send_dev || !recv_dev || block_bytes == 0) return fail(r, "zkgpu_rccl_collective", "bad argument");
    RC_HIP(r, hipSetDevice(r->device));
    if (op == ZKGPU_ALL_GATHER) {
        RC_NCCL(r, ncclAllGather(send_dev, recv_dev, block_bytes, ncclUint8, r->comm, r->stream));
    } else if (op == ZKGPU_ALL_TO_ALL) {
        // one send and one receive per peer in a single group: every pair of GPUs uses its own xGMI link
        const char *s = static_cast<const char *>(send_dev);
        char *d = static_cast<char *>(recv_dev);
        RC_NCCL(r, ncclGroupStart());
        for (int p = 0; p < r->world; p++) {
            RC_NCCL(r, ncclSend(s + (size_t)p * block_bytes, block_bytes, ncclUint8, p, r->comm, r->stream));
            RC_NCCL(r, ncclRecv(d + (size_t)p * block_bytes, block_bytes, ncclUint8, p, r->comm, r->stream));
        }
        RC_NCCL(r, ncclGroupEnd());
    } else {
        return fail(r, "zkgpu_rccl_collective", "unknown operation");
    }
    RC_HIP(r, hipStreamSynchronize(r->stream));
    return 0;
}

int zkgpu_rccl_attach(zkgpu_rccl *r, zkgpu_ctx *ctx, uint64_t shard_base, uint64_t total_points, int share_quotient) {
    if (!r || !ctx) return fail(r, "zkgpu_rccl_attach", "bad argument");
    int rc = zkgpu_prover_set_sharding(ctx, shard_base, total_points, (uint32_t)r->world, zkgpu_rccl_exchange, r);
    if (rc) return fail(r, "zkgpu_prover_set_sharding", zkgpu_last_error(ctx));
    if (share_quotient) {
        rc = zkgpu_set_collective(ctx, (uint32_t)r->rank, (uint32_t)r->world, zkgpu_rccl_collective, r);
        if (rc) return fail(r, "zkgpu_set_collective", zkgpu_last_error(ctx));
    }
    return 0;
}

}  // extern "C"
