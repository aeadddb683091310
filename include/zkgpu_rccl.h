/* zkgpu_rccl.h — ready-made exchange functions for multi-GPU proving, over RCCL (libzkgpu_rccl.so).
 *
 * libzkgpu.so itself never talks to another GPU: it calls back into the application for the two exchanges of a shared
 * proof (zkgpu_exchange_fn, zkgpu_collective_fn in zkgpu.h).  A Python caller implements them with torch.distributed;
 * a native caller (cgo, Rust FFI, C) links this library instead: one process or thread per GPU, one zkgpu_rccl per
 * rank, and the two functions below are passed to zkgpu_prover_set_sharding / zkgpu_set_collective with the
 * zkgpu_rccl pointer as `user` — or zkgpu_rccl_attach does both.
 *
 * No counterpart in the reference, which proves on one host.  All functions return 0 on success and a non-zero code
 * otherwise; zkgpu_rccl_last_error names the failing call. */
#ifndef ZKGPU_RCCL_H
#define ZKGPU_RCCL_H

#include <stddef.h>
#include <stdint.h>

#include "zkgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ZKGPU_RCCL_ID_BYTES 128

typedef struct zkgpu_rccl zkgpu_rccl;

/* Rank 0 creates the communicator's identifier and hands the 128 bytes to every other rank by the application's own
 * means (a file, a socket, MPI, an environment variable). */
int zkgpu_rccl_unique_id(uint8_t id_out[ZKGPU_RCCL_ID_BYTES]);
/* Joins the communicator as `rank` of `world` on GPU `device`; blocks until all ranks have called it. */
int zkgpu_rccl_init(zkgpu_rccl **out, const uint8_t id[ZKGPU_RCCL_ID_BYTES], int rank, int world, int device);
void zkgpu_rccl_destroy(zkgpu_rccl *r);
const char *zkgpu_rccl_last_error(const zkgpu_rccl *r);

/* A zkgpu_exchange_fn: all-gathers the m partial sums (18 limbs each, host memory) of every rank into
 * all[rank][m][18].  Staged through a small device buffer: world * m * 144 bytes per call. */
int zkgpu_rccl_exchange(void *user, const uint64_t *mine, size_t m, uint64_t *all);
/* A zkgpu_collective_fn on device buffers: ZKGPU_ALL_TO_ALL sends block r of send_dev to rank r (2^log_n * 32 /
 * world^2 bytes per peer for the sharded inverse transform), ZKGPU_ALL_GATHER collects one block from every rank.
 * Returns after the data has arrived. */
int zkgpu_rccl_collective(void *user, int op, const void *send_dev, void *recv_dev, size_t block_bytes);

/* Registers both functions on a context: the commitments are shared with this rank holding the SRS points
 * [shard_base, shard_base + its point count) of total_points, and — when share_quotient is non-zero and world is
 * 2, 4 or 8 — the 8n coset of the quotient too (call before zkgpu_prover_key_load). */
int zkgpu_rccl_attach(zkgpu_rccl *r, zkgpu_ctx *ctx, uint64_t shard_base, uint64_t total_points, int share_quotient);

#ifdef __cplusplus
}
#endif
#endif
