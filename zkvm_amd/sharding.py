"""Point-range sharding of a commitment over the GPUs of one node (one process per GPU).

Every rank holds SRS[base, base + len) and the matching slice of the scalars, reduces its range to one projective
point on its GPU, and the 144-byte partial sums are all-gathered (RCCL when the process group's backend is nccl)
and added in rank order, followed by the single conversion to affine — the same sum as
msm_variable_base over the whole range (coset-bls12_381/src/coset/multiscalar_mul.rs:143-220) because G1 addition
is associative and the affine form of a point is unique.
"""
import numpy as np

from . import api


def shard_range(n, rank, world):
    """Contiguous balanced split of [0, n): -> (base, length); the first n % world ranks get one more."""
    q, r = divmod(n, world)
    base = rank * q + min(rank, r)
    return base, q + (1 if rank < r else 0)


def gather_partials(part, device=None):
    """All-gather one projective partial sum (18 uint64 limbs) per rank -> (world, 18) array, rank order."""
    import torch
    import torch.distributed as dist
    part = np.ascontiguousarray(part, dtype=np.uint64).reshape(18)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return part.reshape(1, 18).copy()
    mine = torch.from_numpy(part.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    return np.stack([g.cpu().numpy().view(np.uint64) for g in gathered])


def combine(parts):
    """Sum of the gathered partial sums as an affine point -> (xy limbs (12,), infinity flag)."""
    return api.g1_sum_to_affine(np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, 18))


def sharded_msm(ctx, scalars_shard_dev, device=None):
    """This rank's range sum on its GPU, exchange, combine: the commitment, identical on every rank."""
    part = ctx.msm_partial(scalars_shard_dev, 0)
    return combine(gather_partials(part, device))
