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


def srs_setup_range(ctx, tau, g_xy, base, count):
    """Generates this rank's range of the SRS on its GPU: tau^(base + i) * g = tau^i * (tau^base * g), i < count."""
    from . import circuits
    if base:
        tau_int = int.from_bytes(np.ascontiguousarray(tau, dtype=np.uint64).tobytes(), "little") * circuits.MONT_INV % circuits.R_MOD
        shift = circuits.ints_to_limbs([circuits.to_mont(pow(tau_int, base, circuits.R_MOD))])[0]
        ctx.srs_setup(shift, g_xy, 2)
        g_xy = ctx.srs_export(2)[0][1]
    ctx.srs_setup(tau, g_xy, count)


def enable_prover_sharding(ctx, shard_base, total_points, device=None):
    """Makes every commitment of this context's prover a point-range shard (zkgpu_prover_set_sharding): the partial
    sums travel through torch.distributed.all_gather of the default process group (RCCL when its backend is nccl;
    pass the CUDA device then).  Returns the callback object, which must stay alive as long as the context."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64))
    def exchange(_user, mine, m, out):
        try:
            words = int(m) * 18
            local = np.ctypeslib.as_array(mine, shape=(words,)).copy()
            t = torch.from_numpy(local.view(np.int64))
            if device is not None:
                t = t.to(device)
            gathered = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            dst = np.ctypeslib.as_array(out, shape=(world * words,))
            for r, g in enumerate(gathered):
                dst[r * words:(r + 1) * words] = g.cpu().numpy().view(np.uint64)
            return 0
        except Exception:   # never unwind through the C frames
            return 1

    ctx._exchange_cb = exchange
    ctx._check(ctx._lib.zkgpu_prover_set_sharding(ctx._h, int(shard_base), int(total_points), int(world),
                                                  C.cast(exchange, C.c_void_p), None))
    return exchange
