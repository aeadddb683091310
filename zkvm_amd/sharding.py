"""Point-range sharding of a commitment over the GPUs of one node (one process per GPU).

Every rank holds SRS[base, base + len) and the matching slice of the scalars, reduces its range to one projective
point on its GPU, and the 144-byte partial sums are all-gathered (RCCL when the process group's backend is nccl)
and added in rank order, followed by the single conversion to affine — the same sum as
msm_variable_base over the whole range (coset-bls12_381/src/coset/multiscalar_mul.rs:143-220) because G1 addition
is associative and the affine form of a point is unique.
"""
import numpy as np

from . import api

ALL_TO_ALL, ALL_GATHER = 0, 1


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


def collective_on_tensors(op, send, recv, stage_on_host=False):
    """One collective of the sharded transforms on byte tensors of the default process group: ALL_TO_ALL (send holds
    world blocks, block r for rank r; recv receives block r from rank r) or ALL_GATHER (send holds one block; recv
    receives all ranks' blocks in rank order).  With stage_on_host the data travels through host copies (gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if stage_on_host:
        s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
    else:
        s, r = send, recv
    if op == ALL_TO_ALL:
        dist.all_to_all_single(r, s)
    elif op == ALL_GATHER:
        if stage_on_host or not r.is_cuda:
            dist.all_gather(list(r.chunk(world)), s)
        else:
            dist.all_gather_into_tensor(r, s)
    else:
        raise ValueError("unknown collective %r" % (op,))
    if stage_on_host:
        recv.copy_(r)
    return recv


class _DeviceBytes:
    """A range of device memory owned by the library, presented through __cuda_array_interface__ so that torch wraps
    it as a tensor without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2,
                                         "strides": None}


def enable_collectives(ctx, device=None):
    """Registers torch.distributed as the exchange of this context's sharded transforms (zkgpu_set_collective): RCCL
    on the device buffers when the default process group's backend is nccl (pass its CUDA device), host-staged
    otherwise.  Returns the callback object, which must stay alive as long as the context."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    cuda = torch.device("cuda", ctx.device)
    on_host = device is None

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
    def collective(_user, op, send_ptr, recv_ptr, block_bytes):
        try:
            send_len = block_bytes * (world if op == ALL_TO_ALL else 1)
            send = torch.as_tensor(_DeviceBytes(send_ptr, send_len), device=cuda)
            recv = torch.as_tensor(_DeviceBytes(recv_ptr, block_bytes * world), device=cuda)
            collective_on_tensors(op, send, recv, stage_on_host=on_host)
            torch.cuda.synchronize(cuda)      # the library reads recv on its own stream next
            return 0
        except Exception:   # never unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1

    ctx._collective_cb = collective
    ctx._check(ctx._lib.zkgpu_set_collective(ctx._h, int(rank), int(world), C.cast(collective, C.c_void_p), None))
    return collective
