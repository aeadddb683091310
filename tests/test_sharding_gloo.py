"""World-size-2 rehearsal of the point-range sharded commitment on the CPU (gloo): rank partitioning, the all-gather
of 144-byte partial sums and the host-side combination.  The per-rank range sum, which a GPU computes in the
product, is supplied here by the oracle, so the exchange and combination code is what is under test."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_shard_range_partitions():
    from zkvm_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 7):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(l for _, l in spans) == n
            for (b0, l0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + l0 == b1
            assert max(l for _, l in spans) - min(l for _, l in spans) <= 1


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import model as M
    from oracle import pyoracle as orc
    from zkvm_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pts, _ = orc.g1_multiples_of_generator(1, n)
    sc = M.random_fr_mont(n, seed=21)
    base, length = sharding.shard_range(n, rank, world)
    part = orc.msm(pts[base:base + length], sc[base:base + length])
    parts = sharding.gather_partials(part)
    assert parts.shape == (world, 18)
    assert np.array_equal(parts[rank], part)
    xy, inf = sharding.combine(parts)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.concatenate([xy, [np.uint64(inf)]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 300])
def test_sharded_commitment_world2(tmp_path, orc, n):
    import torch.multiprocessing as mp
    import model as M
    port = 29500 + (os.getpid() % 2000) + n % 7
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    pts, _ = orc.g1_multiples_of_generator(1, n)
    sc = M.random_fr_mont(n, seed=21)
    want_xy, want_inf = orc.g1_to_affine(orc.msm(pts, sc))
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "r%d.npy" % r))
        assert np.array_equal(got[:12], want_xy) and int(got[12]) == int(want_inf)


def test_combine_single_process(orc):
    """Without a process group the gather is the identity and the combination still converts to affine."""
    import model as M
    from zkvm_amd import sharding
    pts, _ = orc.g1_multiples_of_generator(1, 16)
    sc = M.random_fr_mont(16, seed=4)
    part = orc.msm(pts, sc)
    xy, inf = sharding.combine(sharding.gather_partials(part))
    want_xy, want_inf = orc.g1_to_affine(part)
    assert np.array_equal(xy, want_xy) and inf == int(want_inf)


# ---- one transform shared by several ranks --------------------------------------------------------------------------
# The library's sharded transforms (NttEngine::slice_forward / inverse_sharded) restated with big integers around the
# oracle's local transforms, with the exchange steps running through zkvm_amd.sharding.collective_on_tensors over
# gloo: rank c evaluates on the coset (g) w^c <w^world> with one local transform, and the way back is a local inverse
# transform, an all-to-all transpose, a world-point inverse DFT per coefficient and an all-gather.
def _ntt_worker(rank, world, port, log_n, len_in, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import model as M
    from oracle import pyoracle as orc
    from zkvm_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    R = M.R_MOD
    n = 1 << log_n
    m_loc, blk = n // world, n // world // world
    w = M.domain_root(n)
    g = 7
    coeffs = M.random_fr_ints(len_in, 77)                      # the same polynomial on every rank
    s = g * pow(w, rank, R) % R
    # forward: fold modulo X^M - s^M, scale by s^m, plain local transform of M points
    folded = [0] * min(len_in, m_loc)
    for j, c in enumerate(coeffs):
        folded[j % m_loc] = (folded[j % m_loc] + c * pow(s, (j // m_loc) * m_loc, R)) % R
    scaled = M.fr_vec_to_mont([c * pow(s, j, R) % R for j, c in enumerate(folded)])
    mine = orc.fft("fft", scaled, m_loc)
    want = orc.fft("coset_fft", M.fr_vec_to_mont(coeffs), n)
    assert np.array_equal(mine, want[rank::world]), "slice of the forward transform"
    # inverse: local inverse transform over the rank's coset -> the polynomial modulo X^M - s^M
    u = M.fr_vec_from_mont(orc.fft("ifft", mine, m_loc))
    s_inv = pow(s, -1, R)
    u = [v * pow(s_inv, j, R) % R for j, v in enumerate(u)]
    send = torch.from_numpy(M.fr_vec_to_mont(u).view(np.uint8).reshape(-1))
    recv = torch.empty_like(send)
    sharding.collective_on_tensors(sharding.ALL_TO_ALL, send, recv)
    got = M.fr_vec_from_mont(recv.numpy().view(np.uint64).reshape(-1, 4))     # [c][m'] : rank c's block for my range
    w_world_inv = pow(w, -m_loc, R)
    out = []
    for q in range(world):
        scale = pow(pow(g, q * m_loc, R) * world, -1, R)
        for mm in range(blk):
            acc = sum(got[c * blk + mm] * pow(w_world_inv, c * q, R) for c in range(world)) % R
            out.append(acc * scale % R)
    send = torch.from_numpy(M.fr_vec_to_mont(out).view(np.uint8).reshape(-1))
    recv = torch.empty(send.numel() * world, dtype=torch.uint8)
    sharding.collective_on_tensors(sharding.ALL_GATHER, send, recv)
    gathered = M.fr_vec_from_mont(recv.numpy().view(np.uint64).reshape(-1, 4))  # [r][q][m']
    full = [0] * n
    for r in range(world):
        for q in range(world):
            for mm in range(blk):
                full[q * m_loc + r * blk + mm] = gathered[(r * world + q) * blk + mm]
    assert full[:len_in] == coeffs and not any(full[len_in:]), "coefficients after the sharded inverse"
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,len_in", [(2, 6, 64), (2, 7, 19), (4, 6, 35), (4, 5, 32)])
def test_sharded_transform_algebra_over_gloo(tmp_path, orc, world, log_n, len_in):
    import torch.multiprocessing as mp
    port = 29300 + (os.getpid() % 2000) + 17 * world + log_n
    mp.spawn(_ntt_worker, args=(world, port, log_n, len_in, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))
