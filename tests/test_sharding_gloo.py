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
