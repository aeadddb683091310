"""One transform and one quotient over several GPUs, rehearsed with several processes on one GPU (gloo for the
exchanges): every rank's slice of the forward transforms, the four-step inverse with its all-to-all and all-gather, and
whole proofs whose 8n-point work and commitments are both shared - all bit for bit against the CPU oracle."""
import os
import sys

import numpy as np
import pytest

import model as M

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def ctx():
    import zkvm_amd
    c = zkvm_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("log_n,len_in", [(5, 32), (10, 1000), (13, (1 << 10) + 3), (17, (1 << 14) + 3), (16, 1 << 16)])
def test_forward_slices_interleave_to_the_oracle_transform(ctx, orc, world, log_n, len_in):
    """Rank c's local transform over the coset (g) w^c <w^world> gives elements c, c + world, ... of fft / coset_fft:
    no exchange.  Inputs longer than a rank's share are folded first (len_in = n + 3 against 8n / 8)."""
    import torch
    n = 1 << log_n
    x = M.random_fr_mont(len_in, 300 + log_n)
    xd = torch.from_numpy(x.view(np.int64)).cuda()
    for mode, name in ((0, "fft"), (2, "coset_fft")):
        want = orc.fft(name, x, n)
        for rank in range(world):
            got = ctx.ntt_slice(xd, log_n, mode, rank, world)
            ctx.sync()
            assert np.array_equal(got.cpu().numpy().view(np.uint64), want[rank::world]), (name, rank)


def _inverse_worker(rank, world, port, cases, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import model as M
    import zkvm_amd
    from oracle import pyoracle as orc
    from zkvm_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = zkvm_amd.Context(0)
    sharding.enable_collectives(ctx)
    for log_n, seed in cases:
        n = 1 << log_n
        evals = M.random_fr_mont(n, seed)                     # all evaluations; this rank holds every world-th
        for mode, name in ((1, "ifft"), (3, "coset_ifft")):
            mine = torch.from_numpy(np.ascontiguousarray(evals[rank::world]).view(np.int64)).cuda()
            got = ctx.intt_sharded(mine, log_n, mode)
            ctx.sync()
            want = orc.fft(name, evals, n)
            assert np.array_equal(got.cpu().numpy().view(np.uint64), want), (name, log_n, rank)
        # round trip with the forward slice of a short polynomial (the quotient's shape: degree about n / 2)
        coeffs = M.random_fr_mont(n // 2 + 7 if n >= 16 else n, seed + 1)
        cd = torch.from_numpy(coeffs.view(np.int64)).cuda()
        back = ctx.intt_sharded(ctx.ntt_slice(cd, log_n, 2, rank, world), log_n, 3)
        ctx.sync()
        b = back.cpu().numpy().view(np.uint64)
        assert np.array_equal(b[:coeffs.shape[0]], coeffs) and not b[coeffs.shape[0]:].any()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cases", [(2, [(2, 1), (6, 2), (13, 3), (17, 4)]), (4, [(4, 5), (11, 6), (16, 7)]),
                                         (8, [(6, 8), (15, 9)])])
def test_inverse_over_ranks_equals_oracle(tmp_path, world, cases):
    import torch.multiprocessing as mp
    port = 29100 + (os.getpid() % 1500) + 31 * world
    mp.spawn(_inverse_worker, args=(world, port, cases, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def _prove_worker(rank, world, port, n, seed, widgets, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import zkvm_amd
    from zkvm_amd import api, circuits, sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = zkvm_amd.Context(0)
    total = n + 7
    base, count = sharding.shard_range(total, rank, world)
    sharding.srs_setup_range(ctx, circuits.seeded_tau(3), circuits.g1_generator_limbs(), base, count)
    sharding.enable_prover_sharding(ctx, base, total)        # commitments: SRS point ranges
    sharding.enable_collectives(ctx)                          # 8n-point work: cyclic slices of the coset
    c = np.load(os.path.join(out_dir, "circuit.npz"))
    prover = api.Prover(ctx, n, int(c["constraints"]), c["selectors"].reshape(-1, 4), b"shared", [int(i) for i in c["pi_indexes"]])
    assert prover.key_eval8n_len() == 8 * n // world
    sel_xy, sel_inf = prover.selector_commitments()
    proofs = [prover.prove(c["wires"].reshape(-1, 4), c["public_inputs"], circuits.random_blinders(s)) for s in (seed, seed + 1)]
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), sel_xy=sel_xy, sel_inf=sel_inf,
             **{"%s%d" % (k, i): v for i, p in enumerate(proofs) for k, v in p.items()})
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def _merkle_batch(k, height, seed):
    from zkvm_amd import poseidon as P
    tree = P.Tree(height, native=True)
    rng = np.random.default_rng(seed)
    spots = sorted({int(p) for p in rng.integers(0, tree.capacity(), k + 4)})[:k]
    for p in spots:
        tree.insert(p, P.hash_to_scalar(p.to_bytes(8, "little")))
    return P.batch_opening_circuit([tree.opening(p) for p in spots], [tree.leaves[p] for p in spots], threads=4)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,log_n", [(2, 10), (2, 14), (4, 10), (8, 6), (4, "merkle"), (8, 14), (4, 17)])
def test_shared_proof_equals_oracle(tmp_path, orc, world, log_n):
    """zkgpu_set_collective + zkgpu_prover_set_sharding: every rank holds 1 / world of the ProverKey's coset vectors and
    of the SRS, evaluates the quotient on its slice, and all ranks return the oracle's proof.  The "merkle" case is the
    reference's sharded configuration in small: three Poseidon-Merkle openings in one circuit (2^14 rows, three public
    inputs) on four ranks.  (8, 14) is the fold path at size: n + 3 coefficients are more than a rank's 8n / 8 points;
    (4, 17) a proof whose passes take three launches per transform on every rank."""
    import torch.multiprocessing as mp
    from zkvm_amd import circuits
    if log_n == "merkle":
        c = _merkle_batch(3, 4, seed=9)
        n, log_n, seed = c["n"], c["n"].bit_length() - 1, 77
        assert n == 1 << 14
    else:
        n = 1 << log_n
        seed = 40 + log_n
        c = circuits.arithmetic_circuit(n, seed=seed)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    np.savez(os.path.join(str(tmp_path), "circuit.npz"), constraints=c["constraints"], selectors=c["selectors"], wires=c["wires"],
             public_inputs=c["public_inputs"], pi_indexes=np.array(c["pi_indexes"]))
    port = 28800 + (os.getpid() % 1500) + 7 * world + log_n
    mp.spawn(_prove_worker, args=(world, port, n, seed, False, str(tmp_path)), nprocs=world, join=True)
    srs, _ = orc.srs_powers(orc.g1_generator(), circuits.seeded_tau(3), n + 7)
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"shared", c["pi_indexes"], threads=16)
    want_sel, want_sel_inf = op.selector_comms()
    for i, s in enumerate((seed, seed + 1)):
        want = op.prove(c["wires"], c["public_inputs"], circuits.random_blinders(s))
        assert want["rc"] == 0
        for r in range(world):
            got = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
            assert np.array_equal(got["sel_inf"], want_sel_inf)
            assert np.array_equal(got["sel_xy"][want_sel_inf == 0], want_sel[want_sel_inf == 0])
            for key in ("comms", "comms_inf", "evals", "challenges"):
                assert np.array_equal(got["%s%d" % (key, i)], want[key]), (r, key, i)
