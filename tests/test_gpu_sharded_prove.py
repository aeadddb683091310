"""Multi-GPU proving rehearsed with two processes on one GPU (gloo for the exchange): every rank holds half of the SRS,
sums its range of every commitment, and both must return the oracle's proof bit for bit."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, n, out_dir):
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
    sharding.enable_prover_sharding(ctx, base, total)
    c = circuits.arithmetic_circuit(n, seed=9)
    circuits.fill_sigma_rows(c, ctx)
    prover = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"sharded", c["pi_indexes"])
    sel_xy, sel_inf = prover.selector_commitments()
    proof = prover.prove(c["wires"].reshape(-1, 4), c["public_inputs"], circuits.random_blinders(9))
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), sel_xy=sel_xy, sel_inf=sel_inf, **proof)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 1024])
def test_two_rank_sharded_proof_equals_oracle(tmp_path, orc, n):
    import torch.multiprocessing as mp
    from zkvm_amd import circuits
    port = 29700 + (os.getpid() % 1500) + (n % 13)
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    tau = circuits.seeded_tau(3)
    srs, _ = orc.srs_powers(orc.g1_generator(), tau, n + 7)
    c = circuits.arithmetic_circuit(n, seed=9)
    c["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(c["sigma_map"][k]) for k in range(4)])
    op = orc.Prover(n, c["constraints"], c["selectors"], srs, b"sharded", c["pi_indexes"], threads=4)
    want = op.prove(c["wires"], c["public_inputs"], circuits.random_blinders(9))
    assert want["rc"] == 0
    want_sel, want_sel_inf = op.selector_comms()
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert np.array_equal(got["sel_inf"], want_sel_inf)
        assert np.array_equal(got["sel_xy"][want_sel_inf == 0], want_sel[want_sel_inf == 0])
        for key in ("comms", "comms_inf", "evals", "challenges"):
            assert np.array_equal(got[key], want[key]), (r, key)


@pytest.mark.parametrize("extra", [["--log-gates", "10"], ["--log-gates", "10", "--sharded"],
                                   ["--workload", "msm", "--log-points", "14"]])
def test_bench_two_rank_flow(extra):
    """bench.py under torch.distributed.run with two ranks sharing the one GPU (rehearsal mode: gloo collectives):
    barriers, max-over-ranks timing, sharded exchange and the single JSON line of rank 0."""
    import json
    import subprocess
    env = dict(os.environ, ZKGPU_BENCH_REHEARSAL="1")
    port = 29900 + (os.getpid() % 1000) + len(extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["scaling"] == ("weak" if extra == ["--log-gates", "10"] else "strong")
