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
           "--warmup", "1", "--no-cpu-baseline", "--no-sub-records"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["scaling"] == ("weak" if extra == ["--log-gates", "10"] else "strong")


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the script starts its two ranks itself before touching the GPU
    and relays one line.  The prove line of a multi-GPU run carries the strong-scaling record of the range-sharded
    commitment over all ranks and the one-proof-on-all-GPUs record, whose proof must equal the single-GPU proof."""
    import json
    import subprocess
    env = dict(os.environ, ZKGPU_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--log-gates", "12",
           "--log-points", "17"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0 and out["scaling"] == "weak"
    ms = out["msm_scaling"]
    assert ms["ranks_seen"] == 2 and ms["n_points"] == 1 << 17 and ms["points_per_rank"] == 1 << 16 and ms["value"] > 0
    assert ms["scaling"] == "strong"
    sp = out["sharded_prove"]
    assert sp["equals_single_gpu_proof"] is True and sp["ms_per_proof"] > 0
    assert out["roofline"]["bound"] == "valu" and 0 < out["roofline"]["frac"] < 1
    assert "cpu_baseline" not in out          # rank 0 at N = 1 only


def test_bench_one_gpu_line_carries_all_three_metrics():
    """N = 1: the prove line holds the live commitment and transform records beside the proof rate, a roofline entry
    per kernel family and the CPU baseline with the host it ran on."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ZKGPU_BENCH_REHEARSAL")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log-gates", "14", "--log-points", "18"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["msm"]["value"] > 0 and out["msm"]["unit"] == "Mpts/s" and out["msm"]["ranks_seen"] == 1
    assert out["ntt"]["value"] > 0 and out["ntt"]["unit"] == "GB/s"
    kernels = " ".join(r["kernel"] for r in out["rooflines"])
    for name in ("ntt_pass_kernel", "quotient_limb_kernel", "bucket reduction"):
        assert name in kernels
    for r in [out["roofline"]] + out["rooflines"]:
        assert r["avg_launch_ms"] > 0 and 0 < r["frac"] < 1.05, r
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["os_cpu_count"] >= 1 and "cpu_model" in cb
