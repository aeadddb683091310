"""The exchange functions over RCCL itself (torch.distributed backend "nccl"), as far as a one-GPU box allows: a
process group of one rank.  What the multi-GPU benchmark relies on is checked here on the real backend - byte tensors
wrapped around the library's device buffers go through all_to_all_single and all_gather_into_tensor, the 144-byte
partial sums through all_gather on the device - while the multi-rank data movement itself is covered by the gloo tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import numpy as np, torch, torch.distributed as dist
import model as M, zkvm_amd
from oracle import pyoracle as orc
from zkvm_amd import api, circuits, sharding
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = zkvm_amd.Context(0)
sharding.enable_collectives(ctx, dev)
# the sharded inverse transform with its two collectives on RCCL (one rank: the exchanges are copies)
for log_n in (4, 13, 18):
    x = M.random_fr_mont((1 << log_n) - 3, 500 + log_n)
    xd = torch.from_numpy(x.view(np.int64)).cuda()
    ev = ctx.ntt_slice(xd, log_n, 2, 0, 1)
    assert np.array_equal(ev.cpu().numpy().view(np.uint64), orc.fft("coset_fft", x, 1 << log_n))
    back = ctx.intt_sharded(ev, log_n, 3).cpu().numpy().view(np.uint64)
    assert np.array_equal(back[:x.shape[0]], x) and not back[x.shape[0]:].any()
# the byte-tensor collectives on plain CUDA tensors
s = torch.arange(4096, dtype=torch.uint8, device=dev)
r = torch.zeros_like(s)
assert torch.equal(sharding.collective_on_tensors(sharding.ALL_TO_ALL, s, r), s)
r.zero_()
assert torch.equal(sharding.collective_on_tensors(sharding.ALL_GATHER, s, r), s)
# the partial-sum exchange of a sharded commitment on the device
n = 1 << 12
ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n)
sc = M.random_fr_mont(n, 9)
sd = torch.from_numpy(sc.view(np.int64)).cuda()
xy, inf = sharding.sharded_msm(ctx, sd, dev)
want_xy, want_inf = ctx.msm(sd)
assert inf == want_inf and np.array_equal(xy, want_xy)
# a prover with the commitment exchange on RCCL
c = circuits.arithmetic_circuit(256, seed=3)
circuits.fill_sigma_rows(c, ctx)
ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), 256 + 7)
plain = api.Prover(ctx, 256, c["constraints"], c["selectors"].reshape(-1, 4), b"rccl", c["pi_indexes"])
p0 = plain.prove(c["wires"].reshape(-1, 4), c["public_inputs"], circuits.random_blinders(1))
sharding.enable_prover_sharding(ctx, 0, 256 + 7, dev)
shared = api.Prover(ctx, 256, c["constraints"], c["selectors"].reshape(-1, 4), b"rccl", c["pi_indexes"])
p1 = shared.prove(c["wires"].reshape(-1, 4), c["public_inputs"], circuits.random_blinders(1))
assert np.array_equal(p0["comms"], p1["comms"]) and np.array_equal(p0["evals"], p1["evals"])
ctx.close()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
"""


def test_exchanges_run_on_the_rccl_backend():
    port = 28500 + os.getpid() % 1000
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, here=HERE, port=port)], env=env, capture_output=True,
                         text=True, timeout=280)
    assert res.returncode == 0 and "RCCL_SINGLE_RANK_OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])


def test_bench_single_rank_under_the_launcher():
    """bench.py as one rank of torch.distributed.run (the way a multi-GPU run starts it), RCCL process group included."""
    import json
    port = 28400 + os.getpid() % 1000
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ZKGPU_BENCH_REHEARSAL")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log-gates", "12",
           "--log-points", "16", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["msm"]["ranks_seen"] == 1 and out["ntt"]["value"] > 0
