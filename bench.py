#!/usr/bin/env python3
"""Benchmark of the hot path on N GPUs of one node (one process per GPU).

  python bench.py --gpus 1 --steps K --warmup W [--workload prove|ntt|msm]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM.

workload prove (default; BASELINE.json configs[3]): one step = one full PLONK proof of a 2^20-gate synthetic
  arithmetic circuit with the witness resident in HBM; with N GPUs every rank proves its own circuit (weak).
workload ntt (BASELINE.json configs[1]): one step = fft(2^20) + ifft(2^20) + coset_fft of an
  (n+3)-coefficient polynomial on the 8n = 2^23 domain + coset_ifft(2^23), n = 2^20.  Metric: Fr NTT GB/s with
  the algorithmic bytes 2 * N * 32 per N-point transform.  With N GPUs every rank transforms its own batch
  (independent polynomials, no collective): weak scaling.
workload msm (configs[2]): one step = one 2^24-point G1 MSM; with N GPUs the points are range-sharded, every
  rank sums its range and the 144-byte partial sums are all-gathered over RCCL and added: strong scaling.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the workload, timed live with HIP
events on the library's own stream; `cpu_baseline` times the CPU oracle (a C++ restatement of the reference's
algorithm with the reference's parallel structure; the reference itself is Rust and cannot be built here) on a
bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FR_TOP = np.uint64(0x73eda753299d7d48)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
# Integer-VALU ceilings measured on MI355X with tools/ubench.hip (whole chip, dependent chains, >= 2 waves per SIMD):
FR_MUL_PEAK_G = 127.0    # G Fr Montgomery products / s
FQ28_MUL_PEAK_G = 76.0   # G Fq products / s in the MSM's 28-bit-limb form
# Multiply-accumulate work of one mixed bucket addition in units of one full Fq28 product (196 partial products +
# 210 for the Montgomery reduction = 406): 6 products, 2 squarings (105 + 210 = 0.78 each) and Y3's two products under
# one reduction (392 + 210 = 1.48).
PRODUCTS_PER_MIXED_ADD = 6 + 2 * 315 / 406 + 602 / 406


def pmc_traffic(profile_csv, kernel):
    """HBM bytes per launch of `kernel` from a committed hardware-counter summary (tools/pmc_summary.py output:
    FETCH_SIZE doubled for gfx950, WRITE_SIZE as counted), or None when the file or the kernel is missing."""
    import csv
    path = os.path.join(ROOT, "profiles", profile_csv)
    if not os.path.exists(path):
        return None
    for row in csv.DictReader(open(path)):
        if row["kernel"].split("<")[0].endswith(kernel):      # template arguments dropped
            return int(row["hbm_read_bytes_corrected"]) + int(row["hbm_write_bytes"])
    return None


def random_fr(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] %= FR_TOP
    return a


def to_dev(a, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(device)


# ZKGPU_BENCH_REHEARSAL=1 rehearses the multi-rank flow on a machine with ONE GPU: all ranks share cuda:0 and the
# collectives run over gloo on host tensors.  Its numbers mean nothing; it exists to exercise the N > 1 code path.
REHEARSAL = os.environ.get("ZKGPU_BENCH_REHEARSAL") == "1"


def comm_device(device):
    return None if REHEARSAL else device


def dist_setup(n_gpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if REHEARSAL:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("gloo" if REHEARSAL else "nccl", rank=rank, world_size=world)
    assert world == n_gpus, "launch with torch.distributed.run --nproc-per-node %d" % n_gpus
    return rank, world, local


def barrier(world, device):
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize(device)


def max_over_ranks(value, world, device):
    import torch
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=comm_device(device) or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- NTT workload ------------------------------------------------------------------------------------------
LOG_N = 20
NTT_STEP_TRANSFORMS = [(0, LOG_N), (1, LOG_N), (2, LOG_N + 3), (3, LOG_N + 3)]   # (mode, log size)


def ntt_step_bytes():
    return sum(2 * (1 << lg) * 32 for _, lg in NTT_STEP_TRANSFORMS)


def run_ntt(args, rank, world, local, device):
    import torch
    import zkvm_amd
    ctx = zkvm_amd.Context(local)
    n = 1 << LOG_N
    evals = to_dev(random_fr(n, 1 + rank), device)                 # seed 1: forward / inverse input
    coeffs = to_dev(random_fr(n + 3, 2 + 100 * rank), device)      # seed 2: (n+3)-coefficient polynomial
    buf_n = torch.empty((n, 4), dtype=torch.int64, device=device)
    buf_8n = torch.empty((8 * n, 4), dtype=torch.int64, device=device)

    def step():
        ctx.ntt(evals, LOG_N, zkvm_amd.FFT, out=buf_n)
        ctx.ntt(buf_n, LOG_N, zkvm_amd.IFFT, out=buf_n)
        ctx.ntt(coeffs, LOG_N + 3, zkvm_amd.COSET_FFT, out=buf_8n)
        ctx.ntt(buf_8n, LOG_N + 3, zkvm_amd.COSET_IFFT, out=buf_8n)

    for _ in range(max(args.warmup, 1)):
        step()
    ctx.sync()
    barrier(world, device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    barrier(world, device)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, device)

    # the round trip must return the input (cheap end-to-end sanity check inside the bench itself)
    assert torch.equal(buf_n, evals), "ifft(fft(x)) != x"
    assert torch.equal(buf_8n[:n + 3], coeffs) and not bool(buf_8n[n + 3:].any()), "coset round trip failed"

    out = {}
    if rank == 0:
        # dominant kernel: ntt_pass_kernel on the 2^23 coset transforms (3 launches per transform).  HIP events on
        # the library's stream around `reps` coset_fft(2^23) calls.
        reps, passes = 10, 3
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            ctx.ntt(coeffs, LOG_N + 3, zkvm_amd.COSET_FFT, out=buf_8n)
        ms = ctx.timer_stop()
        launch_ms = ms / (reps * passes)
        alg_bytes_per_launch = 2 * (8 * n) * 32 / passes
        achieved = alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9
        fr_products = (8 * n // 2) * (LOG_N + 3) / passes          # butterflies per launch, one Fr product each
        g_products = fr_products / (launch_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "ntt_pass_kernel (2^23-point coset_fft, %d launches per transform)" % passes,
                           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4),
                           "traffic": pmc_traffic("r01_pmc_ntt.csv", "ntt_pass_kernel"),
                           "traffic_note": "mean over the 2^20 and 2^23 launches of one bench step (profiles/r01_pmc_ntt.csv)",
                           "avg_launch_ms": round(launch_ms, 4),
                           "algorithmic_bytes_per_launch": int(alg_bytes_per_launch),
                           "valu": {"unit": "G Fr products/s", "achieved": round(g_products, 1), "peak": FR_MUL_PEAK_G,
                                    "frac": round(g_products / FR_MUL_PEAK_G, 3)},
                           "note": "integer-VALU-bound: one Fr product plus an add and a subtract per butterfly"}
        out["per_transform_ms"] = {}
        for mode, lg, name in ((0, LOG_N, "fft_2^20"), (1, LOG_N, "ifft_2^20"), (2, LOG_N + 3, "coset_fft_2^23"),
                               (3, LOG_N + 3, "coset_ifft_2^23")):
            src, dst = (evals, buf_n) if lg == LOG_N else (coeffs if mode == 2 else buf_8n, buf_8n)
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                ctx.ntt(src, lg, mode, out=dst)
            out["per_transform_ms"][name] = round(ctx.timer_stop() / 5, 4)
    ctx.sync()
    value = world * args.steps * ntt_step_bytes() / elapsed / 1e9
    return value, elapsed, out, ctx


def cpu_baseline_ntt():
    """Serial radix-2 FFT of the oracle (the reference's best_fft is serial: plonk/src/fft/domain.rs:286-340) on
    one 2^20 forward, one 2^20 inverse and one 2^23 coset transform."""
    from oracle import pyoracle as orc
    n = 1 << LOG_N
    x = random_fr(n, 1)
    c = random_fr(n + 3, 2)
    secs = 0.0
    nbytes = 0
    for mode, data, size in (("fft", x, n), ("ifft", x, n), ("coset_fft", c, 8 * n)):
        _, s = orc.fft(mode, data, size, with_time=True)
        secs += s
        nbytes += 2 * size * 32
    return {"value": round(nbytes / secs / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "fft(2^20) + ifft(2^20) + coset_fft(2^20+3 coeffs -> 2^23), serial radix-2 as in the reference; %.1f s" % secs}


# ---- MSM workload ------------------------------------------------------------------------------------------
MSM_LOG = 24   # the largest size BASELINE.json names for the commitment MSM; it fits one GPU (35 GiB table)


def run_msm(args, rank, world, local, device):
    import torch
    import zkvm_amd
    ctx = zkvm_amd.Context(local)
    from zkvm_amd import sharding
    n_total = 1 << args.log_points
    base, shard = sharding.shard_range(n_total, rank, world)
    # this rank's range of the SRS: tau^(base + i) * G = tau^i * (tau^base * G)
    from zkvm_amd import circuits
    tau = circuits.seeded_tau()
    g = circuits.g1_generator_limbs()
    if base:
        tau_int = int.from_bytes(tau.tobytes(), "little") * circuits.MONT_INV % circuits.R_MOD
        shift = circuits.ints_to_limbs([circuits.to_mont(pow(tau_int, base, circuits.R_MOD))])[0]
        ctx.srs_setup(shift, g, 2)
        g = ctx.srs_export(2)[0][1]
    ctx.srs_setup(tau, g, shard)
    scalars = to_dev(random_fr(n_total, 3)[base:base + shard], device)   # seed 3: uniform scalars

    def step():
        return sharding.sharded_msm(ctx, scalars, comm_device(device))

    for _ in range(max(args.warmup, 1)):
        step()
    barrier(world, device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier(world, device)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, device)
    out = {}
    if rank == 0:
        reps = 3
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            ctx.msm_kernels(scalars)
        ms = ctx.timer_stop() / reps
        out["msm_device_phase_ms"] = round(ms, 3)
        out["result_x0"] = hex(int(res[0][0]))
        phases = [ctx.msm_profile(scalars) for _ in range(reps)]
        acc_ms = sum(p["accumulate"] for p in phases) / reps
        out["msm_phase_ms"] = {k: round(sum(p[k] for p in phases) / reps, 3) for k in phases[0]}
        windows = ctx.msm_get_params()[1]
        alg = 128.0 * shard
        achieved = alg / (acc_ms * 1e-3) / 1e9
        products = PRODUCTS_PER_MIXED_ADD * shard * windows / (acc_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "msm_accumulate_kernel (this rank's %d points)" % shard,
                           "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "traffic": (pmc_traffic("r01_pmc_msm%d.csv" % (shard.bit_length() - 1), "msm_accumulate_kernel")
                                       if shard & (shard - 1) == 0 else None),
                           "traffic_note": "profiles/r01_pmc_msm<log2 points>.csv: the same kernel on the same point count",
                           "avg_launch_ms": round(acc_ms, 4), "algorithmic_bytes_per_launch": int(alg),
                           "valu": {"unit": "G Fq product-equivalents/s", "achieved": round(products, 1), "peak": FQ28_MUL_PEAK_G,
                                    "frac": round(products / FQ28_MUL_PEAK_G, 3)},
                           "note": "integer-VALU-bound: %d table points per scalar, one mixed addition (%.2f product-equivalents) each" % (windows, PRODUCTS_PER_MIXED_ADD)}
    value = args.steps * n_total / elapsed / 1e6
    return value, elapsed, out, ctx


def cpu_baseline_msm():
    from oracle import pyoracle as orc
    n = 1 << 22   # about ten seconds of CPU work on 16 threads
    pts = tiled_srs(n)
    sc = random_fr(n, 3)
    cores = min(os.cpu_count() or 1, 16)
    _, secs = orc.msm(pts, sc, threads=cores, with_time=True)
    return {"value": round(n / secs / 1e6, 4), "unit": "Mpts/s", "cores": cores, "kind": "port",
            "sample": "one 2^22-point msm_variable_base (window-parallel as the reference's rayon map); %.1f s" % secs}


# ---- prove workload ----------------------------------------------------------------------------------------
PROVE_LOG = 20


def tiled_srs(n_points):
    """Point table for the CPU baseline only: 2^12 consecutive multiples of G from the oracle, tiled (duplicates do
    not change the CPU's work: its addition formulas are complete and the bucket method never compares points)."""
    from oracle import pyoracle as orc
    base, _ = orc.g1_multiples_of_generator(1, 1 << 12)
    reps = (n_points + (1 << 12) - 1) >> 12
    return np.tile(base, (reps, 1))[:n_points]


def run_prove(args, rank, world, local, device):
    import torch
    import zkvm_amd
    from zkvm_amd import api, circuits
    log_n = args.log_gates
    n = 1 << log_n
    ctx = zkvm_amd.Context(local)
    sharded = args.sharded and world > 1
    t0 = time.perf_counter()
    if sharded:
        # one proof at a time on all GPUs: every rank holds a range of the SRS and sums its share of each commitment
        from zkvm_amd import sharding
        base, count = sharding.shard_range(n + 7, rank, world)
        sharding.srs_setup_range(ctx, circuits.seeded_tau(), circuits.g1_generator_limbs(), base, count)
        sharding.enable_prover_sharding(ctx, base, n + 7, comm_device(device))
    else:
        if args.msm_window or args.msm_segment:
            ctx.msm_set_params(args.msm_window, args.msm_segment)
        ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n + 7)   # tau^i * G, generated on the device
    ctx.sync()
    t_srs = time.perf_counter() - t0
    t0 = time.perf_counter()
    circ = circuits.arithmetic_circuit(n, seed=5 if sharded else 5 + rank)
    circuits.fill_sigma_rows(circ, ctx)
    t_circ = time.perf_counter() - t0
    t0 = time.perf_counter()
    prover = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"bench", circ["pi_indexes"])
    ctx.sync()
    t_key = time.perf_counter() - t0
    circ["selectors"] = None
    wires = to_dev(circ["wires"].reshape(-1, 4), device)
    blinders = circuits.random_blinders(1)

    proof = None
    for _ in range(max(args.warmup, 1)):
        proof = prover.prove(wires, circ["public_inputs"], blinders)
    barrier(world, device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = prover.prove(wires, circ["public_inputs"], blinders)
    barrier(world, device)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, device)
    out = {}
    timed = None
    if sharded or rank == 0:      # a sharded proof is collective: every rank takes part
        timed = prover.prove(wires, circ["public_inputs"], blinders, timings=True)
    if rank == 0:
        assert np.array_equal(timed["comms"], proof["comms"]) and np.array_equal(timed["evals"], proof["evals"])
        out["phase_ms"] = {k: round(v, 3) for k, v in timed["timings_ms"].items()}
        out["setup_s"] = {"srs_gpu": round(t_srs, 3), "circuit_host": round(t_circ, 2), "key_preprocess_gpu": round(t_key, 3)}
        out["proof_digest"] = {"a_comm_x0": hex(int(proof["comms"][0][0])), "z_eval_0": hex(int(proof["evals"][14][0]))}
        # dominant kernel: msm_accumulate_kernel of the 11 commitments; HIP events on the library's stream around the
        # phases of one n-point commitment
        n_local = count if sharded else n      # points this GPU holds
        sc = to_dev(random_fr(n_local, 3), device)
        reps = 5
        ctx.msm_kernels(sc)
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            ctx.msm_kernels(sc)
        msm_ms = ctx.timer_stop() / reps
        phases = [ctx.msm_profile(sc) for _ in range(reps)]
        out["msm_2^%d_device_ms" % log_n] = round(msm_ms, 3)
        out["msm_2^%d_phase_ms" % log_n] = {k: round(sum(p[k] for p in phases) / reps, 3) for k in phases[0]}
        out["_acc_ms"] = sum(p["accumulate"] for p in phases) / reps
        out["_windows"] = ctx.msm_get_params()[1]
        out["_msm_points"] = n_local
    value = (1 if sharded else world) * args.steps / elapsed
    return value, elapsed, out, ctx


def cpu_baseline_prove(log_gates):
    """The oracle's Prover::prove on a 2^17-gate circuit of the same generator with the reference's parallel
    structure (rayon-parallel MSM windows and quotient evaluation, everything else serial)."""
    from oracle import pyoracle as orc
    from zkvm_amd import circuits
    log_s = 17
    n = 1 << log_s
    cores = min(os.cpu_count() or 1, 16)
    circ = circuits.arithmetic_circuit(n, seed=5)
    circ["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(circ["sigma_map"][k]) for k in range(4)])
    op = orc.Prover(n, circ["constraints"], circ["selectors"], tiled_srs(n + 7), b"bench", circ["pi_indexes"], threads=cores)
    t0 = time.perf_counter()
    r = op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(1))
    secs = time.perf_counter() - t0
    assert r["rc"] == 0
    scale = 1 << (log_gates - log_s)
    return {"value": round(1.0 / (secs * scale), 6), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "one 2^%d-gate proof by the CPU oracle took %.2f s on %d threads; scaled linearly in the gate count "
                      "(x%d) to 2^%d gates, which favours the CPU (its FFT is n log n)" % (log_s, secs, cores, scale, log_gates)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="prove", choices=["prove", "ntt", "msm"])
    ap.add_argument("--log-gates", type=int, default=PROVE_LOG)
    ap.add_argument("--log-points", type=int, default=MSM_LOG)
    ap.add_argument("--sharded", action="store_true",
                    help="prove workload on N > 1 GPUs: one proof at a time with point-range sharded commitments "
                         "(strong scaling) instead of independent proofs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--msm-window", type=int, default=0, help="tuning: force the MSM window width (0 = the engine's choice)")
    ap.add_argument("--msm-segment", type=int, default=0, help="tuning: log2 of the buckets per reduction segment (0 = default)")
    args = ap.parse_args()

    import torch
    rank, world, local = dist_setup(args.gpus)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    if args.workload == "prove":
        value, elapsed, extra, ctx = run_prove(args, rank, world, local, device)
        line = {"metric": "PLONK prove @2^%d gates" % args.log_gates, "unit": "proofs/s", "scaling": "weak",
                "dtype": "u32x8 Fr / u32x12 Fq (Montgomery)",
                "config": {"workload": "full PLONK prove, 2^%d gates (wire commits, permutation Z, quotient over the 8n "
                                       "coset, linearization, openings), bit-exact vs the CPU prover" % args.log_gates,
                           "n": 1 << args.log_gates, "parallelism": "independent proofs per GPU"}}
        if args.sharded and world > 1:
            line["scaling"] = "strong"
            line["config"]["parallelism"] = ("one proof on %d GPUs: SRS point ranges per GPU, 144-byte partial sums "
                                             "all-gathered over RCCL per commitment batch, the rest replicated" % world)
        if rank == 0:
            acc_ms = extra.pop("_acc_ms")
            windows = extra.pop("_windows")
            n = extra.pop("_msm_points")
            alg = 128.0 * n            # every point (96 B) and scalar (32 B) once
            achieved = alg / (acc_ms * 1e-3) / 1e9
            products = PRODUCTS_PER_MIXED_ADD * n * windows / (acc_ms * 1e-3) / 1e9
            extra["roofline"] = {"bound": "hbm", "kernel": "msm_accumulate_kernel of one %d-point commitment (11 per proof)" % n,
                                 "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(achieved / HBM_PEAK_GBS, 5),
                                 "traffic": pmc_traffic("r01_pmc_msm20.csv", "msm_accumulate_kernel") if args.log_gates == 20 else None,
                                 "traffic_note": "profiles/r01_pmc_msm20.csv: the same kernel on one 2^20-point commitment",
                                 "avg_launch_ms": round(acc_ms, 4), "algorithmic_bytes_per_launch": int(alg),
                                 "valu": {"unit": "G Fq product-equivalents/s", "achieved": round(products, 1), "peak": FQ28_MUL_PEAK_G,
                                          "frac": round(products / FQ28_MUL_PEAK_G, 3)},
                                 "note": "integer-VALU-bound: %d table points per scalar, one mixed addition (9.03 product-equivalents) each; the HBM fraction "
                                         "only says that memory is not the limit" % windows}
    elif args.workload == "ntt":
        value, elapsed, extra, ctx = run_ntt(args, rank, world, local, device)
        line = {"metric": "Fr NTT GB/s", "unit": "GB/s", "scaling": "weak", "dtype": "u32x8 (255-bit Fr, Montgomery)",
                "config": {"workload": "fft(2^20) + ifft(2^20) + coset_fft(2^20+3 -> 2^23) + coset_ifft(2^23) per step, "
                                       "bit-exact vs EvaluationDomain", "n": 1 << LOG_N,
                           "algorithmic_bytes_per_step": ntt_step_bytes(), "parallelism": "independent polynomials per GPU"}}
    else:
        value, elapsed, extra, ctx = run_msm(args, rank, world, local, device)
        line = {"metric": "BLS12-381 G1 MSM Mpts/s", "unit": "Mpts/s", "scaling": "strong",
                "dtype": "u32x12 (381-bit Fq, Montgomery)",
                "config": {"workload": "KZG10 commit: 2^%d-point G1 MSM, uniform scalars" % args.log_points,
                           "n": 1 << args.log_points,
                           "parallelism": "point-range shards, 144-byte partial sums all-gathered over RCCL"}}
    if rank == 0:
        line.update({"value": round(value, 3), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                     "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "vs_baseline": None,
                     "data": "synthetic (seeded circuits and scalars; SRS = powers of a seeded tau times the G1 generator)"})
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline_prove(args.log_gates) if args.workload == "prove" else
                                    cpu_baseline_ntt() if args.workload == "ntt" else cpu_baseline_msm())
        print(json.dumps(line))
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
