#!/usr/bin/env python3
"""Benchmark of the hot path on N GPUs of one node (one process per GPU).

  python bench.py --gpus N --steps K --warmup W [--workload prove|ntt|msm]

With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (python -m torch.distributed.run)
before it touches the GPU and relays rank 0's line; started under torch.distributed.run it is one of the ranks.

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM.

workload prove (default; BASELINE.json configs[3]): one step = one full PLONK proof of a 2^20-gate synthetic
  arithmetic circuit with the witness resident in HBM; with N GPUs every rank proves its own circuit (weak).  The line
  also carries the other two parts of the metric, timed live in the same run: `msm` (2^24-point commitment, Mpts/s) and
  `ntt` (the transform step, GB/s) at N = 1; `msm_scaling` (the 2^24-point commitment range-sharded over the N GPUs
  with the partial sums all-gathered over RCCL: strong scaling) and `sharded_prove` (one proof on N GPUs) at N > 1.
workload ntt (configs[1]): one step = fft(2^20) + ifft(2^20) + coset_fft of an (n+3)-coefficient polynomial on the
  8n = 2^23 domain + coset_ifft(2^23), n = 2^20.  Metric: Fr NTT GB/s with the algorithmic bytes 2 * N * 32 per N-point
  transform.  With N GPUs every rank transforms its own batch (independent polynomials, no collective): weak scaling.
workload msm (configs[2]): one step = one 2^24-point G1 MSM; with N GPUs the points are range-sharded, every rank
  sums its range and the 144-byte partial sums are all-gathered over RCCL and added: strong scaling.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the workload and `rooflines` the other
kernels of the step, all timed live with HIP events on the library's own stream; `cpu_baseline` times the CPU oracle (a
C++ restatement of the reference's algorithm with the reference's parallel structure; the reference itself is Rust
and cannot be built here) on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FR_TOP = np.uint64(0x73eda753299d7d48)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
# Issue ceiling of v_mad_u64_u32, the instruction every field product is made of, measured with tools/ubench.hip on all
# 1024 SIMDs at 4 and 8 waves per SIMD in launches of 8 ms (profiles/r03_ubench.log): 4.29 cycles per wave-instruction
# per SIMD at the 2.34-2.36 GHz held under that load when every operand has a register of its own and the carry-out
# goes to rotating SGPR pairs (4.35-4.5 with the carry-out in vcc) = 559-563 G/s; in the same harness v_fma_f32 with a
# register of its own per operand issues every 2.4 cycles (the 2-cycle full-rate class), so v_mad_u64_u32 is a
# half-rate instruction: nominally 4 cycles, 614 G/s at 2.4 GHz.
MAD_PEAK_G = 563.0       # G wave-instructions / s, whole chip: the best rate measured for the instruction in any form
# 32 x 32 -> 64-bit multiply-accumulates of one field product on unsaturated limbs (zkvm_amd/csrc/ff.hpp mul_limbs:
# UN^2 partial products + UN (UN + 1) for the Montgomery reduction)
FQ28_MADS = 14 * 14 + 14 * 15        # 406
FR_MADS = 9 * 9 + 9 * 10             # 171
# Multiply-accumulate work of one mixed bucket addition in units of one full Fq28 product: 6 products, 2 squarings
# (105 + 210 = 0.78 each) and Y3's two products under one reduction (392 + 210 = 1.48).
PRODUCTS_PER_MIXED_ADD = 6 + 2 * 315 / 406 + 602 / 406
MADS_PER_MIXED_ADD = PRODUCTS_PER_MIXED_ADD * FQ28_MADS       # 3666
# The same addition priced independently of this implementation: 11 field products (the reference's add_mixed,
# coset-bls12_381/src/g1.rs:701-736) of 2 * 12^2 + 12 = 300 32-bit multiply-accumulates each (CIOS on 12 words).
ALGORITHMIC_MADS_PER_MIXED_ADD = 11 * 300
PROFILE_ROUND = "r03"


def pmc_row(profile_csv, kernel):
    """Row of `kernel` in a committed hardware-counter summary (tools/pmc_summary.py output), or None."""
    import csv
    path = os.path.join(ROOT, "profiles", profile_csv)
    if not os.path.exists(path):
        return None
    for row in csv.DictReader(open(path)):
        if row["kernel"].split("<")[0].endswith(kernel):      # template arguments dropped
            return row
    return None


def pmc_traffic(profile_csv, kernel):
    """HBM bytes per launch of `kernel` from the committed counter summary (FETCH_SIZE doubled for gfx950, WRITE_SIZE
    as counted), or None when the file or the kernel is missing."""
    row = pmc_row(profile_csv, kernel)
    if row is None:
        return None
    return int(row["hbm_read_bytes_corrected"]) + int(row["hbm_write_bytes"])


def pmc_valu_busy(profile_csv, kernel):
    """Fraction of the SIMDs' quad-cycles with a VALU instruction in flight: SQ_ACTIVE_INST_VALU / (1024 SIMDs x
    GRBM_GUI_ACTIVE / 8 XCDs / 4), from the committed counter summary; None when missing."""
    row = pmc_row(profile_csv, kernel)
    try:
        active = float(row["SQ_ACTIVE_INST_VALU_per_dispatch"])
        gui = float(row["GRBM_GUI_ACTIVE_per_dispatch"])
        return round(active / (1024.0 * gui / 8.0 / 4.0), 3) if gui else None
    except (TypeError, KeyError, ValueError):
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


def self_launch(n_gpus):
    """Starts the N ranks of this script as fresh processes (this one has not touched the GPU) and relays rank 0's
    JSON line.  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout.splitlines():
        if out.startswith('{"metric"'):
            line = out
        elif out.strip():
            print(out, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if proc.returncode == 0 and line is None:
        print("bench.py: the ranks finished without printing a result line", file=sys.stderr)
        return 1
    return proc.returncode


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
    if world != n_gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (n_gpus, world))
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


def ranks_seen(world, device):
    """Number of ranks that take part in a collective of the process group (sum of ones)."""
    import torch
    if world == 1:
        return 1
    import torch.distributed as dist
    t = torch.ones(1, dtype=torch.int64, device=comm_device(device) or "cpu")
    dist.all_reduce(t)
    return int(t.item())


def timed_steps(step, steps, warmup, world, device, sync):
    """W untimed steps, then exactly K steps between barriers; the slowest rank's time."""
    for _ in range(max(warmup, 1)):
        step()
    sync()
    barrier(world, device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    barrier(world, device)
    return max_over_ranks(time.perf_counter() - t0, world, device)


# ---- NTT workload ------------------------------------------------------------------------------------------
LOG_N = 20
NTT_STEP_TRANSFORMS = [(0, LOG_N), (1, LOG_N), (2, LOG_N + 3), (3, LOG_N + 3)]   # (mode, log size)


def ntt_step_bytes():
    return sum(2 * (1 << lg) * 32 for _, lg in NTT_STEP_TRANSFORMS)


def ntt_roofline(ctx, coeffs, buf_8n):
    """ntt_pass_kernel on the 2^23-point coset transform (3 launches per transform): HIP events on the library's
    stream around `reps` coset_fft calls."""
    import zkvm_amd
    n = 1 << LOG_N
    reps, passes = 10, 3
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        ctx.ntt(coeffs, LOG_N + 3, zkvm_amd.COSET_FFT, out=buf_8n)
    launch_ms = ctx.timer_stop() / (reps * passes)
    alg_bytes = 2 * (8 * n) * 32 / passes
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    butterflies = (8 * n // 2) * (LOG_N + 3) / passes          # per launch, one Fr product each
    mad_rate = butterflies * FR_MADS / 64 / (launch_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "ntt_pass_kernel (2^23-point coset_fft, %d launches per transform)" % passes,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(PROFILE_ROUND + "_pmc_ntt.csv", "ntt_pass_kernel"),
            "traffic_source": "profiles/%s_pmc_ntt.csv (rocprofv3 --pmc passes of this workload; mean over the 2^20 and "
                              "2^23 launches of a step)" % PROFILE_ROUND,
            "avg_launch_ms": round(launch_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
            "valu": {"unit": "G v_mad_u64_u32 wave-instructions/s", "achieved": round(mad_rate, 1), "peak": MAD_PEAK_G,
                     "frac": round(mad_rate / MAD_PEAK_G, 3), "mads_per_butterfly": FR_MADS,
                     "busy_frac_pmc": pmc_valu_busy(PROFILE_ROUND + "_pmc_ntt.csv", "ntt_pass_kernel")},
            "note": "integer-VALU-bound, not memory-bound: one Fr product (%d multiply-accumulates) plus an add, a subtract, "
                    "two carry sweeps and 18 LDS accesses per butterfly (two stages per LDS round trip, twiddles kept in limb form)"
                    % FR_MADS}


def measure_ntt(ctx, rank, world, device, steps, warmup):
    """-> (GB/s over all ranks, seconds, extra records)"""
    import torch
    import zkvm_amd
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

    elapsed = timed_steps(step, steps, warmup, world, device, ctx.sync)
    # the round trip must return the input (cheap end-to-end sanity check inside the bench itself)
    assert torch.equal(buf_n, evals), "ifft(fft(x)) != x"
    assert torch.equal(buf_8n[:n + 3], coeffs) and not bool(buf_8n[n + 3:].any()), "coset round trip failed"
    out = {}
    if rank == 0:
        out["roofline"] = ntt_roofline(ctx, coeffs, buf_8n)
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
    return world * steps * ntt_step_bytes() / elapsed / 1e9, elapsed, out


def run_ntt(args, rank, world, local, device):
    import zkvm_amd
    ctx = zkvm_amd.Context(local)
    value, elapsed, out = measure_ntt(ctx, rank, world, device, args.steps, args.warmup)
    return value, elapsed, out, ctx


def cpu_info():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count()}


def cpu_threads():
    """Threads the oracle may use: the host cores this process is allowed, at most 16 (the CPU share of one GPU)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        allowed = min(allowed, int(env))
    return max(1, min(allowed, 16))


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
    out = {"value": round(nbytes / secs / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": "fft(2^20) + ifft(2^20) + coset_fft(2^20+3 coeffs -> 2^23), serial radix-2 as in the reference; %.1f s" % secs}
    out.update(cpu_info())
    return out


# ---- MSM workload ------------------------------------------------------------------------------------------
MSM_LOG = 24   # the largest size BASELINE.json names for the commitment MSM; it fits one GPU (35 GiB table)


def msm_rooflines(ctx, scalars, shard):
    """Phase times of one device MSM over this rank's `shard` points from HIP events at the phase boundaries ->
    (records for the JSON line, roofline of msm_accumulate_kernel, rooflines of the sort and reduction kernels)."""
    reps = 3
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        ctx.msm_kernels(scalars)
    device_ms = ctx.timer_stop() / reps
    phases = [ctx.msm_profile(scalars) for _ in range(reps)]
    ph = {k: sum(p[k] for p in phases) / reps for k in phases[0]}
    window, levels = ctx.msm_get_params()
    items = shard * levels                      # (point, digit) pairs = table entries gathered = mixed additions
    acc_ms = ph["accumulate"]
    mad_rate = items * MADS_PER_MIXED_ADD / 64 / (acc_ms * 1e-3) / 1e9
    alg = 128.0 * shard
    log_pts = shard.bit_length() - 1
    pmc_file = "%s_pmc_msm%d.csv" % (PROFILE_ROUND, log_pts)
    main = {"bound": "valu", "kernel": "msm_accumulate_kernel (%d points, %d-bit signed windows, %d table levels)" % (shard, window, levels),
            "achieved": round(mad_rate, 1), "peak": MAD_PEAK_G, "unit": "G v_mad_u64_u32 wave-instructions/s",
            "frac": round(mad_rate / MAD_PEAK_G, 3),
            "frac_algorithmic": round(items * ALGORITHMIC_MADS_PER_MIXED_ADD / 64 / (acc_ms * 1e-3) / 1e9 / MAD_PEAK_G, 3),
            "frac_note": "frac = multiply-accumulates this kernel issues / ceiling; frac_algorithmic = 11 products x 300 "
                         "multiply-accumulates per mixed addition (the reference's formulas on 32-bit words) / ceiling",
            "peak_source": "tools/ubench.hip on this part: v_mad_u64_u32 with a register of its own per operand issues every "
                           "4.29 cycles per SIMD at the 2.35 GHz held under that load, 1024 SIMDs (profiles/r03_ubench.log; "
                           "v_fma_f32 in the same harness: 2.4 cycles)",
            "mads_per_mixed_addition": round(MADS_PER_MIXED_ADD), "mixed_additions_per_launch": int(items),
            "valu_busy_frac_pmc": pmc_valu_busy(pmc_file, "msm_accumulate_kernel") if shard & (shard - 1) == 0 else None,
            "avg_launch_ms": round(acc_ms, 4),
            "traffic": pmc_traffic(pmc_file, "msm_accumulate_kernel") if shard & (shard - 1) == 0 else None,
            "traffic_source": "profiles/%s (rocprofv3 --pmc passes: the same kernel on the same point count)" % pmc_file,
            "hbm": {"algorithmic_bytes_per_launch": int(alg), "achieved": round(alg / (acc_ms * 1e-3) / 1e9, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "note": "bound by integer-VALU issue, not by HBM: one extended-Jacobian mixed addition (%.2f field products of "
                    "%d multiply-accumulates) per gathered table entry; the HBM fraction of the algorithmic 128 bytes per "
                    "point only says that memory is not the limit" % (PRODUCTS_PER_MIXED_ADD, FQ28_MADS)}
    others = []
    sort_ms = ph["count_pass"] + ph["scatter_pass"]
    if sort_ms > 0:
        # count pass reads the keys (4 B per pair), scatter pass reads keys and table indices (8 B) and writes 4 B
        sort_bytes = items * 16.0
        if window > 16:      # two-level sort above 2^15 buckets: count pass, then the LDS-staged final pass
            sort_name, pmc_names = "msm_sort2_kernel<0> + msm_sort3_kernel", ("msm_sort2_kernel", "msm_sort3_kernel")
            note = ("the final pass orders a tile of 2^14 entries by bucket in LDS and stores contiguous runs: bound by HBM and "
                    "LDS traffic, no longer by the transaction rate of lane-scattered 4-byte stores")
        else:
            sort_name, pmc_names = "msm_sort_kernel<0> + <1>", ("msm_sort_kernel",)
            note = "one-level LDS counting sort; scattered 4-byte stores"
        traffic = None
        if shard & (shard - 1) == 0:
            parts = [pmc_traffic(pmc_file, k) for k in pmc_names]
            traffic = sum(parts) if all(p is not None for p in parts) else None
        others.append({"bound": "hbm", "kernel": "%s (count and placing pass over %d pairs)" % (sort_name, items),
                       "avg_launch_ms": round(sort_ms / 2, 4), "algorithmic_bytes_per_launch": int(sort_bytes / 2),
                       "achieved": round(sort_bytes / (sort_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(sort_bytes / (sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "traffic": traffic, "note": note})
    buckets = 1 << (window - 1)
    red_adds = 2.0 * buckets                    # running sums: two accumulator additions per bucket, then the trees
    red_rate = red_adds * 12 * FQ28_MADS / 64 / (ph["reduce"] * 1e-3) / 1e9
    others.append({"bound": "valu", "kernel": "bucket reduction: msm_segment_kernel + msm_group_partial_kernel + "
                                              "msm_group_final_kernel (%d buckets)" % buckets,
                   "avg_launch_ms": round(ph["reduce"] / 3, 4), "phase_ms": round(ph["reduce"], 4),
                   "achieved": round(red_rate, 1), "peak": MAD_PEAK_G, "unit": "G v_mad_u64_u32 wave-instructions/s",
                   "frac": round(red_rate / MAD_PEAK_G, 3),
                   "note": "about 2 accumulator additions (12 products each) per bucket; latency-bound: chains of dependent "
                           "additions on few waves"})
    rec = {"device_ms": round(device_ms, 3), "phase_ms": {k: round(v, 3) for k, v in ph.items()}}
    return rec, main, others


def shard_srs(ctx, n_total, rank, world):
    """Generates this rank's range of the SRS tau^i * G on its GPU -> (base, count)."""
    from zkvm_amd import circuits, sharding
    base, shard = sharding.shard_range(n_total, rank, world)
    sharding.srs_setup_range(ctx, circuits.seeded_tau(), circuits.g1_generator_limbs(), base, shard)
    return base, shard


def measure_msm(ctx, rank, world, device, log_points, steps, warmup):
    """Range-sharded commitment over all ranks -> (Mpts/s, seconds, extra records)."""
    from zkvm_amd import sharding
    n_total = 1 << log_points
    base, shard = shard_srs(ctx, n_total, rank, world)
    scalars = to_dev(random_fr(n_total, 3)[base:base + shard], device)   # seed 3: uniform scalars
    res = []

    def step():
        res[:] = [sharding.sharded_msm(ctx, scalars, comm_device(device))]

    elapsed = timed_steps(step, steps, warmup, world, device, ctx.sync)
    out = {}
    if rank == 0:
        rec, main, others = msm_rooflines(ctx, scalars, shard)
        out["msm_device_phase_ms"] = rec["device_ms"]
        out["msm_phase_ms"] = rec["phase_ms"]
        out["result_x0"] = hex(int(res[0][0][0]))
        out["roofline"] = main
        out["rooflines"] = others
    return steps * n_total / elapsed / 1e6, elapsed, out


def run_msm(args, rank, world, local, device):
    import zkvm_amd
    ctx = zkvm_amd.Context(local)
    value, elapsed, out = measure_msm(ctx, rank, world, device, args.log_points, args.steps, args.warmup)
    return value, elapsed, out, ctx


def cpu_baseline_msm():
    from oracle import pyoracle as orc
    n = 1 << 22   # about ten seconds of CPU work on 16 threads
    pts = tiled_srs(n)
    sc = random_fr(n, 3)
    cores = cpu_threads()
    _, secs = orc.msm(pts, sc, threads=cores, with_time=True)
    out = {"value": round(n / secs / 1e6, 4), "unit": "Mpts/s", "cores": cores, "kind": "port",
           "sample": "one 2^22-point msm_variable_base (window-parallel as the reference's rayon map); %.1f s" % secs}
    out.update(cpu_info())
    return out


# ---- prove workload ----------------------------------------------------------------------------------------
PROVE_LOG = 20


def tiled_srs(n_points):
    """Point table for the CPU baseline only: 2^12 consecutive multiples of G from the oracle, tiled (duplicates do
    not change the CPU's work: its addition formulas are complete and the bucket method never compares points)."""
    from oracle import pyoracle as orc
    base, _ = orc.g1_multiples_of_generator(1, 1 << 12)
    reps = (n_points + (1 << 12) - 1) >> 12
    return np.tile(base, (reps, 1))[:n_points]


def quotient_roofline(ctx, n, coset=8):
    """quotient_limb_kernel (arithmetic-only circuits) of the most recent proof on `ctx`: one thread per point of the
    coset the proof ran over (`coset` x n points: 4n for keys with the arithmetic and permutation terms only, the
    reference's 8n otherwise)."""
    ms = ctx.quotient_kernel_ms()
    streams = 19       # a b c d z | q_m q_l q_r q_o q_f q_c q_arith | sigma 1-4 | X, L1 and PI on the coset
    points = float(coset) * n
    alg = points * 32 * (streams + 1)
    products = 25      # 7 arithmetic gate, 17 permutation argument (beta X twice: plain and times 32), 1 for 1 / Z_H
    mad_rate = points * products * FR_MADS / 64 / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "quotient_limb_kernel (%dn = %d points, %d streams read, 1 written)" % (coset, int(points), streams),
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(alg),
            "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(PROFILE_ROUND + ("_pmc_prove.csv" if coset == 4 else "_pmc_prove_8n.csv"), "quotient_limb_kernel"),
            "valu": {"unit": "G v_mad_u64_u32 wave-instructions/s", "achieved": round(mad_rate, 1), "peak": MAD_PEAK_G,
                     "frac": round(mad_rate / MAD_PEAK_G, 3), "field_products_per_point": products}}


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
        base, count = shard_srs(ctx, n + 7, rank, world)
        sharding.enable_prover_sharding(ctx, base, n + 7, comm_device(device))
        sharding.enable_collectives(ctx, comm_device(device))       # 2, 4 or 8 ranks: the 8n-point work is shared too
    else:
        if args.msm_window or args.msm_segment:
            ctx.msm_set_params(args.msm_window, args.msm_segment)
        if args.full_coset:
            ctx.force_full_coset(True)
        ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n + 7)   # tau^i * G, generated on the device
    ctx.sync()
    t_srs = time.perf_counter() - t0
    t0 = time.perf_counter()
    # `--circuit widgets` (one GPU): the timed proofs themselves are those of a key with every widget, e.g. for a
    # kernel trace or counter passes of its quotient kernels; the line's config says so
    widgets_main = args.circuit == "widgets" and world == 1 and n >= (1 << 14)
    circ = circuits.widget_circuit(n) if widgets_main else circuits.arithmetic_circuit(n, seed=5 if sharded else 5 + rank)
    circuits.fill_sigma_rows(circ, ctx)
    t_circ = time.perf_counter() - t0
    t0 = time.perf_counter()
    prover = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"bench", circ["pi_indexes"])
    ctx.sync()
    t_key = time.perf_counter() - t0
    selectors = circ["selectors"] if (world > 1 and not sharded and rank == 0) else None   # rank 0 reuses them below
    circ["selectors"] = None
    wires = to_dev(circ["wires"].reshape(-1, 4), device)
    blinders = circuits.random_blinders(1)

    proofs = []

    def step():
        proofs[:] = [prover.prove(wires, circ["public_inputs"], blinders)]

    elapsed = timed_steps(step, args.steps, args.warmup, world, device, ctx.sync)
    proof = proofs[0]
    out = {}
    timed = None
    if sharded or rank == 0:      # a sharded proof is collective: every rank takes part
        timed = prover.prove(wires, circ["public_inputs"], blinders, timings=True)
    if rank == 0:
        assert np.array_equal(timed["comms"], proof["comms"]) and np.array_equal(timed["evals"], proof["evals"])
        out["phase_ms"] = {k: round(v, 3) for k, v in timed["timings_ms"].items()}
        out["phase_note"] = ("laps between the transcript's waits: the coset transforms of the wires and of z run behind their "
                             "commitments and count into the first two laps")
        out["setup_s"] = {"srs_gpu": round(t_srs, 3), "circuit_host": round(t_circ, 2), "key_preprocess_gpu": round(t_key, 3)}
        out["proof_digest"] = {"a_comm_x0": hex(int(proof["comms"][0][0])), "z_eval_0": hex(int(proof["evals"][14][0]))}
        out["quotient_coset"] = ("%dn points" % timed["coset"]) + (
            " (every second point of the reference's 8n coset; the seven top coefficients of the quotient restored from "
            "those of the round polynomials: the same coefficients)" if timed["coset"] == 4 else " (the reference's coset)")
        quot = quotient_roofline(ctx, n, timed["coset"])
        if widgets_main:   # the time spans the four widget launches as well; their own roofline is in the `widgets` record
            quot["kernel"] = "quotient of a key with every widget: quotient_widget_kernel<0..3> + " + quot["kernel"]
            quot["note"] = "streams and products counted for the arithmetic and permutation launch only"
        # dominant kernel: msm_accumulate_kernel of the 11 commitments; HIP events on the library's stream around the
        # phases of one n-point commitment
        n_local = count if sharded else n      # points this GPU holds
        sc = to_dev(random_fr(n_local, 3), device)
        ctx.msm_kernels(sc)
        rec, main, others = msm_rooflines(ctx, sc, n_local)
        main["kernel"] += ": one commitment, 11 per proof"
        out["msm_2^%d_device_ms" % log_n] = rec["device_ms"]
        out["msm_2^%d_phase_ms" % log_n] = rec["phase_ms"]
        out["roofline"] = main
        out["rooflines"] = others + [quot]
        del sc
    value = (1 if sharded else world) * args.steps / elapsed
    if rank == 0 and not sharded and not args.no_sub_records:
        # extras beside the headline: a failure in one of them is written into the line instead of losing it
        def extra(name, make):
            try:
                out[name] = make()
            except Exception as e:   # noqa: BLE001
                out.setdefault("sub_record_error", {})[name] = "%s: %s" % (type(e).__name__, e)
        extra("with_h2d", lambda: prove_with_transfers(prover, circ, blinders, n, proof, 1e3 * elapsed / args.steps, max(args.steps // 2, 4)))
        extra("proofs_in_flight", lambda: proofs_in_flight(prover, wires, circ, blinders, proof, max(args.steps // 2, 4), elapsed / args.steps))
        if (args.circuit == "widgets" or not args.no_widget_record) and n >= (1 << 14):   # eight rounds of every widget: 2^12 rows
            extra("widgets", lambda: widget_key_record(ctx, n, device, max(args.steps // 3, 3)))
            if "widgets" in out:
                out["rooflines"].append(out["widgets"].pop("roofline"))

    # ---- the other parts of the metric, in the same run -----------------------------------------------------------
    if not args.no_sub_records and not sharded:
        if world == 1:
            import gc
            ntt_value, ntt_elapsed, ntt_out = measure_ntt(ctx, rank, world, device, max(args.steps // 2, 3), 2)
            out["ntt"] = {"metric": "Fr NTT GB/s", "value": round(ntt_value, 1), "unit": "GB/s",
                          "ms_per_step": round(1e3 * ntt_elapsed / max(args.steps // 2, 3), 4),
                          "workload": "fft(2^20) + ifft(2^20) + coset_fft(2^20+3 -> 2^23) + coset_ifft(2^23) per step",
                          "per_transform_ms": ntt_out["per_transform_ms"]}
            out["rooflines"].append(ntt_out["roofline"])
            gc.collect()
            torch.cuda.empty_cache()
        # With several ranks the records below are extras beside the headline line: a failure in one of them (they
        # are the only code here that needs more than one GPU) is recorded in the line instead of losing it.
        try:
            mctx = zkvm_amd.Context(local)
            m_steps = max(args.steps // 4, 3)
            m_value, m_elapsed, m_out = measure_msm(mctx, rank, world, device, args.log_points, m_steps, 1)
            seen = ranks_seen(world, device)
            if rank == 0:
                rec = {"metric": "BLS12-381 G1 MSM Mpts/s", "value": round(m_value, 2), "unit": "Mpts/s", "n_points": 1 << args.log_points,
                       "steps": m_steps, "ms_per_msm": round(1e3 * m_elapsed / m_steps, 3), "scaling": "strong",
                       "ranks_seen": seen, "points_per_rank": (1 << args.log_points) // world,
                       "phase_ms": m_out["msm_phase_ms"], "result_x0": m_out["result_x0"]}
                if world == 1:
                    out["msm"] = rec
                    out["msm"]["roofline"] = m_out["roofline"]
                else:
                    rec["parallelism"] = ("point ranges of the SRS per GPU, 144-byte partial sums all-gathered over %s and added "
                                          "in rank order" % ("gloo (rehearsal)" if REHEARSAL else "RCCL"))
                    out["msm_scaling"] = rec
            mctx.close()
            del mctx
            if world > 1:
                # one proof on all GPUs: rank 0's circuit again (seed 5), every commitment range-sharded
                from zkvm_amd import sharding
                sctx = zkvm_amd.Context(local)
                base, count = shard_srs(sctx, n + 7, rank, world)
                sharding.enable_prover_sharding(sctx, base, n + 7, comm_device(device))
                sharding.enable_collectives(sctx, comm_device(device))
                sel_shape = [15, n, 4]
                if rank == 0:
                    sel = np.ascontiguousarray(selectors)
                    w_all = circ["wires"]
                    pub = circ["public_inputs"]
                else:
                    c0 = circuits.arithmetic_circuit(n, seed=5)
                    circuits.fill_sigma_rows(c0, sctx)
                    sel, w_all, pub = c0["selectors"], c0["wires"], c0["public_inputs"]
                assert list(sel.shape) == sel_shape
                sp = api.Prover(sctx, n, circ["constraints"], sel.reshape(-1, 4), b"bench", circ["pi_indexes"])
                sw = to_dev(w_all.reshape(-1, 4), device)
                sproofs = []

                def sstep():
                    sproofs[:] = [sp.prove(sw, pub, blinders)]

                s_steps = max(args.steps // 2, 3)
                s_elapsed = timed_steps(sstep, s_steps, 1, world, device, sctx.sync)
                if rank == 0:
                    same = bool(np.array_equal(sproofs[0]["comms"], proof["comms"]) and np.array_equal(sproofs[0]["evals"], proof["evals"]))
                    assert same, "the sharded proof differs from the single-GPU proof of the same circuit"
                    out["sharded_prove"] = {"ms_per_proof": round(1e3 * s_elapsed / s_steps, 3), "steps": s_steps,
                                            "equals_single_gpu_proof": same,
                                            "coset_points_per_gpu": sp.key_eval8n_len(),
                                            "parallelism": "one proof on %d GPUs: SRS point ranges per GPU with the partial sums "
                                                           "all-gathered per commitment batch; with 2, 4 or 8 GPUs the 8n coset is "
                                                           "dealt out cyclically (key vectors, coset transforms and quotient on 1/N "
                                                           "of the points, inverse transform through an all-to-all transpose and an "
                                                           "all-gather); n-point work replicated" % world}
                sctx.close()
        except Exception as exc:   # noqa: BLE001
            if world == 1:
                raise
            import traceback
            traceback.print_exc()
            out.setdefault("sub_record_error", {})["multi_gpu_records"] = "%s: %s" % (type(exc).__name__, exc)
    return value, elapsed, out, ctx


def prove_with_transfers(prover, circ, blinders, n, resident_proof, resident_ms, steps):
    """The same proof with the witness starting in HOST memory, as the reference's service has it (one fresh witness per
    proof, merkle-plonk/src/lib.rs:275-332): from pageable memory, from pinned memory, and from pinned memory with the
    next proof's witness uploaded beside the current proof (zkgpu_witness_prefetch, two alternating host buffers).
    Wall-clock milliseconds per proof; the resident figure is the headline's."""
    from zkvm_amd import api
    ctx = prover.ctx
    w_host = np.ascontiguousarray(circ["wires"].reshape(-1, 4))
    pub = circ["public_inputs"]

    def timed(fn, reps):
        fn(0)
        ctx.sync()
        t0 = time.perf_counter()
        for i in range(reps):
            last = fn(i + 1)
        ctx.sync()
        return 1e3 * (time.perf_counter() - t0) / reps, last

    pageable_ms, p0 = timed(lambda i: prover.prove(w_host, pub, blinders), steps)
    pins = [api.PinnedFr(4 * n) for _ in range(2)]
    for p in pins:
        p.array[:] = w_host
    pinned_ms, p1 = timed(lambda i: prover.prove(pins[i & 1].array, pub, blinders), steps)
    prover.prefetch(pins[0].array)          # the loop: upload witness i + 1, prove witness i

    def overlapped(i):
        prover.prefetch(pins[(i + 1) & 1].array)
        return prover.prove(pins[i & 1].array, pub, blinders)

    overlap_ms, p2 = timed(overlapped, steps)
    prover.prove(pins[(steps + 1) & 1].array, pub, blinders)      # consumes the last upload
    for p in (p0, p1, p2):
        assert np.array_equal(p["comms"], resident_proof["comms"]) and np.array_equal(p["evals"], resident_proof["evals"])
    for p in pins:
        p.close()
    return {"unit": "ms per proof, wall clock, witness starting in host memory (128 MiB at 2^20 gates)",
            "resident": round(resident_ms, 3), "pageable": round(pageable_ms, 3), "pinned": round(pinned_ms, 3),
            "pinned_prefetch": round(overlap_ms, 3), "steps": steps,
            "pinned_prefetch_over_resident": round(overlap_ms / resident_ms, 4),
            "note": "pinned_prefetch: zkgpu_witness_prefetch uploads the next witness on a copy stream beside the current "
                    "proof (two device buffers); never the headline value, which has the witness resident in HBM"}


def widget_key_record(ctx, n, device, steps):
    """A key in which every widget has gates (range, logic, fixed-base scalar multiplication, curve addition:
    zkvm_amd.circuits.widget_circuit), same row count: its quotient is one launch per widget plus the launch of the
    arithmetic and permutation terms, over the 4n coset like an arithmetic key's (forced over the reference's 8n coset
    for the second figure).  On clones of the benchmark's context (the SRS table is shared)."""
    from zkvm_amd import api, circuits
    circ = circuits.widget_circuit(n)
    sel_ready = [False]

    def run(full_coset):
        wctx = ctx.clone()
        try:
            wctx.force_full_coset(full_coset)
            if not sel_ready[0]:
                circuits.fill_sigma_rows(circ, wctx)
                sel_ready[0] = True
            t0 = time.perf_counter()
            wp = api.Prover(wctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"bench-widgets", circ["pi_indexes"])
            wctx.sync()
            t_key = time.perf_counter() - t0
            w = to_dev(circ["wires"].reshape(-1, 4), device)
            bl = circuits.random_blinders(2)
            first = wp.prove(w, circ["public_inputs"], bl)
            wctx.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                last = wp.prove(w, circ["public_inputs"], bl)
            wctx.sync()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            assert last["coset"] == (8 if full_coset else 4) and np.array_equal(first["comms"], last["comms"])
            return ms, wctx.quotient_kernel_ms(), t_key, last
        finally:
            wctx.close()

    ms, q_ms, t_key, proof4 = run(False)
    ms8, q_ms8, _, proof8 = run(True)
    assert all(np.array_equal(proof4[k], proof8[k]) for k in ("comms", "evals", "challenges"))
    points = 4.0 * n
    # field products per point as the kernels are written (poly.hip): range 13, logic 18, fixed-base 19, curve addition
    # 12, arithmetic and permutation 25
    products = 13 + 18 + 19 + 12 + 25
    streams = 19 + 4 + 3 + 2       # the arithmetic kernel's, four widget selectors, three "next row" reads, the sum twice
    alg = points * 32 * (streams + 1)
    mad_rate = points * products * FR_MADS / 64 / (q_ms * 1e-3) / 1e9
    roof = {"bound": "valu", "kernel": "quotient of a key with every widget: quotient_widget_kernel<0..3> + quotient_limb_kernel "
                                       "(4n = %d points, 5 launches)" % int(points),
            "avg_launch_ms": round(q_ms / 5, 4), "phase_ms": round(q_ms, 4),
            "achieved": round(mad_rate, 1), "peak": MAD_PEAK_G, "unit": "G v_mad_u64_u32 wave-instructions/s",
            "frac": round(mad_rate / MAD_PEAK_G, 3), "field_products_per_point": products,
            "hbm": {"algorithmic_bytes": int(alg), "achieved": round(alg / (q_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(alg / (q_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    return {"circuit": "every widget in use (zkvm_amd.circuits.widget_circuit), 2^%d rows" % (n.bit_length() - 1),
            "ms_per_proof": round(ms, 3), "quotient_coset": "4n points (every second point of the reference's 8n coset; the same proof)",
            "quotient_kernels_ms": round(q_ms, 4), "over_8n": {"ms_per_proof": round(ms8, 3), "quotient_kernels_ms": round(q_ms8, 4)},
            "key_preprocess_s": round(t_key, 3), "steps": steps, "roofline": roof}


def proofs_in_flight(prover, wires, circ, blinders, resident_proof, steps, one_at_a_time_s):
    """Two proofs at a time on one GPU: the benchmark's context and a clone of it (zkgpu_ctx_clone: the same SRS table
    and prover key by reference, a stream and per-proof buffers of its own), one host thread each, `steps` proofs per
    thread.  Throughput, not the latency of the headline."""
    import threading
    ctx = prover.ctx
    clone = ctx.clone()
    try:
        provers = [prover, prover.on(clone)]
        last, errors = [None, None], []

        def run(t, reps):
            try:
                for _ in range(reps):
                    last[t] = provers[t].prove(wires, circ["public_inputs"], blinders)
            except Exception as exc:   # noqa: BLE001 - reported by the caller's assert
                errors.append(exc)

        run(1, 1)                       # the clone's first proof builds its twiddle tables and buffers
        ctx.sync()
        clone.sync()
        threads = [threading.Thread(target=run, args=(t, steps)) for t in range(2)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        ctx.sync()
        clone.sync()
        secs = time.perf_counter() - t0
        assert not errors, errors
        for p in last:
            assert np.array_equal(p["comms"], resident_proof["comms"]) and np.array_equal(p["evals"], resident_proof["evals"])
        shared = clone.device_bytes_shared()
        return {"contexts": 2, "proofs_per_s": round(2 * steps / secs, 3), "one_at_a_time_proofs_per_s": round(1.0 / one_at_a_time_s, 3),
                "steps_per_context": steps, "shared_table_and_key_bytes": shared,
                "clone_private_bytes": clone.device_bytes() - shared,
                "note": "the second context is a zkgpu_ctx_clone of the first: SRS table and prover key are held once"}
    finally:
        clone.close()


CPU_BASELINE_RECORD = "r03_cpu_baseline_prove.json"


class progress_lines:
    """While the CPU oracle works for minutes nothing else would be printed: a line on stderr every 30 seconds shows a
    watching harness that the run is alive."""

    def __init__(self, what):
        self.what = what

    def __enter__(self):
        import threading
        self.stop = threading.Event()
        t0 = time.time()

        def beat():
            while not self.stop.wait(30.0):
                print("[bench] %s: %d s" % (self.what, time.time() - t0), file=sys.stderr, flush=True)

        self.thread = threading.Thread(target=beat, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        return False


def oracle_prove_seconds(log_gates, cores):
    """One proof of the benchmark's circuit family by the CPU oracle with the reference's parallel structure
    (rayon-parallel MSM windows and quotient evaluation, everything else serial) -> seconds of Prover::prove."""
    from oracle import pyoracle as orc
    from zkvm_amd import circuits
    n = 1 << log_gates
    circ = circuits.arithmetic_circuit(n, seed=5)
    circ["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(circ["sigma_map"][k]) for k in range(4)])
    with progress_lines("CPU oracle, 2^%d-gate key and proof" % log_gates):
        op = orc.Prover(n, circ["constraints"], circ["selectors"], tiled_srs(n + 7), b"bench", circ["pi_indexes"], threads=cores)
        circ["selectors"] = None
        t0 = time.perf_counter()
        r = op.prove(circ["wires"], circ["public_inputs"], circuits.random_blinders(1))
        secs = time.perf_counter() - t0
    assert r["rc"] == 0
    return secs


def cpu_baseline_prove(log_gates, full=False):
    """Times the oracle's Prover::prove.  By default on a 2^17-gate circuit of the same generator (a bounded sample:
    ten seconds of CPU work) scaled to 2^log_gates by the ratio MEASURED between the two sizes on the same kind of host
    (profiles/r03_cpu_baseline_prove.json, written by `bench.py --cpu-baseline-full`); with full=True the 2^log_gates
    proof itself is timed (minutes)."""
    cores = cpu_threads()
    log_s = min(17, log_gates)
    secs_s = oracle_prove_seconds(log_s, cores)
    out = {"unit": "proofs/s", "cores": cores, "kind": "port"}
    out.update(cpu_info())
    if log_s == log_gates:
        out["value"] = round(1.0 / secs_s, 6)
        out["sample"] = "one 2^%d-gate proof by the CPU oracle, measured in this run: %.2f s on %d threads" % (log_gates, secs_s, cores)
        return out
    if full:
        secs_full = oracle_prove_seconds(log_gates, cores)
        out["value"] = round(1.0 / secs_full, 6)
        out["sample"] = "one 2^%d-gate proof by the CPU oracle, measured: %.2f s on %d threads (2^%d gates: %.2f s)" % (
            log_gates, secs_full, cores, log_s, secs_s)
        out["measured"] = {"log_gates": log_gates, "seconds": round(secs_full, 3), "sample_log_gates": log_s,
                           "sample_seconds": round(secs_s, 3), "ratio": round(secs_full / secs_s, 4)}
        return out
    ratio, source = float(1 << (log_gates - log_s)), "linear in the gate count (no measured record found)"
    path = os.path.join(ROOT, "profiles", CPU_BASELINE_RECORD)
    if os.path.exists(path):
        rec = json.load(open(path))["cpu_baseline"]["measured"]
        if rec["log_gates"] == log_gates and rec["sample_log_gates"] == log_s:
            ratio = float(rec["ratio"])
            source = ("ratio %.3f measured between a 2^%d-gate (%.1f s) and a 2^%d-gate (%.1f s) oracle proof on the same "
                      "kind of host, profiles/%s" % (ratio, log_gates, rec["seconds"], log_s, rec["sample_seconds"],
                                                     CPU_BASELINE_RECORD))
    out["value"] = round(1.0 / (secs_s * ratio), 6)
    out["sample"] = "one 2^%d-gate proof by the CPU oracle took %.2f s on %d threads in this run; scaled to 2^%d gates by %s" % (
        log_s, secs_s, cores, log_gates, source)
    out["sample_seconds"] = round(secs_s, 3)
    out["scale"] = round(ratio, 4)
    return out


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
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time the CPU oracle on the full 2^log-gates proof (minutes) instead of the bounded sample")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="prove workload: leave out the msm / ntt / msm_scaling / sharded_prove records")
    ap.add_argument("--circuit", default="arithmetic", choices=["arithmetic", "widgets"],
                    help="prove workload: `widgets` times proofs of a key with every widget instead of the arithmetic circuit (one GPU) and "
                         "keeps the widget sub-record even under --no-widget-record")
    ap.add_argument("--no-widget-record", action="store_true", help="prove workload: leave out the widget-key record")
    ap.add_argument("--full-coset", action="store_true",
                    help="prove workload: evaluate the quotient over the reference's 8n coset even where 4n points suffice")
    ap.add_argument("--msm-window", type=int, default=0, help="tuning: force the MSM window width (0 = the engine's choice)")
    ap.add_argument("--msm-segment", type=int, default=0, help="tuning: log2 of the buckets per reduction segment (0 = default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no GPU call has been made in this process: the ranks start as fresh children
        sys.exit(self_launch(args.gpus))

    import torch
    rank, world, local = dist_setup(args.gpus)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    if args.workload == "prove":
        value, elapsed, extra, ctx = run_prove(args, rank, world, local, device)
        line = {"metric": "PLONK prove @2^%d gates" % args.log_gates, "unit": "proofs/s", "scaling": "weak",
                "dtype": "u32x8 Fr / u32x12 Fq (Montgomery)",
                "config": {"workload": "full PLONK prove, 2^%d gates (wire commits, permutation Z, quotient, "
                                       "linearization, openings), bit-exact vs the CPU prover" % args.log_gates,
                           "n": 1 << args.log_gates, "parallelism": "independent proofs per GPU"}}
        if args.circuit == "widgets" and world == 1 and args.log_gates >= 14:
            line["config"]["workload"] += "; circuit: every widget in use (zkvm_amd.circuits.widget_circuit)"
        if args.sharded and world > 1:
            line["scaling"] = "strong"
            line["config"]["parallelism"] = ("one proof on %d GPUs: SRS point ranges per GPU, 144-byte partial sums "
                                             "all-gathered over RCCL per commitment batch; the 8n coset dealt out cyclically "
                                             "(sharded quotient, four-step inverse transform with all-to-all); n-point work "
                                             "replicated" % world)
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
        from zkvm_amd import _lib
        line["library"] = os.path.relpath(_lib.LIB_PATH, ROOT)
        if _lib.LIB_OVERRIDDEN:
            line["library_note"] = "ZKGPU_LIB override: an A/B measurement, not this tree's library"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline_prove(args.log_gates, args.cpu_baseline_full) if args.workload == "prove" else
                                    cpu_baseline_ntt() if args.workload == "ntt" else cpu_baseline_msm())
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
