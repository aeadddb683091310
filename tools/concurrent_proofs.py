"""Throughput of small proofs with several contexts (one per host thread, each with its own stream) on one GPU:
a 2^16-gate proof is bound by launch and dependency latency, so independent proofs overlap.
    python tools/concurrent_proofs.py LOG_GATES THREADS [PROOFS_PER_THREAD] [START_AT]
START_AT (seconds since the epoch) holds the timed loop back until that moment, so that several processes started
together (each has its own hardware queues) measure over the same window; their rates then add up."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import zkvm_amd
    from zkvm_amd import api, circuits
    log_n, threads = int(sys.argv[1]), int(sys.argv[2])
    per_thread = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    start_at = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    n = 1 << log_n
    circ = circuits.arithmetic_circuit(n, seed=3)
    workers = []
    for t in range(threads):
        ctx = zkvm_amd.Context(0)
        ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n + 7)
        if t == 0:
            circuits.fill_sigma_rows(circ, ctx)
        gp = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"conc", circ["pi_indexes"])
        wires = torch.from_numpy(circ["wires"].reshape(-1, 4).view(np.int64)).cuda()
        workers.append((ctx, gp, wires))
    bl = circuits.random_blinders(1)
    digests = [None] * threads

    def run(k, count):
        ctx, gp, wires = workers[k]
        for _ in range(count):
            out = gp.prove(wires, circ["public_inputs"], bl)
        digests[k] = out["comms"].tobytes()

    for k in range(threads):
        run(k, 2)                       # warm-up, serially
    ts = [threading.Thread(target=run, args=(k, per_thread)) for k in range(threads)]
    if start_at:
        assert time.time() < start_at, "set up too late for the common start"
        time.sleep(start_at - time.time())
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    elapsed = time.perf_counter() - t0
    assert all(d == digests[0] for d in digests), "proofs differ between contexts"
    total = threads * per_thread
    print("2^%d gates, %d threads: %d proofs in %.3f s -> %.1f proofs/s (%.2f ms per proof per thread)%s"
          % (log_n, threads, total, elapsed, total / elapsed, elapsed / per_thread * 1e3,
             "  [window %.3f .. %.3f]" % (start_at, start_at + elapsed) if start_at else ""), flush=True)
    for ctx, _, _ in workers:
        ctx.close()


if __name__ == "__main__":
    main()
