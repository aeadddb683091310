"""The reference's own workload on the GPU prover: a Poseidon-Merkle tree, one circuit compiled once, then one proof
per opened leaf, each checked by the verifier (merkle-plonk/src/lib.rs:275-332 proves one opening per proof in a
loop; circuit: merkle-plonk/src/lib.rs:117-131).
    python tools/merkle_opening.py HEIGHT [OPENINGS] [CONTEXTS]
A tree of height H has 4^H leaves; its opening circuit has about 1 010 gates per level.  With CONTEXTS > 1 a second
leg runs the same loop on that many host threads, each with a context and a prover of its own on the one GPU and each
generating its witnesses itself: the openings per second of a service that proves one request after another."""
import os
import random
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import model as M
    import plonk_check
    import zkvm_amd
    from oracle import pyoracle as orc
    from zkvm_amd import api, circuits, poseidon as P
    height = int(sys.argv[1])
    openings = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    contexts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    label = b"merkle-opening"

    tree = P.Tree(height)
    rng = random.Random(height)
    positions = sorted({rng.randrange(tree.capacity()) for _ in range(openings)})
    t0 = time.time()
    for p in positions:
        tree.insert(p, P.digest(P.Domain.OTHER, [p])[0])
    root = tree.root()
    print("tree of height %d, %d leaves set, root %#x  (%.2f s)" % (height, len(positions), root, time.time() - t0),
          flush=True)

    # compile once: the default circuit of the reference opens leaf 0 of a tree holding one empty item
    blank = P.Tree(height)
    blank.insert(0, 0)
    t0 = time.time()
    circ = P.opening_circuit(blank.opening(0), 0).tables()
    n = circ["n"]
    print("circuit: %d gates, n = 2^%d  (%.2f s)" % (circ["constraints"], n.bit_length() - 1, time.time() - t0), flush=True)

    ctx = zkvm_amd.Context(0)
    tau = M.random_fr_mont(1, seed=0xdea1)[0]
    gxy, _ = orc.g1_to_affine(orc.g1_generator())
    ctx.srs_setup(tau, gxy, n + 7)
    circuits.fill_sigma_rows(circ, ctx)
    t0 = time.time()
    gp = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), label, circ["pi_indexes"])
    ctx.sync()
    print("key %.3f s" % (time.time() - t0), flush=True)
    sxy, sinf = gp.selector_commitments()
    tau_int = M.fr_from_mont(tau)
    h, tau_h = plonk_check.opening_key_g2(tau_int)

    times = []
    buffers = None
    for k, p in enumerate(positions):
        # the witness of one opening from the library's native front-end: the wire columns and the public root, which
        # is all that changes from one opening to the next (the reference reruns the circuit per proof for them,
        # plonk/src/composer/prover.rs:214-238).  Checked on the first leaf against the gate-by-gate Python Composer.
        t0 = time.time()
        buffers = P.batch_opening_witness([tree.opening(p)], [tree.leaves[p]], threads=1, out=buffers)
        host_s = time.time() - t0
        this = dict(wires=buffers[0], public_inputs=buffers[1])
        if k == 0:
            slow = P.opening_circuit(tree.opening(p), tree.leaves[p]).tables()
            assert np.array_equal(slow["wires"], this["wires"]) and np.array_equal(slow["public_inputs"], this["public_inputs"])
            assert np.array_equal(slow["selectors"][:11], circ["selectors"][:11])
            assert np.array_equal(slow["sigma_map"], circ["sigma_map"]) and slow["pi_indexes"] == circ["pi_indexes"]
        wires = torch.from_numpy(this["wires"].reshape(-1, 4).view(np.int64)).cuda()
        bl = circuits.random_blinders(0xdea1 + k)
        if k == 0:
            gp.prove(wires, this["public_inputs"], bl)                     # warm-up
        proof = gp.prove(wires, this["public_inputs"], bl, timings=True)
        times.append(proof["timings_ms"]["total"])
        assert M.fr_from_mont(this["public_inputs"][0]) == root
        ok = plonk_check.verify(n, circ["constraints"], label, proof, sxy, sinf, this["public_inputs"],
                                circ["pi_indexes"], gxy, h, tau_h, orc)
        assert ok, "verifier rejects the opening of leaf %d" % p
        print("leaf %d: witness %.1f ms on the host, proof %.2f ms, accepted" % (p, 1e3 * host_s, times[-1]), flush=True)
    print("height %d: %d openings proved and verified, %.2f ms per proof (median)" % (height, len(times), float(np.median(times))),
          flush=True)
    if contexts > 1:
        service(contexts, height, tree, positions, circ, label, tau, gxy, root, (sxy, sinf, h, tau_h))
    ctx.close()


def service(contexts, height, tree, positions, circ, label, tau, gxy, root, verifier_side, rounds=24):
    """`contexts` host threads, each: witness of the next leaf (one core), copy to the GPU, prove.  Every thread's last
    proof goes through the verifier."""
    import torch
    import model as M
    import plonk_check
    import zkvm_amd
    from oracle import pyoracle as orc
    from zkvm_amd import api, circuits, poseidon as P
    n = circ["n"]
    sxy, sinf, h, tau_h = verifier_side
    # one context loads the SRS and the key; the others are clones of it (zkgpu_ctx_clone = the reference's
    # `#[derive(Clone)] Prover`): a stream and per-proof buffers each, one SRS table and one key on the GPU
    first = zkvm_amd.Context(0)
    first.srs_setup(tau, gxy, n + 7)
    gp0 = api.Prover(first, n, circ["constraints"], circ["selectors"].reshape(-1, 4), label, circ["pi_indexes"])
    first.sync()
    workers = [(first, gp0)]
    for _ in range(contexts - 1):
        ctx = first.clone()
        workers.append((ctx, gp0.on(ctx)))
    print("%d contexts: %.2f GiB of table and key held once, %.2f GiB of buffers per clone" % (
        contexts, first.device_bytes_shared() / 2.0**30,
        (workers[-1][0].device_bytes() - workers[-1][0].device_bytes_shared()) / 2.0**30), flush=True)
    openings = [(tree.opening(p), tree.leaves[p]) for p in positions]
    last = [None] * contexts

    def run(k, count):
        _, gp = workers[k]
        buffers = None
        for i in range(count):
            opening, leaf = openings[(k + i * contexts) % len(openings)]
            buffers = P.batch_opening_witness([opening], [leaf], threads=1, out=buffers)
            wires = torch.from_numpy(buffers[0].reshape(-1, 4).view(np.int64)).cuda()
            proof = gp.prove(wires, buffers[1], circuits.random_blinders(k * 1000 + i))
        last[k] = (proof, buffers[1].copy())

    for k in range(contexts):
        run(k, 2)                       # warm-up, serially
    threads = [threading.Thread(target=run, args=(k, rounds)) for k in range(contexts)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    elapsed = time.perf_counter() - t0
    for k, (proof, pi) in enumerate(last):
        assert M.fr_from_mont(pi[0]) == root
        assert plonk_check.verify(n, circ["constraints"], label, proof, sxy, sinf, pi, circ["pi_indexes"], gxy, h, tau_h, orc), k
    total = contexts * rounds
    print("height %d, %d contexts on one GPU, witnesses generated in the loop: %d openings in %.3f s -> %.0f openings/s; "
          "the last proof of every thread accepted" % (height, contexts, total, elapsed, total / elapsed), flush=True)
    for ctx, _ in workers:
        ctx.close()


if __name__ == "__main__":
    main()
