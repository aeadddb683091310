"""The reference's own workload on the GPU prover: a Poseidon-Merkle tree, one circuit compiled once, then one proof
per opened leaf, each checked by the verifier (merkle-plonk/src/lib.rs:275-332 proves one opening per proof in a
loop; circuit: merkle-plonk/src/lib.rs:117-131).
    python tools/merkle_opening.py HEIGHT [OPENINGS]
A tree of height H has 4^H leaves; its opening circuit has about 1 010 gates per level."""
import os
import random
import sys
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
    ctx.close()


if __name__ == "__main__":
    main()
