#!/usr/bin/env python3
"""Times the proof and the quotient kernels of a 2^log-row key in which every widget has gates
(zkvm_amd.circuits.widget_circuit) with the library ZKGPU_LIB names (default: this tree's), for comparing builds:

    python tools/widget_quotient_ab.py [log_rows]                              # this tree
    ZKGPU_LIB=tools/ab_libzkgpu_r02.so python tools/widget_quotient_ab.py     # an older build
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import zkvm_amd
    from zkvm_amd import api, circuits
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << log_n
    ctx = zkvm_amd.Context(0)
    ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n + 7)
    circ = circuits.widget_circuit(n)
    circuits.fill_sigma_rows(circ, ctx)
    prover = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"ab", circ["pi_indexes"])
    w = torch.from_numpy(circ["wires"].reshape(-1, 4).view(np.int64)).cuda()
    bl = circuits.random_blinders(2)
    first = prover.prove(w, circ["public_inputs"], bl)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        last = prover.prove(w, circ["public_inputs"], bl)
    ctx.sync()
    ms = 1e3 * (time.perf_counter() - t0) / 5
    assert np.array_equal(first["comms"], last["comms"])
    print("library %s: 2^%d rows, every widget: %.3f ms per proof, quotient kernels %.3f ms, a_comm %s z_eval %s" % (
        zkvm_amd.LIB_PATH, log_n, ms, ctx.quotient_kernel_ms(), hex(int(last["comms"][0][0])), hex(int(last["evals"][14][0]))))
    ctx.close()


if __name__ == "__main__":
    main()
