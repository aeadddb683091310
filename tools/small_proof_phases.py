"""Phase times of one small proof (median of several):  python tools/small_proof_phases.py LOG_GATES"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import zkvm_amd
    from zkvm_amd import api, circuits
    log_n = int(sys.argv[1])
    n = 1 << log_n
    circ = circuits.arithmetic_circuit(n, seed=3)
    ctx = zkvm_amd.Context(0)
    ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n + 7)
    circuits.fill_sigma_rows(circ, ctx)
    gp = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"phases", circ["pi_indexes"])
    wires = torch.from_numpy(circ["wires"].reshape(-1, 4).view(np.int64)).cuda()
    bl = circuits.random_blinders(1)
    runs = [gp.prove(wires, circ["public_inputs"], bl, timings=True)["timings_ms"] for _ in range(15)][3:]
    med = {k: float(np.median([r[k] for r in runs])) for k in runs[0]}
    print("2^%d gates: " % log_n + "  ".join("%s %.2f" % kv for kv in med.items()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
