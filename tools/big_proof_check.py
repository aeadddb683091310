"""Proves one arithmetic circuit of 2^LOG_N gates on the GPU and runs the reference verifier's check on the proof
(tests/plonk_check.py: the final equation under the SRS trapdoor, then the whole verifier with its pairing):
    python tools/big_proof_check.py LOG_N
For sizes the test-suite does not reach (2^22, 2^24: the 8n domain has 2^27 points there)."""
import os
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
    from zkvm_amd import api, circuits
    log_n = int(sys.argv[1])
    n = 1 << log_n
    ctx = zkvm_amd.Context(0)
    tau = M.random_fr_mont(1, seed=4040)[0]
    gxy, _ = orc.g1_to_affine(orc.g1_generator())
    t0 = time.time()
    ctx.srs_setup(tau, gxy, n + 7)
    print("srs %.2f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    c = circuits.arithmetic_circuit(n, seed=log_n)
    circuits.fill_sigma_rows(c, ctx)
    print("circuit %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    gp = api.Prover(ctx, n, c["constraints"], c["selectors"].reshape(-1, 4), b"big-proof", c["pi_indexes"])
    print("key %.2f s, device memory %.1f GiB" % (time.time() - t0, ctx.device_bytes() / 2**30), flush=True)
    sxy, sinf = gp.selector_commitments()
    c["selectors"] = None
    wires = torch.from_numpy(c["wires"].reshape(-1, 4).view(np.int64)).cuda()
    bl = circuits.random_blinders(log_n)
    proof = gp.prove(wires, c["public_inputs"], bl, timings=True)
    proof = gp.prove(wires, c["public_inputs"], bl, timings=True)
    print("prove %.1f ms" % proof["timings_ms"]["total"], proof["timings_ms"], flush=True)
    pi0 = M.fr_from_mont(c["public_inputs"][0])
    tau_int = M.fr_from_mont(tau)
    left, right = plonk_check.verifier_sides(n, proof, sxy, sinf, pi0, tau_int, gxy)
    assert left == right, "verifier equation fails"
    left, right = plonk_check.verifier_sides(n, proof, sxy, sinf, pi0, tau_int, gxy, tamper=1)
    assert left != right
    h, tau_h = plonk_check.opening_key_g2(tau_int)
    ok = plonk_check.verify(n, c["constraints"], b"big-proof", proof, sxy, sinf, c["public_inputs"], c["pi_indexes"], gxy,
                            h, tau_h, orc)
    assert ok, "pairing verifier rejects"
    print("2^%d gates: proof accepted by the verifier" % log_n, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
