"""Randomised differential check of the whole prover: circuits built from random sequences of Composer operations
(arithmetic gates, booleans, selects, range and logic components, fixed-base multiplications, curve additions, public
inputs), proved on the
GPU and by the CPU oracle with the same blinders; every commitment, evaluation and challenge must agree.  One circuit
in seven gets a corrupted witness value or public input: both sides must then agree on success or failure as well
(the GPU decides there between its 4n-coset and 8n-coset quotient).
    python tools/fuzz_prove.py [ITERATIONS] [SEED]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_circuit(rng):
    from zkvm_amd import composer as C
    R = C.R
    c = C.Composer.initialized()
    pool = [c.append_witness(int.from_bytes(rng.bytes(32), "little") % R) for _ in range(3)]
    small = [c.append_witness(int(rng.integers(0, 1 << 30))) for _ in range(2)]
    points = []
    ops = int(rng.integers(1, 40))
    use_widgets = rng.random() < 0.5
    for _ in range(ops):
        u = rng.random()
        a, b = (pool[int(rng.integers(0, len(pool)))] for _ in range(2))
        coef = lambda: int(rng.integers(-5, 6))
        if u < 0.3:
            pool.append(c.gate_mul(C.Constraint().mult(coef() or 1).constant(coef()).a(a).b(b)))
        elif u < 0.55:
            d = pool[int(rng.integers(0, len(pool)))]
            pool.append(c.gate_add(C.Constraint().left(coef()).right(coef()).fourth(coef()).constant(coef()).a(a).b(b).d(d)))
        elif u < 0.62:
            bit = c.append_witness(int(rng.integers(0, 2)))
            c.component_boolean(bit)
            pool.append(c.component_select(bit, a, b))
        elif u < 0.7:
            pool.append(c.append_public(c[a]))
            c.assert_equal(pool[-1], a)
        elif u < 0.75:
            pool.append(c.append_constant(int(rng.integers(0, 1000))))
        elif use_widgets and u < 0.83:
            w = small[int(rng.integers(0, len(small)))]
            c.component_range(w, 16)
        elif use_widgets and u < 0.92:
            x, y = small[int(rng.integers(0, len(small)))], small[int(rng.integers(0, len(small)))]
            small.append(c.append_logic_xor(x, y, 16) if rng.random() < 0.5 else c.append_logic_and(x, y, 16))
        elif use_widgets:
            if len(points) < 2:
                points.append(c.append_point(C.jubjub_mul(C.JUBJUB_GENERATOR, int(rng.integers(1, 1 << 40)))))
            elif len(points) == 2 and rng.random() < 0.5:
                points.append(c.component_mul_generator(c.append_witness(int(rng.integers(1, 1 << 62)))))
            else:
                points.append(c.component_add_point(points[int(rng.integers(0, len(points)))], points[-1]))
    return c.tables()


def main():
    import model as M
    import zkvm_amd
    from oracle import pyoracle as orc
    from zkvm_amd import api, circuits
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.Generator(np.random.PCG64(seed))
    tau = M.random_fr_mont(1, seed=seed + 1000)[0]
    srs, _ = orc.srs_powers(orc.g1_generator(), tau, 2048 + 7)
    ctx = zkvm_amd.Context(0)
    routes = {0: 0, 4: 0, 8: 0}
    corrupted = 0
    for it in range(iters):
        circ = random_circuit(rng)
        n = circ["n"]
        if rng.random() < 1 / 7:
            corrupted += 1
            u = rng.random()
            if u < 0.4:
                # one variable changed in every cell of its copy cycle: the copy constraints still hold, only gates break
                # (in a widget row: only the row-wise widget check tells the GPU to leave the 4n coset)
                circ["wires"] = circ["wires"].copy()
                sigma = circ["sigma_map"].reshape(-1)
                cells = [int(rng.integers(0, 4)) * n + int(rng.integers(0, circ["constraints"]))]
                while int(sigma[cells[-1]]) != cells[0]:
                    cells.append(int(sigma[cells[-1]]))
                value = (M.fr_from_mont(circ["wires"][cells[0] // n, cells[0] % n]) + 1 + int(rng.integers(0, 7))) % M.R_MOD
                for cell in cells:
                    circ["wires"][cell // n, cell % n] = M.fr_to_mont(value)
            elif u < 0.85 or not len(circ["public_inputs"]):
                circ["wires"] = circ["wires"].copy()
                circ["wires"][int(rng.integers(0, 4)), int(rng.integers(0, circ["constraints"])), 0] ^= np.uint64(1 << int(rng.integers(0, 60)))
            else:
                circ["public_inputs"] = circ["public_inputs"].copy()
                circ["public_inputs"][int(rng.integers(0, len(circ["public_inputs"]))), 0] ^= np.uint64(4)
        circ["selectors"][11:15] = np.stack([orc.sigma_to_lagrange(circ["sigma_map"][k]) for k in range(4)])
        op = orc.Prover(n, circ["constraints"], circ["selectors"], srs[:n + 7], b"fuzz", circ["pi_indexes"], threads=4)
        bl = circuits.random_blinders(it)
        want = op.prove(circ["wires"], circ["public_inputs"], bl)
        ctx.srs_load(srs[:n + 7])
        rc = 0
        try:
            gp = api.Prover(ctx, n, circ["constraints"], circ["selectors"].reshape(-1, 4), b"fuzz", circ["pi_indexes"])
            got = gp.prove(circ["wires"].reshape(-1, 4), circ["public_inputs"], bl)
        except zkvm_amd.ZkGpuError as e:
            rc, got = e.code, None
        except (api.PolynomialDegreeTooLarge, api.PolynomialDegreeIsZero) as e:   # the reference's own error variants
            rc, got = type(e).__name__, None
        routes[got["coset"] if got is not None else 0] += 1
        ok = (want["rc"] == 0) == (got is not None)
        if ok and got is not None:
            ok = all(np.array_equal(got[k], want[k]) for k in ("challenges", "evals", "comms_inf", "comms"))
        if not ok:
            print("MISMATCH at iteration", it, "seed", seed, "n", n, "oracle rc", want["rc"], "gpu rc", rc)
            sys.exit(1)
        if it % 20 == 0:
            print("iteration", it, "ok (n = %d, %d gates)" % (n, circ["constraints"]), flush=True)
    ctx.close()
    print("fuzz ok:", iters, "circuits, seed", seed, "- proofs over the 4n coset:", routes[4], "over the 8n coset:", routes[8],
          "refused by both sides:", routes[0], "- corrupted inputs:", corrupted)


if __name__ == "__main__":
    main()
