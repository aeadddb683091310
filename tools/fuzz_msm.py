"""Randomised differential check of the MSM against the CPU oracle: small point sets with repeated and negated
points, points at infinity, scalars that collide (equal, opposite, 0, 1, r - 1, small) and random window widths; every
case also as a ragged batch of up to four commitments (zkgpu_commit_many).
    python tools/fuzz_msm.py [ITERATIONS] [SEED]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import model as M
    import zkvm_amd
    from oracle import pyoracle as orc
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.Generator(np.random.PCG64(seed))
    ctx = zkvm_amd.Context(0)
    base_pts, _ = orc.g1_multiples_of_generator(1, 64)
    neg = base_pts.copy()
    for i in range(64):                                     # -P: y -> p - y (Montgomery form is linear)
        y = M.limbs_to_int(neg[i, 6:])
        neg[i, 6:] = M.int_to_limbs((M.P_MOD - y) % M.P_MOD, 6)
    pool = np.concatenate([base_pts, neg])
    specials = [0, 1, 2, M.R_MOD - 1, M.R_MOD - 2, (1 << 20) - 1, 1 << 20, (1 << 255) % M.R_MOD]
    for it in range(iters):
        n = int(rng.integers(1, 400))
        idx = rng.integers(0, 128 if rng.random() < 0.5 else 4, size=n)
        pts = pool[idx].copy()
        inf = (rng.random(n) < 0.05).astype(np.uint8)
        vals = []
        common = int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62))
        for k in range(n):
            u = rng.random()
            if u < 0.25:
                vals.append(specials[int(rng.integers(0, len(specials)))])
            elif u < 0.5:
                vals.append(common if rng.random() < 0.5 else (M.R_MOD - common) % M.R_MOD)
            elif u < 0.6:
                vals.append(int(rng.integers(0, 1 << 16)))
            else:
                vals.append(int.from_bytes(rng.bytes(32), "little") % M.R_MOD)
        sc = M.fr_vec_to_mont(vals)
        window = int(rng.choice([0, 0, 4, 5, 7, 9, 12, 13, 16, 17, 20]))
        seg = int(rng.integers(0, 5))
        ctx.msm_set_params(window, seg)
        ctx.srs_load(pts, inf)
        got_xy, got_inf = ctx.msm(sc)
        want_xy, want_inf = orc.g1_to_affine(orc.msm(pts, sc, inf=inf, threads=4))
        if got_inf != want_inf or not np.array_equal(got_xy, want_xy):
            print("MISMATCH at iteration", it, "seed", seed, "n", n, "window", window, "segment", seg)
            np.savez("gpurun_out/fuzz_fail.npz", pts=pts, inf=inf, sc=sc, window=window, seg=seg)
            sys.exit(1)
        # the same scalars cut into a ragged batch of up to four commitments (one batched pass on the GPU)
        m = int(rng.integers(2, 5))
        polys, lens = [], []
        for _ in range(m):
            k = int(rng.integers(2, n + 1)) if n >= 2 else 0
            if k < 2:
                break
            poly = sc[:k].copy()
            if not poly[1:].any():
                poly[k - 1] = M.fr_to_mont(3)          # degree at least one
            polys.append(poly)
            lens.append(k)
        if polys:
            try:
                got = ctx.commit_many(polys)
            except (zkvm_amd.ZkGpuError, ValueError) as e:
                print("ERROR in commit_many at iteration", it, "seed", seed, "lens", lens, e)
                sys.exit(1)
            for j, poly in enumerate(polys):
                top = max(i for i in range(lens[j]) if poly[i].any()) + 1
                wx, wi = orc.g1_to_affine(orc.msm(pts[:top], poly[:top], inf=inf[:top], threads=4))
                if got[j][1] != wi or not np.array_equal(got[j][0], wx):
                    print("BATCH MISMATCH at iteration", it, "seed", seed, "lens", lens, "j", j, "window", window, "segment", seg)
                    sys.exit(1)
        if it % 50 == 0:
            print("iteration", it, "ok", flush=True)
    ctx.msm_set_params(0, 0)
    ctx.close()
    print("fuzz ok:", iters, "iterations, seed", seed)


if __name__ == "__main__":
    main()
