"""Device time of a 2^k-point MSM by phase for uniform and for witness-shaped scalars (30 % zeros, 20 % ones, 20 %
below 2^16, 30 % uniform): repeated scalars pile their digits on few buckets and few sort ranges.
    python tools/skew_msm.py [LOG_POINTS]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import model as M
    import zkvm_amd
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << log_n
    ctx = zkvm_amd.Context(0)
    from zkvm_amd import circuits
    ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n)
    rng = np.random.Generator(np.random.PCG64(5))
    uni = M.random_fr_mont(n, 3)
    wit = uni.copy()
    kind = rng.random(n)
    wit[kind < 0.3] = 0
    wit[(kind >= 0.3) & (kind < 0.5)] = M.fr_to_mont(1)
    mask = (kind >= 0.5) & (kind < 0.7)
    wit[mask] = M.fr_vec_to_mont([int(v) for v in rng.integers(0, 1 << 16, size=int(mask.sum()))])
    for name, sc in (("uniform", uni), ("witness-shaped", wit)):
        sd = torch.from_numpy(sc.view(np.int64)).cuda()
        ctx.msm_profile(sd)
        best = None
        for _ in range(3):
            ph = ctx.msm_profile(sd)
            tot = sum(v for k, v in ph.items() if k != "sort")
            if best is None or tot < best[0]:
                best = (tot, ph)
        print(name, "total %.3f ms" % best[0], {k: round(v, 3) for k, v in best[1].items()}, flush=True)


if __name__ == "__main__":
    main()
