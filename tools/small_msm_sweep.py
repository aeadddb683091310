"""Phase times of small MSMs against the window width and the segment width of the bucket reduction:
    python tools/small_msm_sweep.py LOG_N [WINDOWS [SEGMENTS]]      (comma-separated lists; 0 = automatic)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import zkvm_amd
    from zkvm_amd import circuits
    log_n = int(sys.argv[1])
    lists = [[int(x) for x in a.split(",")] for a in sys.argv[2:]]
    windows = lists[0] if len(lists) > 0 else [0]
    segments = lists[1] if len(lists) > 1 else [0]
    n = 1 << log_n
    sc = torch.from_numpy(circuits.random_mont(np.random.default_rng(log_n), n).view(np.int64)).cuda()
    want = None
    for window in windows:
        ctx = zkvm_amd.Context(0)
        ctx.msm_set_params(window, 0)
        ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n)
        rows = []
        for seg in segments:
            ctx.msm_set_params(window, seg)
            got = ctx.msm(sc)
            want = want if want is not None else got
            assert np.array_equal(got[0], want[0]), "result changed"
            best = None
            for _ in range(5):
                ms = ctx.msm_profile(sc)
                if best is None or sum(ms.values()) < sum(best.values()):
                    best = ms
            rows.append((sum(best.values()), seg, best))
        for total, seg, best in sorted(rows, key=lambda r: r[0])[:4]:
            print("2^%d window %s seg %d: " % (log_n, ctx.msm_get_params(), seg)
                  + "  ".join("%s %.3f" % kv for kv in best.items()) + "  total %.3f" % total, flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
