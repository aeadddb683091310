"""Randomised differential check of the NTT against the CPU oracle: random sizes up to 2^17, random input lengths
(shorter and longer than the domain), all four modes, random tile geometry.
    python tools/fuzz_ntt.py [ITERATIONS] [SEED]"""
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
    modes = ["fft", "ifft", "coset_fft", "coset_ifft"]
    tiles = [(0, 0), (6, 1), (7, 2), (8, 1), (9, 1), (9, 2), (10, 2), (11, 1), (12, 2)]
    for it in range(iters):
        log_n = int(rng.integers(0, 18))
        n = 1 << log_n
        kind = rng.random()
        if kind < 0.4:
            length = int(rng.integers(0, n + 1))
        elif kind < 0.6:
            length = n
        elif kind < 0.8:
            length = min(n, n // 8 + int(rng.integers(0, 8)))
        else:
            length = n + int(rng.integers(1, 50))
        x = M.random_fr_mont(max(length, 1), seed * 100003 + it)[:length]
        if length and rng.random() < 0.2:
            x[rng.integers(0, length, size=max(1, length // 3))] = 0
        mode = int(rng.integers(0, 4))
        tile = tiles[int(rng.integers(0, len(tiles)))]
        ctx.ntt_set_params(*tile)
        got = ctx.ntt(x, log_n, mode)
        want = orc.fft(modes[mode], x, n)
        if not np.array_equal(got, want):
            print("MISMATCH at iteration", it, "seed", seed, "log_n", log_n, "length", length, modes[mode], "tile", tile)
            sys.exit(1)
        if it % 100 == 0:
            print("iteration", it, "ok", flush=True)
    ctx.ntt_set_params(0, 0)
    ctx.close()
    print("fuzz ok:", iters, "transforms, seed", seed)


if __name__ == "__main__":
    main()
