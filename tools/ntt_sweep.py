"""Times the four transforms over tile sizes:  python tools/ntt_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import zkvm_amd
    device = torch.device("cuda", 0)
    ctx = zkvm_amd.Context(0)
    n = 1 << 20
    x = bench.to_dev(bench.random_fr(n, 1), device)
    c = bench.to_dev(bench.random_fr(n + 3, 2), device)
    b1 = torch.empty((n, 4), dtype=torch.int64, device=device)
    b8 = torch.empty((8 * n, 4), dtype=torch.int64, device=device)
    ref = {}
    combos = [(12, 1), (12, 2), (11, 1), (11, 2), (10, 1), (10, 2), (9, 1)]
    if len(sys.argv) > 1:
        combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for _ in range(30):                       # let the clocks settle before anything is timed
        ctx.ntt(c, 23, 2, out=b8)
    ctx.sync()
    for tile, ept in combos + combos:
        ctx.ntt_set_params(tile, ept)
        out = []
        for mode, lg, src, dst in ((0, 20, x, b1), (1, 20, x, b1), (2, 23, c, b8), (3, 23, b8, b8)):
            ctx.ntt(src, lg, mode, out=dst)
            ctx.sync()
            if mode in (0, 2):
                chk = int(dst.sum().item())
                assert ref.setdefault(mode, chk) == chk, "result changed with the tile size"
            ctx.timer_start()
            for _ in range(10):
                ctx.ntt(src, lg, mode, out=dst)
            out.append(ctx.timer_stop() / 10)
        gb = (2 * 2 * n * 32 + 2 * 2 * 8 * n * 32) / 1e9
        print("tile 2^%d, %d elems/thread: fft20 %.3f  ifft20 %.3f  coset_fft23 %.3f  coset_ifft23 %.3f ms  -> %.0f GB/s"
              % (tile, 1 << ept, *out, gb / (sum(out) * 1e-3)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
