"""Times the MSM device phase over window and segment widths:  python tools/msm_sweep.py LOG_N [c ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import zkvm_amd
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cs = [int(x) for x in sys.argv[2:]] or [16, 18, 19, 20, 21]
    n = 1 << log_n
    device = torch.device("cuda", 0)
    ctx = zkvm_amd.Context(0)
    pts = bench.tiled_srs(n, device)
    sc = bench.to_dev(bench.random_fr(n, 3), device)
    for c in cs:
        for seg in (2, 3, 4):
            ctx.msm_set_params(c, seg)
            if seg == 2:
                ctx.sync()
                ctx.timer_start()
                ctx.srs_load(pts)
                t_load = ctx.timer_stop()
            ctx.msm_kernels(sc)
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                ctx.msm_kernels(sc)
            ms = ctx.timer_stop() / 5
            print("log_n=%d c=%2d seg=%d  device %.3f ms  (%.1f Mpts/s)  table build %.1f ms  device bytes %.2f GiB"
                  % (log_n, c, seg, ms, n / ms / 1e3, t_load, ctx.device_bytes() / 2**30), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
