"""Phase times of one MSM over window widths, for tuning MsmEngine::choose_window:
   python tools/msm_window_sweep.py LOG_N c1 c2 ...     (c = 0: the engine's own choice)
   DUP=k in the environment makes the first k scalars equal (k-entry buckets on every level)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import zkvm_amd
    from zkvm_amd import circuits
    log_n = int(sys.argv[1])
    cs = [int(x) for x in sys.argv[2:]]
    n = 1 << log_n
    device = torch.device("cuda", 0)
    ctx = zkvm_amd.Context(0)
    host = bench.random_fr(n, 3)
    dup = int(os.environ.get("DUP", "0"))
    if dup:
        host[:dup] = host[0]
    sc = bench.to_dev(host, device)
    for c in cs:
        ctx.msm_set_params(c, 0)
        ctx.srs_setup(circuits.seeded_tau(), circuits.g1_generator_limbs(), n)
        for _ in range(3):
            ctx.msm_kernels(sc)
        ctx.sync()
        ctx.timer_start()
        for _ in range(10):
            ctx.msm_kernels(sc)
        ms = ctx.timer_stop() / 10
        ph = [ctx.msm_profile(sc) for _ in range(5)]
        avg = {k: sum(p[k] for p in ph) / 5 for k in ph[0]}
        print("log_n=%d c=%2d K=%2d  device %.3f ms   " % (log_n, c, ctx.msm_get_params()[1], ms)
              + "  ".join("%s %.3f" % kv for kv in avg.items()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
