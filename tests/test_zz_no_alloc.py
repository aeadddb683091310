"""zkgpu_ntt_dev_noalloc - the transform for callers that cannot allocate - against zkgpu_ntt_dev on the same inputs,
bit for bit, through the C ABI: two calls in a row per mode, the second on a smaller input (the static scratch array
keeps the first call's contents), against the oracle at every size for the forward transform (the allocating entry
point shares the pass code, so a common bug would hide behind the self-comparison), with the library's device bytes and
the GPU's free memory unchanged across each call, with two contexts on two host threads sharing the GPU's scratch array,
plus the argument checks."""
import threading

import numpy as np
import pytest

import model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import zkvm_amd
    c = zkvm_amd.Context(0)
    yield c
    c.close()


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_noalloc_matches_allocating_entry_point(ctx, orc, mode):
    import torch
    name = ["fft", "ifft", "coset_fft", "coset_ifft"][mode]
    # big first, then smaller: sizes chosen so that one, two and three passes occur (tiles hold 2^9 elements)
    for log_n, len_in in ((20, (1 << 20) - 5), (14, 1 << 14), (18, (1 << 17) + 3), (9, 300), (4, 16)):
        n = 1 << log_n
        x = M.random_fr_mont(len_in, 7000 + 10 * log_n + mode)
        xd = dev(x)
        want = ctx.ntt(xd, log_n, mode)                      # allocating entry point; also leaves the tables resident
        got = torch.full((n, 4), -1, dtype=torch.int64, device="cuda")
        ctx.sync()
        torch.cuda.synchronize()
        held, free = ctx.device_bytes(), torch.cuda.mem_get_info()[0]
        ctx.ntt_noalloc(xd, log_n, mode, got)
        ctx.sync()
        assert ctx.device_bytes() == held and torch.cuda.mem_get_info()[0] == free, "the entry point allocated"
        assert torch.equal(got, want), (name, log_n)
        # the oracle's serial transform needs ~1 s at 2^20: every size for fft, the small ones for the other modes
        if log_n <= 14 or mode == 0 or (mode == 2 and log_n == 18):
            assert np.array_equal(got.cpu().numpy().view(np.uint64), orc.fft(name, x, n)), (name, log_n)
    # in place
    x = M.random_fr_mont(1 << 12, 99)
    xd = dev(x)
    want = ctx.ntt(xd, 12, mode)
    ctx.ntt_noalloc(xd, 12, mode, xd)
    ctx.sync()
    assert torch.equal(xd, want)


def test_two_contexts_on_two_threads_share_the_scratch(orc):
    """Each context has its own stream; the GPU's one scratch array is handed from call to call by the library's
    per-device event, so transforms issued at once from two host threads come out right (multi-pass sizes: the array is
    live between the passes)."""
    import torch
    import zkvm_amd
    ctxs = [zkvm_amd.Context(0), zkvm_amd.Context(0)]
    try:
        sizes = (18, 20, 19, 20, 18)
        ins = [[dev(M.random_fr_mont((1 << lg) - 3 * t, 8100 + 10 * t + i)) for i, lg in enumerate(sizes)] for t in range(2)]
        modes = [[(i + t) % 3 for i in range(len(sizes))] for t in range(2)]      # fft, ifft, coset_fft: tables serve any size
        want = [[ctxs[t].ntt(ins[t][i], lg, modes[t][i]) for i, lg in enumerate(sizes)] for t in range(2)]
        for c in ctxs:
            c.sync()
        outs = [[torch.full((1 << lg, 4), -1, dtype=torch.int64, device="cuda") for lg in sizes] for t in range(2)]
        errors = []
        start = threading.Barrier(2)

        def run(t):
            try:
                start.wait()
                for rep in range(6):
                    for i, lg in enumerate(sizes):
                        ctxs[t].ntt_noalloc(ins[t][i], lg, modes[t][i], outs[t][i])
                ctxs[t].sync()
            except Exception as e:   # noqa: BLE001 - reported below, in the main thread
                errors.append(e)

        threads = [threading.Thread(target=run, args=(t,)) for t in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        for t in range(2):
            for i, lg in enumerate(sizes):
                assert torch.equal(outs[t][i], want[t][i]), (t, i, lg)
        x = ins[0][1].cpu().numpy().view(np.uint64)
        name = ["fft", "ifft", "coset_fft"][modes[0][1]]
        assert np.array_equal(outs[0][1].cpu().numpy().view(np.uint64), orc.fft(name, x, 1 << 20))
    finally:
        for c in ctxs:
            c.close()


def test_noalloc_rejects_what_it_cannot_serve():
    import torch
    import zkvm_amd
    c = zkvm_amd.Context(0)
    try:
        x = dev(M.random_fr_mont(1 << 10, 5))
        out = torch.empty((1 << 10, 4), dtype=torch.int64, device="cuda")
        with pytest.raises(zkvm_amd.ZkGpuError) as e:      # fresh context: no tables yet
            c.ntt_noalloc(x, 10, 0, out)
        assert e.value.code == -7
        c.ntt(x, 10, 0)
        c.ntt_noalloc(x, 10, 0, out)
        c.ntt_noalloc(x, 8, 1, out[:256])                   # the tables of a larger size serve a smaller one
        with pytest.raises(zkvm_amd.ZkGpuError) as e:      # coset_ifft tables are per size
            c.ntt_noalloc(x, 10, 3, out)
        assert e.value.code == -7
        big = torch.empty((1 << 21, 4), dtype=torch.int64, device="cuda")
        with pytest.raises(zkvm_amd.ZkGpuError) as e:      # beyond the static scratch
            c.ntt_noalloc(x, 21, 0, big)
        assert e.value.code == -1 and "2^20" in str(e.value)
    finally:
        c.close()
