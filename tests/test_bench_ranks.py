"""bench.py's N > 1 bookkeeping under a world_size-2 gloo group with the GPU calls stubbed: the barrier + synchronize bracket and the
MAX-over-ranks clock of timed(), per_rank_kernel_ms, the whole-job `value`, ms_per_step_with_closing_barrier, rccl_world, and the shard the C-ABI assigns each
rank — so that the shape of the N > 1 JSON line is checked here before a driver runs it on hardware (N > 1 is UNMEASURED on hardware: no multi-GPU node
was available to the build).  Reference model of the host side: one superloop per device, /root/reference/src/main.c:40-81."""
import importlib.util
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEvent:
    """torch.cuda.Event on a machine without a GPU: the 'device' finishes a step when the host has made it"""
    def __init__(self, enable_timing=True):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def query(self):
        return True

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _FakeCuda:
    Event = _FakeEvent

    @staticmethod
    def synchronize():
        pass


class _FakeTorch:
    """what bench.timed() / gather_per_rank() use of torch, with the tensors on the CPU"""
    cuda = _FakeCuda
    float64 = torch.float64

    @staticmethod
    def tensor(data, dtype=None, device=None):
        return torch.tensor(data, dtype=dtype, device="cpu")

    @staticmethod
    def zeros_like(t):
        return torch.zeros_like(t)


def _worker(rank, world, port, q):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    pkg = importlib.import_module("stm32f7-rtlsdr_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b.DIST_DEVICE["d"] = "cpu"
        steps, per_step = 20, (0.002 if rank == 0 else 0.004)          # rank 1 is the slow one: 40 ms against 80 ms
        made = []

        def step(i):
            made.append(i)
            time.sleep(per_step)
        elapsed, ms = b.timed(_FakeTorch, dist, True, None, step, steps)
        with_barrier = b.TIMED_WITH_BARRIER["s"]
        per_rank = b.gather_per_rank(_FakeTorch, dist, world, ms, device="cpu")
        ns, nsamp = 512, 240000
        sf = b.scaling_fields(world, dist.get_world_size(), ns, nsamp, steps, elapsed, with_barrier, True)
        lo, hi = pkg.fanout.shard_range(world * ns, rank, world)
        q.put((rank, len(made), elapsed, ms, with_barrier, per_rank, sf, (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_rank_bookkeeping_of_the_bench_line_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, el0, ms0, wb0, pr0, sf0, sh0), (r1, n1, el1, ms1, wb1, pr1, sf1, sh1) = res
    assert (r0, r1) == (0, 1) and n0 == n1 == 20                      # EXACTLY K steps on every rank
    # the clock is the MAX over ranks (the slow rank's 20 x 4 ms), the same figure on both ranks; the closing barrier is outside it
    assert el0 == el1 and 0.080 <= el0 < 0.2, (el0, el1)
    assert wb0 == wb1 and wb0 >= el0
    # every rank's own kernel time reaches every rank
    assert pr0 == pr1 and len(pr0) == 2 and 1.9 <= pr0[0] < 3.9 <= pr0[1] < 8.0, pr0
    assert abs(ms0 - pr0[0]) < 1e-3 and abs(ms1 - pr0[1]) < 1e-3
    # the line: whole-job throughput over BOTH ranks' streams / the max-over-ranks time; weak scaling; the world RCCL (here gloo) reports
    for sf in (sf0, sf1):
        assert sf["n_gpus"] == 2 and sf["rccl_world"] == 2 and sf["scaling"] == "weak" and sf["unit"] == "MSamples/s"
        assert abs(sf["value"] - 2 * 512 * 240000 * 20 / el0 / 1e6) <= 0.06
        assert abs(sf["ms_per_step"] - el0 / 20 * 1e3) <= 1e-4
        assert sf["ms_per_step_with_closing_barrier"] >= sf["ms_per_step"] and "MAX over ranks" in sf["timing"]
    # contiguous shards of the 1024 streams, one definition (sdrfm_shard_range)
    assert sh0 == (0, 512) and sh1 == (512, 1024)


def test_single_rank_line_has_no_distributed_fields():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    sf = b.scaling_fields(1, 1, 256, 240000, 20, 0.000524, None, False)
    assert sf["n_gpus"] == 1 and "timing" not in sf and "ms_per_step_with_closing_barrier" not in sf
    assert abs(sf["value"] - 256 * 240000 * 20 / 0.000524 / 1e6) < 0.06
    # the read basis the >= 70 % target is defined on: 2 B per IQ sample against 8 TB/s
    assert b.read_basis(256 * 240000, 0.02194) == pytest.approx(0.70, abs=2e-4)
    assert b.read_basis(256 * 240000, 0.0) is None
