"""world_size-2 gloo test of the stream fan-out / audio fan-in (the only collectives of the path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, nbytes, q):
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    pkg = importlib.import_module("stm32f7-rtlsdr_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy((np.arange(n_streams * nbytes) % 253).astype(np.uint8).reshape(n_streams, nbytes))
        root_buf = full if rank == 0 else None
        local = pkg.fanout.scatter_streams(root_buf, n_streams, nbytes, "cpu")
        lo, hi = pkg.fanout.shard_range(n_streams, rank, world)
        ok = bool(torch.equal(local, full[lo:hi]))
        # "audio": a per-stream function of the shard so that the gather can be checked exactly
        audio_local = local[:, :7].to(torch.float32) * 0.5 + torch.arange(lo, hi, dtype=torch.float32)[:, None]
        gathered = pkg.fanout.gather_audio(audio_local, n_streams)
        if rank == 0:
            want = full[:, :7].to(torch.float32) * 0.5 + torch.arange(n_streams, dtype=torch.float32)[:, None]
            ok = ok and bool(torch.equal(gathered, want))
        else:
            ok = ok and gathered is None
        q.put((rank, ok, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_streams", [8, 5])
def test_scatter_gather_world2(n_streams):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_streams, 96, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == n_streams


def test_shard_range_partitions(pkg):
    for n in (0, 1, 7, 256, 4096, 1000):
        for w in (1, 2, 4, 8):
            spans = [pkg.fanout.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_range_is_one_definition_in_c(pkg):
    """sdrfm_shard_range (C, csrc/rtlctl.c) is what both hosts use: examples/multi_gpu_main.c directly, fanout.shard_range through
    ctypes.  Contiguous, exhaustive, the first n % world ranks own one stream more; bad arguments are refused."""
    import ctypes as C
    lib = pkg.load_library()
    for n in (0, 1, 7, 256, 512, 1000, 4096, 4099):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            for rank in range(world):
                first, count = C.c_uint32(), C.c_uint32()
                assert lib.sdrfm_shard_range(n, world, rank, C.byref(first), C.byref(count)) == 0
                q, r = divmod(n, world)
                assert (first.value, count.value) == (rank * q + min(rank, r), q + (1 if rank < r else 0))
                assert first.value == pos
                pos += count.value
                assert pkg.fanout.shard_range(n, rank, world) == (first.value, first.value + count.value)
            assert pos == n
    first, count = C.c_uint32(), C.c_uint32()
    assert lib.sdrfm_shard_range(8, 0, 0, C.byref(first), C.byref(count)) == 16      # SDRFM_EINVAL
    assert lib.sdrfm_shard_range(8, 2, 2, C.byref(first), C.byref(count)) == 16
    assert lib.sdrfm_shard_range(8, 2, 1, None, C.byref(count)) == 16
