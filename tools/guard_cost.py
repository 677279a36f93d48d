#!/usr/bin/env python3
"""tools/guard_cost.py [reps] — what the conditioning guard's radius costs: the default (statistical) radius against
SDRFM_CFG_GUARD_WORST_CASE, alternating on ONE box, `reps` runs per cell (default 6).  Cells: a batch of carriers (the FM test signal: BASELINE configs[2]),
the thin-spot classes of tools/q_classes.py built around the worst-case radius, and batches with 10 % / 25 % noise-only streams (per-stream routing settled).
Every run is a process of its own.  Per run: 100 warm-up calls (routing settles at call 64), then five regions of 200 calls, serial and overlapped; reported: the median region, us per call.
Output: one table on stdout (profiles/r06_guard_worst_case.txt).  Measurement tool; the oracle is not involved."""
import ctypes as C
import importlib
import os
import statistics as st
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (before the HIP runtime loads: this tool creates many handles in turn, and overlapped calls need hardware queues of their own — INTEGRATION.md section 3)
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
import q_classes as qc  # noqa: E402

NS, NSAMP, NB = 256, 240000, 5


def rotate(base, nb):
    """nb device batches [NS, 2 NSAMP]: consecutive pieces of NS captures (rows beyond the distinct ones are byte-rotations of them)"""
    distinct, total = base.shape[0], base.shape[1] // 2
    dev = torch.from_numpy(base).cuda()
    out = [torch.empty((NS, 2 * NSAMP), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    for r0 in range(0, NS, distinct):
        rows = dev if r0 == 0 else torch.roll(dev, shifts=2 * (7919 * (r0 // distinct) % total), dims=1)
        n = min(distinct, NS - r0)
        for b in range(nb):
            out[b][r0:r0 + n] = rows[:n, 2 * b * NSAMP:2 * (b + 1) * NSAMP]
    return out


def cell_inputs(kind, h, r_wc):
    if kind == "carriers":
        return rotate(pkg.make_iq(32, NB * NSAMP, mode="fm", first_id=0), NB)
    if kind == "thin spots":
        rng = np.random.default_rng(5)
        return rotate(np.stack([qc.make_row(c, NB * NSAMP, h, r_wc, rng) for c in qc.CLASSES for _ in range(3)]), NB)
    pct = int(kind.split()[0])
    b = rotate(pkg.make_iq(32, NB * NSAMP, mode="fm", first_id=0), NB)
    rows = torch.tensor([s for s in range(NS) if ((s + 1) * pct) // 100 > (s * pct) // 100], dtype=torch.long, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(99)
    for x in b:
        x[rows] = torch.randint(0, 256, (rows.numel(), 2 * NSAMP), dtype=torch.uint8, device="cuda", generator=gen)
    return b


STREAM = {}


def run(batches, h, g, wc):
    stream = STREAM.setdefault("s", torch.cuda.Stream())
    aud = [torch.zeros((NS, NSAMP // 50 + 1), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    res = {}
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=NS, guard_worst_case=wc)) as dm:
        dm.set_stream(stream.cuda_stream)
        n = [0]

        def step(ovl):
            dm.process_batch_device(batches[n[0] % NB], aud[n[0] & 1], overlap=ovl)
            n[0] += 1
        for ovl in (False, True):
            for _ in range(100):
                step(ovl)
            dm.flush(); stream.synchronize()
            regs = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(200):
                    step(ovl)
                dm.flush()
                e1.record(stream); e1.synchronize()
                regs.append(e0.elapsed_time(e1) / 200 * 1e3)
            res["overlapped" if ovl else "serial"] = st.median(regs)
        res["kernel"] = dm.kernel_name
        res["routed"] = int(dm.route().sum())
        res["guard_r"] = dm.q_guard()["guard_r"]
        dm.synchronize()
        dm.set_stream(None)
    return res


def one_run(kind, wc):
    """child process: one cell, one radius -> a JSON line (a fresh process per run: handles created one after the other in ONE process end up with
    internal streams that share hardware queues, and the overlapped calls then run one after the other — 44 us per call instead of 22)"""
    import json
    h, g = pkg.default_config(64)
    r1, a1 = C.c_float(), C.c_float()
    pkg.load_library().sdrfm_q_guard2(h.ctypes.data, h.size, g.ctypes.data, g.size, 1, C.byref(r1), C.byref(a1))
    print(json.dumps(run(cell_inputs(kind, h, r1.value), h, g, wc)), flush=True)


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "--one":
        return one_run(sys.argv[2], sys.argv[3] == "1")
    import json
    import subprocess
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    h, g = pkg.default_config(64)
    r0, a0, r1, a1 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    lib = pkg.load_library()
    lib.sdrfm_q_guard2(h.ctypes.data, h.size, g.ctypes.data, g.size, 0, C.byref(r0), C.byref(a0))
    lib.sdrfm_q_guard2(h.ctypes.data, h.size, g.ctypes.data, g.size, 1, C.byref(r1), C.byref(a1))
    print("guard radius: statistical %.3f, worst case %.3f (of 127.5 full scale); both cut at pi - %.2e" % (r0.value, r1.value, np.pi - a0.value))
    print("%-22s %-11s | serial us/call: mean (min .. max) | overlapped us/call: mean (min .. max) | streams routed | last kernel" % ("cell", "radius"))
    for kind in ("carriers", "thin spots", "10 % noise", "25 % noise"):
        acc = {False: [], True: []}
        for _ in range(reps):
            for wc in (False, True):
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", kind, "1" if wc else "0"], capture_output=True, text=True, timeout=600)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                if out.returncode != 0 or not line:
                    print("run failed:", kind, wc, out.stderr[-400:], file=sys.stderr)
                    continue
                acc[wc].append(json.loads(line[-1]))
        for wc in (False, True):
            s = [r["serial"] for r in acc[wc]]; o = [r["overlapped"] for r in acc[wc]]
            print("%-22s %-11s | %7.2f (%.2f .. %.2f)            | %7.2f (%.2f .. %.2f)                | %3d            | %s" % (
                kind, "worst case" if wc else "statistical", st.mean(s), min(s), max(s), st.mean(o), min(o), max(o), acc[wc][-1]["routed"], acc[wc][-1]["kernel"]), flush=True)


if __name__ == "__main__":
    main()
