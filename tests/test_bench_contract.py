"""CPU checks of bench.py's bookkeeping: which committed PMC summary feeds `roofline.traffic`, and how many input batches the
timed loop rotates over (cold-HBM rule).  The GPU side of the contract is exercised by the driver's own bench run."""
import importlib.util
import json
import os
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_traffic_lookup_matches_kernel_and_workload_size():
    b = _bench()
    head = json.load(open(os.path.join(ROOT, "profiles", "traffic_r02.json")))
    t = b.latest_traffic(head["kernel_name"], head["algorithmic_bytes_per_launch"])
    # (the newest committed pass of this kernel and size wins: round 3 re-profiled design S as the bit-exact handle's kernel)
    assert t is not None and t["file"].startswith(("traffic_r0", "r0")) and abs(t["hbm_bytes_per_launch"] / head["hbm_bytes_per_launch"] - 1.0) < 0.01
    # same kernel, twice the streams: the 512-stream profile, not the 256-stream one
    t512 = b.latest_traffic(head["kernel_name"], 2 * head["algorithmic_bytes_per_launch"])
    assert t512 is not None and abs(t512["hbm_bytes_per_launch"] / t["hbm_bytes_per_launch"] - 2.0) < 0.02
    # a workload size nobody profiled, or another kernel: no figure rather than a wrong one
    assert b.latest_traffic(head["kernel_name"], 12345.0) is None
    assert b.latest_traffic("no such kernel", head["algorithmic_bytes_per_launch"]) is None


def test_newest_round_traffic_feeds_the_headline_kernel():
    """The default bench line (design Q, configs[2]) takes `roofline.traffic` from the newest committed counter passes of the same kernel and size
    (round 6: re-taken by tools/profile_round.sh at the kernel with the angle-difference discriminator and the peeled last step)."""
    b = _bench()
    head = json.load(open(os.path.join(ROOT, "profiles", "traffic_r06.json")))
    assert head["kernel_name"].startswith("fast-q") and "k_mfir" in head["rocprof_kernel"]
    t = b.latest_traffic(head["kernel_name"], head["algorithmic_bytes_per_launch"])
    assert t is not None and t["file"].startswith(("traffic_r06", "r06_")) and abs(t["hbm_bytes_per_launch"] / head["hbm_bytes_per_launch"] - 1.0) < 0.01
    assert 1.0 <= t["hbm_bytes_per_launch"] / head["algorithmic_bytes_per_launch"] < 1.06       # warm-up re-reads: ~2 % of the bytes
    # the pipe figures of the kernels that are not bound by the HBM come from the same round's passes: `valu_frac` in their bench lines
    d = b.latest_pmc_derived("wbfm-fused (k_wbfm_steps<8,10>)")
    assert d is not None and d["file"] == "r06_wbfm_pmc.json" and 0.5 < d["valu_issue_busy_fraction"] < 0.9
    d = b.latest_pmc_derived("k_spectrum_chain<10, 12, 2>")
    assert d is not None and d["file"] == "r06_spectrum_pmc.json" and 0.5 < d["valu_issue_busy_fraction"] < 0.9
    t = b.latest_traffic("k_spectrum_chain<10, 12, 2>", 256 * 234 * 1024 * 2.0 + 256 * 1024 * 4.0)
    assert t is not None and 1.0 <= t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"] < 1.1
    blk = b.bound_block("valu", "wbfm-fused (k_wbfm_steps<8,10>)", 0.108, 121241600.0, None)
    assert blk["bound"] == "valu" and blk["valu_frac"] == blk["valu_issue_busy_fraction"] and 0.6 < blk["valu_frac"] < 0.8


def test_traffic_never_below_algorithmic_bytes():
    """A committed summary whose traffic is below the algorithmic bytes would mean a broken counter pass."""
    import glob
    for fn in glob.glob(os.path.join(ROOT, "profiles", "r0[23456]*_pmc.json")):
        d = json.load(open(fn))
        assert d["hbm_bytes_per_launch"] >= 0.999 * d["algorithmic_bytes_per_launch"], fn
        assert d.get("commit") and d["commit"] != "wip", fn


def test_rotation_exceeds_the_infinity_cache():
    b = _bench()
    for bytes_per_batch in (256 * 480000, 512 * 480000, 128 * 640000, 64 * 480000):
        nb = b.pick_batches(types.SimpleNamespace(batches=0), bytes_per_batch)
        assert nb >= 3 and (nb - 1) * bytes_per_batch > 1.5 * b.L3_BYTES      # between two uses of a batch: more than the L3 holds
    assert b.pick_batches(types.SimpleNamespace(batches=7), 1) == 7


def test_mixed_input_class_spreads_the_noisy_streams_evenly():
    """bench.py --iq-class mixed:P: P per cent of the streams, evenly spread, hold noise (the per-stream routing's workload)."""
    b = _bench()
    for ns, pct in ((256, 5), (256, 10), (256, 25), (512, 10)):
        rows = b.noisy_rows(ns, pct)
        assert abs(len(rows) - ns * pct / 100.0) <= 1.0, (ns, pct, len(rows))
        gaps = [y - x for x, y in zip(rows, rows[1:])]
        assert max(gaps) - min(gaps) <= 1, (ns, pct)
