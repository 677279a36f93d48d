"""CPU tests of design Q's conditioning guard (csrc/sdrfm_q.hip, thresholds: csrc/qtaps.c sdrfm_q_guard): the numpy emulation of the
matrix-pipe arithmetic (tools/q_emulate.py) with and without the guard, against the oracle, on the two round-3 soak cases in which an
unguarded design Q left the 1e-5 tolerance (tests/golden/q_guard_*.npz, make_golden_q_guard.py) and on the input classes of SURVEY.md 8d.
The GPU twin of this file is tests/test_q_guard_gpu.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import TOL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def guard_of(pkg, h, g):
    lib = pkg.load_library()
    r, a = C.c_float(), C.c_float()
    h, g = np.ascontiguousarray(h, np.float32), np.ascontiguousarray(g, np.float32)
    assert lib.sdrfm_q_guard(h.ctypes.data, h.size, g.ctypes.data, g.size, C.byref(r), C.byref(a)) == 0
    return r.value, a.value


def _emulate(z, guard, k3="diff"):
    import q_emulate as qe
    sizes = [int(x) for x in z["sizes"]]
    # (the emulation runs one stream from reset: the branch-cut case's second call stands alone behind its reset)
    iq, ref = (z["iq"][2 * sizes[0]:], z["audio"][sizes[0] // 50:]) if len(z["reset_before"]) else (z["iq"], z["audio"])
    stats = {}
    got, want, _ = qe.design_q_audio(iq, z["h"], z["g"], guard=guard, stats=stats, k3=k3)
    assert np.array_equal(ref, want.astype(np.float32)), "the fixture's expected audio is the oracle's"
    e = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
    return e, stats


def test_guard_thresholds_of_the_baseline_taps(pkg):
    for T, lo, hi in ((64, 4.0, 5.5), (32, 2.3, 3.2), (16, 1.9, 2.7)):
        h, g = pkg.default_config(T)
        r, a = guard_of(pkg, h, g)
        assert lo < r < hi, (T, r)                         # a carrier at 4 % of full scale already clears it
        assert 5e-5 < np.pi - a < 1.2e-4, (T, a)
    # audio taps of zero: nothing to guard
    assert guard_of(pkg, pkg.default_config(64)[0], np.zeros(32, np.float32)) == (0.0, 4.0)


def test_worst_case_radius_is_the_written_bound(pkg):
    """SDRFM_CFG_GUARD_WORST_CASE (csrc/qtaps.c sdrfm_q_guard2): R = 2 max|g| E / 5e-6 with E = (T + 4) 127.5 sum|h| 2^-24 + 64 T max|h| / (127 * 65793) — every rounding of the
    definition's chain and of the recombination the same way, every tap's quantisation error against a full-scale byte —; the branch-cut margin does not depend on E;
    the statistical radius is the same function with E = 1.25 sqrt(T) 127.5 sum|h| 2^-24."""
    lib = pkg.load_library()
    for T in (16, 32, 64):
        h, g = pkg.default_config(T)
        out = {}
        for wc in (0, 1):
            r, a = C.c_float(), C.c_float()
            assert lib.sdrfm_q_guard2(h.ctypes.data, h.size, g.ctypes.data, g.size, wc, C.byref(r), C.byref(a)) == 0
            out[wc] = (r.value, a.value)
        habs, hmax, gmax = float(np.abs(h.astype(np.float64)).sum()), float(np.abs(h).max()), float(np.abs(g).max())
        e_wc = (T + 4) * 127.5 * habs * 2.0 ** -24 + 64.0 * T * hmax / (127 * 65793)
        e_st = 1.25 * np.sqrt(T) * 127.5 * habs * 2.0 ** -24
        assert out[1][0] == pytest.approx(2 * gmax * e_wc / 5e-6, rel=1e-6) and out[0][0] == pytest.approx(2 * gmax * e_st / 5e-6, rel=1e-6)
        assert out[0][1] == out[1][1] == pytest.approx(np.pi - 2 * 5e-6 / gmax, abs=1e-6)
        assert out[0] == guard_of(pkg, h, g)                              # sdrfm_q_guard = the statistical one
        assert out[1][0] < 0.33 * 127.5 * abs(float(h.astype(np.float64).sum()))   # a carrier at a third of full scale clears it (sdrfm_create's gate for the flag)
    # the proven |dy| bound really bounds what the emulation sees, on the class with the largest partial sums
    import q_emulate as qe
    h, g = pkg.default_config(64)
    iq = pkg.make_iq(2, 60000, mode="fm", first_id=5)
    _, _, ymax_err = qe.design_q_audio(iq[0], h, g)
    assert ymax_err < (64 + 4) * 127.5 * float(np.abs(h.astype(np.float64)).sum()) * 2.0 ** -24


@pytest.mark.parametrize("name,n_bad,worst", [("q_guard_branch_cut_T64", 7, 0.7), ("q_guard_deep_fade_T32", 1, 1.1e-5)])
def test_the_soak_cases_fail_without_the_guard_and_pass_with_it(pkg, name, n_bad, worst):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    e, _ = _emulate(z, None, k3="product")
    assert int((e > TOL).sum()) == n_bad and e.max() > worst     # what round 3's kernel did (its conjugate-product discriminator, no guard: profiles/r03b_fuzz_q.txt)
    e, _ = _emulate(z, None)
    assert int((e > TOL).sum()) >= 1 and e.max() > worst         # today's discriminator (a difference of angles) without the guard: out of tolerance too
    e, st = _emulate(z, guard_of(pkg, z["h"], z["g"]))
    assert e.max() <= 1e-6, e.max()                               # repaired: two orders inside the tolerance again
    assert 0 < st["repaired"] < st["pairs"] // 4


@pytest.mark.parametrize("T", [16, 64])
@pytest.mark.parametrize("mode", ["fm", "random", "const", "counter"])
def test_guarded_emulation_on_every_input_class(pkg, T, mode):
    import q_emulate as qe
    h, g = pkg.default_config(T)
    guard = guard_of(pkg, h, g)
    iq = pkg.make_iq(2, 120000, mode=mode, first_id=31)
    st = {}
    for s in range(2):
        got, want, _ = qe.design_q_audio(iq[s], h, g, guard=guard, stats=st)
        e = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
        assert e.max() <= 1e-6, (mode, s, e.max())
    frac = st["repaired"] / st["pairs"]
    if mode == "fm":
        assert frac < 1e-3, frac                               # a carrier never meets the guard (but for the stream's first outputs)
    if mode == "const":
        assert frac == 1.0                                     # |y| = 0.5 |sum h| sqrt 2: everything goes the definition's way


@pytest.mark.parametrize("T,D,Da,fs", [(64, 8, 8, 2.048e6), (64, 16, 5, 3.2e6)])
def test_guarded_emulation_at_the_other_front_end_rates(pkg, T, D, Da, fs):
    """The same arithmetic at the 2.048 and 3.2 MS/s geometries (the guard's thresholds do not depend on D; the windows do)."""
    import q_emulate as qe
    h, g = pkg.default_config(T, fs=fs, fir_decim=D, audio_taps=32, audio_decim=Da)
    guard = guard_of(pkg, h, g)
    for mode in ("fm", "random"):
        iq = pkg.make_iq(1, 8 * D * Da * 150, mode=mode, fs=fs, first_id=47)[0]
        got, want, _ = qe.design_q_audio(iq, h, g, D=D, Da=Da, guard=guard)
        e = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
        assert e.max() <= 1e-6, (mode, e.max())


@pytest.mark.parametrize("T", [64, 16])
@pytest.mark.parametrize("cls", ["oob_carrier", "oob_carrier_fm", "weak_inband", "adjacent_plus_weak", "periodic"])
def test_guarded_emulation_where_the_guard_is_thinnest(pkg, T, cls):
    """strong out-of-band carriers with A |H(f)| between one and three guard radii (a neighbouring station: large partial sums, |y| just
    above the radius, for whole audio windows), weak in-band carriers of 2 .. 8 LSB, both together, and periodic byte patterns whose chain roundings are
    systematic (tools/q_classes.py) — through the emulation with the DEVICE's arctangent and its OWN chain repair, against the oracle at the plain criterion."""
    import q_classes as qc
    import q_emulate as qe
    h, g = pkg.default_config(T)
    guard = guard_of(pkg, h, g)
    rng = np.random.default_rng(1000 * T + len(cls))
    worst, st = 0.0, {}
    for _ in range(3):
        iq = qc.make_row(cls, 60000, h, guard[0], rng)
        got, want, _ = qe.design_q_audio(iq, h, g, guard=guard, stats=st)
        e = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
        worst = max(worst, float(e.max()))
    assert worst <= TOL, (cls, T, worst, st)
    assert worst <= 2e-6, (cls, T, worst, st)                      # (measured: <= 1e-6; the tolerance is 1e-5)


def test_the_guard_fixtures_regenerate_from_a_clean_checkout():
    """the expected audio of tests/golden/q_guard_*.npz is re-derived from the bytes the fixtures themselves hold (no soak dump needed)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden_q_guard.py"), "--regenerate", "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("bit-identical") == 2, r.stdout
