"""Design Q's conditioning guard on the device (csrc/sdrfm_q.hip): where the phase of y[m] conj(y[m-1]) is ill-conditioned — a deep fade,
or a discriminator input within reach of the branch cut — the matrix-pipe kernel recomputes the pair of d's with the definition's own fmaf
chain from the raw bytes, so that the 1e-5 tolerance of north_star holds for ANY input, uniform random bytes included (SURVEY.md 8d's
worst-case class).  All comparisons at the plain criterion |a - b| <= 1e-5 max(|b|, 1): no scaling, nothing left out.
CPU twin (numpy emulation of the same arithmetic): tests/test_q_guard.py."""
import os

import numpy as np
import pytest

from conftest import scaled_err, TOL

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run_fixture(pkg, z, ns, **cfg):
    sizes, resets = [int(x) for x in z["sizes"]], {int(x) for x in z["reset_before"]}
    iq = np.tile(z["iq"][None, :], (ns, 1))
    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=z["h"], audio_coeffs=z["g"], n_streams=ns, max_bytes_per_call=2 * max(sizes), **cfg))
    pos, parts, names = 0, [], []
    for ci, n in enumerate(sizes):
        if ci in resets:
            dm.reset()
        parts.append(dm.process_batch(iq[:, 2 * pos:2 * (pos + n)]))
        names.append(dm.kernel_name)
        pos += n
    return dm, np.concatenate(parts, axis=1), names


@pytest.mark.parametrize("name", ["q_guard_branch_cut_T64", "q_guard_deep_fade_T32"])
def test_round3_soak_cases_on_a_default_handle(pkg, oracle_mod, name):
    """The two committed cases (tests/golden/make_golden_q_guard.py) in which round 3's unguarded kernel was 0.78 and 1.16e-5 off."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    dm, got, names = _run_fixture(pkg, z, 64)
    assert all(n.startswith("fast-q") for n in names), names        # the matrix-pipe kernel, not a fallback
    assert np.array_equal(got[0].view(np.uint32), got[63].view(np.uint32))
    assert scaled_err(got[0], z["audio"]) <= TOL
    assert scaled_err(got[0], z["audio"]) <= 1e-6                    # ... and as close as everywhere else
    st = dm.q_guard()
    assert st["lanes"] > 0 and st["passes"] > 0, st                  # the repair path is what did it
    dm.close()


def test_the_branch_cut_case_bites_without_the_guard(pkg, monkeypatch):
    """Same bytes with the guard switched off (development library, SDRFM_Q_GUARD_R=0 / SDRFM_Q_GUARD_A=4): round 3's error is back —
    the fixture really exercises the guard."""
    if not os.path.exists(pkg.library_path(dev=True)):
        pytest.skip("libsdrfm_dev.so not built (make -C stm32f7-rtlsdr_amd/csrc dev)")
    monkeypatch.setenv("SDRFM_Q_GUARD_R", "0")
    monkeypatch.setenv("SDRFM_Q_GUARD_A", "4")
    z = np.load(os.path.join(GOLD, "q_guard_branch_cut_T64.npz"))
    dm, got, names = _run_fixture(pkg, z, 64, dev_library=True)
    assert all(n.startswith("fast-q") for n in names), names
    assert dm.q_guard()["lanes"] == 0
    assert scaled_err(got[0], z["audio"]) > 0.5
    dm.close()


@pytest.mark.parametrize("T", [64, 16])
def test_six_million_random_class_outputs_against_a_bit_exact_twin(pkg, T):
    """BASELINE configs[2]'s shape filled with uniform random bytes generated on the device, every row different: 256 x 24 000 = 6.1 M
    decimated outputs through a default handle and through a bit-exact twin (whose audio is the definition's but for atan2's last ulps).
    An unguarded design Q meets an ill-conditioned phase about once per 1e7 outputs of this class (profiles/r03b_fuzz_q.txt)."""
    import torch
    ns, nsamp = 256, 240000
    h, g = pkg.default_config(T)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + T)
    worst, lanes = 0.0, 0
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as fast, \
         pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, bit_exact=True)) as exact:
        for call in range(2):                                         # the second call on carried state
            iq = torch.randint(0, 256, (ns, 2 * nsamp), dtype=torch.uint8, device="cuda", generator=gen)
            a1 = torch.zeros((ns, nsamp // 50), dtype=torch.float32, device="cuda"); a2 = torch.zeros_like(a1)
            torch.cuda.synchronize()
            assert fast.process_batch_device(iq, a1) == nsamp // 50 and exact.process_batch_device(iq, a2) == nsamp // 50
            fast.synchronize(); exact.synchronize()
            assert fast.kernel_name.startswith("fast-q") and not exact.kernel_name.startswith("fast-q")
            e = ((a1 - a2).abs() / a2.abs().clamp(min=1.0)).max().item()
            worst = max(worst, e)
        lanes = fast.q_guard()["lanes"]
    assert worst <= TOL, worst
    assert worst <= 2e-6, worst                                       # the measured margin, not only the tolerance
    assert lanes > 30000                                              # noise-only input does meet the guard (one lane in ten at T = 64, one in forty at T = 16)


def test_guard_at_call_boundaries_and_kernel_changes(pkg, oracle_mod):
    """Random-class rows cut into calls so that ill-conditioned pairs fall on the first outputs of a call: design Q after design Q (the
    64 raw samples it left), design Q after a bit-exact kernel (their T - 1 samples and their y[-1]), a bit-exact kernel after design Q
    (y[-1] recomputed for it), overlapped calls (the previous call's buffer), a reset in between.  Every distinct row against the oracle."""
    import torch
    h, g = pkg.default_config(64)
    ns, nd = 256, 16
    calls = [(2400, False), (2400, False), (1001, False), (399, False), (2400, False), (4800, True), (4800, True), (777, False), (2823, False),
             (2400, True), ("reset", False), (4800, False), (2400, True)]
    total = sum(n for n, _ in calls if n != "reset")
    rows = np.concatenate([pkg.make_iq(nd - 2, total, mode="random", first_id=8000), pkg.make_iq(1, total, mode="counter", first_id=8100),
                           pkg.make_iq(1, total, mode="fm", first_id=8200)])
    dev = torch.from_numpy(np.tile(rows, (ns // nd, 1))).cuda()
    bufs = [torch.zeros((ns, 128), dtype=torch.float32, device="cuda") for _ in range(len(calls))]
    torch.cuda.synchronize()
    orcs = [oracle_mod.Oracle(h, g) for _ in range(nd)]
    names, pos, want, counts = [], 0, [[] for _ in range(nd)], []
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * 4800)) as dm:
        for k, (n, ovl) in enumerate(calls):
            if n == "reset":
                dm.reset(); [o.reset() for o in orcs]; names.append("reset"); counts.append(0)
                continue
            counts.append(dm.process_batch_device(dev[:, 2 * pos:], bufs[k], nbytes=2 * n, overlap=ovl))
            names.append(dm.kernel_name)
            for s in range(nd):
                want[s].append(orcs[s].process(rows[s, 2 * pos:2 * (pos + n)]))
            pos += n
        dm.synchronize()
        st = dm.q_guard()
    got = np.concatenate([b.cpu().numpy()[:, :c] for b, c in zip(bufs, counts)], axis=1)
    assert sum(n.startswith("fast-q") for n in names) >= 8 and sum("overlapped" in n for n in names) >= 3, names
    assert any(n.startswith("generic") or n.startswith("fast-b") for n in names), names
    assert st["lanes"] > 1000
    for s in range(nd):
        assert scaled_err(got[s], np.concatenate(want[s])) <= TOL, (s, names)
    assert np.array_equal(got[:nd].view(np.uint32), got[nd:2 * nd].view(np.uint32))


def test_guard_does_not_touch_a_carrier(pkg):
    """An FM carrier at the synthetic level never meets the guard after the stream's first outputs: the repair path stays out of the
    headline workload's way (the bench line's figure is the guarded kernel's)."""
    import torch
    ns, nsamp = 256, 240000
    h, g = pkg.default_config(64)
    iq = torch.from_numpy(np.tile(pkg.make_iq(16, nsamp, mode="fm", first_id=1000), (ns // 16, 1))).cuda()
    audio = torch.zeros((ns, 4800), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm:
        dm.process_batch_device(iq, audio)                            # from reset: the first outputs see the zero history's bytes
        first = dm.q_guard()["lanes"]
        for _ in range(3):
            dm.process_batch_device(iq, audio)
        assert dm.q_guard()["lanes"] == first
        assert first <= 8 * ns


def test_noise_only_streams_move_to_the_bit_exact_kernels_and_back(pkg, oracle_mod):
    """The repair path at every audio stage costs three times a carrier's call; the bit-exact kernels cost 1.4 times.  A handle whose
    last window of design-Q calls was mostly repair work is served by them for a while (csrc/sdrfm.hip: SDRFM_Q_ADAPT_*), then design Q
    is tried again; a handle fed carriers never leaves design Q.  The audio is the oracle's throughout, across both changes of kernel."""
    import torch
    h, g = pkg.default_config(64)
    ns, nsamp, ncalls = 256, 24000, 80
    for mode, expect_switch in (("random", True), ("fm", False)):
        rows = pkg.make_iq(8, ncalls * nsamp, mode=mode, first_id=4242)
        dev = torch.from_numpy(np.tile(rows, (ns // 8, 1))).cuda()
        out = torch.zeros((ncalls, ns, nsamp // 50), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        names = []
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
            for k in range(ncalls):
                assert dm.process_batch_device(dev[:, 2 * k * nsamp:], out[k], nbytes=2 * nsamp) == nsamp // 50
                names.append(dm.kernel_name.split()[0])
                if k % 4 == 3:
                    dm.synchronize()                                  # (a caller that looks at its audio now and then: the statistics arrive)
            dm.synchronize()
        assert names[0] == "fast-q", names
        if expect_switch:
            first = min(k for k, n in enumerate(names) if n != "fast-q")
            assert first == 64, (first, names)                        # the first window of 16 design-Q calls takes effect at the first call of the fifth: a FIXED call (round 6)
            assert all(n in ("fast-s", "fast-b") for n in names[first:]), names   # ... and for the 1024 calls that follow
        else:
            assert all(n == "fast-q" for n in names), names
        got = out.cpu().numpy()
        for s in range(8):
            want = oracle_mod.Oracle(h, g).process(rows[s])
            assert scaled_err(np.concatenate([got[k, s] for k in range(ncalls)]), want) <= TOL, (mode, s)


@pytest.mark.parametrize("D,Da,fs", [(10, 5, 2.4e6), (8, 8, 2.048e6), (16, 5, 3.2e6)])
def test_matrix_pipe_kernel_with_repairs_writes_only_its_audio(pkg, D, Da, fs):
    """Canary (no GPU sanitizer on the pool): the audio of a machine-filling call on rows of noise — the matrix-pipe kernel with its repair path
    at work — sits in a larger allocation filled with a sentinel; every word outside the documented output region keeps it, at every rate."""
    import torch
    SENT = -12345.0
    h, g = pkg.default_config(64, fs=fs, fir_decim=D, audio_taps=32, audio_decim=Da)
    unit = 8 * D * Da
    ns, nsamp = 256, unit * 47
    A = nsamp // (D * Da)
    iq = torch.from_numpy(pkg.make_iq(ns, nsamp, mode="random", first_id=77)).cuda()
    pad, stride = 64, A + 38
    big = torch.full((pad + ns * stride + pad,), SENT, dtype=torch.float32, device="cuda")
    audio = big[pad:pad + ns * stride].view(ns, stride)
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, fir_decim=D, audio_decim=Da, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
        for rep in range(2):
            assert dm.process_batch_device(iq, audio) == A
            dm.synchronize()
            assert dm.kernel_name.startswith("fast-q"), dm.kernel_name
        assert dm.q_guard()["lanes"] > 1000
    host = big.cpu().numpy()
    assert np.all(host[:pad] == SENT) and np.all(host[-pad:] == SENT)
    body = host[pad:-pad].reshape(ns, stride)
    assert np.all(body[:, A:] == SENT)
    assert np.isfinite(body[:, :A]).all() and not np.any(body[:, :A] == SENT)


def test_measured_read_ceiling_hook(pkg):
    """include/sdrfm_dev.h sdrfm_debug_read_ceiling (what bench.py prints as roofline.peak_measured): a read-only LDS-DMA stream over buffers
    that together exceed the Infinity Cache lands between the guide's cold-HBM figure and the 8 TB/s specification."""
    import ctypes as C
    import torch
    bufs = [torch.zeros(123 << 20, dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    ptrs = (C.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
    out = C.c_double()
    assert pkg.load_library().sdrfm_debug_read_ceiling(0, ptrs, len(bufs), bufs[0].numel(), 40, C.byref(out)) == 0
    assert 4000.0 < out.value < 8000.0, out.value
    assert pkg.load_library().sdrfm_debug_read_ceiling(0, ptrs, 0, bufs[0].numel(), 40, C.byref(out)) == 16        # SDRFM_EINVAL


@pytest.mark.parametrize("T", [64, 16])
def test_matrix_pipe_kernel_where_the_guard_is_thinnest(pkg, oracle_mod, T):
    """VERDICT r04 item 3, on the device: strong out-of-band carriers with A |H(f)| between one and three guard radii (with and without FM), weak in-band
    carriers of 2 .. 8 LSB, a strong adjacent carrier plus a weak wanted one, periodic byte patterns (tools/q_classes.py) — every distinct row against the
    oracle at the plain criterion and against a bit-exact twin, on a default handle served by the matrix-pipe kernel."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "tools"))
    import q_classes as qc
    h, g = pkg.default_config(T)
    ns, nsamp, ncalls = 256, 48000, 3
    rng = np.random.default_rng(77 + T)
    lib = pkg.load_library()
    import ctypes as C
    r, a = C.c_float(), C.c_float()
    assert lib.sdrfm_q_guard(h.ctypes.data, h.size, g.ctypes.data, g.size, C.byref(r), C.byref(a)) == 0
    rows = np.stack([qc.make_row(c, ncalls * nsamp, h, r.value, rng) for c in qc.CLASSES for _ in range(3)])
    nd = rows.shape[0]
    dev = torch.from_numpy(np.tile(rows, ((ns + nd - 1) // nd, 1))[:ns]).cuda()
    outs = {}
    for tag, kw in (("q", {}), ("x", {"bit_exact": True})):
        out = torch.zeros((ncalls, ns, nsamp // 50), dtype=torch.float32, device="cuda")
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, **kw)) as dm:
            for k in range(ncalls):
                dm.process_batch_device(dev[:, 2 * k * nsamp:], out[k], nbytes=2 * nsamp, overlap=(tag == "q" and k > 0))
                if tag == "q":
                    assert dm.kernel_name.startswith("fast-q") and "+" not in dm.kernel_name, dm.kernel_name
            dm.synchronize()
            if tag == "q":
                assert dm.q_guard()["lanes"] > 0
        outs[tag] = out.cpu().numpy()
    worst = 0.0
    for s in range(nd):
        want = oracle_mod.Oracle(h, g).process(rows[s])
        got = np.concatenate([outs["q"][k, s] for k in range(ncalls)])
        e = scaled_err(got, want)
        worst = max(worst, e)
        assert e <= TOL, (qc.CLASSES[s // 3], s, e)
    assert worst <= 2e-6, worst
    assert scaled_err(outs["q"].ravel(), outs["x"].ravel()) <= 2e-6


@pytest.mark.parametrize("T", [64, 16])
def test_worst_case_guard_flag_on_the_thin_spot_classes(pkg, oracle_mod, T):
    """SDRFM_CFG_GUARD_WORST_CASE derives the guard's radius from the PROVEN worst case of |y_fast-q - y_definition| (csrc/qtaps.c
    sdrfm_q_guard2: every rounding of the chain and of the recombination the same way, every tap's quantisation error against a full-scale byte) instead of
    the statistical bound — 6.9 x the radius at 64 taps.  The thin-spot classes (tools/q_classes.py) built around THAT radius, on a handle with the flag:
    still the matrix-pipe kernel, every distinct row the oracle's at the plain criterion; a batch of carriers never meets the guard under it either."""
    import ctypes as C
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "tools"))
    import q_classes as qc
    h, g = pkg.default_config(T)
    lib = pkg.load_library()
    r0, a0, r1, a1 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    assert lib.sdrfm_q_guard2(h.ctypes.data, h.size, g.ctypes.data, g.size, 0, C.byref(r0), C.byref(a0)) == 0
    assert lib.sdrfm_q_guard2(h.ctypes.data, h.size, g.ctypes.data, g.size, 1, C.byref(r1), C.byref(a1)) == 0
    assert a0.value == a1.value and 3.0 < r1.value / r0.value < 9.0, (r0.value, r1.value)     # (T + 4) / (1.25 sqrt T) and the taps' term: 6.9 at 64 taps, 4.1 at 16
    assert r1.value < 0.33 * 127.5                                                              # a carrier at a third of full scale clears it
    ns, nsamp, ncalls = 256, 48000, 3
    rng = np.random.default_rng(177 + T)
    rows = np.stack([qc.make_row(c, ncalls * nsamp, h, r1.value, rng) for c in qc.CLASSES for _ in range(3)])
    nd = rows.shape[0]
    dev = torch.from_numpy(np.tile(rows, ((ns + nd - 1) // nd, 1))[:ns]).cuda()
    out = torch.zeros((ncalls, ns, nsamp // 50), dtype=torch.float32, device="cuda")
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, guard_worst_case=True)) as dm:
        assert abs(dm.q_guard()["guard_r"] - r1.value) <= 1e-6 * r1.value
        for k in range(ncalls):
            dm.process_batch_device(dev[:, 2 * k * nsamp:], out[k], nbytes=2 * nsamp, overlap=(k > 0))
            assert dm.kernel_name.startswith("fast-q") and "+" not in dm.kernel_name, dm.kernel_name
        dm.synchronize()
        assert dm.q_guard()["lanes"] > 0
    got = out.cpu().numpy()
    worst = 0.0
    for s in range(nd):
        want = oracle_mod.Oracle(h, g).process(rows[s])
        e = scaled_err(np.concatenate([got[k, s] for k in range(ncalls)]), want)
        worst = max(worst, e)
        assert e <= TOL, (qc.CLASSES[s // 3], s, e)
    assert worst <= 2e-6, worst
    # carriers (the FM test signal of SURVEY 8d: amplitude 100) under the flag: the guard does not fire behind the streams' first outputs
    iq = torch.from_numpy(pkg.make_iq(ns, 2 * nsamp, mode="fm", first_id=4242)).cuda()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, guard_worst_case=True)) as dm:
        dm.process_batch_device(iq, out[0], nbytes=2 * nsamp)
        dm.synchronize()
        first = dm.q_guard()["lanes"]
        dm.process_batch_device(iq[:, 2 * nsamp:], out[1], nbytes=2 * nsamp)
        dm.synchronize()
        assert dm.q_guard()["lanes"] == first, (first, dm.q_guard())
