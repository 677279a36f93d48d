"""Per-stream routing between the matrix-pipe kernel (design Q) and the bit-exact kernels (csrc/sdrfm.hip: the handle's comment).

256 real dongles are not all tuned to a station: the reference's front end tunes one fixed frequency
(Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Src/tuner_e4k.c:1097) and hands over whatever is there (usbh_rtlsdr.c:1058-1101).  A noise-only stream
sends design Q to its repair path at almost every audio stage, and a kernel lasts as long as its slowest wave — so the streams whose windows of design-Q
calls are mostly repair work are served by the bit-exact kernels: design-B workgroups inside design Q's launch (csrc/sdrfm_q.hip k_mix: every shape both designs
have an instance for), or a launch of their own ahead of design Q's (the others: the generic kernel).  Checked here: the
statistics find exactly the noise-only streams, without the host ever waiting; every distinct row equals the oracle at the plain 1e-5 across the change of
kernel; a stream's audio is bit-identical to what its kernel gives alone; calls return without blocking."""
import os
import time

import numpy as np
import pytest

from conftest import scaled_err, TOL

pytestmark = pytest.mark.gpu


def _mixed_rows(pkg, ns, nsamp, noisy_every, first_id=900):
    """8 distinct carrier rows and 4 distinct noise rows tiled over ns streams; stream s is noise-only when s % noisy_every == 1."""
    fm = pkg.make_iq(8, nsamp, mode="fm", first_id=first_id)
    rnd = pkg.make_iq(4, nsamp, mode="random", first_id=first_id + 100)
    mask = np.array([1 if (noisy_every and s % noisy_every == 1) else 0 for s in range(ns)], dtype=np.uint8)
    src = [("r", s % 4) if mask[s] else ("f", s % 8) for s in range(ns)]
    iq = np.stack([rnd[i] if k == "r" else fm[i] for k, i in src])
    return iq, mask, src, fm, rnd


@pytest.mark.parametrize("overlap", [True, False])
def test_statistics_route_exactly_the_noise_only_streams_and_the_audio_stays_the_oracles(pkg, oracle_mod, overlap):
    import torch
    h, g = pkg.default_config(64)
    ns, nsamp, ncalls = 256, 24000, 80
    iq, mask, src, fm, rnd = _mixed_rows(pkg, ns, ncalls * nsamp, 8)          # 32 of 256 streams (12.5 %) hold noise
    dev = torch.from_numpy(iq).cuda()
    out = torch.zeros((ncalls, ns, nsamp // 50), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    names = []
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
        for k in range(ncalls):
            # (a view of the capture per call: consecutive calls read different, adjacent rows — what SDRFM_F_OVERLAP asks for)
            assert dm.process_batch_device(dev[:, 2 * k * nsamp:], out[k], nbytes=2 * nsamp, overlap=overlap) == nsamp // 50
            names.append(dm.kernel_name)
            if k % 8 == 7:
                time.sleep(0.005)                                              # (the device's report arrives while the host does something else: no synchronisation)
        routed = dm.route()
        dm.synchronize()
    assert names[0].startswith("fast-q") and "+" not in names[0], names[0]
    assert np.array_equal(routed, mask), (routed.sum(), mask.sum())
    first = min(k for k, n in enumerate(names) if "+" in n)
    assert first == 64, (first, names)                                         # a window of 16 calls takes effect when its counter set comes round again: four windows later, at a FIXED call
    assert all(("+" in n and "(32 streams) in one launch" in n) for n in names[first:]), names[first:]   # (k_mix)
    got = out.cpu().numpy()
    seen = set()
    for s in range(ns):
        if src[s] in seen:
            continue
        seen.add(src[s])
        row = rnd[src[s][1]] if src[s][0] == "r" else fm[src[s][1]]
        want = oracle_mod.Oracle(h, g).process(row)
        assert scaled_err(np.concatenate([got[k, s] for k in range(ncalls)]), want) <= TOL, (s, src[s])
    # every copy of a row gives the same bits whatever else the batch holds
    for s in range(ns):
        t = next(t for t in range(ns) if src[t] == src[s])
        assert np.array_equal(got[:, s].view(np.uint32), got[:, t].view(np.uint32)), (s, t)


@pytest.mark.parametrize("taps,overlap", [(64, True), (64, False), (16, True), (16, False), (48, True), (48, False)])
def test_a_routed_streams_audio_is_bit_identical_to_what_its_kernel_gives_alone(pkg, taps, overlap):
    """Assignment set by the test hook (every fourth stream to the bit-exact kernels, carriers or not): the design-Q streams equal an all-design-Q handle's
    audio bit for bit, the others a SDRFM_CFG_BIT_EXACT handle's — also across the call at which the assignment changes.  64 taps: both kinds of workgroup in
    one launch (an overlapped call's design-B workgroups read what lies before the call from the previous call's buffer, not from the carried state);
    16 taps (the reference's RTLSDR_FIR): the same; 48 taps (no design-B instance: the generic kernel serves the routed streams): two launches."""
    import torch
    h, g = pkg.default_config(taps)
    ns, nsamp, ncalls = 256, 48000, 6
    iq, _, _, _, _ = _mixed_rows(pkg, ns, ncalls * nsamp, 5, first_id=1300)    # some rows of noise among them: the repair path runs on both sides
    dev = torch.from_numpy(iq).cuda()
    mask = np.array([1 if s % 4 == 2 else 0 for s in range(ns)], dtype=np.uint8)
    outs = {}
    for tag, cfg, route_at in (("q", {}, None), ("x", {"bit_exact": True}, None), ("mixed", {}, 2)):
        out = torch.zeros((ncalls, ns, nsamp // 50), dtype=torch.float32, device="cuda")
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, **cfg)) as dm:
            for k in range(ncalls):
                if route_at is not None and k == route_at:
                    assert np.array_equal(dm.route(mask), mask)
                dm.process_batch_device(dev[:, 2 * k * nsamp:], out[k], nbytes=2 * nsamp, overlap=overlap)
                if tag == "mixed" and k >= route_at:
                    assert "+" in dm.kernel_name and "(64 streams)" in dm.kernel_name, dm.kernel_name
                    assert ("in one launch" in dm.kernel_name) == (taps != 48), dm.kernel_name
            dm.synchronize()
        outs[tag] = out.cpu().numpy().view(np.uint32)
    q, x, m = outs["q"], outs["x"], outs["mixed"]
    assert np.array_equal(m[:2], q[:2])                                        # before the change: design Q for all
    on_q = mask == 0
    assert np.array_equal(m[3:][:, ~on_q], x[3:][:, ~on_q])                          # the routed streams: the bit-exact kernels' bits
    assert np.array_equal(m[3:][:, on_q], q[3:][:, on_q])                            # the others: design Q's bits, whatever shares the batch
    # the call at which the assignment changes: the routed streams take over with the definition's y[-1] (as a SDRFM_CFG_BIT_EXACT handle carries it);
    # the design-Q streams' first outputs after it meet that y[-1] instead of design Q's own (within 1e-4 of it): equal within the tolerance
    # (... and with the 31 discriminator outputs design Q left, which differ from the bit-exact kernels' in the last ulps of the arctangent)
    for got, ref in ((m[2][~on_q], x[2][~on_q]), (m[2][on_q], q[2][on_q])):
        a, b = got.view(np.float32), ref.view(np.float32)
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) <= 2e-6
        assert np.array_equal(got[:, 8:], ref[:, 8:])


def test_calls_return_without_waiting_for_the_device(pkg):
    """include/sdrfm.h: a SDRFM_F_DEVICE_PTRS call returns without synchronising — also when a window of statistics closes while earlier windows are still
    running (rounds 3 - 4 waited on an event there): 96 calls of ~50 us of device time each are issued in a fraction of the time the device needs for them."""
    import torch
    h, g = pkg.default_config(64)
    ns, nsamp, ncalls = 512, 240000, 96
    iq = torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", first_id=77)).cuda()
    bufs = [iq, iq.clone(), iq.clone()]
    aud = [torch.zeros((ns, nsamp // 50), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
        for ovl in (False, True):
            for k in range(8):
                dm.process_batch_device(bufs[k % 3], aud[k & 1], overlap=ovl)
            dm.synchronize()
            t0 = time.perf_counter()
            for k in range(ncalls):
                dm.process_batch_device(bufs[k % 3], aud[k & 1], overlap=ovl)
            t_issue = time.perf_counter() - t0
            dm.synchronize()
            t_all = time.perf_counter() - t0
            assert t_issue < 0.5 * t_all, (ovl, t_issue, t_all)


def test_a_stream_is_tried_on_design_q_again_and_reset_starts_over(pkg, oracle_mod):
    """A noisy stream leaves design Q for SDRFM_Q_ADAPT_BACKOFF (1024) calls, is then tried again (and sent back when it is still noise); sdrfm_reset puts
    every stream back on design Q at once — including right behind overlapped calls on rows of noise (the statistics of kernels still in flight
    must not reach the windows after the reset)."""
    import torch
    h, g = pkg.default_config(64)
    ns, nsamp = 256, 24000
    iq, mask, src, fm, rnd = _mixed_rows(pkg, ns, 4 * nsamp, 16, first_id=2100)
    dev = torch.from_numpy(iq).cuda()
    out = [torch.zeros((ns, nsamp // 50), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
        hist = []
        for k in range(1200):
            dm.process_batch_device(dev[:, 2 * (k % 4) * nsamp:], out[k & 1], nbytes=2 * nsamp, overlap=True)
            if k % 8 == 7:
                time.sleep(0.002)
            hist.append(int(dm.route().sum()))
        assert hist[14] == 0 and max(hist) == int(mask.sum())                 # (the first window closes with the sixteenth call)
        up = hist.index(int(mask.sum()))
        assert up == 64, up                                                    # ... and takes effect at the first call of the fifth window: a fixed call
        down = next(k for k in range(up, len(hist)) if hist[k] == 0)
        assert up + 1000 <= down <= up + 1100, (up, down)                       # tried again after the back-off ...
        assert hist[-1] == int(mask.sum()) or max(hist[down:]) == int(mask.sum())   # ... and found noisy again
        # reset right behind overlapped calls on noise: every stream is design Q's again, and the carriers that follow stay there
        for k in range(6):
            dm.process_batch_device(dev[:, 2 * (k % 4) * nsamp:], out[k & 1], nbytes=2 * nsamp, overlap=True)
        dm.reset()
        assert int(dm.route().sum()) == 0
        clean = torch.from_numpy(np.tile(fm[:, :2 * 4 * nsamp], (ns // 8, 1))).cuda()
        for k in range(40):
            dm.process_batch_device(clean[:, 2 * (k % 4) * nsamp:], out[k & 1], nbytes=2 * nsamp, overlap=(k > 0))
            if k % 8 == 7:
                time.sleep(0.002)
            assert dm.kernel_name.startswith("fast-q") and "+" not in dm.kernel_name, (k, dm.kernel_name)
        assert int(dm.route().sum()) == 0
        dm.synchronize()


@pytest.mark.parametrize("T,D,Da", [(64, 8, 8), (16, 8, 8), (64, 16, 5), (32, 10, 5)])
def test_the_one_launch_kernel_at_the_other_front_end_rates(pkg, oracle_mod, T, D, Da):
    """k_mix has an instance for every shape both designs serve (2.048 MS/s: D = 8, audio / 8; 3.2 MS/s: D = 16; 32 taps): with every third stream
    routed by the test hook, serial and overlapped calls give each stream the bits its kernel gives alone, and the audio is the oracle's at 1e-5."""
    import torch
    h, g = pkg.default_config(T, fir_decim=D, audio_taps=32, audio_decim=Da)
    ns, nsamp, ncalls = 192, D * Da * 8 * 150, 5
    iq, _, _, _, _ = _mixed_rows(pkg, ns, ncalls * nsamp, 7, first_id=3100 + T + D)
    dev = torch.from_numpy(iq).cuda()
    mask = np.array([1 if s % 3 == 1 else 0 for s in range(ns)], dtype=np.uint8)
    na = nsamp // (D * Da)
    outs = {}
    for tag, cfg, ovl in (("q", {}, False), ("x", {"bit_exact": True}, False), ("serial", {}, False), ("overlapped", {}, True)):
        out = torch.zeros((ncalls, ns, na), dtype=torch.float32, device="cuda")
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, fir_decim=D, audio_decim=Da, n_streams=ns, max_bytes_per_call=2 * nsamp, **cfg)) as dm:
            for k in range(ncalls):
                if tag in ("serial", "overlapped") and k == 1:
                    assert np.array_equal(dm.route(mask), mask)
                assert dm.process_batch_device(dev[:, 2 * k * nsamp:], out[k], nbytes=2 * nsamp, overlap=ovl) == na
                if tag in ("serial", "overlapped") and k >= 1:
                    assert "(64 streams) in one launch" in dm.kernel_name, dm.kernel_name
                    assert ("overlapped" in dm.kernel_name) == ovl, dm.kernel_name
            dm.synchronize()
        outs[tag] = out.cpu().numpy()
    on_q = mask == 0
    for tag in ("serial", "overlapped"):
        m = outs[tag].view(np.uint32)
        assert np.array_equal(m[2:][:, ~on_q], outs["x"].view(np.uint32)[2:][:, ~on_q]), tag
        assert np.array_equal(m[2:][:, on_q], outs["q"].view(np.uint32)[2:][:, on_q]), tag
    for s in (0, 1, 8, 22):                                                     # carriers and noise, on either kind of workgroup
        want = oracle_mod.Oracle(h, g, D=D, Da=Da).process(iq[s])
        for tag in ("serial", "overlapped"):
            assert scaled_err(np.concatenate([outs[tag][k, s] for k in range(ncalls)]), want) <= TOL, (tag, s)


@pytest.mark.parametrize("seed", range(11, 11 + int(os.environ.get("SDRFM_ROUTE_SOAK", "6"))))   # (tools: SDRFM_ROUTE_SOAK=120 for a longer soak)
def test_random_call_sequences_with_the_assignment_changing_under_them(pkg, oracle_mod, seed):
    """Soak of the routing: random stream counts, call lengths (every one a length design Q serves), SDRFM_F_OVERLAP on or off per call, and the
    assignment changed by the test hook at random calls (up to 45 % of the streams routed: both kinds of workgroup in one launch; sometimes more than
    half: the bit-exact kernels take the batch).  Every distinct row is the oracle's at 1e-5 over the whole sequence; copies of a row that share an
    assignment history give the same bits."""
    import torch
    rng = np.random.default_rng(seed)
    T, D, Da = [(64, 10, 5), (16, 10, 5), (64, 8, 8), (64, 16, 5), (32, 10, 5), (64, 10, 5)][seed % 6]
    h, g = pkg.default_config(T, fir_decim=D, audio_taps=32, audio_decim=Da)
    ns = int(rng.choice([96, 160, 256]))
    unit = D * Da * 8
    lens = [int(unit * rng.integers(60, 220)) for _ in range(14)]
    total = sum(lens)
    fm = pkg.make_iq(6, total, mode="fm", first_id=5000 + seed)
    rnd = pkg.make_iq(3, total, mode="random", first_id=5100 + seed)
    kind = rng.integers(0, 9, size=ns)                                          # rows 0 - 5: carriers, 6 - 8: noise
    group = rng.integers(0, 3, size=ns)                                         # streams of one group share their assignment history
    iq = np.stack([fm[k] if k < 6 else rnd[k - 6] for k in kind])
    dev = torch.from_numpy(iq).cuda()
    bufs = [torch.zeros((ns, n // (D * Da)), dtype=torch.float32, device="cuda") for n in lens]
    torch.cuda.synchronize()                                                    # (the fills run on torch's stream, the calls on the handle's)
    outs = []
    names = []
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, fir_decim=D, audio_decim=Da, n_streams=ns, max_bytes_per_call=2 * max(lens))) as dm:
        off = 0
        for k, n in enumerate(lens):
            if k and rng.random() < 0.5:
                on = rng.random(3) < [0.5, 0.35, 0.2]                            # which groups go to the bit-exact kernels from this call on
                mask = on[group].astype(np.uint8)
                assert np.array_equal(dm.route(mask), mask)
            na = dm.process_batch_device(dev[:, 2 * off:], bufs[k], nbytes=2 * n, overlap=bool(rng.random() < 0.6))
            assert na == n // (D * Da)
            outs.append(bufs[k])
            names.append(dm.kernel_name)
            off += n
        dm.synchronize()
    got = torch.cat(outs, dim=1).cpu().numpy()
    seen = {}
    for s in range(ns):
        key = (int(kind[s]), int(group[s]))
        if key in seen:
            assert np.array_equal(got[s].view(np.uint32), got[seen[key]].view(np.uint32)), (s, seen[key], key)
            continue
        seen[key] = s
        want = oracle_mod.Oracle(h, g, D=D, Da=Da).process(iq[s])
        assert scaled_err(got[s], want) <= TOL, (s, key, names)


@pytest.mark.parametrize("overlap", [False, True])
def test_a_call_design_q_cannot_take_behind_mixed_calls(pkg, oracle_mod, overlap):
    """a mixed call leaves design Q's own y[-1] for the clean streams only; a following call design Q cannot serve (here: a length that
    is no whole number of audio periods, so the bit-exact kernels take the whole batch) must turn exactly THOSE into the definition's y[-1] — the routed
    streams' are the definition's already, and design Q left no raw samples for them.  (The bug: all streams were 'fixed' from stale rows; d[0] of the
    routed streams, hence ~Ta / Da audio outputs each, left the tolerance.)  Carriers and noise on both sides of the assignment, every distinct row
    against the oracle over the whole sequence."""
    import torch
    h, g = pkg.default_config(64)
    D, Da = 10, 5
    unit = D * Da * 8
    ns = 256
    lens = [unit * 120, unit * 120, unit * 120, unit * 97 + D * Da, unit * 120 - D * Da, unit * 120, unit * 60 + 2 * D, unit * 120]
    total = sum(lens)
    iq, _, src, fm, rnd = _mixed_rows(pkg, ns, (total + 7) // 8 * 8, 5, first_id=7300)   # (rows 16 bytes apart in whole multiples: what design Q asks of a row)
    dev = torch.from_numpy(iq).cuda()
    mask = np.array([1 if s % 4 == 2 else 0 for s in range(ns)], dtype=np.uint8)     # routed by the test hook: carriers and noise alike
    bufs = [torch.zeros((ns, n // (D * Da) + 2), dtype=torch.float32, device="cuda") for n in lens]
    torch.cuda.synchronize()
    names, counts = [], []
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * max(lens))) as dm:
        off = 0
        for k, n in enumerate(lens):
            if k == 1:
                assert np.array_equal(dm.route(mask), mask)
            counts.append(dm.process_batch_device(dev[:, 2 * off:], bufs[k], nbytes=2 * n, overlap=overlap))
            names.append(dm.kernel_name)
            off += n
        dm.synchronize()
    assert "+" in names[1] and "+" in names[2] and "+" in names[5], names          # mixed calls ...
    assert not names[3].startswith("fast-q") and not names[6].startswith("fast-q"), names   # ... and calls design Q cannot take right behind them
    got = np.concatenate([b.cpu().numpy()[:, :c] for b, c in zip(bufs, counts)], axis=1)
    seen = {}
    for s in range(ns):
        key = (src[s], int(mask[s]))
        if key in seen:
            assert np.array_equal(got[s].view(np.uint32), got[seen[key]].view(np.uint32)), (s, seen[key], key)
            continue
        seen[key] = s
        row = rnd[src[s][1]] if src[s][0] == "r" else fm[src[s][1]]
        want = oracle_mod.Oracle(h, g).process(row[:2 * total])
        assert scaled_err(got[s], want) <= TOL, (s, key, names)


def test_two_runs_of_one_capture_give_the_same_bits_whatever_the_hosts_timing(pkg):
    """which kernel serves a stream at which call is a function of the bytes and the call sequence alone (the reference's model is one
    deterministic superloop: /root/reference/src/main.c:72-80).  The same 200-call capture — carriers and noise-only streams that the statistics route away, serial
    and overlapped calls mixed — runs three times with different random host sleeps and once with none at all (the host far ahead of the device); the audio is
    bit-identical and so is the sequence of kernels."""
    import torch
    h, g = pkg.default_config(64)
    ns, nsamp, ncalls = 256, 8000, 200
    iq, mask, _, _, _ = _mixed_rows(pkg, ns, ncalls * nsamp, 6, first_id=8800)
    dev = torch.from_numpy(iq).cuda()
    runs = []
    for trial, seed in enumerate((None, 1, 2, 3)):
        rng = np.random.default_rng(seed) if seed is not None else None
        out = torch.zeros((ncalls, ns, nsamp // 50), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        names = []
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
            for k in range(ncalls):
                dm.process_batch_device(dev[:, 2 * k * nsamp:], out[k], nbytes=2 * nsamp, overlap=(k % 7 != 3))
                names.append(dm.kernel_name)
                if rng is not None and rng.random() < 0.3:
                    time.sleep(float(rng.choice([0.0002, 0.001, 0.004, 0.012])))
            routed = dm.route()
            dm.synchronize()
        runs.append((out.cpu().numpy().view(np.uint32), names, routed))
    assert np.array_equal(runs[0][2], mask)                                    # (the noise-only streams were found)
    assert any("+" in n for n in runs[0][1])
    for got, names, routed in runs[1:]:
        assert names == runs[0][1]
        assert np.array_equal(routed, runs[0][2])
        assert np.array_equal(got, runs[0][0])
