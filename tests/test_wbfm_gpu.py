"""GPU parity of the multi-channel WBFM path (BASELINE configs[4]) against its oracle, through the C-ABI."""
import numpy as np
import pytest

from conftest import scaled_err, TOL

pytestmark = pytest.mark.gpu


def _taps(pkg):
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
    g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
    return p, g


def _check_all(got, want, where=""):
    """EVERY band, EVERY sample within the north-star tolerance.  The DFT graph is op-for-op the oracle's, so c_b is bit-exact,
    re / im of the discriminator are bit-identical on both sides and a +-pi branch-cut flip cannot happen — also not in the
    noise-only bands."""
    assert got.shape == want.shape, (where, got.shape, want.shape)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)) / np.maximum(np.abs(want), 1.0)
    assert err.max() <= TOL, (where, float(err.max()), np.unravel_index(int(np.argmax(err)), err.shape))


def test_single_stream_matches_oracle(pkg, oracle_mod):
    p, g = _taps(pkg)
    iq = pkg.make_iq(1, 320000, mode="fm", fs=3.2e6, first_id=31)[0]
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g))
    got = dm.process_batch(iq)[0]
    want = oracle_mod.WbfmOracle(p, g).process(iq)
    assert got.shape == want.shape == (16, 4800)
    _check_all(got, want)
    dm.close()


def test_configs4_shard_128_streams_matches_oracle_everywhere(pkg, oracle_mod):
    """BASELINE configs[4], one GPU's share: 128 streams x 640 000 bytes (0.1 s at 3.2 MS/s) in ONE device-resident call,
    every stream, every band, every sample against the oracle; a second call checks the carried state at that size."""
    import torch
    p, g = _taps(pkg)
    ns, nsamp = 128, 320000
    iq_host = np.concatenate([pkg.make_iq(64, nsamp, mode="fm", fs=3.2e6, first_id=4000),
                              pkg.make_iq(32, nsamp, mode="random", fs=3.2e6, first_id=4100),
                              pkg.make_iq(32, nsamp, mode="fm", fs=3.2e6, first_id=4200)])
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp))
    assert dm.kernel_name.startswith("wbfm-fused")
    iq = torch.from_numpy(iq_host).cuda()
    audio = torch.zeros((ns, 16, 4800), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n = dm.process_batch_device(iq, audio)
    dm.synchronize()
    assert n == 4800
    first = audio.cpu().numpy()
    n2 = dm.process_batch_device(iq, audio)                    # same bytes again: state (history, c_prev, resampler phase) carried
    dm.synchronize()
    second = audio.cpu().numpy()[:, :, :n2]
    for s_ in range(ns):
        o = oracle_mod.WbfmOracle(p, g)
        _check_all(first[s_], o.process(iq_host[s_]), "stream %d call 1" % s_)
        _check_all(second[s_], o.process(iq_host[s_]), "stream %d call 2" % s_)
    dm.close()


def test_ragged_chunks_bitwise_equal_one_shot(pkg, oracle_mod):
    p, g = _taps(pkg)
    iq = pkg.make_iq(1, 200003, mode="fm", fs=3.2e6, first_id=32)[0]
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g))
    one = dm.process_batch(iq)[0]
    dm.reset()
    rng = np.random.default_rng(6)
    parts, pos = [], 0
    while pos < iq.size:
        n = 2 * int(rng.choice([0, 1, 7, 16, 17, 255, 4096, 30001]))
        parts.append(dm.process_batch(iq[pos:pos + n])[0])
        pos += n
    got = np.concatenate(parts, axis=1)
    assert got.shape == one.shape
    assert np.array_equal(got.view(np.uint32), one.view(np.uint32))
    dm.close()


def test_batch_and_band_placement(pkg, oracle_mod):
    """Several streams at once; an FM tone placed in band 5 comes out of band 5 of its stream and nowhere else."""
    p, g = _taps(pkg)
    fs, n = 3.2e6, 160000
    t = np.arange(n)
    rows = []
    for band in (5, 11, 0):
        fc = band * fs / 16 if band < 8 else (band - 16) * fs / 16
        ph = 2 * np.pi * fc * t / fs + (40e3 / 3000.0) * np.sin(2 * np.pi * 3000.0 * t / fs)
        iq = np.empty(2 * n, np.uint8)
        iq[0::2] = np.clip(np.rint(127.5 + 90 * np.cos(ph)), 0, 255)
        iq[1::2] = np.clip(np.rint(127.5 + 90 * np.sin(ph)), 0, 255)
        rows.append(iq)
    iq = np.stack(rows)
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=3))
    got = dm.process_batch(iq)
    assert got.shape == (3, 16, 2400)
    for s, band in enumerate((5, 11, 0)):
        want = oracle_mod.WbfmOracle(p, g).process(iq[s])
        _check_all(got[s], want, "stream %d" % s)
        amp = np.std(got[s, :, 300:], axis=1)
        assert abs(amp[band] - 2 * np.pi * 40e3 / 200e3 / np.sqrt(2)) < 0.05
    dm.close()


def test_device_buffers_and_errors(pkg, oracle_mod):
    import torch
    p, g = _taps(pkg)
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=4, max_bytes_per_call=64000))
    iq_host = pkg.make_iq(4, 32000, mode="fm", fs=3.2e6, first_id=50)
    iq = torch.from_numpy(iq_host).cuda()
    audio = torch.zeros((4, 16, 500), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()    # allocations / fills ran on torch's stream; the library uses its own
    n = dm.process_batch_device(iq, audio)
    dm.synchronize()
    assert n == 480
    for s in range(4):
        want = oracle_mod.WbfmOracle(p, g).process(iq_host[s])
        _check_all(audio[s, :, :n].cpu().numpy(), want, "stream %d" % s)
    with pytest.raises(pkg.SdrfmError) as e:
        dm.process_batch(np.zeros((4, 7), np.uint8))
    assert e.value.status == 17
    with pytest.raises(pkg.SdrfmError) as e:
        dm.process_batch(np.zeros((4, 70000), np.uint8))
    assert e.value.status == 18
    dm.close()


def _run(pkg, iq, n_streams, **flags):
    p, g = _taps(pkg)
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=n_streams, **flags))
    name = dm.kernel_name
    out = dm.process_batch(iq)
    dm.close()
    return name, out


@pytest.mark.parametrize("n_streams,nsamp", [(1, 100000), (5, 40007), (8, 25601)])
def test_fused_kernel_bitwise_equals_generic_kernels(pkg, n_streams, nsamp):
    """The step kernel (one lane per channelizer step, DFT in registers), the one-lane-per-branch kernel (DFT across lanes by
    DPP) and the two-kernel generic path evaluate the same frozen chains: identical bits, for stream counts that do and do not
    fill a wave's 4 groups and an odd number of steps."""
    iq = pkg.make_iq(n_streams, nsamp, mode="fm", fs=3.2e6, first_id=70)
    nf, fused = _run(pkg, iq, n_streams)
    nb, branch = _run(pkg, iq, n_streams, branch_lanes=True)
    ng, generic = _run(pkg, iq, n_streams, force_generic=True)
    assert "k_wbfm_steps" in nf and "k_wbfm_fused" in nb and ng.startswith("wbfm-generic")
    assert np.array_equal(fused.view(np.uint32), generic.view(np.uint32))
    assert np.array_equal(branch.view(np.uint32), generic.view(np.uint32))


def test_fused_kernel_output_independent_of_run_length(pkg):
    iq = pkg.make_iq(4, 64000, mode="random", fs=3.2e6, first_id=80)
    ref = None
    for branch_lanes in (False, True):
        for nt in (64, 66, 130, 1000, 8000):
            _, out = _run(pkg, iq, 4, run_steps=nt, branch_lanes=branch_lanes)
            if ref is None:
                ref = out
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (branch_lanes, nt)


def test_degenerate_inputs_match_oracle_exactly_in_sign(pkg, oracle_mod):
    """Constant and silent inputs drive band outputs to exact zeros, where the sign of a zero picks +pi or -pi in K3:
    both GPU paths must land on the oracle's side."""
    p, g = _taps(pkg)
    n = 20000
    rows = [np.full(2 * n, 128, np.uint8), np.tile(np.array([255, 0], np.uint8), n), np.tile(np.array([127, 128, 128, 127], np.uint8), n // 2)]
    iq = np.stack(rows)
    for flags in ({}, {"branch_lanes": True}, {"force_generic": True}):
        _, got = _run(pkg, iq, 3, **flags)
        for s in range(3):
            want = oracle_mod.WbfmOracle(p, g).process(iq[s])
            _check_all(got[s], want, "%s stream %d" % (flags, s))


def test_golden_vector_on_gpu(pkg):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wbfm_three_carriers.npz"))
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=z["p"], resamp_coeffs=z["g"], resamp_up=int(z["L"]), resamp_down=int(z["M"])))
    got = dm.process_batch(z["iq"])[0]
    assert got.shape == z["audio"].shape
    _check_all(got, z["audio"], "golden")                             # all 16 bands, every sample
    dm.close()


@pytest.mark.parametrize("byte_off,pad", [(0, 0), (2, 6), (6, 2), (14, 10)])
def test_device_path_with_unaligned_rows_matches_generic_bitwise(pkg, byte_off, pad):
    """Device-resident rows that start at any even byte address and have any even stride (the step kernel fetches 32 bytes per
    lane with plain 16-byte loads), fed in ragged chunks so that the step phase moves too; bitwise against the generic kernels."""
    import torch
    p, g = _taps(pkg)
    ns, nsamp = 3, 52000
    iq_host = pkg.make_iq(ns, nsamp, mode="fm", fs=3.2e6, first_id=900)
    stride = 2 * nsamp + pad
    buf = torch.zeros(ns * stride + 64, dtype=torch.uint8, device="cuda")
    rows = buf[byte_off:byte_off + ns * stride].view(ns, stride)
    rows[:, :2 * nsamp] = torch.from_numpy(iq_host).cuda()
    outs = {}
    for name, flags in (("steps", {}), ("generic", {"force_generic": True})):
        dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, max_bytes_per_call=1 << 17, **flags))
        audio = torch.zeros((ns, 16, 4096), dtype=torch.float32, device="cuda")
        got, pos = [], 0
        for c in (2 * 5000, 2 * 5001, 2 * 17, 2 * 20000, 2 * 3, 2 * 21979):
            n = dm.process_batch_device(rows[:, pos:pos + c], audio, nbytes=c)
            dm.synchronize()
            got.append(audio[:, :, :n].cpu().numpy().copy())
            pos += c
        assert pos == 2 * nsamp
        outs[name] = np.concatenate(got, axis=2)
        if name == "steps":
            assert "k_wbfm_steps" in dm.kernel_name
        dm.close()
    assert outs["steps"].shape == outs["generic"].shape == (ns, 16, 780)
    assert np.array_equal(outs["steps"].view(np.uint32), outs["generic"].view(np.uint32))


def test_kernel_name_follows_the_call_and_run_steps_is_bounded(pkg):
    """sdrfm_wbfm_kernel_name tells which kernel served the LAST call: a call too short for the fused kernels (< 64 channelizer steps)
    runs on the generic pair and says so; the run-length test hook is range-checked at create."""
    p, g = _taps(pkg)
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=2))
    assert "k_wbfm_steps" in dm.kernel_name                         # what the configuration selects
    iq = pkg.make_iq(2, 64000, mode="fm", fs=3.2e6, first_id=7)
    dm.process_batch(iq[:, :2 * 40 * 16])                           # 40 steps
    assert dm.kernel_name.startswith("wbfm-generic"), dm.kernel_name
    dm.process_batch(iq[:, 2 * 40 * 16:])
    assert "k_wbfm_steps" in dm.kernel_name, dm.kernel_name
    dm.close()
    for bad in (2, 32, 8194, 60000):
        with pytest.raises(pkg.SdrfmError) as e:
            pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=2, run_steps=bad))
        assert e.value.status == 16
