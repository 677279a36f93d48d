"""CPU tests of the WBFM (channelizer) oracle: band selectivity, rate, streaming invariance, independent restatement."""
import numpy as np
import pytest


def _taps(pkg):
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)          # prototype: cutoff 0.8 x half a band
    g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0      # resampler prototype at the 6x-upsampled rate, gain L
    return p, g


def _tone(fs, f, n, dev=0.0, fm=1000.0):
    t = np.arange(n)
    ph = 2 * np.pi * f * t / fs + (dev / fm) * np.sin(2 * np.pi * fm * t / fs)
    iq = np.empty(2 * n, np.uint8)
    iq[0::2] = np.clip(np.rint(127.5 + 100 * np.cos(ph)), 0, 255)
    iq[1::2] = np.clip(np.rint(127.5 + 100 * np.sin(ph)), 0, 255)
    return iq


def test_rates_and_band_selectivity(pkg, oracle_mod):
    fs = 3.2e6
    p, g = _taps(pkg)
    for band in (0, 3, 8, 13):
        fc = band * fs / 16 if band < 8 else (band - 16) * fs / 16      # band b is centred on b*fs/16 (mod fs)
        iq = _tone(fs, fc, 320000, dev=30e3, fm=2000.0)
        a = oracle_mod.WbfmOracle(p, g).process(iq)
        assert a.shape == (16, 4800)                                      # 0.1 s -> 48 kHz x 0.1 s per band
        # the FM tone (2 kHz, 30 kHz deviation) is demodulated in its own band: audio = 2*pi*dev/200k*cos(...) per sample
        amp = np.std(a[:, 400:], axis=1)
        expect = 2 * np.pi * 30e3 / 200e3 / np.sqrt(2)
        assert abs(amp[band] - expect) / expect < 0.05, (band, amp[band], expect)
        spec = np.abs(np.fft.rfft(a[band, 400:] * np.hanning(4400)))
        assert abs(np.fft.rfftfreq(4400, 1 / 48000.0)[np.argmax(spec)] - 2000.0) < 25.0


def test_chunked_equals_one_shot_bit_exact(pkg, oracle_mod):
    p, g = _taps(pkg)
    iq = pkg.make_iq(1, 100003, mode="fm", fs=3.2e6, first_id=4)[0]
    one = oracle_mod.WbfmOracle(p, g).process(iq)
    o = oracle_mod.WbfmOracle(p, g)
    rng = np.random.default_rng(2)
    parts, pos = [], 0
    while pos < iq.size:
        n = 2 * int(rng.integers(0, 3000))
        parts.append(o.process(iq[pos:pos + n]))
        pos += n
    got = np.concatenate(parts, axis=1)
    assert got.shape == one.shape
    assert np.array_equal(got.view(np.uint32), one.view(np.uint32))


def test_channelizer_matches_direct_dft_definition(pkg, oracle_mod):
    """Independent restatement in float64: c_b[t] = sum_k p[k] x[16t+15-k] e^{+j 2 pi b k/16}; discriminator; resampler."""
    p, g = _taps(pkg)
    L, M = 6, 25
    iq = pkg.make_iq(1, 16 * 700, mode="fm", fs=3.2e6, first_id=9)[0]
    a = oracle_mod.WbfmOracle(p, g).process(iq).astype(np.float64)
    x = (iq[0::2].astype(np.float64) - 127.5) + 1j * (iq[1::2].astype(np.float64) - 127.5)
    xp = np.concatenate([np.zeros(127, complex), x])
    T = x.size // 16
    k = np.arange(128)
    c = np.zeros((16, T), complex)
    for t in range(T):
        seg = xp[127 + 16 * t + 15 - k]                      # x[16t+15-k]
        for b in range(16):
            c[b, t] = np.sum(p.astype(np.float64) * seg * np.exp(2j * np.pi * b * k / 16))
    prev = np.concatenate([np.zeros((16, 1), complex), c[:, :-1]], axis=1)
    prod = c * np.conj(prev)
    d = np.where(prod == 0, 0.0, np.angle(prod))
    n_out = a.shape[1]
    want = np.zeros((16, n_out))
    for j in range(n_out):
        nj, phi = (j * M) // L, (j * M) % L
        for i in range((len(g) - 1 - phi) // L + 1):
            if nj - i >= 0:
                want[:, j] += float(g[phi + L * i]) * d[:, nj - i]
    assert n_out == (T * L + M - 1) // M or n_out == ((T - 1) * L) // M + 1
    # fp32 path vs float64 definition: the +-pi wrap of a noisy empty band can flip, so compare the occupied band only
    occ = int(np.argmax(np.mean(np.abs(c), axis=1)))
    assert np.max(np.abs(a[occ] - want[occ])) < 2e-4


def test_golden_vector(oracle_mod):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wbfm_three_carriers.npz"))
    got = oracle_mod.WbfmOracle(z["p"], z["g"], int(z["L"]), int(z["M"])).process(z["iq"])
    assert got.shape == z["audio"].shape
    assert np.array_equal(got.view(np.uint32), z["audio"].view(np.uint32))
