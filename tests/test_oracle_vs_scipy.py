"""The oracles against an INDEPENDENT third-party implementation of the same textbook operations (scipy.signal in float64):
decimating FIR = upfirdn(h, x, down=D) with the spec's alignment, discriminator = angle(y[m] * conj(y[m-1])), audio decimator =
upfirdn(g, d, down=Da); channelizer = a mixer + upfirdn per band.  This pins the oracles' alignment, tap order, sign
conventions and phase carry against code that is not ours; the fp32 chains are then held to ~1e-5 of the float64 result."""
import numpy as np
import pytest

scipy_signal = pytest.importorskip("scipy.signal")


def _x(iq):
    return (iq[0::2].astype(np.float64) - 127.5) + 1j * (iq[1::2].astype(np.float64) - 127.5)


def _decimate(taps, x, down):
    """y[m] = sum_k taps[k] x[(m+1)*down - 1 - k], x[n<0] = 0  — via scipy.signal.upfirdn (full convolution, then phase pick)."""
    full = scipy_signal.upfirdn(np.asarray(taps, np.float64), x, up=1, down=1)      # full[n] = sum_k taps[k] x[n-k]
    return full[down - 1:len(x):down]


@pytest.mark.parametrize("T,D,Ta,Da", [(64, 10, 32, 5), (16, 10, 32, 5), (7, 3, 5, 4), (128, 16, 64, 6)])
def test_fm_oracle_matches_scipy_pipeline(pkg, oracle_mod, T, D, Ta, Da):
    if (T, D) in ((64, 10), (16, 10)):
        h, g = pkg.default_config(T, audio_taps=Ta)
    else:
        rng = np.random.default_rng(T)
        h = (rng.standard_normal(T) / T).astype(np.float32)
        g = (rng.standard_normal(Ta) / Ta).astype(np.float32)
    iq = pkg.make_iq(1, 60007, mode="fm", first_id=4)[0]
    y = _decimate(h, _x(iq), D)
    prev = np.concatenate([[0.0 + 0.0j], y[:-1]])
    d = np.angle(y * np.conj(prev))
    d[0] = 0.0                                                          # y[-1] = 0 -> re = im = 0 -> 0 by definition
    want = _decimate(g, d, Da)
    got = oracle_mod.Oracle(h, g, D, Da).process(iq)
    assert got.size == want.size
    # fp32 chains vs float64: the discriminator of a strong FM signal is well conditioned
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) < 2e-5


def test_wbfm_oracle_band_matches_scipy_mixer_filter_decimator(pkg, oracle_mod):
    """Band b of the critically sampled channelizer == mix by exp(+j 2 pi b n/16)... here stated the textbook way: filter the
    signal shifted DOWN by band b with the prototype and keep every 16th sample; the discriminator removes the constant
    phase the polyphase derivation leaves on each band."""
    fs, n, band = 3.2e6, 64000, 3
    t = np.arange(n)
    fc = band * fs / 16 + 7e3
    ph = 2 * np.pi * fc * t / fs + (30e3 / 2000.0) * np.sin(2 * np.pi * 2000.0 * t / fs)
    iq = np.empty(2 * n, np.uint8)
    iq[0::2] = np.clip(np.rint(127.5 + 90 * np.cos(ph)), 0, 255)
    iq[1::2] = np.clip(np.rint(127.5 + 90 * np.sin(ph)), 0, 255)
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
    g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
    x = _x(iq) * np.exp(-2j * np.pi * band * t / 16)                    # shift band b to DC
    c = _decimate(p, x, 16)                                             # 200 kS/s
    d = np.angle(c[1:] * np.conj(c[:-1]))
    d = np.concatenate([[0.0], d])
    want = scipy_signal.upfirdn(np.asarray(g, np.float64), d, up=6, down=25)
    got = oracle_mod.WbfmOracle(p, g).process(iq)[band]
    m = min(got.size, want.size)
    assert m > 900
    err = np.abs(got[:m] - want[:m])                                    # (measured 1.5e-7 on a signal of amplitude 1.5 rad)
    assert np.max(err) < 5e-6, float(np.max(err))
