"""CPU tests of the spectrum-view oracle (oracle/sdrfm_spectrum_oracle.c): it is the build-defined spec of the FFT view
(the reference has no FFT code: parity unpinned), so it is pinned against an independent float64 FFT and against the
committed golden vectors."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ref(iq, nfft, window):
    x = (iq[0::2].astype(np.float64) - 127.5) + 1j * (iq[1::2].astype(np.float64) - 127.5)
    frames = x.size // nfft
    acc = np.zeros(nfft)
    for f in range(frames):
        acc += np.abs(np.fft.fft(x[f * nfft:(f + 1) * nfft] * window.astype(np.float64))) ** 2
    return np.fft.fftshift(acc / max(frames, 1)), frames


@pytest.mark.parametrize("nfft", [64, 256, 1024, 4096])
def test_oracle_matches_float64_fft(pkg, oracle_mod, nfft):
    iq = pkg.make_iq(1, 5 * nfft + 17, mode="fm", first_id=3)[0]
    hann = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(nfft) / nfft)).astype(np.float32)
    got, frames = oracle_mod.SpectrumOracle(nfft).process(iq)
    want, wf = _ref(iq, nfft, hann)
    assert frames == wf == 5
    assert np.max(np.abs(got - want)) <= 2e-6 * want.max()            # fp32 FFT against float64: ~log2(N) * 2^-24 relative
    got_w, _ = oracle_mod.SpectrumOracle(nfft, hann).process(iq)      # explicit window == default window
    assert np.array_equal(got_w.view(np.uint32), got.view(np.uint32))


def test_tone_lands_in_its_bin_dc_in_the_middle(oracle_mod):
    nfft, k = 1024, 100                                                # +100 bins above DC
    n = np.arange(4 * nfft)
    ph = 2 * np.pi * k * n / nfft
    iq = np.empty(2 * n.size, np.uint8)
    iq[0::2] = np.clip(np.rint(127.5 + 100 * np.cos(ph)), 0, 255)
    iq[1::2] = np.clip(np.rint(127.5 + 100 * np.sin(ph)), 0, 255)
    p, frames = oracle_mod.SpectrumOracle(nfft).process(iq)
    assert frames == 4 and int(p.argmax()) == nfft // 2 + k
    # Hann: coherent gain 0.5 -> peak power (100 * N/2)^2, and the two neighbours are 6 dB down
    assert abs(p[nfft // 2 + k] / (100 * nfft / 2) ** 2 - 1) < 0.02
    assert abs(p[nfft // 2 + k + 1] / p[nfft // 2 + k] - 0.25) < 0.02
    neg = np.empty_like(iq)                                            # conjugate signal -> mirrored bin
    neg[0::2], neg[1::2] = iq[0::2], 255 - iq[1::2]
    assert int(oracle_mod.SpectrumOracle(nfft).process(neg)[0].argmax()) == nfft // 2 - k


def test_edge_cases(oracle_mod):
    o = oracle_mod.SpectrumOracle(256)
    p, frames = o.process(np.zeros(2 * 255, np.uint8))                 # shorter than one frame
    assert frames == 0 and np.all(p == 0)
    p, frames = o.process(np.zeros(0, np.uint8))
    assert frames == 0 and np.all(p == 0)
    with pytest.raises(ValueError):
        o.process(np.zeros(7, np.uint8))                               # odd byte count
    with pytest.raises(ValueError):
        oracle_mod.SpectrumOracle(1000)                                # not a power of two
    mid = np.full(2 * 256 * 3, 128, np.uint8)                          # +0.5 DC on both rails -> only the DC bin (+ Hann skirts)
    p, _ = o.process(mid)
    assert int(p.argmax()) == 128 and p[128 + 2:].max() < 1e-6 * p[128]


def test_golden_vectors(oracle_mod):
    files = sorted(glob.glob(os.path.join(GOLD, "spectrum_*.npz")))
    assert len(files) >= 4
    for fn in files:
        z = np.load(fn)
        win = z["window"] if z["window"].size else None
        got, frames = oracle_mod.SpectrumOracle(int(z["nfft"]), win).process(z["iq"])
        assert frames == int(z["frames"])
        assert np.array_equal(got.view(np.uint32), z["power"].view(np.uint32)), fn
