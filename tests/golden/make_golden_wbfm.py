"""Generates tests/golden/wbfm_*.npz from the build's own WBFM oracle (the reference has no channelizer, hence no vectors).
Run from the repo root:  python tests/golden/make_golden_wbfm.py
The input carries FM carriers in bands 0, 5 and 11 so that those bands are well conditioned; the file stores the input
bytes, both tap sets, L/M and the oracle's fp32 audio of all 16 bands."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle.oracle import WbfmOracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    fs, n = 3.2e6, 24000
    t = np.arange(n)
    sig = np.zeros(n, np.complex128)
    for band, ftone, amp in ((0, 1000.0, 35.0), (5, 2500.0, 30.0), (11, 4000.0, 30.0)):
        fc = band * fs / 16 if band < 8 else (band - 16) * fs / 16
        sig += amp * np.exp(1j * (2 * np.pi * fc * t / fs + (40e3 / ftone) * np.sin(2 * np.pi * ftone * t / fs)))
    iq = np.empty(2 * n, np.uint8)
    iq[0::2] = np.clip(np.rint(127.5 + sig.real), 0, 255)
    iq[1::2] = np.clip(np.rint(127.5 + sig.imag), 0, 255)
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
    g = (pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0).astype(np.float32)
    audio = WbfmOracle(p, g, 6, 25).process(iq)
    np.savez_compressed(os.path.join(HERE, "wbfm_three_carriers.npz"), iq=iq, p=p, g=g, L=6, M=25, audio=audio, bands=np.array([0, 5, 11]))
    print("wbfm_three_carriers", audio.shape, [float(np.std(audio[b, 50:])) for b in (0, 5, 11)])


if __name__ == "__main__":
    main()
