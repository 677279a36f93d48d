"""Generates tests/golden/spectrum_*.npz from the build's own spectrum oracle (the reference has no FFT code, hence no
vectors).  Run from the repo root:  python tests/golden/make_golden_spectrum.py
Each case stores the INPUT bytes, nfft, the window (empty = library default Hann) and the oracle's fp32 power spectrum."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle.oracle import SpectrumOracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("spectrum_fm_1024", "fm", 1024, 8 * 1024 + 37, None), ("spectrum_random_256_rect", "random", 256, 5 * 256, "rect"),
         ("spectrum_counter_2048", "counter", 2048, 3 * 2048 + 1000, None), ("spectrum_const_64", "const", 64, 640, None)]


def main():
    for name, mode, nfft, n, win in CASES:
        iq = pkg.make_iq(1, n, mode=mode, first_id=21)[0]
        window = np.ones(nfft, np.float32) if win == "rect" else None
        power, frames = SpectrumOracle(nfft, window).process(iq)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), iq=iq, nfft=nfft, window=np.zeros(0, np.float32) if window is None else window,
                            power=power, frames=frames)
        print(name, "frames", frames, "peak bin", int(power.argmax()))


if __name__ == "__main__":
    main()
