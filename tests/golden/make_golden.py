"""Generates tests/golden/*.npz from the build's own oracle (the reference has no vectors for this path).

Run from the repo root:  python tests/golden/make_golden.py
Each case stores the INPUT bytes, the chunking, the taps and the oracle's fp32 audio (bit pattern is what is checked).
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle.oracle import Oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    # name, mode, n_samples, T, D, Ta, Da, chunks (bytes; None = one shot)
    ("cfg1_fm_T16_oneshot", "fm", 24000, 16, 10, 32, 5, None),
    ("cfg3_fm_T64_urb512", "fm", 12800, 64, 10, 32, 5, 512),
    ("random_T64_ragged", "random", 9001, 64, 10, 32, 5, "ragged"),
    ("counter_T16_urb1024", "counter", 8192, 16, 10, 32, 5, 1024),
    ("const_T64", "const", 4000, 64, 10, 32, 5, None),
    ("odd_geometry_T7_D3", "fm", 5000, 7, 3, 5, 4, "ragged"),
]


def main():
    index = {"generator": "tests/golden/make_golden.py", "cases": []}
    for name, mode, n, T, D, Ta, Da, chunks in CASES:
        if (T, D) in ((16, 10), (64, 10)):
            h, g = pkg.default_config(T, audio_taps=Ta)
        else:
            rng = np.random.default_rng(T * 31 + D)
            h = (rng.standard_normal(T) / T).astype(np.float32)
            g = (rng.standard_normal(Ta) / Ta).astype(np.float32)
        iq = pkg.make_iq(1, n, mode=mode, first_id=7)[0]
        if chunks is None:
            sizes = [iq.size]
        elif chunks == "ragged":
            rng = np.random.default_rng(n)
            sizes, pos = [], 0
            while pos < iq.size:
                c = min(2 * int(rng.integers(0, 400)), iq.size - pos)
                sizes.append(c)
                pos += c
        else:
            sizes = [min(chunks, iq.size - p) for p in range(0, iq.size, chunks)]
        o = Oracle(h, g, D, Da)
        audio, pos = [], 0
        for c in sizes:
            audio.append(o.process(iq[pos:pos + c]))
            pos += c
        audio = np.concatenate(audio)
        fn = name + ".npz"
        np.savez_compressed(os.path.join(HERE, fn), iq=iq, h=h, g=g, D=D, Da=Da, chunks=np.array(sizes, np.int64), audio=audio)
        index["cases"].append({"name": name, "file": fn, "n_audio": int(audio.size)})
        print(name, "n_audio", audio.size)
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
