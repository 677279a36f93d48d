#!/usr/bin/env python3
"""LDS bank-conflict model for the spectrum kernel's FFT (csrc/sdrfm_spectrum.hip): enumerates every access pattern of the
64 lanes of a wave — the bit-reversed scatter, the 4 point reads/writes of every fused two-stage pass, the single last stage
of odd log2(N), the twiddle reads — and counts the bank-slot rows each one needs (a float2 occupies one of 16 slots of the 32
4-byte banks) for a given index padding.  pad(i) = i + (i >> 4) + (i >> 8) reaches the conflict-free ideal for N >= 256.

    python tools/fft_lds_padding.py
"""
import numpy as np


def brev(n, bits):
    return int(format(n, "0%db" % bits)[::-1], 2)


def point_patterns(logn):
    n, pats = 1 << logn, []
    for q in range(max(n // 64, 1)):
        pats.append([brev(lane + 64 * q, logn) for lane in range(64) if lane + 64 * q < n])
    s = 1
    while s + 1 <= logn:
        h = 1 << (s - 1)
        for g0 in range(0, max(n // 4, 1), 64):
            for off in range(4):
                pats.append([(((g >> (s - 1)) << (s + 1)) + (g & (h - 1))) + off * h for g in range(g0, min(g0 + 64, n // 4))])
        s += 2
    if s <= logn:
        h = 1 << (s - 1)
        for j0 in range(0, n // 2, 64):
            for off in range(2):
                pats.append([(((j >> (s - 1)) << s) + (j & (h - 1))) + off * h for j in range(j0, min(j0 + 64, n // 2))])
    return pats


def twiddle_patterns(logn):
    n, pats = 1 << logn, []
    s = 1
    while s + 1 <= logn:
        h = 1 << (s - 1)
        for g0 in range(0, max(n // 4, 1), 64):
            pos = [g & (h - 1) for g in range(g0, min(g0 + 64, n // 4))]
            pats += [[p << (logn - s) for p in pos], [p << (logn - s - 1) for p in pos], [(p + h) << (logn - s - 1) for p in pos]]
        s += 2
    return pats


def rows(pats, pad, distinct_only=False):
    total = 0
    for idx in pats:
        addrs = set(pad(i) for i in idx) if distinct_only else [pad(i) for i in idx]     # equal addresses broadcast
        total += int(max(np.bincount([a % 16 for a in addrs], minlength=16)))
    return total


if __name__ == "__main__":
    pads = {"none": lambda i: i, "i+(i>>4)": lambda i: i + (i >> 4), "i+(i>>4)+(i>>8)": lambda i: i + (i >> 4) + (i >> 8)}
    for logn in range(6, 13):
        pp, tp = point_patterns(logn), twiddle_patterns(logn)
        ideal = sum((len(i) + 15) // 16 for i in pp)
        print("N=%4d  points: ideal %4d" % (1 << logn, ideal),
              " ".join("%s=%d" % (k, rows(pp, f)) for k, f in pads.items()),
              "| twiddles:", " ".join("%s=%d" % (k, rows(tp, f, True)) for k, f in pads.items()))
