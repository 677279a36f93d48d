#!/usr/bin/env python3
"""Gate (a) for a matrix-pipe version of the WBFM channelizer (what design Q is for the FM path): would an EXACT evaluation of
c_b[t] = sum_k W[b,k] x[k] (integer arithmetic from fixed-point taps, one rounding) stay within the 1e-5 parity tolerance of the oracle's
fp32 definition (polyphase fmaf chains + radix-2 DFT), on the input classes the tests use?  numpy float64 stands in for "exact".
Prints, per input class: max scaled audio error in the occupied band, over all 16 bands, and the share of audio samples beyond 1e-5.
usage: wq_emulate.py [n_samples]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
g = (pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0).astype(np.float32)
L, M, NB, P, Tg = 6, 25, 16, 128, 60
HD = (Tg + L - 1) // L
k = np.arange(P)
W = p.astype(np.float64)[None, :] * np.exp(2j * np.pi * np.arange(NB)[:, None] * (k[None, :] % NB) / NB)   # c_b = sum_k W[b,k] x[16 t + 15 - k]
for bits in (None, 24, 32):
    if bits is None:
        Wq, label = W, "exact taps (float64 of the fp32 products)"
    else:
        q = np.abs(W).max() / (2 ** (bits - 1) - 1)
        Wq, label = (np.round(W.real / q) + 1j * np.round(W.imag / q)) * q, "taps rounded to %d-bit fixed point" % bits
    for mode, fid in (("fm", 31), ("fm", 4000), ("random", 4100), ("counter", 5), ("const", 6)):
        iq = pkg.make_iq(1, n, mode=mode, fs=3.2e6, first_id=fid)[0]
        orc = O.WbfmOracle(p, g, L, M)
        want = orc.process(iq)                                    # [16, A]
        x = (iq[0::2].astype(np.float64) - 127.5) + 1j * (iq[1::2].astype(np.float64) - 127.5)
        xp = np.concatenate([np.zeros(P - 1, complex), x])
        Tn = n // NB
        # windows: step t uses x[16 t + 15 - k], k = 0..127  ->  xp index (P-1) + 16 t + 15 - k
        idx = (P - 1) + 16 * np.arange(Tn)[:, None] + 15 - k[None, :]
        c = xp[idx] @ Wq.T                                        # [Tn, 16]
        c32 = c.real.astype(np.float32).astype(np.float64) + 1j * c.imag.astype(np.float32).astype(np.float64)   # one rounding to fp32
        prev = np.vstack([np.zeros((1, NB), complex), c32[:-1]])
        z = c32 * np.conj(prev)
        d = np.where(z == 0, 0.0, np.angle(z)).astype(np.float32).astype(np.float64)      # [Tn, 16]
        A = want.shape[1]
        dd = np.vstack([np.zeros((HD, NB)), d])
        out = np.zeros((NB, A))
        for j in range(A):
            nj, phi = (j * M) // L, (j * M) % L
            imax = (Tg - 1 - phi) // L
            ii = np.arange(imax + 1)
            out[:, j] = (g[phi + L * ii].astype(np.float64)[:, None] * dd[HD + nj - ii, :]).sum(axis=0)
        e = np.abs(out - want) / np.maximum(np.abs(want), 1.0)
        pw = (np.abs(c32) ** 2).mean(axis=0)
        occ = int(np.argmax(pw))
        print("%-44s %-8s occupied band %2d: max %.2e | all bands: max %.2e, beyond 1e-5: %.4f %% | median |c| per band min %.3g max %.3g"
              % (label, mode, occ, e[occ].max(), e.max(), 100.0 * (e > 1e-5).mean(), np.sqrt(pw.min()), np.sqrt(pw.max())), flush=True)
