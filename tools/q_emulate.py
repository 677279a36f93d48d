"""Gate (a) of design Q: a numpy emulation of the i8-matrix-pipe FIR arithmetic against the oracle.

Design Q evaluates K2 as  y = q * (S0 + 2^8 S1 + 2^16 S2) + 0.5 * sum(h),  S_t = sum_k digit_t(h[k]) * (byte - 128):
  * the input bytes enter the matrix pipe as i8 = byte XOR 0x80 = byte - 128 (exact); the spec's x = byte - 127.5 = i8 + 0.5,
    so the missing half is the constant 0.5 * sum(h);
  * every tap is the fixed-point integer H = round(h / q), |H| <= 127 * 65793, written in NDIG balanced base-256 digits
    (each in [-128, 127]): three i8 operands;
  * v_mfma_i32_16x16x64_i8 accumulates each S_t exactly in i32 (|S_t| <= 64 * 128 * 128 = 2^20);
  * the digits are combined in fp32: S01 = S0 + (S1 << 8) in i32 (exact, < 2^29), then
    y = fma(float(S01), q, fma(float(S2), 65536 q, c)).
Everything after K2 (discriminator, audio FIR) is the spec's arithmetic unchanged.

Prints the worst scaled audio error |a - oracle| / max(|oracle|, 1) per (taps, input class); the tolerance is 1e-5.
Runs on the CPU only (no GPU minutes); the oracle is used here as the checker, as in tests/.
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle.oracle import Oracle  # noqa: E402

DIGMAX = {2: 127 * 257, 3: 127 * 65793, 4: 127 * 16843009}


def quantize_taps(h, ndig):
    """-> (q as python float, digits int64 [ndig, T]) with h ~= q * sum_t 256^t digits[t]."""
    h = np.asarray(h, dtype=np.float64)
    q = float(np.max(np.abs(h))) / DIGMAX[ndig]
    H = np.rint(h / q).astype(np.int64)
    digs = []
    for _ in range(ndig):
        d = ((H + 128) % 256) - 128
        digs.append(d)
        H = (H - d) // 256
    assert np.all(H == 0)
    return q, np.stack(digs)


def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    return f32(f32(a).astype(np.float64) * f32(b).astype(np.float64) + f32(c).astype(np.float64))


def design_q_audio(iq, h, g, ndig=3, D=10, Da=5, guard=None, stats=None):
    """One stream from reset, zero history handled like the product does: the first outputs come from the exact spec (the
    generic kernel patches them), so only steady-state arithmetic is judged here.
    guard = (guard_r, guard_a) emulates the kernel's conditioning guard (csrc/sdrfm_q.hip): outputs are held in pairs (2 i, 2 i + 1) by one
    lane; a pair one of whose three y's (y[2 i - 1], y[2 i], y[2 i + 1]) has max(|re|, |im|) < guard_r, or one of whose |d|'s exceeds
    guard_a, gets both d's from the definition's own y's (the oracle's).  stats (a dict) receives the number of repaired pairs."""
    T, Ta = len(h), len(g)
    b = iq.astype(np.int64)
    xi8 = np.stack([b[0::2] - 128, b[1::2] - 128], axis=1)          # [N, 2]
    N = xi8.shape[0]
    M = N // D
    q, digs = quantize_taps(h, ndig)
    # window matrix of output m: samples (m+1)D-1-k, k = 0..T-1; outputs whose window starts before sample 0 are patched below
    m = np.arange(M)
    idx = (m[:, None] + 1) * D - 1 - np.arange(T)[None, :]            # [M, T]
    ok = idx[:, -1] >= 0
    idxc = np.clip(idx, 0, N - 1)
    S = np.einsum("tk,mkc->tmc", digs, xi8[idxc])                    # exact int64, [ndig, M, 2]
    assert np.abs(S).max() < 2 ** 31
    qf = np.float32(q)
    cst = np.float32(0.5 * np.sum(np.asarray(h, dtype=np.float64)))
    if ndig == 3:
        s01 = S[0] + 256 * S[1]
        assert np.abs(s01).max() < 2 ** 31
        y = fma32(f32(s01), qf, fma32(f32(S[2]), np.float32(65536.0) * qf, cst))
    elif ndig == 4:
        s01 = S[0] + 256 * S[1]
        s23 = S[2] + 256 * S[3]
        assert max(np.abs(s01).max(), np.abs(s23).max()) < 2 ** 31
        y = fma32(f32(s01), qf, fma32(f32(s23), np.float32(65536.0) * qf, cst))
    else:
        s01 = S[0] + 256 * S[1]
        y = fma32(f32(s01), qf, cst)
    # exact-spec values where the window reaches before the stream start (zero history)
    orc = Oracle(h, g, D, Da)
    want = orc.process(iq)
    yo, _ = orc.last_stage()
    y[~ok] = yo[~ok]
    # K3 (spec arithmetic, fp32)
    prev = np.vstack([np.zeros((1, 2), np.float32), y[:-1]])
    yr, yi, pr, pi = y[:, 0], y[:, 1], prev[:, 0], prev[:, 1]
    re = fma32(yr, pr, f32(yi * pi))
    im = f32(f32(yi * pr) - f32(yr * pi))
    d = np.where((re == 0) & (im == 0), np.float32(0), np.arctan2(im, re).astype(np.float32))
    if guard is not None:
        gr_, ga_ = np.float32(guard[0]), np.float32(guard[1])
        linf = np.maximum(np.abs(y[:, 0]), np.abs(y[:, 1]))
        linf_p = np.maximum(np.abs(prev[:, 0]), np.abs(prev[:, 1]))
        Mp = (M // 2) * 2
        small = np.minimum(np.minimum(linf_p[0:Mp:2], linf[0:Mp:2]), linf[1:Mp:2]) < gr_
        cut = np.maximum(np.abs(d[0:Mp:2]), np.abs(d[1:Mp:2])) > ga_
        flag = np.repeat(small | cut, 2)
        yprev_o = np.vstack([np.zeros((1, 2), np.float32), yo[:-1]])
        re_o = fma32(yo[:, 0], yprev_o[:, 0], f32(yo[:, 1] * yprev_o[:, 1]))
        im_o = f32(f32(yo[:, 1] * yprev_o[:, 0]) - f32(yo[:, 0] * yprev_o[:, 1]))
        d_o = np.where((re_o == 0) & (im_o == 0), np.float32(0), np.arctan2(im_o, re_o).astype(np.float32))
        d[:Mp][flag] = d_o[:Mp][flag]
        if stats is not None:
            stats["pairs"] = stats.get("pairs", 0) + Mp // 2
            stats["repaired"] = stats.get("repaired", 0) + int((small | cut).sum())
            stats["branch_cut"] = stats.get("branch_cut", 0) + int(cut.sum())
    # K4 (spec chain, oldest first)
    A = M // Da
    dd = np.concatenate([np.zeros(Ta - 1, np.float32), d])
    acc = np.zeros(A, np.float32)
    j = np.arange(A)
    for k in range(Ta):                                             # window element k (oldest first) times g[Ta-1-k]
        acc = fma32(np.float32(g[Ta - 1 - k]), dd[(j + 1) * Da - 1 + k], acc)
    ymax_err = float(np.max(np.abs(y[ok].astype(np.float64) - yo[ok].astype(np.float64))))
    return acc, want, ymax_err


def main():
    nstreams = int(os.environ.get("QEMU_STREAMS", "8"))
    nsamp = int(os.environ.get("QEMU_SAMPLES", "240000"))
    rows = []
    for T in (16, 32, 64):
        h, g = pkg.default_config(T)
        for ndig in (2, 3, 4):
            for mode in ("fm", "random", "const", "counter"):
                iq = pkg.make_iq(nstreams, nsamp, mode=mode, first_id=40)
                worst, over, yerr = 0.0, 0, 0.0
                for s in range(nstreams):
                    got, want, ye = design_q_audio(iq[s], h, g, ndig)
                    e = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
                    worst = max(worst, float(e.max()))
                    over += int((e > 1e-5).sum())
                    yerr = max(yerr, ye)
                rows.append({"T": T, "digits": ndig, "mode": mode, "streams": nstreams, "samples": nsamp,
                             "max_scaled_err": worst, "n_over_1e-5": over, "max_abs_y_err": yerr})
                print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    main()
