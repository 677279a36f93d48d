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


def q_atan2_dev(im, re):
    """design Q's own arctangent (csrc/sdrfm_q.hip q_discriminate), fp32 operation by operation: the larger magnitude clamped at 2^-120, v = min / max through
    a reciprocal (numpy's correctly rounded one: the device's v_rcp_f32 is within 1 ulp of it), the 6-coefficient minimax polynomial in v^2, the two
    reflections, the sign of im.  (0, 0) -> 0 through the clamp."""
    re, im = f32(re), f32(im)
    ax, ay = np.abs(re), np.abs(im)
    mx = np.maximum(np.maximum(ax, ay), np.float32(2.0 ** -120))
    mn = np.minimum(ax, ay)
    v = f32(mn * f32(np.float32(1.0) / mx))
    s2 = f32(v * v)
    q = np.full_like(v, np.float32(float.fromhex("0x1.e34882p-8")))
    for c in ("-0x1.22fc74p-5", "0x1.509024p-4", "-0x1.12688cp-3", "0x1.96c562p-3", "-0x1.554086p-2"):
        q = fma32(q, s2, np.float32(float.fromhex(c)))
    a = fma32(v, f32(s2 * q), v)
    a = np.where(ay > ax, f32(np.float32(float.fromhex("0x1.921fb6p+0")) - a), a)
    a = np.where(re < 0, f32(np.float32(float.fromhex("0x1.921fb6p+1")) - a), a)
    return np.copysign(a, im).astype(np.float32)


def q_wrap_dev(x):
    """x in (-2 pi, 2 pi) -> x - 2 pi rint(x / 2 pi) as the kernel evaluates it (csrc/sdrfm_q.hip q_wrap): the rounding through 1.5 * 2^23, three fp32 operations"""
    big = np.float32(12582912.0)
    t = fma32(x, np.float32(float.fromhex("0x1.45f306p-3")), big)
    k = f32(t - big)
    return fma32(k, np.float32(float.fromhex("-0x1.921fb6p+2")), x)


def chain_y(iq, h, D, m):
    """the DEFINITION's y[m] for the output indices m (all windows inside the stream): acc = fmaf(h[k], x, acc), oldest sample first, x = byte - 127.5 —
    the repair path's arithmetic (csrc/sdrfm_q.hip repair_flagged), which is the oracle's (oracle/sdrfm_oracle.c) written out here independently"""
    T = len(h)
    b = iq.astype(np.float32)
    x = np.stack([b[0::2], b[1::2]], axis=1) - np.float32(127.5)
    m = np.asarray(m, dtype=np.int64)
    acc = np.zeros((m.size, 2), np.float32)
    for k in range(T - 1, -1, -1):                                  # oldest sample first: tap k meets sample (m + 1) D - 1 - k
        acc = fma32(np.float32(h[k]), x[(m + 1) * D - 1 - k], acc)
    return acc


def host_discriminate(lib, yr, yi, pr, pi):
    """the definition's K3 as the device's repair path evaluates it (sdrfm_math.h sdrfm_discriminate, same source and rounding: the library's host hook)"""
    return np.array([lib.sdrfm_host_discriminate(float(a), float(b), float(c), float(d)) for a, b, c, d in zip(yr, yi, pr, pi)], dtype=np.float32)


def design_q_audio(iq, h, g, ndig=3, D=10, Da=5, guard=None, stats=None, device_atan=True, k3="diff", recomb="cvt"):
    """One stream from reset, zero history handled like the product does: the first outputs come from the exact spec (the
    generic kernel patches them), so only steady-state arithmetic is judged here.
    guard = (guard_r, guard_a) emulates the kernel's conditioning guard (csrc/sdrfm_q.hip): outputs are held in pairs (2 i, 2 i + 1) by one
    lane; a pair one of whose three y's (y[2 i - 1], y[2 i], y[2 i + 1]) has max(|re|, |im|) < guard_r, or one of whose |d|'s exceeds
    guard_a, gets both d's from the definition's chain recomputed HERE from the raw bytes (chain_y) and the definition's discriminator (round 5;
    round 4 took them from the oracle, which checked the flagging rule but not the repair arithmetic).  device_atan: the unflagged d's use design Q's
    own 6-coefficient arctangent (q_atan2_dev), not libm's.  stats (a dict) receives the number of repaired pairs.
    k3: "diff" (round 6, the kernel's default) — d[m] = wrap(theta[m] - theta[m-1]), theta = the device's arctangent of y itself (q_angle / q_wrap) —
    or "product" (rounds 3 - 5: the arctangent of the conjugate product).  recomb: "cvt" (S0 + 256 S1 in i32, two conversions, two fmas) or "magic"
    (each digit sum as an exact float, three fmas: the SDRFM_Q_MAGIC experiment)."""
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
    if ndig == 3 and recomb == "magic":
        assert np.abs(S).max() < 2 ** 22
        y = fma32(f32(S[0]), qf, fma32(f32(S[1]), np.float32(256.0) * qf, fma32(f32(S[2]), np.float32(65536.0) * qf, cst)))
    elif ndig == 3:
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
    if device_atan and k3 == "diff":
        th = q_atan2_dev(yi, yr)
        thp = np.concatenate([np.zeros(1, np.float32), th[:-1]])
        d = q_wrap_dev(f32(th - thp))
    else:
        d = q_atan2_dev(im, re) if device_atan else np.where((re == 0) & (im == 0), np.float32(0), np.arctan2(im, re).astype(np.float32))
    if guard is not None:
        gr_, ga_ = np.float32(guard[0]), np.float32(guard[1])
        linf = np.maximum(np.abs(y[:, 0]), np.abs(y[:, 1]))
        linf_p = np.maximum(np.abs(prev[:, 0]), np.abs(prev[:, 1]))
        Mp = (M // 2) * 2
        small = np.minimum(np.minimum(linf_p[0:Mp:2], linf[0:Mp:2]), linf[1:Mp:2]) < gr_
        cut = np.maximum(np.abs(d[0:Mp:2]), np.abs(d[1:Mp:2])) > ga_
        flag = np.repeat(small | cut, 2)
        # the repair: the three y's under a flagged pair by the definition's chain FROM THE BYTES (not taken from the oracle), the definition's discriminator as
        # the library evaluates it; the stream's first outputs (windows reaching before sample 0) keep the oracle's y, as the product's generic kernel patches them
        pairs = np.nonzero(small | cut)[0]
        if pairs.size:
            mm = np.unique(np.concatenate([2 * pairs - 1, 2 * pairs, 2 * pairs + 1]))
            mm = mm[(mm >= 0) & (mm < M)]
            inside = ok[mm]
            yc = np.array(yo[mm], dtype=np.float32)
            if inside.any():
                yc[inside] = chain_y(iq, h, D, mm[inside])
            lut = {int(v): i for i, v in enumerate(mm)}
            lib = pkg.load_library()
            for o in (0, 1):
                mo = 2 * pairs + o
                cur = yc[[lut[int(v)] for v in mo]]
                prv = np.array([yc[lut[int(v) - 1]] if int(v) > 0 else (0.0, 0.0) for v in mo], dtype=np.float32).reshape(-1, 2)
                d[mo] = host_discriminate(lib, cur[:, 0], cur[:, 1], prv[:, 0], prv[:, 1])
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
