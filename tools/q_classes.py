"""Input classes that probe design Q's conditioning guard where it is thinnest (csrc/qtaps.c sdrfm_q_guard: the guard's E is a
1.25 sqrt(T) bound on the difference between design Q's y and the definition's fmaf chain, not the chain's worst case):

  oob_carrier        a strong carrier OUT of the channel filter's pass band, amplitude 100 .. 120, at a frequency where A |H(f)| lies between 1 and 3
                     guard radii: large partial sums (the chain's rounding error at its largest) with |y| just above the radius, sustained over whole
                     audio windows — a neighbouring FM station.  With and without frequency modulation.
  weak_inband        a carrier of 2 .. 8 LSB amplitude within +-20 kHz plus a little noise: |y| of the order of the radius for the whole capture
  adjacent_plus_weak a strong adjacent carrier (as oob_carrier) plus a weak wanted one (2 .. 8 LSB): the strong one sets the partial sums, the weak one the phase
  periodic           periodic byte patterns — square waves between two byte levels and short repeating byte sequences — whose chain roundings are
                     systematic rather than random

Byte format: interleaved u8 I / Q, offset binary, as the reference's buffer holds them
(Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Inc/usbh_rtlsdr.h:165-173).  Test infrastructure only (tools/fuzz_q.py, tests/test_q_guard*.py)."""
import numpy as np

CLASSES = ("oob_carrier", "oob_carrier_fm", "weak_inband", "adjacent_plus_weak", "periodic")


def response(h, f, fs):
    """|H(f)| of real taps h at frequency f (Hz)"""
    k = np.arange(len(h))
    return float(np.abs(np.sum(np.asarray(h, np.float64) * np.exp(-2j * np.pi * f * k / fs))))


def _oob_frequency(rng, h, fs, guard_r, amp):
    """a frequency outside the pass band where amp |H(f)| lies in [guard_r, 3 guard_r] (None when the response never gets there)"""
    fgrid = np.linspace(0.02 * fs, 0.49 * fs, 1500)
    k = np.arange(len(h))
    H = np.abs(np.exp(-2j * np.pi * np.outer(fgrid, k) / fs) @ np.asarray(h, np.float64))
    ok = np.nonzero((amp * H >= guard_r) & (amp * H <= 3.0 * guard_r) & (H < 0.5 * H[0] + 0.5 * abs(np.sum(h))))[0]
    if ok.size == 0:
        return None
    f = float(fgrid[rng.choice(ok)])
    return f if rng.random() < 0.5 else -f


def _bytes(ci, cq, rng, noise):
    n = ci.size
    vi = 127.5 + ci + (rng.standard_normal(n) * noise if noise else 0.0)
    vq = 127.5 + cq + (rng.standard_normal(n) * noise if noise else 0.0)
    out = np.empty(2 * n, np.uint8)
    out[0::2] = np.clip(np.rint(vi), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(vq), 0, 255).astype(np.uint8)
    return out


def make_row(cls, n, h, guard_r, rng, fs=2.4e6):
    """one stream of n IQ samples of class `cls` for channel taps h and guard radius guard_r"""
    t = np.arange(n) / fs
    if cls in ("oob_carrier", "oob_carrier_fm", "adjacent_plus_weak"):
        amp = float(rng.uniform(100.0, 120.0))
        f = _oob_frequency(rng, h, fs, guard_r, amp)
        if f is None:                                              # (taps whose stop band is deeper than that: the strongest out-of-band point there is)
            f = 0.25 * fs
        ph = 2 * np.pi * f * t + rng.uniform(0, 2 * np.pi)
        if cls == "oob_carrier_fm":
            fa = float(rng.uniform(300.0, 8000.0))
            ph = ph + (float(rng.uniform(2e3, 30e3)) / fa) * np.sin(2 * np.pi * fa * t)
        ci, cq = amp * np.cos(ph), amp * np.sin(ph)
        if cls == "adjacent_plus_weak":
            a2 = float(rng.uniform(2.0, 8.0)); f2 = float(rng.uniform(-20e3, 20e3)); fa = float(rng.uniform(500.0, 7000.0))
            p2 = 2 * np.pi * f2 * t + (75e3 / fa) * np.sin(2 * np.pi * fa * t) * float(rng.uniform(0.1, 1.0))
            ci, cq = ci + a2 * np.cos(p2), cq + a2 * np.sin(p2)
        return _bytes(ci, cq, rng, float(rng.choice([0.0, 0.5, 2.0])))
    if cls == "weak_inband":
        a = float(rng.uniform(2.0, 8.0)); f = float(rng.uniform(-20e3, 20e3)); fa = float(rng.uniform(500.0, 7000.0))
        ph = 2 * np.pi * f * t + (75e3 / fa) * np.sin(2 * np.pi * fa * t) * float(rng.uniform(0.0, 1.0))
        return _bytes(a * np.cos(ph), a * np.sin(ph), rng, float(rng.choice([0.0, 0.3, 1.0])))
    if cls == "periodic":
        out = np.empty(2 * n, np.uint8)
        if rng.random() < 0.5:                                     # square waves between two byte levels, I and Q with periods and phases of their own
            for c in (0, 1):
                lo, hi = sorted(int(x) for x in rng.integers(0, 256, 2))
                per = int(rng.integers(2, 200)); off = int(rng.integers(0, per))
                out[c::2] = np.where(((np.arange(n) + off) // max(per // 2, 1)) % 2 == 0, lo, hi).astype(np.uint8)
        else:                                                      # a short byte sequence, repeated
            per = int(rng.integers(1, 65)) * 2
            seq = rng.integers(0, 256, per).astype(np.uint8)
            if rng.random() < 0.5:
                seq = (128 + np.rint(float(rng.uniform(1, 127)) * np.sin(2 * np.pi * np.arange(per) / per * int(rng.integers(1, 5)))).astype(np.int64)).clip(0, 255).astype(np.uint8)
            out[:] = np.tile(seq, (2 * n + per - 1) // per)[:2 * n]
        return out
    raise ValueError(cls)
