#!/usr/bin/env python3
"""tools/pcm_chain_emulate.py — (CPU) the arithmetic of the PCM sink inside the demodulator's launch (csrc/sdrfm_sink_chain.h, the PCM block of csrc/sdrfm_q.hip's
flush_audio), restated in numpy with the kernel's operation order: every run of `run_len` audio outputs is sunk on its own from state 0 — chunks of 8 per lane, the
chunk's contribution as a dot product with alpha d^(7-q), six Hillis-Steele steps over the 64 lanes, the exact chain from the true carry-in —, publishes its end state
and finishes its first 64 outputs with d^(k+1) * (its predecessor's end state).  fp32 throughout (fused multiply-adds through float64: exact products, one rounding
that differs from a true fma's in ~1e-9 of the cases).  Used by tests/test_pcm_chain_cpu.py to hold the scheme to the host routine sdrfm_pcm_deemph_s16 within 1 LSB
without a GPU, and to show why the scheme needs (1 - alpha)^64 below rounding (SDRFM_CHAIN_MIN_ALPHA).  Run as a script: prints the comparison for 75 us / 50 us.
Test infrastructure: nothing here is on a product path."""
import numpy as np

F = np.float32
FIX, CH = 64, 8


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)


def pcm_word(v):
    c = np.clip(v.astype(F), F(-32768.0), F(32767.0))
    return np.rint(c).astype(np.int32)                           # (round-half-even, as the 1.5 * 2^23 addition does)


def sink_flush(x, yrun, alpha, gain, w, pc):
    """One flush of up to 512 outputs: returns (y of every output from the run's state so far, the state behind the last one)."""
    n = x.size
    assert 0 < n <= 64 * CH
    xp = np.zeros(64 * CH, F)
    xp[:n] = x
    xr = xp.reshape(64, CH)
    sc = np.zeros(64, F)
    for q in range(CH):
        sc = fma(w[q], xr[:, q], sc)
    pw = F(pc)
    sc[0] = fma(pw, yrun, sc[0])
    d = 1
    while d < 64:
        o = np.concatenate([np.zeros(d, F), sc[:-d]])
        sn = fma(pw, o, sc)
        sc = np.where(np.arange(64) >= d, sn, sc).astype(F)
        pw = F(pw * pw)
        d <<= 1
    y = np.concatenate([[F(yrun)], sc[:-1]]).astype(F)
    ys = np.zeros((64, CH), F)
    valid = (np.arange(64)[:, None] * CH + np.arange(CH)[None, :]) < n
    for q in range(CH):
        yn = fma(alpha, (xr[:, q] - y).astype(F), y)
        ys[:, q] = yn
        y = np.where(valid[:, q], yn, y).astype(F)
    flat = ys.reshape(-1)[:n]
    return flat, F(flat[-1])


def chain_emulate(x, alpha, gain, run_len=400, state0=0.0, min_alpha_check=True):
    """PCM (int32 values, one per output) and the final state, by the in-launch scheme, for one stream's call cut into runs of run_len outputs."""
    alpha, gain = F(alpha), F(gain)
    d = 1.0 - float(alpha)
    w = [F(float(alpha) * d ** (CH - 1 - q)) for q in range(CH)]
    pc = F(d ** CH)
    dpow = np.array([d ** (k + 1) for k in range(FIX)], F)
    x = np.asarray(x, F)
    out = np.zeros(x.size, np.int32)
    carry = F(state0)                                            # the predecessor's published end state
    pos = 0
    while pos < x.size:
        n = min(run_len, x.size - pos)
        if x.size - (pos + n) < FIX and x.size - (pos + n) > 0:  # (the host never leaves a run shorter than its predecessor's reach)
            n = x.size - pos
        yrun, first, yloc = F(0.0), True, None
        ys_all = np.zeros(n, F)
        q0 = 0
        while q0 < n:
            m = min(64 * CH, n - q0)
            ys, yrun = sink_flush(x[pos + q0:pos + q0 + m], yrun, alpha, gain, w, pc)
            ys_all[q0:q0 + m] = ys
            q0 += m
        nfix = min(FIX, n)
        ys_all[:nfix] = fma(dpow[:nfix], carry, ys_all[:nfix])   # the run's first outputs, finished with the predecessor's state
        out[pos:pos + n] = pcm_word((ys_all * gain).astype(F))
        carry = yrun                                             # published: the run's OWN end state (what d^n * carry would add is below rounding)
        pos += n
    return out, float(carry)


def main():
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    pkg = importlib.import_module("stm32f7-rtlsdr_amd")
    rng = np.random.default_rng(5)
    t = np.arange(48000) / 48000.0
    x = (1.2 * np.sin(2 * np.pi * 1000 * t) + 0.5 * np.sin(2 * np.pi * 7300 * t) + 0.05 * rng.standard_normal(t.size)).astype(F)
    gain = F(32767.0 / (2 * np.pi * 75e3 / 240e3))
    for tau in (75e-6, 50e-6):
        alpha = float(pkg.load_library().sdrfm_pcm_alpha(48000.0, tau))
        want, st = pkg.pcm_deemph_s16_host(x, alpha, gain)
        got, st2 = chain_emulate(x, alpha, gain)
        dd = np.abs(got - want[0::2].astype(np.int32))
        print("tau %.0f us: alpha %.4f, (1 - alpha)^64 = %.2e: max |PCM difference| %d LSB, %.3f %% of the outputs differ, state %.3e relative"
              % (tau * 1e6, alpha, (1 - alpha) ** 64, dd.max(), 100.0 * (dd > 0).mean(), abs(st2 - st) / max(abs(st), 0.25)))


if __name__ == "__main__":
    main()
