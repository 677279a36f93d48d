"""Host-side audio sink: de-emphasis + int16 stereo PCM in the layout BSP_AUDIO_OUT_Play takes."""
import ctypes as C

import numpy as np


def _ref(x, alpha, gain, y0):
    y, out = np.float32(y0), np.empty(2 * x.size, np.int16)
    for i, v in enumerate(x):
        # fmaf(alpha, v - y, y): the difference rounds to fp32 first, the multiply-add rounds once
        d = np.float32(v - y)
        y = np.float32(np.float64(alpha) * np.float64(d) + np.float64(y))
        s = np.clip(np.float32(y * np.float32(gain)), -32768.0, 32767.0)
        out[2 * i] = out[2 * i + 1] = np.int16(np.rint(s))
    return out, y


def test_deemphasis_and_pcm_layout(pkg):
    lib = pkg.load_library()
    alpha = lib.sdrfm_pcm_alpha(48000.0, 75e-6)
    assert abs(alpha - (1 - np.exp(-1 / (48000 * 75e-6)))) < 1e-6
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(3000) * 1.2).astype(np.float32)
    gain = np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3))
    st = C.c_float(0.0)
    pcm = np.zeros(2 * x.size, np.int16)
    # two calls with carried state == one call
    assert lib.sdrfm_pcm_deemph_s16(x.ctypes.data, 1000, alpha, gain, C.byref(st), pcm.ctypes.data) == 0
    assert lib.sdrfm_pcm_deemph_s16(x[1000:].ctypes.data, 2000, alpha, gain, C.byref(st), pcm[2000:].ctypes.data) == 0
    want, y_end = _ref(x, alpha, gain, 0.0)
    assert np.array_equal(pcm[0::2], pcm[1::2])                       # L == R
    assert np.max(np.abs(pcm.astype(np.int32) - want.astype(np.int32))) <= 1
    assert abs(st.value - float(y_end)) < 1e-5
    assert pcm.max() == 32767 or pcm.min() == -32768 or np.abs(pcm).max() < 32767   # saturates, never wraps
    assert lib.sdrfm_pcm_deemph_s16(x.ctypes.data, 10, 0.0, gain, C.byref(st), pcm.ctypes.data) == 16
