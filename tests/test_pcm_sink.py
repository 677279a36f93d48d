"""Audio sink: de-emphasis + int16 stereo PCM in the layout BSP_AUDIO_OUT_Play takes
(Utilities/STM32746G-Discovery/stm32746g_discovery_audio.c:224).  The host routine (csrc/pcm_sink.c) is pinned here, BIT FOR
BIT, by an exact-rational Python restatement of its definition; the device sink is held to the host routine in
tests/test_pcm_sink_gpu.py."""
import ctypes as C

import numpy as np

from conftest import fma32


def pcm_reference(x, alpha, gain, y0):
    """y = fmaf(alpha, x - y, y) (one rounded difference, one fused multiply-add); pcm = rint(clamp(y * gain)) half-even."""
    y, out = np.float32(y0), np.empty(2 * x.size, np.int16)
    alpha, gain = np.float32(alpha), np.float32(gain)
    for i, v in enumerate(x):
        y = fma32(alpha, np.float32(np.float32(v) - y), y)
        s = np.float32(y * gain)
        s = min(max(s, np.float32(-32768.0)), np.float32(32767.0))
        out[2 * i] = out[2 * i + 1] = np.int16(np.rint(s))        # numpy rint = round-half-even = lrintf's default mode
    return out, y


def test_deemphasis_and_pcm_layout(pkg):
    lib = pkg.load_library()
    alpha = lib.sdrfm_pcm_alpha(48000.0, 75e-6)
    assert abs(alpha - (1 - np.exp(-1 / (48000 * 75e-6)))) < 1e-6
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(3000) * 1.2).astype(np.float32)
    x[100:110] = [0.5 / 3.0, -0.5 / 3.0, 10.0, 10.0, 0, 0, 1e-30, -1e-30, 3.0, -3.0]   # saturation, zeros, tiny values
    x[200:204] = -10.0
    gain = np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3))
    st = C.c_float(0.0)
    pcm = np.zeros(2 * x.size, np.int16)
    # two calls with carried state == one call
    assert lib.sdrfm_pcm_deemph_s16(x.ctypes.data, 1000, alpha, gain, C.byref(st), pcm.ctypes.data) == 0
    assert lib.sdrfm_pcm_deemph_s16(x[1000:].ctypes.data, 2000, alpha, gain, C.byref(st), pcm[2000:].ctypes.data) == 0
    want, y_end = pcm_reference(x, alpha, gain, 0.0)
    assert np.array_equal(pcm[0::2], pcm[1::2])                       # L == R
    assert np.array_equal(pcm, want)                                  # integer output: bit-exact, no LSB of slack
    assert np.float32(st.value).view(np.uint32) == np.float32(y_end).view(np.uint32)
    assert pcm.max() == 32767 and pcm.min() == -32768                 # saturates, never wraps
    assert lib.sdrfm_pcm_deemph_s16(x.ctypes.data, 10, 0.0, gain, C.byref(st), pcm.ctypes.data) == 16


def test_half_way_cases_round_to_even(pkg):
    lib = pkg.load_library()
    # alpha = 1 makes y = x exactly; gain = 1 -> pcm = rint(x): ties go to the even integer
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 32766.5, 32767.5, -32768.5], np.float32)
    st = C.c_float(0.0)
    pcm = np.zeros(2 * x.size, np.int16)
    assert lib.sdrfm_pcm_deemph_s16(x.ctypes.data, x.size, 1.0, 1.0, C.byref(st), pcm.ctypes.data) == 0
    assert pcm[0::2].tolist() == [0, 2, 2, 0, -2, -2, 32766, 32767, -32768]
