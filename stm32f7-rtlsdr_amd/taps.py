"""Tap fixtures.

RTLSDR_FIR is the only filter table the reference contains: the RTL2832's on-chip 32-tap symmetric FIR, of which the
firmware stores the first 16 coefficients (Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Inc/usbh_rtlsdr.h:340-345).
It is used here, divided by its sum, purely as the deterministic "16-tap" fixture of BASELINE configs 1-2.
"""
import ctypes as C
import numpy as np

from .siggen import _siggen

RTLSDR_FIR = (-54, -36, -41, -40, -32, -14, 14, 53, 101, 156, 215, 273, 327, 372, 404, 421)


def rtlsdr_fir16():
    t = np.asarray(RTLSDR_FIR, dtype=np.float64)
    return (t / t.sum()).astype(np.float32)  # sum = 2119


def lowpass_taps(n, cutoff_over_fs):
    """Hamming-windowed sinc, unity DC gain (tools/siggen/siggen.c: siggen_lowpass)."""
    out = np.zeros(int(n), dtype=np.float32)
    _siggen().siggen_lowpass(out.ctypes.data_as(C.c_void_p), C.c_uint32(int(n)), C.c_double(float(cutoff_over_fs)))
    return out


def default_config(fir_taps=64, fs=2.4e6, fir_decim=10, audio_taps=32, audio_decim=5):
    """Taps of the BASELINE configurations: T=16 -> RTLSDR_FIR/2119; otherwise a 100 kHz Hamming-sinc at fs;
    audio filter: 15 kHz Hamming-sinc at fs/fir_decim."""
    h = rtlsdr_fir16() if fir_taps == 16 else lowpass_taps(fir_taps, 100e3 / fs)
    g = lowpass_taps(audio_taps, 15e3 / (fs / fir_decim))
    return h, g
