"""Known-answer tests of the host-side RTL2832 configuration arithmetic (SURVEY.md §8c values, produced by the reference's
own code: RTLSDR_set_fir CALC state and RTLSDR_set_sample_rate state 0)."""
import ctypes as C

import pytest


def test_fir_register_image(pkg):
    lib = pkg.load_library()
    fir = (C.c_int * 16)(*pkg.RTLSDR_FIR)
    out = (C.c_uint8 * 20)()
    assert lib.sdrfm_rtl_pack_fir(fir, out) == 0
    assert bytes(out).hex() == "cadcd7d8e0f20e3506509c0d7111147174194 1a5".replace(" ", "")
    bad = (C.c_int * 16)(*([200] + list(pkg.RTLSDR_FIR[1:])))
    assert lib.sdrfm_rtl_pack_fir(bad, out) == 16          # the firmware only logs "Invalid FIR coefficient!"


@pytest.mark.parametrize("rate,ratio,real", [
    (240000, 0x0e000000, 0x1e000000),        # the rate the firmware programs (usbh_rtlsdr.c:898)
    (1024000, 0x07080000, 0x07080000), (2048000, 0x03840000, 0x03840000),
    (2400000, 0x03000000, 0x03000000), (3200000, 0x02400000, 0x02400000),
])
def test_resampler_ratio(pkg, rate, ratio, real):
    lib = pkg.load_library()
    r, rr, f = C.c_uint32(), C.c_uint32(), C.c_double()
    assert lib.sdrfm_rtl_resampler(rate, 28800000, C.byref(r), C.byref(rr), C.byref(f)) == 0
    assert r.value == ratio and rr.value == real
    if rate == 240000:
        assert f.value == 240000.0


def test_resampler_validity_window(pkg):
    lib = pkg.load_library()
    r, rr, f = C.c_uint32(), C.c_uint32(), C.c_double()
    for bad in (225000, 300001, 900000, 3200001, 0):
        assert lib.sdrfm_rtl_resampler(bad, 28800000, C.byref(r), C.byref(rr), C.byref(f)) == 16
    for ok in (225001, 300000, 900001, 3200000):
        assert lib.sdrfm_rtl_resampler(ok, 28800000, C.byref(r), C.byref(rr), C.byref(f)) == 0
