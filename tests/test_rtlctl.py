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


class _E4kPll(C.Structure):
    _fields_ = [("fosc", C.c_uint32), ("intended_flo", C.c_uint32), ("flo", C.c_uint32), ("x", C.c_uint16),
                ("z", C.c_uint8), ("r", C.c_uint8), ("r_idx", C.c_uint8), ("threephase", C.c_uint8)]


@pytest.mark.parametrize("want_hz,flo,z,x,r,r_idx,three", [
    (99700000, 99699993, 110, 50972, 32, 13, 1),          # E4K_compute_pll_params(fosc = 28.8 MHz) run here
    (433920000, 433919970, 90, 26214, 6, 2, 0),
    (1090000000, 1089999975, 151, 25486, 4, 1, 0),
])
def test_e4k_pll_known_answers(pkg, want_hz, flo, z, x, r, r_idx, three):
    lib = pkg.load_library()
    p = _E4kPll()
    assert lib.sdrfm_e4k_pll_params(28800000, want_hz, C.byref(p)) == 0
    assert (p.flo, p.z, p.x, p.r, p.r_idx, p.threephase) == (flo, z, x, r, r_idx, three)
    assert p.fosc == 28800000 and p.intended_flo == want_hz
    assert 0 <= want_hz - p.flo < 28800000 / 65536 / r + 1   # the fraction has 1/65536 resolution, truncated


def test_e4k_pll_bands_and_errors(pkg):
    lib = pkg.load_library()
    p = _E4kPll()
    for hz, r in ((60000000, 48), (72399999, 48), (72400000, 40), (200000000, 16), (349999999, 8), (350000000, 8),
                  (500000000, 6), (1199999999, 4), (1200000000, 2), (1700000000, 2)):
        assert lib.sdrfm_e4k_pll_params(28800000, hz, C.byref(p)) == 0
        assert p.r == r, hz
        assert p.threephase == (1 if hz < 350000000 else 0)
        assert abs(p.flo - hz) <= 500
    assert lib.sdrfm_e4k_pll_params(15999999, 100000000, C.byref(p)) == 16
    assert lib.sdrfm_e4k_pll_params(30000001, 100000000, C.byref(p)) == 16
    assert lib.sdrfm_e4k_pll_params(28800000, 100000000, None) == 16
