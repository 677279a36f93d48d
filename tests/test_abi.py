"""CPU tests of the C-ABI library: it loads, exports everything include/sdrfm.h declares, fails loudly without a GPU,
and its K3 arithmetic (evaluated on the host through the test hooks) is accurate."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sdrfm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdrfm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 11
    for sym in declared:
        assert hasattr(lib, sym), "libsdrfm.so does not export %s" % sym
    assert sorted(pkg.ABI_SYMBOLS) == declared
    assert lib.sdrfm_abi_version() == 1


def test_test_hooks_live_in_their_own_header(pkg):
    """include/sdrfm.h is the drop-in boundary and nothing else; the hooks tests use are declared in include/sdrfm_dev.h.  The
    product library exports the arithmetic hooks (they compute nothing for a caller) but no instrumented-kernel accessor."""
    import subprocess
    boundary = _declared_symbols()
    src = open(os.path.join(ROOT, "include", "sdrfm_dev.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    hooks = sorted(set(re.findall(r"\b(sdrfm_[a-z0-9_]+)\s*\(", src)))
    assert sorted(hooks) == sorted(pkg.lib.TEST_HOOK_SYMBOLS + pkg.lib.DEV_ONLY_SYMBOLS)
    assert not set(hooks) & set(boundary)
    assert not [s for s in boundary if "debug" in s or "_host_" in s or "_dev_" in s]
    dyn = subprocess.run(["nm", "-D", pkg.library_path()], capture_output=True, text=True, check=True).stdout
    for sym in pkg.lib.TEST_HOOK_SYMBOLS:
        assert (" T " + sym) in dyn, sym
    for sym in pkg.lib.DEV_ONLY_SYMBOLS:
        assert sym not in dyn, sym
    dev = pkg.library_path(dev=True)
    if os.path.exists(dev):
        ddyn = subprocess.run(["nm", "-D", dev], capture_output=True, text=True, check=True).stdout
        for sym in pkg.lib.TEST_HOOK_SYMBOLS + pkg.lib.DEV_ONLY_SYMBOLS:
            assert (" T " + sym) in ddyn, sym


def test_status_codes_match_usbh_status_enum(pkg):
    # USBH_OK=0, USBH_BUSY, USBH_FAIL, USBH_NOT_SUPPORTED, USBH_UNRECOVERED_ERROR
    # (Middlewares/ST/STM32_USB_Host_Library/Core/Inc/usbh_def.h:303-311)
    assert [pkg.STATUS[i] for i in range(5)] == ["SDRFM_OK", "SDRFM_BUSY", "SDRFM_FAIL", "SDRFM_NOT_SUPPORTED",
                                                 "SDRFM_UNRECOVERED_ERROR"]
    lib = pkg.load_library()
    for code in pkg.STATUS:
        assert lib.sdrfm_strerror(code).decode() not in ("", "unknown status")
    assert lib.sdrfm_strerror(12345).decode() == "unknown status"


def test_create_rejects_bad_arguments_before_touching_the_gpu(pkg):
    from importlib import import_module
    L = import_module("stm32f7-rtlsdr_amd.lib")
    lib = pkg.load_library()
    h = np.ones(4, np.float32)
    c = L.Config()
    out = C.c_void_p()
    assert lib.sdrfm_create(None, C.byref(out)) == L.EINVAL
    c.struct_size = 3
    assert lib.sdrfm_create(C.byref(c), C.byref(out)) == L.EINVAL
    c.struct_size = C.sizeof(L.Config)
    c.n_streams, c.fir_taps, c.fir_decim, c.audio_taps, c.audio_decim = 1, 4, 10, 4, 5
    c.fir_coeffs = h.ctypes.data_as(C.POINTER(C.c_float))
    c.audio_coeffs = h.ctypes.data_as(C.POINTER(C.c_float))
    c.fir_taps = 257
    assert lib.sdrfm_create(C.byref(c), C.byref(out)) == L.EINVAL
    c.fir_taps, c.fir_decim = 4, 0
    assert lib.sdrfm_create(C.byref(c), C.byref(out)) == L.EINVAL
    c.fir_decim = 10
    bad = np.array([1, np.nan, 0, 0], np.float32)
    c.fir_coeffs = bad.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.sdrfm_create(C.byref(c), C.byref(out)) == L.EINVAL
    assert not out.value


def test_no_gpu_means_loud_failure_not_a_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h, g = pkg.default_config(16)
    with pytest.raises(pkg.SdrfmError) as e:
        pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g))
    assert e.value.status == 19  # SDRFM_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_host_evaluation_of_device_atan2_is_accurate(pkg):
    lib = pkg.load_library()
    rng = np.random.default_rng(0)
    ys = np.concatenate([rng.standard_normal(20000) * 10 ** rng.uniform(-6, 4, 20000), [0.0, -0.0, 1.0, -1.0, 1e-30, 3.0]])
    xs = np.concatenate([rng.standard_normal(20000) * 10 ** rng.uniform(-6, 4, 20000), [-1.0, -1.0, 0.0, 0.0, -1e30, 3.0]])
    ys, xs = ys.astype(np.float32), xs.astype(np.float32)
    got = np.array([lib.sdrfm_host_atan2f(float(y), float(x)) for y, x in zip(ys, xs)], dtype=np.float64)
    want = np.arctan2(ys.astype(np.float64), xs.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - want) / ulp) <= 4.0
    assert np.max(np.abs(got - want)) <= 8e-7
    # signed zeros behave like libm
    assert lib.sdrfm_host_atan2f(0.0, -1.0) == pytest.approx(np.pi)
    assert lib.sdrfm_host_atan2f(-0.0, -1.0) == pytest.approx(-np.pi)
    assert lib.sdrfm_host_atan2f(0.0, 0.0) == 0.0


def test_ordering_calls_reject_a_null_handle_without_touching_the_gpu(pkg):
    """sdrfm_flush / sdrfm_flush_previous / sdrfm_wait_previous / sdrfm_synchronize / sdrfm_set_stream (the ordering half of the boundary, include/sdrfm.h) answer
    SDRFM_EINVAL for a NULL handle — no device needed; what they do on a device is in tests/test_overlap_gpu.py."""
    lib = pkg.load_library()
    for fn in (lib.sdrfm_flush, lib.sdrfm_flush_previous, lib.sdrfm_synchronize):
        assert fn(None) == 16
    assert lib.sdrfm_set_stream(None, None) == 16
    assert lib.sdrfm_wait_previous(None, None) == 16                       # (round 6: a consumer's own stream behind call k - 1)
    n = __import__("ctypes").c_uint32()
    assert lib.sdrfm_process_batch(None, None, 0, 0, None, 0, __import__("ctypes").byref(n), 3) == 16
    assert lib.sdrfm_process_batch_pcm(None, None, None, 0, 0, None, 0, None, 0, __import__("ctypes").byref(n), 0) == 16   # (round 6: demodulator and PCM sink in one call)


def test_discriminator_conventions(pkg):
    lib = pkg.load_library()
    assert lib.sdrfm_host_discriminate(3.0, -2.0, 0.0, 0.0) == 0.0          # y[-1] = 0  ->  d[0] = 0
    assert lib.sdrfm_host_discriminate(5.0, 7.0, 5.0, 7.0) == 0.0           # identical samples -> exactly 0
    assert lib.sdrfm_host_discriminate(0.0, 1.0, 1.0, 0.0) == pytest.approx(np.pi / 2, abs=1e-6)
    assert lib.sdrfm_host_discriminate(1.0, 0.0, 0.0, 1.0) == pytest.approx(-np.pi / 2, abs=1e-6)


def test_product_library_holds_no_development_kernels_or_environment_knobs(pkg):
    """libsdrfm.so = result-correct kernels only: no ablation / instrumented / design-A instantiation and not one SDRFM_*
    environment variable name; those live in libsdrfm_dev.so (built with -DSDRFM_DEV)."""
    import subprocess
    path = pkg.library_path()
    syms = subprocess.run(["nm", "-C", path], capture_output=True, text=True, check=True).stdout
    kernels = re.findall(r"k_fastb?<[^>]*>", syms)
    assert kernels, "no specialised kernel found in the product library"
    for k in kernels:
        assert k.startswith("k_fastb<") and k.rstrip(">").split(",")[-1].strip() == "0", k    # MODE == 0 only
    blob = open(path, "rb").read()
    for knob in (b"SDRFM_ABLATE", b"SDRFM_PHASE_PROFILE", b"SDRFM_FAST_KIND", b"SDRFM_FAST_R", b"SDRFM_AUDIO_BATCH", b"SDRFM_WARM_AHEAD",
                 b"SDRFM_WAVES_PER_CU", b"SDRFM_MIN_SUBTILES", b"SDRFM_NO_PRIO", b"SDRFM_NO_FOLD", b"SDRFM_NO_ZEROCOPY",
                 b"SDRFM_WBFM_GENERIC", b"SDRFM_WBFM_NT", b"SDRFM_END_PRIO", b"SDRFM_NO_STREAM", b"SDRFM_STREAM_PROFILE", b"SDRFM_WBFM_PROFILE"):
        assert knob not in blob, knob
    # no development export, no getenv at all
    dyn = subprocess.run(["nm", "-D", path], capture_output=True, text=True, check=True).stdout
    assert "dev_read_debug" not in dyn and " getenv" not in dyn and "secure_getenv" not in dyn, "the product library reads no environment variable"
    dev = pkg.library_path(dev=True)
    if os.path.exists(dev):
        assert b"SDRFM_PHASE_PROFILE" in open(dev, "rb").read()
        ddyn = subprocess.run(["nm", "-D", dev], capture_output=True, text=True, check=True).stdout
        assert "sdrfm_dev_read_debug" in ddyn and "sdrfm_wbfm_dev_read_debug" in ddyn
