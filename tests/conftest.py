import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build native pieces once per session if they are missing (no-op when the .so files travelled with the repo)."""
    need = [os.path.join(ROOT, "stm32f7-rtlsdr_amd", "csrc", "libsdrfm.so"),
            os.path.join(ROOT, "tools", "siggen", "libsiggen.so"), os.path.join(ROOT, "oracle", "libsdrfm_oracle.so"),
            os.path.join(ROOT, "oracle", "libsdrfm_wbfm_oracle.so"), os.path.join(ROOT, "oracle", "libsdrfm_spectrum_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session")
def pkg(_built):
    return importlib.import_module("stm32f7-rtlsdr_amd")


@pytest.fixture(scope="session")
def oracle_mod(_built):
    from oracle import oracle
    return oracle


TOL = 1e-5  # north-star tolerance: |a-b| <= 1e-5 * max(|b|, 1)   (audio is in radians, full scale pi)


def scaled_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if got.size == 0:
        return 0.0
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)))


@pytest.fixture(scope="session")
def tol():
    return TOL


# ---- exact fp32 arithmetic in pure Python (small cases only; shared by the oracle and sink tests) ----------------------
def round_f32(fr):
    """Correctly rounded (nearest-even) Fraction -> float32, no double rounding."""
    from fractions import Fraction
    if fr == 0:
        return np.float32(0.0)
    c = np.float32(float(fr))  # candidate (double-rounded, at most one ulp off)
    best = None
    for cand in (np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))):
        err = abs(Fraction(float(cand)) - fr)
        key = (err, int(np.float32(cand).view(np.uint32)) & 1)
        if best is None or key < best[0]:
            best = (key, cand)
    return np.float32(best[1])


def fma32(a, b, c):
    from fractions import Fraction
    return round_f32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))
