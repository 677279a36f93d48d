"""ctypes wrapper of oracle/libsdrfm_oracle.so — TEST INFRASTRUCTURE (tests/, smoke(), bench.py's cpu_baseline only)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libsdrfm_oracle.so")
        if not os.path.exists(path):
            raise ImportError("%s missing: run `make -C oracle` or __graft_entry__.build()" % path)
        lib = C.CDLL(path)
        vp, f32p = C.c_void_p, C.c_void_p
        lib.sdrfm_oracle_create.argtypes = [C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_uint32, f32p]
        lib.sdrfm_oracle_create.restype = vp
        lib.sdrfm_oracle_destroy.argtypes = [vp]
        lib.sdrfm_oracle_destroy.restype = None
        lib.sdrfm_oracle_reset.argtypes = [vp]
        lib.sdrfm_oracle_reset.restype = None
        lib.sdrfm_oracle_process.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
        lib.sdrfm_oracle_process.restype = C.c_long
        lib.sdrfm_oracle_process_f64.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
        lib.sdrfm_oracle_process_f64.restype = C.c_long
        lib.sdrfm_oracle_last_stage.argtypes = [vp, vp, vp, C.c_size_t]
        lib.sdrfm_oracle_last_stage.restype = C.c_size_t
        _lib = lib
    return _lib


class Oracle:
    """One stream of the scalar-C oracle."""

    def __init__(self, h, g, D=10, Da=5):
        self._lib = _load()
        self.h = np.ascontiguousarray(h, dtype=np.float32)
        self.g = np.ascontiguousarray(g, dtype=np.float32)
        self.D, self.Da = int(D), int(Da)
        self._o = self._lib.sdrfm_oracle_create(self.h.size, self.D, self.h.ctypes.data, self.g.size, self.Da,
                                                self.g.ctypes.data)
        if not self._o:
            raise ValueError("sdrfm_oracle_create failed")

    def close(self):
        if self._o:
            self._lib.sdrfm_oracle_destroy(self._o)
            self._o = None

    def __del__(self):
        self.close()

    def reset(self):
        self._lib.sdrfm_oracle_reset(self._o)

    def _cap(self, nbytes):
        return nbytes // 2 // self.D // self.Da + 2

    def process(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
        out = np.empty(self._cap(iq.size), dtype=np.float32)
        n = self._lib.sdrfm_oracle_process(self._o, iq.ctypes.data, iq.size, out.ctypes.data, out.size)
        if n < 0:
            raise ValueError("sdrfm_oracle_process rejected its arguments")
        return out[:n].copy()

    def process_f64(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
        out = np.empty(self._cap(iq.size), dtype=np.float64)
        n = self._lib.sdrfm_oracle_process_f64(self._o, iq.ctypes.data, iq.size, out.ctypes.data, out.size)
        if n < 0:
            raise ValueError("sdrfm_oracle_process_f64 rejected its arguments")
        return out[:n].copy()

    def last_stage(self):
        n = self._lib.sdrfm_oracle_last_stage(self._o, None, None, 0)
        y = np.empty(2 * n, dtype=np.float32)
        d = np.empty(n, dtype=np.float32)
        self._lib.sdrfm_oracle_last_stage(self._o, y.ctypes.data, d.ctypes.data, n)
        return y.reshape(-1, 2), d


def process_batch(h, g, iq2d, D=10, Da=5):
    """[n_streams, nbytes] -> [n_streams, n_audio] with a fresh oracle per stream."""
    outs = []
    for s in range(iq2d.shape[0]):
        o = Oracle(h, g, D, Da)
        outs.append(o.process(iq2d[s]))
        o.close()
    return np.stack(outs)


# ---- multi-channel WBFM (BASELINE configs[4]) --------------------------------------------------------------------
_wlib = None


def _wload():
    global _wlib
    if _wlib is None:
        path = os.path.join(_HERE, "libsdrfm_wbfm_oracle.so")
        if not os.path.exists(path):
            raise ImportError("%s missing: run `make -C oracle`" % path)
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.sdrfm_wbfm_oracle_create.argtypes = [C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        lib.sdrfm_wbfm_oracle_create.restype = vp
        lib.sdrfm_wbfm_oracle_destroy.argtypes = [vp]
        lib.sdrfm_wbfm_oracle_destroy.restype = None
        lib.sdrfm_wbfm_oracle_process.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
        lib.sdrfm_wbfm_oracle_process.restype = C.c_long
        _wlib = lib
    return _wlib


class WbfmOracle:
    """One stream of the 16-band channelizer + per-band FM demod + L/M resampler oracle."""
    NB = 16

    def __init__(self, p, g, L=6, M=25):
        self._lib = _wload()
        self.p = np.ascontiguousarray(p, dtype=np.float32)
        self.g = np.ascontiguousarray(g, dtype=np.float32)
        self.L, self.M = int(L), int(M)
        self._o = self._lib.sdrfm_wbfm_oracle_create(self.p.size, self.p.ctypes.data, self.g.size, self.L, self.M, self.g.ctypes.data)
        if not self._o:
            raise ValueError("sdrfm_wbfm_oracle_create failed")

    def close(self):
        if self._o:
            self._lib.sdrfm_wbfm_oracle_destroy(self._o)
            self._o = None

    def __del__(self):
        self.close()

    def process(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
        cap = (iq.size // 2 // self.NB + 2) * self.L // self.M + 2
        out = np.zeros((self.NB, cap), dtype=np.float32)
        n = self._lib.sdrfm_wbfm_oracle_process(self._o, iq.ctypes.data, iq.size, out.ctypes.data, cap)
        if n < 0:
            raise ValueError("sdrfm_wbfm_oracle_process rejected its arguments")
        return out[:, :n].copy()


_slib = None


def _sload():
    global _slib
    if _slib is None:
        path = os.path.join(_HERE, "libsdrfm_spectrum_oracle.so")
        if not os.path.exists(path):
            raise ImportError("%s missing: run `make -C oracle`" % path)
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.sdrfm_spectrum_oracle_create.argtypes = [C.c_uint32, vp]
        lib.sdrfm_spectrum_oracle_create.restype = vp
        lib.sdrfm_spectrum_oracle_destroy.argtypes = [vp]
        lib.sdrfm_spectrum_oracle_destroy.restype = None
        lib.sdrfm_spectrum_oracle_process.argtypes = [vp, vp, C.c_uint64, vp]
        lib.sdrfm_spectrum_oracle_process.restype = C.c_long
        _slib = lib
    return _slib


class SpectrumOracle:
    """Averaged windowed power spectrum of one IQ buffer (oracle/sdrfm_spectrum_oracle.c); window=None -> periodic Hann."""

    def __init__(self, nfft=1024, window=None):
        self._lib = _sload()
        self.nfft = int(nfft)
        self.window = None if window is None else np.ascontiguousarray(window, dtype=np.float32)
        if self.window is not None and self.window.size != self.nfft:
            raise ValueError("window length != nfft")
        self._o = self._lib.sdrfm_spectrum_oracle_create(self.nfft, None if self.window is None else self.window.ctypes.data)
        if not self._o:
            raise ValueError("sdrfm_spectrum_oracle_create failed (nfft must be a power of two)")

    def close(self):
        if self._o:
            self._lib.sdrfm_spectrum_oracle_destroy(self._o)
            self._o = None

    def __del__(self):
        self.close()

    def process(self, iq):
        """-> (power[nfft] float32, DC in the middle; number of frames)"""
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
        out = np.zeros(self.nfft, dtype=np.float32)
        n = self._lib.sdrfm_spectrum_oracle_process(self._o, iq.ctypes.data, iq.size, out.ctypes.data)
        if n < 0:
            raise ValueError("sdrfm_spectrum_oracle_process rejected its arguments")
        return out, int(n)
