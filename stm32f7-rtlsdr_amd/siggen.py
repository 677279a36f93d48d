"""Synthetic IQ byte streams (tools/siggen/siggen.c) — inputs only, shared by tests and bench."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {"fm": 0, "random": 1, "const": 2, "counter": 3}
_lib = None


def _siggen():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(_HERE), "tools", "siggen", "libsiggen.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: run __graft_entry__.build()" % path)
        lib = C.CDLL(path)
        lib.siggen_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_double]
        lib.siggen_fill.restype = None
        lib.siggen_lowpass.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
        lib.siggen_lowpass.restype = None
        _lib = lib
    return _lib


def make_iq(n_streams, n_samples, mode="fm", fs=2.4e6, first_id=0, threads=None, out=None):
    """uint8 array [n_streams, 2*n_samples] of interleaved I/Q; stream s uses PRNG id first_id+s."""
    lib = _siggen()
    m = MODES[mode]
    if out is None:
        out = np.empty((n_streams, 2 * n_samples), dtype=np.uint8)
    assert out.shape == (n_streams, 2 * n_samples) and out.dtype == np.uint8 and out.flags.c_contiguous

    def one(s):
        lib.siggen_fill(out[s].ctypes.data, n_samples, first_id + s, m, float(fs))

    threads = threads or min(32, os.cpu_count() or 1)
    if n_streams == 1 or threads == 1:
        for s in range(n_streams):
            one(s)
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one, range(n_streams)))
    return out
