"""SpectrumView — Python mirror of the sdrfm_spectrum_* C entry points: the averaged windowed power spectrum of an IQ buffer
(the reference's own next task, README.md:29), DC in the middle."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import lib as _l


@dataclass
class SpectrumConfig:
    nfft: int = 1024                  # power of two, 64..4096
    window: Optional[np.ndarray] = None   # nfft floats; None = periodic Hann (computed by the library)
    n_streams: int = 1
    max_bytes_per_call: int = 1 << 20
    device: int = 0
    dev_library: bool = False         # tools only: csrc/libsdrfm_dev.so (phase stamps, SDRFM_SPEC_VARIANT)


class SpectrumView:
    def __init__(self, cfg: SpectrumConfig):
        self._lib = _l.load_library(dev=cfg.dev_library)
        self.cfg = cfg
        c = _l.SpectrumConfig()
        c.struct_size = C.sizeof(_l.SpectrumConfig)
        c.n_streams, c.nfft = cfg.n_streams, cfg.nfft
        self._win = None if cfg.window is None else np.ascontiguousarray(cfg.window, dtype=np.float32)
        if self._win is not None and self._win.size != cfg.nfft:
            raise ValueError("window length != nfft")
        c.window = None if self._win is None else self._win.ctypes.data_as(C.POINTER(C.c_float))
        c.max_bytes_per_call, c.device, c.flags = cfg.max_bytes_per_call, cfg.device, 0
        self._h = C.c_void_p()
        st = self._lib.sdrfm_spectrum_create(C.byref(c), C.byref(self._h))
        if st != _l.OK:
            self._h = None
            raise _l.SdrfmError(st, "sdrfm_spectrum_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdrfm_spectrum_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, where):
        if st != _l.OK:
            raise _l.SdrfmError(st, where)

    def set_stream(self, ptr):
        self._ck(self._lib.sdrfm_spectrum_set_stream(self._h, C.c_void_p(int(ptr) if ptr else None)), "sdrfm_spectrum_set_stream")

    def synchronize(self):
        self._ck(self._lib.sdrfm_spectrum_synchronize(self._h), "sdrfm_spectrum_synchronize")

    @property
    def kernel_name(self):
        """the kernel the last call launched (sdrfm_spectrum_kernel_name)"""
        return self._lib.sdrfm_spectrum_kernel_name(self._h).decode()

    def process_batch(self, iq: np.ndarray):
        """host memory: iq [n_streams, nbytes] uint8 -> (power [n_streams, nfft] float32, frames averaged)"""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        if iq.ndim == 1:
            iq = iq[None, :]
        assert iq.shape[0] == self.cfg.n_streams
        out = np.zeros((iq.shape[0], self.cfg.nfft), dtype=np.float32)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_spectrum_process_batch(self._h, iq.ctypes.data, iq.shape[1], iq.shape[1], out.ctypes.data,
                                                        self.cfg.nfft, C.byref(n), 0), "sdrfm_spectrum_process_batch")
        return out, n.value

    def process_batch_device(self, iq, power, nbytes=None):
        """device tensors: iq uint8 [n_streams, >=nbytes], power float32 [n_streams, >=nfft]; enqueue only -> frames"""
        assert iq.is_cuda and power.is_cuda and power.dim() == 2 and power.stride(1) == 1
        nbytes = iq.shape[1] if nbytes is None else int(nbytes)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_spectrum_process_batch(self._h, C.c_void_p(iq.data_ptr()), iq.stride(0), nbytes,
                                                        C.c_void_p(power.data_ptr()), power.stride(0), C.byref(n),
                                                        _l.F_DEVICE_PTRS), "sdrfm_spectrum_process_batch(device)")
        return n.value


def power_db(power, full_scale=None):
    """10*log10 of a power spectrum relative to full_scale (default: its own maximum); display helper, host side."""
    power = np.asarray(power, dtype=np.float64)
    ref = float(power.max()) if full_scale is None else float(full_scale)
    return 10.0 * np.log10(np.maximum(power, 1e-30) / max(ref, 1e-30))
