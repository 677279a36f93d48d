"""WbfmDemod — Python mirror of the sdrfm_wbfm_* C entry points (multi-channel WBFM, BASELINE configs[4])."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import lib as _l

NB = 16


@dataclass
class WbfmConfig:
    proto_coeffs: np.ndarray          # p[0..P), P multiple of 16
    resamp_coeffs: np.ndarray         # g[0..Tg) at the L-times-upsampled band rate
    resamp_up: int = 6                # 200 kS/s * 6/25 = 48 kHz
    resamp_down: int = 25
    n_streams: int = 1
    max_bytes_per_call: int = 1 << 20
    device: int = 0
    force_generic: bool = False       # SDRFM_WBFM_CFG_FORCE_GENERIC (tests): never run the fused kernel
    run_steps: int = 0                # SDRFM_WBFM_CFG_RUN_STEPS (tests): fixed run length of the fused kernel, 0 = per call
    branch_lanes: bool = False        # SDRFM_WBFM_CFG_BRANCH_LANES (tests): the one-lane-per-branch fused kernel instead of one lane per step
    dev_library: bool = False         # load csrc/libsdrfm_dev.so (instrumented build; tools only)


class WbfmDemod:
    def __init__(self, cfg: WbfmConfig):
        self._lib = _l.load_library(dev=cfg.dev_library)
        self.cfg = cfg
        p = np.ascontiguousarray(cfg.proto_coeffs, dtype=np.float32)
        g = np.ascontiguousarray(cfg.resamp_coeffs, dtype=np.float32)
        c = _l.WbfmConfig()
        c.struct_size = C.sizeof(_l.WbfmConfig)
        c.n_streams = cfg.n_streams
        c.proto_taps, c.proto_coeffs = p.size, p.ctypes.data_as(C.POINTER(C.c_float))
        c.resamp_taps, c.resamp_coeffs = g.size, g.ctypes.data_as(C.POINTER(C.c_float))
        c.resamp_up, c.resamp_down = cfg.resamp_up, cfg.resamp_down
        c.max_bytes_per_call, c.device = cfg.max_bytes_per_call, cfg.device
        c.flags = (1 if cfg.force_generic else 0) | (2 if cfg.branch_lanes else 0) | (int(cfg.run_steps) << 8)
        self._h = C.c_void_p()
        st = self._lib.sdrfm_wbfm_create(C.byref(c), C.byref(self._h))
        if st != _l.OK:
            self._h = None
            raise _l.SdrfmError(st, "sdrfm_wbfm_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdrfm_wbfm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, where):
        if st != _l.OK:
            raise _l.SdrfmError(st, where)

    def reset(self):
        self._ck(self._lib.sdrfm_wbfm_reset(self._h), "sdrfm_wbfm_reset")

    def audio_count(self, nbytes):
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_wbfm_audio_count(self._h, int(nbytes), C.byref(n)), "sdrfm_wbfm_audio_count")
        return n.value

    def set_stream(self, ptr):
        self._ck(self._lib.sdrfm_wbfm_set_stream(self._h, C.c_void_p(int(ptr) if ptr else None)), "sdrfm_wbfm_set_stream")

    @property
    def kernel_name(self):
        return self._lib.sdrfm_wbfm_kernel_name(self._h).decode()

    def synchronize(self):
        self._ck(self._lib.sdrfm_wbfm_synchronize(self._h), "sdrfm_wbfm_synchronize")

    def process_batch(self, iq: np.ndarray) -> np.ndarray:
        """host memory: iq [n_streams, nbytes] uint8 -> audio [n_streams, 16, n_audio] float32"""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        if iq.ndim == 1:
            iq = iq[None, :]
        assert iq.shape[0] == self.cfg.n_streams
        nbytes = iq.shape[1]
        cap = max(self.audio_count(nbytes & ~1), 1)
        out = np.zeros((iq.shape[0], NB, cap), dtype=np.float32)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_wbfm_process_batch(self._h, iq.ctypes.data, nbytes, nbytes, out.ctypes.data, cap,
                                                    C.byref(n), 0), "sdrfm_wbfm_process_batch")
        return out[:, :, : n.value]

    def process_batch_device(self, iq, audio, nbytes=None):
        """device tensors: iq uint8 [n_streams, >=nbytes], audio float32 [n_streams, 16, cap]; enqueue only."""
        assert iq.is_cuda and audio.is_cuda and audio.dim() == 3 and audio.shape[1] == NB and audio.is_contiguous()
        nbytes = iq.shape[1] if nbytes is None else int(nbytes)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_wbfm_process_batch(self._h, C.c_void_p(iq.data_ptr()), iq.stride(0), nbytes,
                                                    C.c_void_p(audio.data_ptr()), audio.stride(1), C.byref(n),
                                                    _l.F_DEVICE_PTRS), "sdrfm_wbfm_process_batch(device)")
        return n.value
