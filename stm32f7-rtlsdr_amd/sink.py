"""PcmSink — Python mirror of the sdrfm_pcm_sink_* C entry points: de-emphasis + int16 stereo PCM on the device, in the
layout BSP_AUDIO_OUT_Play takes (Utilities/STM32746G-Discovery/stm32746g_discovery_audio.c:224)."""
import ctypes as C

import numpy as np

from . import lib as _l


def pcm_deemph_s16_host(audio, alpha, gain, state=0.0):
    """sdrfm_pcm_deemph_s16 (the host-side C routine) on one stream: returns (pcm int16 [2n], new state)."""
    lib = _l.load_library()
    x = np.ascontiguousarray(audio, dtype=np.float32)
    st = C.c_float(state)
    pcm = np.zeros(2 * x.size, np.int16)
    rc = lib.sdrfm_pcm_deemph_s16(x.ctypes.data, x.size, alpha, gain, C.byref(st), pcm.ctypes.data)
    if rc != _l.OK:
        raise _l.SdrfmError(rc, "sdrfm_pcm_deemph_s16")
    return pcm, st.value


PCM_F_EXACT = 4   # SDRFM_PCM_F_EXACT (include/sdrfm.h)


class PcmSink:
    """exact=False (default): the blocked-scan kernel (PCM within 1 LSB of the exact form); exact=True: SDRFM_PCM_F_EXACT, one lane per stream,
    bit-identical to the host routine sdrfm_pcm_deemph_s16."""

    def __init__(self, n_streams, alpha, gain, device=0, exact=False):
        self._lib = _l.load_library()
        self._xf = PCM_F_EXACT if exact else 0
        self.n_streams = int(n_streams)
        self._h = C.c_void_p()
        st = self._lib.sdrfm_pcm_sink_create(self.n_streams, alpha, gain, device, C.byref(self._h))
        if st != _l.OK:
            self._h = None
            raise _l.SdrfmError(st, "sdrfm_pcm_sink_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdrfm_pcm_sink_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, st, where):
        if st != _l.OK:
            raise _l.SdrfmError(st, where)

    def reset(self):
        self._ck(self._lib.sdrfm_pcm_sink_reset(self._h), "sdrfm_pcm_sink_reset")

    def set_stream(self, ptr):
        self._ck(self._lib.sdrfm_pcm_sink_set_stream(self._h, C.c_void_p(int(ptr) if ptr else None)), "sdrfm_pcm_sink_set_stream")

    def synchronize(self):
        self._ck(self._lib.sdrfm_pcm_sink_synchronize(self._h), "sdrfm_pcm_sink_synchronize")

    def synchronize_status(self):
        """sdrfm_pcm_sink_synchronize's status as it is (0 = fine; SDRFM_FAIL when a run of a demodulator launch waited in vain for its predecessor's state)."""
        return int(self._lib.sdrfm_pcm_sink_synchronize(self._h))

    def state(self):
        out = np.zeros(self.n_streams, np.float32)
        self._ck(self._lib.sdrfm_pcm_sink_get_state(self._h, out.ctypes.data_as(C.POINTER(C.c_float))), "sdrfm_pcm_sink_get_state")
        return out

    def process_batch(self, audio: np.ndarray) -> np.ndarray:
        """host memory: audio [n_streams, n] float32 -> pcm [n_streams, 2n] int16 (L = R)"""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        if a.ndim == 1:
            a = a[None, :]
        assert a.shape[0] == self.n_streams
        n = a.shape[1]
        pcm = np.zeros((self.n_streams, 2 * n), np.int16)
        self._ck(self._lib.sdrfm_pcm_sink_process_batch(self._h, a.ctypes.data, n, n, pcm.ctypes.data, 2 * n, self._xf),
                 "sdrfm_pcm_sink_process_batch")
        return pcm

    def process_batch_device(self, audio, pcm, n=None):
        """torch tensors on the device: audio float32 [n_streams, >=n], pcm int16 [n_streams, >=2n]; only enqueues."""
        assert audio.is_cuda and pcm.is_cuda and audio.stride(1) == 1 and pcm.stride(1) == 1
        n = audio.shape[1] if n is None else int(n)
        self._ck(self._lib.sdrfm_pcm_sink_process_batch(self._h, C.c_void_p(audio.data_ptr()), audio.stride(0), n,
                                                        C.c_void_p(pcm.data_ptr()), pcm.stride(0), _l.F_DEVICE_PTRS | self._xf),
                 "sdrfm_pcm_sink_process_batch(device)")
