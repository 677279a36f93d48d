"""FmDemod — Python mirror of the sdrfm_* C entry points (include/sdrfm.h).

Host-buffer calls take/return numpy arrays; device-buffer calls take torch CUDA(HIP) tensors and only enqueue work on a
HIP stream (torch is used for device memory and streams only).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import lib as _l


@dataclass
class FmConfig:
    fir_coeffs: np.ndarray          # h[0..T)
    audio_coeffs: np.ndarray        # g[0..Ta)
    fir_decim: int = 10             # 2.4 MS/s -> 240 kS/s, the rate the firmware programs (usbh_rtlsdr.c:898)
    audio_decim: int = 5            # 240 kS/s -> 48 kHz
    n_streams: int = 1
    max_bytes_per_call: int = 1 << 20
    device: int = 0
    force_generic: bool = False     # SDRFM_CFG_FORCE_GENERIC: never use a (T,D)-specialised kernel (tests)
    bit_exact: bool = False         # SDRFM_CFG_BIT_EXACT: only kernels bit-identical to the fmaf-chain definition (never "fast-q")
    no_zerocopy: bool = False       # SDRFM_CFG_NO_ZEROCOPY: URB-sized host calls take the staged copy path (tests)
    guard_worst_case: bool = False  # SDRFM_CFG_GUARD_WORST_CASE: "fast-q"'s conditioning guard from the proven worst-case bound on |dy| (6.9 x the radius at 64 taps)
    dev_library: bool = False       # load csrc/libsdrfm_dev.so (instrumented / ablation kernels, SDRFM_* environment knobs)


class FmDemod:
    """One sdrfm_t handle. Not thread-safe (same model as the reference's single superloop, src/main.c:72-80)."""

    def __init__(self, cfg: FmConfig):
        self._lib = _l.load_library(dev=cfg.dev_library)
        self.cfg = cfg
        h = np.ascontiguousarray(cfg.fir_coeffs, dtype=np.float32)
        g = np.ascontiguousarray(cfg.audio_coeffs, dtype=np.float32)
        c = _l.Config()
        c.struct_size = C.sizeof(_l.Config)
        c.n_streams = cfg.n_streams
        c.fir_taps, c.fir_decim = h.size, cfg.fir_decim
        c.fir_coeffs = h.ctypes.data_as(C.POINTER(C.c_float))
        c.audio_taps, c.audio_decim = g.size, cfg.audio_decim
        c.audio_coeffs = g.ctypes.data_as(C.POINTER(C.c_float))
        c.max_bytes_per_call = cfg.max_bytes_per_call
        c.device = cfg.device
        c.flags = (1 if cfg.force_generic else 0) | (2 if cfg.no_zerocopy else 0) | (4 if cfg.bit_exact else 0) | (8 if cfg.guard_worst_case else 0)
        self._h = C.c_void_p()
        st = self._lib.sdrfm_create(C.byref(c), C.byref(self._h))
        if st != _l.OK:
            self._h = None
            raise _l.SdrfmError(st, "sdrfm_create")

    # -- lifecycle --------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdrfm_destroy(self._h)
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
        self._ck(self._lib.sdrfm_reset(self._h), "sdrfm_reset")

    @property
    def kernel_name(self):
        return self._lib.sdrfm_kernel_name(self._h).decode()

    def audio_count(self, nbytes):
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_audio_count(self._h, int(nbytes), C.byref(n)), "sdrfm_audio_count")
        return n.value

    def set_stream(self, hip_stream_ptr):
        self._ck(self._lib.sdrfm_set_stream(self._h, C.c_void_p(int(hip_stream_ptr) if hip_stream_ptr else None)),
                 "sdrfm_set_stream")

    def synchronize(self):
        self._ck(self._lib.sdrfm_synchronize(self._h), "sdrfm_synchronize")

    def q_guard(self):
        """sdrfm_debug_q_guard (include/sdrfm_dev.h): the matrix-pipe kernel's conditioning guard on this handle — dict with its two
        thresholds and the lanes repaired / repair passes run since create; None when the handle has no matrix-pipe kernel."""
        r, a, lanes, passes = C.c_float(), C.c_float(), C.c_uint64(), C.c_uint64()
        st = self._lib.sdrfm_debug_q_guard(self._h, C.byref(r), C.byref(a), C.byref(lanes), C.byref(passes))
        if st == 3:
            return None
        self._ck(st, "sdrfm_debug_q_guard")
        return {"guard_r": r.value, "guard_a": a.value, "lanes": lanes.value, "passes": passes.value}

    def route(self, mask=None):
        """sdrfm_debug_route (include/sdrfm_dev.h): which streams the bit-exact kernels serve beside the matrix-pipe kernel — a numpy array of 0 / 1 per stream,
        after setting it (mask: one entry per stream, non-zero = the bit-exact kernels) or after taking in the statistics that have arrived (mask=None); None when
        the handle has no matrix-pipe kernel."""
        import numpy as np
        ns = self.cfg.n_streams
        out = (C.c_uint8 * ns)()
        n = C.c_uint32()
        m = None
        if mask is not None:
            m = (C.c_uint8 * ns)(*[1 if x else 0 for x in mask])
        st = self._lib.sdrfm_debug_route(self._h, m, C.byref(n), out)
        if st == 3:
            return None
        self._ck(st, "sdrfm_debug_route")
        return np.frombuffer(out, dtype=np.uint8).copy()

    def phase_cycles(self):
        """sdrfm_debug_phase_cycles: dict of cumulative shader cycles per kernel phase (profiling builds only)."""
        out = (C.c_uint64 * 8)()
        self._ck(self._lib.sdrfm_debug_phase_cycles(self._h, out), "sdrfm_debug_phase_cycles")
        names = ["stage", "fir", "disc", "audio", "carry", "subtiles", "waves", "prologue"]
        return {n: int(out[i]) for i, n in enumerate(names)}

    # -- host buffers -----------------------------------------------------------------------------------------
    def process(self, iq: np.ndarray) -> np.ndarray:
        """sdrfm_process: single stream, uint8 interleaved I/Q in, float32 audio out."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
        # capacity from the handle itself; odd lengths are passed through so the C side reports SDRFM_EODD
        cap = self.audio_count(iq.size & ~1) + 1
        out = np.empty(cap, dtype=np.float32)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_process(self._h, iq.ctypes.data, iq.size, out.ctypes.data, cap, C.byref(n)),
                 "sdrfm_process")
        return out[: n.value]

    def process_batch(self, iq: np.ndarray) -> np.ndarray:
        """sdrfm_process_batch on host memory: iq [n_streams, nbytes] uint8 -> audio [n_streams, n_audio] float32."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        assert iq.ndim == 2 and iq.shape[0] == self.cfg.n_streams
        nbytes = iq.shape[1]
        cap = max(self.audio_count(nbytes & ~1), 1)
        out = np.empty((iq.shape[0], cap), dtype=np.float32)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_process_batch(self._h, iq.ctypes.data, nbytes, nbytes, out.ctypes.data, cap,
                                               C.byref(n), 0), "sdrfm_process_batch")
        return out[:, : n.value]

    # -- device buffers (torch tensors on cfg.device) -----------------------------------------------------------
    def process_batch_device(self, iq, audio, nbytes=None, overlap=False):
        """Enqueue one batch on device-resident buffers (SDRFM_F_DEVICE_PTRS); returns n_audio per stream.

        iq: torch.uint8 [n_streams, >=nbytes] (row stride = iq.stride(0)); audio: torch.float32 [n_streams, cap].
        Nothing is synchronised; the work runs on the stream given to set_stream() (or the handle's own).
        overlap=True adds SDRFM_F_OVERLAP (include/sdrfm.h): consecutive calls may run concurrently; the previous call's `iq` must stay
        intact and `audio` must alternate between two buffers; flush() / synchronize() order the stream behind them.
        """
        assert iq.is_cuda and audio.is_cuda and iq.dim() == 2 and audio.dim() == 2
        assert iq.stride(1) == 1 and audio.stride(1) == 1
        nbytes = iq.shape[1] if nbytes is None else int(nbytes)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_process_batch(self._h, C.c_void_p(iq.data_ptr()), iq.stride(0), nbytes,
                                               C.c_void_p(audio.data_ptr()), audio.stride(0), C.byref(n),
                                               _l.F_DEVICE_PTRS | (_l.F_OVERLAP if overlap else 0)), "sdrfm_process_batch(device)")
        return n.value

    def process_batch_pcm(self, sink, iq, want_audio=False):
        """sdrfm_process_batch_pcm on HOST buffers (synchronous): iq uint8 [n_streams, nbytes] -> int16 PCM [n_streams, 2 * n_audio] (L = R, de-emphasised by the
        device sink `sink`), and the float audio too when want_audio.  The reference superloop's two steps on one filled buffer."""
        import numpy as np
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        if iq.ndim == 1:
            iq = iq[None, :]
        ns, nbytes = iq.shape
        assert ns == self.cfg.n_streams
        cap = self.audio_count(nbytes)
        pcm = np.zeros((ns, 2 * cap), np.int16)
        audio = np.zeros((ns, cap), np.float32) if want_audio else None
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_process_batch_pcm(self._h, sink._h, iq.ctypes.data_as(C.c_void_p), iq.strides[0], nbytes,
                                                   audio.ctypes.data_as(C.c_void_p) if want_audio else None, cap, pcm.ctypes.data_as(C.c_void_p), 2 * cap,
                                                   C.byref(n), 0), "sdrfm_process_batch_pcm(host)")
        return (pcm[:, :2 * n.value], audio[:, :n.value]) if want_audio else pcm[:, :2 * n.value]

    def process_batch_pcm_device(self, sink, iq, audio, pcm, nbytes=None, overlap=False):
        """sdrfm_process_batch_pcm: one call of the demodulator and of the device PCM sink `sink` (a PcmSink of this device and this many streams) —
        where design Q serves the call the sink's chain runs inside its launch (kernel_name ends in "+ pcm"), any other call is followed by the sink's
        own kernel on the handle's stream.  pcm: torch.int16 [n_streams, >= 2 * n_audio]; with overlap=True rotate pcm like audio.  Returns n_audio."""
        assert iq.is_cuda and pcm.is_cuda and iq.dim() == 2 and pcm.dim() == 2 and iq.stride(1) == 1 and pcm.stride(1) == 1
        assert audio is None or (audio.is_cuda and audio.dim() == 2 and audio.stride(1) == 1)      # (None: the PCM is all the call leaves)
        nbytes = iq.shape[1] if nbytes is None else int(nbytes)
        n = C.c_uint32()
        self._ck(self._lib.sdrfm_process_batch_pcm(self._h, sink._h, C.c_void_p(iq.data_ptr()), iq.stride(0), nbytes,
                                                   C.c_void_p(audio.data_ptr() if audio is not None else None), audio.stride(0) if audio is not None else 0,
                                                   C.c_void_p(pcm.data_ptr()), pcm.stride(0),
                                                   C.byref(n), _l.F_DEVICE_PTRS | (_l.F_OVERLAP if overlap else 0)), "sdrfm_process_batch_pcm")
        return n.value

    def wait_previous(self, hip_stream_ptr):
        """sdrfm_wait_previous: the caller's stream `hip_stream_ptr` is ordered behind every overlapped call but the most recent one; the handle's own stream is
        left alone (a consumer of call k-1's audio on its own stream, beside call k)."""
        self._ck(self._lib.sdrfm_wait_previous(self._h, C.c_void_p(int(hip_stream_ptr))), "sdrfm_wait_previous")

    def flush(self, keep_last=False):
        """Order the handle's stream behind every overlapped call made so far (sdrfm_flush) — or, with keep_last, behind all but the most
        recent one (sdrfm_flush_previous); does not block the host."""
        if keep_last:
            self._ck(self._lib.sdrfm_flush_previous(self._h), "sdrfm_flush_previous")
        else:
            self._ck(self._lib.sdrfm_flush(self._h), "sdrfm_flush")
