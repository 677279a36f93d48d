"""ctypes binding of csrc/libsdrfm.so — one Python function per C entry point of include/sdrfm.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

STATUS = {
    0: "SDRFM_OK", 1: "SDRFM_BUSY", 2: "SDRFM_FAIL", 3: "SDRFM_NOT_SUPPORTED", 4: "SDRFM_UNRECOVERED_ERROR",
    16: "SDRFM_EINVAL", 17: "SDRFM_EODD", 18: "SDRFM_ECAPACITY", 19: "SDRFM_NO_DEVICE", 20: "SDRFM_ENOMEM",
}
OK, EINVAL, EODD, ECAPACITY, NO_DEVICE = 0, 16, 17, 18, 19
F_DEVICE_PTRS = 1
F_OVERLAP = 2

# every symbol include/sdrfm.h declares
ABI_SYMBOLS = [
    "sdrfm_create", "sdrfm_destroy", "sdrfm_reset", "sdrfm_audio_count", "sdrfm_process", "sdrfm_process_batch",
    "sdrfm_set_stream", "sdrfm_synchronize", "sdrfm_flush", "sdrfm_flush_previous", "sdrfm_wait_previous", "sdrfm_process_batch_pcm", "sdrfm_kernel_name", "sdrfm_abi_version", "sdrfm_strerror",
    "sdrfm_wbfm_create", "sdrfm_wbfm_destroy", "sdrfm_wbfm_reset", "sdrfm_wbfm_audio_count", "sdrfm_wbfm_process_batch",
    "sdrfm_wbfm_set_stream", "sdrfm_wbfm_synchronize", "sdrfm_wbfm_kernel_name", "sdrfm_rtl_pack_fir", "sdrfm_rtl_resampler", "sdrfm_e4k_pll_params",
    "sdrfm_shard_range",
    "sdrfm_spectrum_create", "sdrfm_spectrum_destroy", "sdrfm_spectrum_process_batch", "sdrfm_spectrum_set_stream",
    "sdrfm_spectrum_synchronize", "sdrfm_spectrum_kernel_name",
    "sdrfm_pcm_deemph_s16", "sdrfm_pcm_alpha",
    "sdrfm_pcm_sink_create", "sdrfm_pcm_sink_destroy", "sdrfm_pcm_sink_reset", "sdrfm_pcm_sink_process_batch",
    "sdrfm_pcm_sink_set_stream", "sdrfm_pcm_sink_synchronize", "sdrfm_pcm_sink_get_state",
    "sdrfm_ring_create", "sdrfm_ring_destroy", "sdrfm_ring_submit", "sdrfm_ring_collect",
]


# include/sdrfm_dev.h: test hooks the product library also exports / development-library-only aids (not the drop-in boundary)
TEST_HOOK_SYMBOLS = ["sdrfm_host_atan2f", "sdrfm_host_discriminate", "sdrfm_debug_discriminate", "sdrfm_q_build", "sdrfm_q_guard", "sdrfm_q_guard2", "sdrfm_debug_q_guard", "sdrfm_debug_read_ceiling", "sdrfm_debug_route"]
DEV_ONLY_SYMBOLS = ["sdrfm_debug_phase_cycles", "sdrfm_debug_raw", "sdrfm_dev_read_debug"]


class SdrfmError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = int(status)
        name = STATUS.get(self.status, "status %d" % self.status)
        msg = ""
        try:
            msg = load_library().sdrfm_strerror(self.status).decode()
        except Exception:  # pragma: no cover
            pass
        super().__init__("%s%s (%s)" % (where + ": " if where else "", name, msg))


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_streams", C.c_uint32), ("fir_taps", C.c_uint32), ("fir_decim", C.c_uint32),
        ("fir_coeffs", C.POINTER(C.c_float)), ("audio_taps", C.c_uint32), ("audio_decim", C.c_uint32),
        ("audio_coeffs", C.POINTER(C.c_float)), ("max_bytes_per_call", C.c_uint32), ("device", C.c_int32),
        ("flags", C.c_uint32),
    ]


class WbfmConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_streams", C.c_uint32), ("proto_taps", C.c_uint32),
        ("proto_coeffs", C.POINTER(C.c_float)), ("resamp_taps", C.c_uint32), ("resamp_up", C.c_uint32),
        ("resamp_down", C.c_uint32), ("resamp_coeffs", C.POINTER(C.c_float)), ("max_bytes_per_call", C.c_uint32),
        ("device", C.c_int32), ("flags", C.c_uint32),
    ]


class SpectrumConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_streams", C.c_uint32), ("nfft", C.c_uint32), ("window", C.POINTER(C.c_float)),
        ("max_bytes_per_call", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
    ]


def library_path(dev=False):
    """csrc/libsdrfm.so (the product) or, with dev=True, csrc/libsdrfm_dev.so (same sources built with -DSDRFM_DEV: adds the
    instrumented / ablation kernels and the SDRFM_* environment knobs that the scripts under tools/ use)."""
    return os.path.join(_HERE, "csrc", "libsdrfm_dev.so" if dev else "libsdrfm.so")


_libs = {}


def load_library(dev=False):
    """Load libsdrfm.so (or the development build) or raise — never substitute anything else for it."""
    if dev in _libs:
        return _libs[dev]
    path = library_path(dev)
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C stm32f7-rtlsdr_amd/csrc%s`). There is no fallback implementation." % (path, " dev" if dev else ""))
    lib = C.CDLL(path)
    vp, u32, u32p = C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)
    lib.sdrfm_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.sdrfm_create.restype = C.c_int
    lib.sdrfm_destroy.argtypes = [vp]
    lib.sdrfm_destroy.restype = None
    lib.sdrfm_reset.argtypes = [vp]
    lib.sdrfm_reset.restype = C.c_int
    lib.sdrfm_audio_count.argtypes = [vp, u32, u32p]
    lib.sdrfm_audio_count.restype = C.c_int
    lib.sdrfm_process.argtypes = [vp, vp, u32, vp, u32, u32p]
    lib.sdrfm_process.restype = C.c_int
    lib.sdrfm_process_batch.argtypes = [vp, vp, C.c_size_t, u32, vp, C.c_size_t, u32p, u32]
    lib.sdrfm_process_batch.restype = C.c_int
    lib.sdrfm_set_stream.argtypes = [vp, vp]
    lib.sdrfm_set_stream.restype = C.c_int
    lib.sdrfm_synchronize.argtypes = [vp]
    lib.sdrfm_synchronize.restype = C.c_int
    lib.sdrfm_flush.argtypes = [vp]
    lib.sdrfm_flush.restype = C.c_int
    lib.sdrfm_flush_previous.argtypes = [vp]
    lib.sdrfm_flush_previous.restype = C.c_int
    lib.sdrfm_wait_previous.argtypes = [vp, vp]
    lib.sdrfm_wait_previous.restype = C.c_int
    lib.sdrfm_process_batch_pcm.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_uint32), C.c_uint32]
    lib.sdrfm_process_batch_pcm.restype = C.c_int
    lib.sdrfm_kernel_name.argtypes = [vp]
    lib.sdrfm_kernel_name.restype = C.c_char_p
    lib.sdrfm_abi_version.argtypes = []
    lib.sdrfm_abi_version.restype = u32
    lib.sdrfm_strerror.argtypes = [C.c_int]
    lib.sdrfm_strerror.restype = C.c_char_p
    lib.sdrfm_host_atan2f.argtypes = [C.c_float, C.c_float]
    lib.sdrfm_host_atan2f.restype = C.c_float
    lib.sdrfm_host_discriminate.argtypes = [C.c_float] * 4
    lib.sdrfm_host_discriminate.restype = C.c_float
    if dev:
        lib.sdrfm_debug_phase_cycles.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.sdrfm_debug_phase_cycles.restype = C.c_int
        lib.sdrfm_debug_raw.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.sdrfm_debug_raw.restype = C.c_int
    lib.sdrfm_q_build.argtypes = [vp, u32, u32, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), u32p]
    lib.sdrfm_q_build.restype = C.c_int
    lib.sdrfm_debug_discriminate.argtypes = [C.c_int] + [vp] * 6 + [u32]
    lib.sdrfm_debug_discriminate.restype = C.c_int
    lib.sdrfm_q_guard.argtypes = [vp, u32, vp, u32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.sdrfm_q_guard.restype = C.c_int
    lib.sdrfm_q_guard2.argtypes = [vp, u32, vp, u32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.sdrfm_q_guard2.restype = C.c_int
    lib.sdrfm_debug_read_ceiling.argtypes = [C.c_int, C.POINTER(vp), u32, C.c_size_t, u32, C.POINTER(C.c_double)]
    lib.sdrfm_debug_read_ceiling.restype = C.c_int
    lib.sdrfm_debug_q_guard.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.sdrfm_debug_q_guard.restype = C.c_int
    lib.sdrfm_debug_route.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
    lib.sdrfm_debug_route.restype = C.c_int
    lib.sdrfm_wbfm_create.argtypes = [C.POINTER(WbfmConfig), C.POINTER(vp)]
    lib.sdrfm_wbfm_create.restype = C.c_int
    lib.sdrfm_wbfm_destroy.argtypes = [vp]
    lib.sdrfm_wbfm_destroy.restype = None
    lib.sdrfm_wbfm_reset.argtypes = [vp]
    lib.sdrfm_wbfm_reset.restype = C.c_int
    lib.sdrfm_wbfm_audio_count.argtypes = [vp, u32, u32p]
    lib.sdrfm_wbfm_audio_count.restype = C.c_int
    lib.sdrfm_wbfm_process_batch.argtypes = [vp, vp, C.c_size_t, u32, vp, C.c_size_t, u32p, u32]
    lib.sdrfm_wbfm_process_batch.restype = C.c_int
    lib.sdrfm_wbfm_set_stream.argtypes = [vp, vp]
    lib.sdrfm_wbfm_set_stream.restype = C.c_int
    lib.sdrfm_wbfm_synchronize.argtypes = [vp]
    lib.sdrfm_wbfm_synchronize.restype = C.c_int
    lib.sdrfm_wbfm_kernel_name.argtypes = [vp]
    lib.sdrfm_wbfm_kernel_name.restype = C.c_char_p
    lib.sdrfm_spectrum_create.argtypes = [C.POINTER(SpectrumConfig), C.POINTER(vp)]
    lib.sdrfm_spectrum_create.restype = C.c_int
    lib.sdrfm_spectrum_destroy.argtypes = [vp]
    lib.sdrfm_spectrum_destroy.restype = None
    lib.sdrfm_spectrum_process_batch.argtypes = [vp, vp, C.c_size_t, u32, vp, C.c_size_t, u32p, u32]
    lib.sdrfm_spectrum_process_batch.restype = C.c_int
    lib.sdrfm_spectrum_set_stream.argtypes = [vp, vp]
    lib.sdrfm_spectrum_set_stream.restype = C.c_int
    lib.sdrfm_spectrum_synchronize.argtypes = [vp]
    lib.sdrfm_spectrum_synchronize.restype = C.c_int
    lib.sdrfm_spectrum_kernel_name.argtypes = [vp]
    lib.sdrfm_spectrum_kernel_name.restype = C.c_char_p
    lib.sdrfm_rtl_pack_fir.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_uint8)]
    lib.sdrfm_rtl_pack_fir.restype = C.c_int
    lib.sdrfm_rtl_resampler.argtypes = [u32, u32, u32p, u32p, C.POINTER(C.c_double)]
    lib.sdrfm_rtl_resampler.restype = C.c_int
    lib.sdrfm_e4k_pll_params.argtypes = [u32, u32, vp]
    lib.sdrfm_e4k_pll_params.restype = C.c_int
    lib.sdrfm_shard_range.argtypes = [u32, u32, u32, u32p, u32p]
    lib.sdrfm_shard_range.restype = C.c_int
    lib.sdrfm_pcm_deemph_s16.argtypes = [vp, u32, C.c_float, C.c_float, C.POINTER(C.c_float), vp]
    lib.sdrfm_pcm_deemph_s16.restype = C.c_int
    lib.sdrfm_pcm_alpha.argtypes = [C.c_float, C.c_float]
    lib.sdrfm_pcm_alpha.restype = C.c_float
    lib.sdrfm_ring_create.argtypes = [vp, u32, u32, C.POINTER(vp)]
    lib.sdrfm_ring_create.restype = C.c_int
    lib.sdrfm_ring_destroy.argtypes = [vp]
    lib.sdrfm_ring_destroy.restype = None
    lib.sdrfm_ring_submit.argtypes = [vp, vp, u32]
    lib.sdrfm_ring_submit.restype = C.c_int
    lib.sdrfm_ring_collect.argtypes = [vp, vp, u32, u32p, C.c_int]
    lib.sdrfm_ring_collect.restype = C.c_int
    lib.sdrfm_pcm_sink_create.argtypes = [u32, C.c_float, C.c_float, C.c_int32, C.POINTER(vp)]
    lib.sdrfm_pcm_sink_create.restype = C.c_int
    lib.sdrfm_pcm_sink_destroy.argtypes = [vp]
    lib.sdrfm_pcm_sink_destroy.restype = None
    lib.sdrfm_pcm_sink_reset.argtypes = [vp]
    lib.sdrfm_pcm_sink_reset.restype = C.c_int
    lib.sdrfm_pcm_sink_process_batch.argtypes = [vp, vp, C.c_size_t, u32, vp, C.c_size_t, u32]
    lib.sdrfm_pcm_sink_process_batch.restype = C.c_int
    lib.sdrfm_pcm_sink_set_stream.argtypes = [vp, vp]
    lib.sdrfm_pcm_sink_set_stream.restype = C.c_int
    lib.sdrfm_pcm_sink_synchronize.argtypes = [vp]
    lib.sdrfm_pcm_sink_synchronize.restype = C.c_int
    lib.sdrfm_pcm_sink_get_state.argtypes = [vp, C.POINTER(C.c_float)]
    lib.sdrfm_pcm_sink_get_state.restype = C.c_int
    _libs[dev] = lib
    return lib
