"""MI355X-native IQ -> FM-audio path ("sdrfm") — Python plumbing over the C-ABI in include/sdrfm.h.

The product is csrc/libsdrfm.so (hand-written HIP for gfx950 + C host shim).  This package only loads it with ctypes,
mirrors its entry points one-to-one, and adds torch plumbing (device buffers, streams, torch.distributed fan-out).
There is no CPU fallback anywhere in this package: without the built library or without a gfx950 device every compute
call raises.

The directory name contains a hyphen (it is fixed by the build contract), so import it with
``importlib.import_module("stm32f7-rtlsdr_amd")``.
"""
# (Importing this package does not touch the process environment.  A host that runs RCCL beside overlapped calls wants GPU_MAX_HW_QUEUES=8 set BEFORE the HIP
# runtime loads — INTEGRATION.md section 3 says why; bench.py and examples/multi_gpu_main.c set it themselves.)
from .lib import SdrfmError, load_library, library_path, STATUS, ABI_SYMBOLS
from .demod import FmDemod, FmConfig
from .wbfm import WbfmDemod, WbfmConfig
from .spectrum import SpectrumView, SpectrumConfig, power_db
from .sink import PcmSink, pcm_deemph_s16_host
from .taps import RTLSDR_FIR, rtlsdr_fir16, lowpass_taps, default_config
from .siggen import make_iq, MODES
from .frontend import ReplayFrontEnd, XferState
from . import fanout
from . import lib

__all__ = [
    "SdrfmError", "load_library", "library_path", "STATUS", "ABI_SYMBOLS", "FmDemod", "FmConfig", "WbfmDemod", "WbfmConfig", "SpectrumView", "SpectrumConfig", "power_db", "PcmSink", "pcm_deemph_s16_host", "RTLSDR_FIR",
    "rtlsdr_fir16", "lowpass_taps", "default_config", "make_iq", "MODES", "ReplayFrontEnd", "XferState", "fanout",
]
