"""MI355X-native IQ -> FM-audio path ("sdrfm") — Python plumbing over the C-ABI in include/sdrfm.h.

The product is csrc/libsdrfm.so (hand-written HIP for gfx950 + C host shim).  This package only loads it with ctypes,
mirrors its entry points one-to-one, and adds torch plumbing (device buffers, streams, torch.distributed fan-out).
There is no CPU fallback anywhere in this package: without the built library or without a gfx950 device every compute
call raises.

The directory name contains a hyphen (it is fixed by the build contract), so import it with
``importlib.import_module("stm32f7-rtlsdr_amd")``.
"""
import os as _os

# Overlapped calls (SDRFM_F_OVERLAP) need two hardware queues of their own beside the caller's stream; the HIP runtime maps a process's streams onto a pool of 4
# by default and RCCL's streams take some of them.  Effective only when this package is imported before anything loads the runtime (torch);
# a host that imports torch first sets the variable itself — bench.py does.
if "WORLD_SIZE" in _os.environ or "TORCHELASTIC_RUN_ID" in _os.environ:   # (a rank of a torch.distributed run: RCCL will be in the process)
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
