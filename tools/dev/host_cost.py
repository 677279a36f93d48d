#!/usr/bin/env python3
"""Host cost of one device-pointer call: a workload small enough that the GPU finishes a call faster than the host can issue one, so
wall time per call = the host's own cost (Python wrapper + C shim + HIP launch).  Serial calls vs SDRFM_F_OVERLAP calls."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

pkg = importlib.import_module("stm32f7-rtlsdr_amd")
ns, nsamp = 256, 4000
h, g = pkg.default_config(64)
iqs = [torch.from_numpy(pkg.make_iq(ns, nsamp, first_id=b)).cuda() for b in range(3)]
audio = [torch.zeros((ns, 80), dtype=torch.float32, device="cuda") for _ in range(2)]
st = torch.cuda.Stream()
for own in (False, True):
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm:
        if not own:
            dm.set_stream(st.cuda_stream)
        for ovl in (False, True):
            for rep in range(2):
                for i in range(200):
                    dm.process_batch_device(iqs[i % 3], audio[i & 1], overlap=ovl)
                dm.synchronize()
                t0 = time.perf_counter()
                for i in range(2000):
                    dm.process_batch_device(iqs[i % 3], audio[i & 1], overlap=ovl)
                t1 = time.perf_counter()
                dm.synchronize()
                t2 = time.perf_counter()
            print("own_stream=%d overlap=%d  issue %.2f us/call, with drain %.2f us/call  (%s)" % (own, ovl, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6, dm.kernel_name))
