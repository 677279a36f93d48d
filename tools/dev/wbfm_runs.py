import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
ns, nsamp = int(os.environ.get("NS", "128")), 320000
p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8); g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
base = torch.from_numpy(pkg.make_iq(64, nsamp, mode="fm", fs=3.2e6)).cuda()
batches = [torch.cat([torch.roll(base, shifts=2 * (7919 * (b * 2 + r) % nsamp), dims=1) for r in range((ns + 63) // 64)])[:ns].contiguous() for b in range(5)]
for rs in [int(x) for x in os.environ.get("RUNS", "0,2500,1250,834,626,418,314,210,0,1250,626,314").split(",")]:
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, run_steps=rs))
    audio = torch.zeros((ns, 16, dm.audio_count(2 * nsamp) + 8), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream(); dm.set_stream(st.cuda_stream); torch.cuda.synchronize()
    for i in range(10): dm.process_batch_device(batches[i % 5], audio)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(st)
    for i in range(50): dm.process_batch_device(batches[i % 5], audio)
    e1.record(st); torch.cuda.synchronize()
    print("run_steps", rs, "us/launch %.1f" % (e0.elapsed_time(e1) * 1e3 / 50))
    dm.set_stream(None); dm.close()
