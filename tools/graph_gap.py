import importlib, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
h, g = pkg.default_config(64)
ns, nsamp = 256, 240000
dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2*nsamp))
s = torch.cuda.Stream(); dm.set_stream(s.cuda_stream)
with torch.cuda.stream(s):
    iq = torch.from_numpy(np.tile(pkg.make_iq(8, nsamp), (32, 1))).cuda()
    audio = torch.zeros((ns, 4808), dtype=torch.float32, device="cuda")
    for _ in range(10): dm.process_batch_device(iq, audio)
s.synchronize()
def run(n):
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(n): dm.process_batch_device(iq, audio)
    s.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("plain us/step", run(200), run(200))
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    for _ in range(20): dm.process_batch_device(iq, audio)
def rung(n):
    t0 = time.perf_counter()
    for _ in range(n): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * 20) * 1e6
print("graph us/step", rung(10), rung(10))
