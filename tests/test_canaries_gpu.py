"""Out-of-bounds write detection without a GPU sanitizer (not available on the pool): device outputs are embedded in larger
allocations filled with a sentinel; after a call every byte outside the documented output region must still be the sentinel,
and the kernels' own state must not have been disturbed by a neighbouring handle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SENT = -12345.0


def test_fm_batch_writes_only_its_audio(pkg, oracle_mod):
    import torch
    h, g = pkg.default_config(64)
    ns, nsamp = 6, 50006                                             # even, not a multiple of the sub-tile
    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp))
    iq_host = pkg.make_iq(ns, nsamp, mode="random", first_id=5)
    iq = torch.from_numpy(iq_host).cuda()
    n_expected = dm.audio_count(2 * nsamp)
    pad, stride = 64, n_expected + 37
    big = torch.full((pad + ns * stride + pad,), SENT, dtype=torch.float32, device="cuda")
    audio = big[pad:pad + ns * stride].view(ns, stride)
    torch.cuda.synchronize()                                         # fills ran on torch's stream, the library uses its own
    for rep in range(3):                                             # first call (fix-up launch) and steady state
        n = dm.process_batch_device(iq, audio)
        dm.synchronize()
        assert n == (n_expected if rep == 0 else n)
        host = big.cpu().numpy()
        assert np.all(host[:pad] == SENT) and np.all(host[-pad:] == SENT)
        rows = host[pad:pad + ns * stride].reshape(ns, stride)
        assert np.all(rows[:, n:] == SENT), "wrote past the audio of a stream"
        assert not np.any(rows[:, :n] == SENT)
    dm.close()


def test_wbfm_batch_writes_only_its_audio(pkg):
    import torch
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
    g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
    ns, nsamp = 5, 64010                                             # a partial quad of streams, ragged length
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp))
    iq = torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", fs=3.2e6, first_id=9)).cuda()
    n_expected = dm.audio_count(2 * nsamp)
    pad, stride = 64, n_expected + 21
    big = torch.full((pad + ns * 16 * stride + pad,), SENT, dtype=torch.float32, device="cuda")
    audio = big[pad:pad + ns * 16 * stride].view(ns, 16, stride)
    torch.cuda.synchronize()
    for _ in range(2):
        n = dm.process_batch_device(iq, audio)
        dm.synchronize()
        host = big.cpu().numpy()
        assert np.all(host[:pad] == SENT) and np.all(host[-pad:] == SENT)
        rows = host[pad:pad + ns * 16 * stride].reshape(ns * 16, stride)
        assert np.all(rows[:, n:] == SENT), "wrote past the audio of a band"
        assert not np.any(rows[:, :n] == SENT)
    dm.close()


def test_neighbouring_handles_do_not_disturb_each_other(pkg, oracle_mod):
    """Two handles interleaved on the same device keep separate streaming state (history, phases)."""
    h, g = pkg.default_config(16)
    a = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g))
    b = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g))
    ia = pkg.make_iq(1, 30000, mode="fm", first_id=1)[0]
    ib = pkg.make_iq(1, 30000, mode="random", first_id=2)[0]
    outa, outb = [], []
    for k in range(0, 60000, 6000):
        outa.append(a.process(ia[k:k + 6000]))
        outb.append(b.process(ib[k:k + 6000]))
    wa, wb = oracle_mod.Oracle(h, g).process(ia), oracle_mod.Oracle(h, g).process(ib)
    assert np.max(np.abs(np.concatenate(outa) - wa) / np.maximum(np.abs(wa), 1)) <= 1e-5
    assert np.max(np.abs(np.concatenate(outb) - wb) / np.maximum(np.abs(wb), 1)) <= 1e-5
    a.close(); b.close()
