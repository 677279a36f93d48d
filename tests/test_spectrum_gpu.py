"""GPU parity of the spectrum view (sdrfm_spectrum_*) against its oracle, through the C-ABI.  Both sides evaluate the same
radix-2 graph with the same fp32 operations, so the comparison is bit for bit."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("nfft", [64, 128, 256, 512, 1024, 2048, 4096])
def test_every_size_bitwise_equals_oracle(pkg, oracle_mod, nfft):
    iq = pkg.make_iq(1, 6 * nfft + 11, mode="fm", first_id=40 + nfft)[0]
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft))
    got, frames = sv.process_batch(iq)
    want, wf = oracle_mod.SpectrumOracle(nfft).process(iq)
    assert frames == wf == 6
    assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32))
    sv.close()


@pytest.mark.parametrize("mode", ["fm", "random", "const", "counter"])
def test_input_classes_and_batch(pkg, oracle_mod, mode):
    ns, nfft = 5, 1024
    iq = pkg.make_iq(ns, 20 * nfft + 500, mode=mode, first_id=60)
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns))
    got, frames = sv.process_batch(iq)
    assert frames == 20 and got.shape == (ns, nfft)
    for s in range(ns):
        want, _ = oracle_mod.SpectrumOracle(nfft).process(iq[s])
        assert np.array_equal(got[s].view(np.uint32), want.view(np.uint32)), (mode, s)
    sv.close()


def test_golden_vectors_custom_window_and_device_buffers(pkg):
    import torch
    for fn in sorted(glob.glob(os.path.join(GOLD, "spectrum_*.npz"))):
        z = np.load(fn)
        nfft = int(z["nfft"])
        win = z["window"] if z["window"].size else None
        sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, window=win))
        got, frames = sv.process_batch(z["iq"])
        assert frames == int(z["frames"])
        assert np.array_equal(got[0].view(np.uint32), z["power"].view(np.uint32)), fn
        iq = torch.from_numpy(z["iq"][None, :].copy()).cuda()
        power = torch.full((1, nfft + 8), -1.0, dtype=torch.float32, device="cuda")     # padded rows: stride > nfft
        torch.cuda.synchronize()    # allocations / fills ran on torch's stream; the library uses its own
        assert sv.process_batch_device(iq, power) == frames
        sv.synchronize()
        assert np.array_equal(power[0, :nfft].cpu().numpy().view(np.uint32), z["power"].view(np.uint32))
        assert torch.all(power[0, nfft:] == -1.0)
        sv.close()


def test_edges_and_errors(pkg):
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=256, n_streams=2, max_bytes_per_call=4096))
    got, frames = sv.process_batch(np.full((2, 510), 200, np.uint8))     # shorter than one frame -> zeros
    assert frames == 0 and np.all(got == 0)
    with pytest.raises(pkg.SdrfmError) as e:
        sv.process_batch(np.zeros((2, 7), np.uint8))
    assert e.value.status == 17
    with pytest.raises(pkg.SdrfmError) as e:
        sv.process_batch(np.zeros((2, 5000), np.uint8))
    assert e.value.status == 18
    sv.close()
    for bad in (1000, 32, 8192):
        with pytest.raises(pkg.SdrfmError) as e:
            pkg.SpectrumView(pkg.SpectrumConfig(nfft=bad))
        assert e.value.status == 16
