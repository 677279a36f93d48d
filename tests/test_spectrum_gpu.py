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


@pytest.mark.parametrize("nfft", [64, 128, 256, 512, 1024])
def test_frame_counts_around_the_runs_of_the_chained_kernel(pkg, oracle_mod, nfft):
    """N <= 1024 run k_spectrum_chain: 12 waves per stream take runs of two 1024-point blocks and hand the running sum on.  Frame counts
    below, at and above one block, one run and one cycle of the twelve waves (24 blocks), counts that leave a last block partly filled
    (N < 1024: several frames per block), three streams with different data: bit for bit against the oracle."""
    fpb = 1024 // nfft                                           # frames per block
    counts = sorted(set([1, 2, 3, 4, 5, fpb - 1, fpb, fpb + 1, 2 * fpb + 1, 23 * fpb, 24 * fpb - 1, 24 * fpb, 24 * fpb + 1, 25 * fpb, 47 * fpb + 1, 48 * fpb, 49 * fpb, 50 * fpb + fpb // 2, 100]) - {0})
    ns = 3
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns, max_bytes_per_call=2 * nfft * (max(counts) + 1)))
    for F in counts:
        iq = pkg.make_iq(ns, F * nfft + 7, mode="fm" if F % 2 else "random", first_id=300 + F)
        got, frames = sv.process_batch(iq)
        assert frames == F
        for s in range(ns):
            want, _ = oracle_mod.SpectrumOracle(nfft).process(iq[s])
            assert np.array_equal(got[s].view(np.uint32), want.view(np.uint32)), (nfft, F, s)
    sv.close()


@pytest.mark.parametrize("nfft", [256, 512, 1024])
@pytest.mark.parametrize("offset,stride_extra", [(1, 0), (0, 1), (1, 3), (2, 2), (0, 0), (2, 0), (0, 2)])
def test_device_buffers_at_odd_addresses_and_strides(pkg, oracle_mod, nfft, offset, stride_extra):
    """The raw-dword kernel wants iq and iq_stride even; anything else is served by the typed-load kernel.  Same results either way."""
    import torch
    ns, F = 4, 27
    nbytes = 2 * nfft * F
    iq = pkg.make_iq(ns, nfft * F, mode="fm", first_id=900 + nfft)
    stride = nbytes + 6 + stride_extra
    buf = torch.zeros(offset + ns * stride, dtype=torch.uint8, device="cuda")
    view = buf[offset:offset + ns * stride].view(ns, stride)[:, :nbytes]
    view.copy_(torch.from_numpy(iq))
    assert view.data_ptr() % 2 == offset % 2 and view.stride(0) == stride
    power = torch.zeros((ns, nfft), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns, max_bytes_per_call=nbytes))
    assert sv.process_batch_device(view, power) == F
    sv.synchronize()
    for s in range(ns):
        want, _ = oracle_mod.SpectrumOracle(nfft).process(iq[s])
        assert np.array_equal(power[s].cpu().numpy().view(np.uint32), want.view(np.uint32)), (nfft, offset, stride_extra, s)
    sv.close()


def test_bench_shape_streams_agree_with_the_oracle_and_with_each_other(pkg, oracle_mod):
    """256 streams x 234 frames x 1024 points (bench.py's spectrum workload): streams 0, 100, 255 against the oracle, and the launch repeated on
    the same buffers gives the same bits (no dependence on wave timing in the hand-over chain)."""
    import torch
    ns, nfft, F = 256, 1024, 234
    rows = pkg.make_iq(8, nfft * F, mode="fm", first_id=700)
    iq = np.tile(rows, (ns // 8, 1))
    d_iq = torch.from_numpy(iq).cuda()
    power = [torch.zeros((ns, nfft), dtype=torch.float32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns, max_bytes_per_call=2 * nfft * F))
    for pw in power:
        assert sv.process_batch_device(d_iq, pw) == F
    sv.synchronize()
    got = power[0].cpu().numpy()
    for s in (0, 100, 255):
        want, _ = oracle_mod.SpectrumOracle(nfft).process(iq[s])
        assert np.array_equal(got[s].view(np.uint32), want.view(np.uint32)), s
    for rep in range(1, ns // 8):
        assert np.array_equal(got[8 * rep:8 * rep + 8].view(np.uint32), got[:8].view(np.uint32)), rep
    for pw in power[1:]:
        assert torch.equal(pw, power[0])
    sv.close()


def test_kernel_names(pkg):
    """sdrfm_spectrum_kernel_name: up to 1024 points run the chained kernel (the typed-load kernel when iq or iq_stride is odd: the test
    above), 2048 and 4096 points k_spectrum<log2 N>."""
    import torch
    nfft = 1024
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=2, max_bytes_per_call=2 * nfft * 5))
    assert sv.kernel_name.startswith("k_spectrum_chain<10")
    buf = torch.zeros(2 * (2 * nfft * 5 + 1), dtype=torch.uint8, device="cuda")
    power = torch.zeros((2, nfft), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    assert sv.process_batch_device(buf.view(2, -1)[:, :2 * nfft * 5], power) == 5     # odd stride
    assert sv.kernel_name == "k_spectrum<10>"
    assert sv.process_batch_device(buf[:2 * 2 * nfft * 5].view(2, -1), power) == 5
    assert sv.kernel_name.startswith("k_spectrum_chain<10")
    sv.synchronize()
    sv.close()
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=256))
    assert sv.kernel_name.startswith("k_spectrum_chain<8")
    sv.close()
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=2048))
    assert sv.kernel_name == "k_spectrum<11>"
    sv.close()
