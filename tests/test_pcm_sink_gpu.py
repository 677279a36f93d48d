"""Device-side PCM sink (csrc/sdrfm_sink.hip) against the host routine sdrfm_pcm_deemph_s16, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(pkg):
    lib = pkg.load_library()
    return lib.sdrfm_pcm_alpha(48000.0, 75e-6), np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3))


@pytest.mark.parametrize("ns,n", [(1, 4800), (3, 1), (64, 63), (65, 130), (256, 4800)])
def test_device_sink_equals_host_routine_bitwise(pkg, ns, n):
    alpha, gain = _params(pkg)
    rng = np.random.default_rng(ns * 1000 + n)
    x = (rng.standard_normal((ns, 2 * n)) * 1.5).astype(np.float32)
    x[0, : min(6, 2 * n)] = [9.0, -9.0, 0.0, 1e-30, 0.5, -0.5][: min(6, 2 * n)]
    with pkg.PcmSink(ns, alpha, gain) as sink:
        a = sink.process_batch(x[:, :n])                         # two calls: the state is carried per stream
        b = sink.process_batch(x[:, n:])
        st_dev = sink.state()
    got = np.concatenate([a, b], axis=1)
    for s in range(ns):
        want, st = pkg.pcm_deemph_s16_host(x[s], alpha, gain)
        assert np.array_equal(got[s], want), (s, int(np.argmax(got[s] != want)))
        assert np.float32(st).view(np.uint32) == st_dev[s].view(np.uint32)


def test_device_sink_after_the_demodulator_on_one_stream(pkg):
    """The batched path end to end on the device: IQ -> sdrfm_process_batch -> sdrfm_pcm_sink_process_batch, both enqueued on
    the same HIP stream, PCM equal to host-sinking the same audio."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp = 16, 240000
    h, g = pkg.default_config(64)
    iq = torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", first_id=500)).cuda()
    audio = torch.zeros((ns, 4800), dtype=torch.float32, device="cuda")
    pcm = torch.zeros((ns, 9600), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm, pkg.PcmSink(ns, alpha, gain) as sink:
        dm.set_stream(stream.cuda_stream)
        sink.set_stream(stream.cuda_stream)
        for _ in range(2):                                        # second pass: carried state in both stages
            n = dm.process_batch_device(iq, audio)
            sink.process_batch_device(audio, pcm, n)
        stream.synchronize()
        a = audio.cpu().numpy()
        got = pcm.cpu().numpy()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm2:
        a1 = dm2.process_batch(iq.cpu().numpy())
    for s in range(ns):
        _, st = pkg.pcm_deemph_s16_host(a1[s], alpha, gain)
        want, _ = pkg.pcm_deemph_s16_host(a[s], alpha, gain, st)
        assert np.array_equal(got[s], want), s


def test_sink_argument_errors(pkg):
    alpha, gain = _params(pkg)
    with pytest.raises(pkg.SdrfmError) as e:
        pkg.PcmSink(4, 0.0, gain)
    assert e.value.status == 16
    with pkg.PcmSink(2, alpha, gain) as sink:
        assert sink.process_batch(np.zeros((2, 0), np.float32)).shape == (2, 0)
