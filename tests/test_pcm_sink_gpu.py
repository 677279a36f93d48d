"""Device-side PCM sink (csrc/sdrfm_sink.hip): its exact form (SDRFM_PCM_F_EXACT) against the host routine sdrfm_pcm_deemph_s16 and an exact-rational
restatement, bit for bit; its default form (the blocked scan) held to 1 LSB of int16 against the exact one."""
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
    with pkg.PcmSink(ns, alpha, gain, exact=True) as sink:
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
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm, pkg.PcmSink(ns, alpha, gain, exact=True) as sink:
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


def test_device_sink_against_the_exact_rational_definition(pkg):
    """The device kernel against an INDEPENDENT restatement (exact rationals, correctly rounded to fp32 at every operation:
    tests/test_pcm_sink.py), not against the product's own host routine: 96 samples per stream, saturation, ties and tiny values
    included, state carried across two calls."""
    from test_pcm_sink import pcm_reference
    alpha, gain = _params(pkg)
    rng = np.random.default_rng(7)
    ns, n = 3, 48
    x = (rng.standard_normal((ns, 2 * n)) * 1.5).astype(np.float32)
    x[0, :8] = [10.0, 10.0, -10.0, -10.0, 0.0, 1e-30, -1e-30, 0.5 / 3.0]
    x[1, :4] = [np.float32(0.5) / gain, np.float32(1.5) / gain, np.float32(-2.5) / gain, 0.0]
    with pkg.PcmSink(ns, alpha, gain, exact=True) as sink:
        got = np.concatenate([sink.process_batch(x[:, :n]), sink.process_batch(x[:, n:])], axis=1)
        st_dev = sink.state()
    for s in range(ns):
        want, y_end = pcm_reference(x[s], alpha, gain, 0.0)
        assert np.array_equal(got[s], want), (s, int(np.argmax(got[s] != want)))
        assert np.float32(y_end).view(np.uint32) == st_dev[s].view(np.uint32)


def test_device_sink_on_its_own_stream_behind_an_event(pkg):
    """The arrangement DESIGN.md 4.7 describes: the sink runs on ITS OWN stream, ordered behind the demodulator's launch by an
    event, while the demodulator's stream goes on with the next batch (double-buffered audio).  PCM equal to sinking the same
    audio on the host, for every batch."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp, nbatch = 256, 24000, 4
    h, g = pkg.default_config(64)
    iqs = [torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", first_id=800 + 7 * b)).cuda() for b in range(nbatch)]
    audio = [torch.zeros((ns, 480), dtype=torch.float32, device="cuda") for _ in range(2)]
    pcm = [torch.zeros((ns, 960), dtype=torch.int16, device="cuda") for _ in range(nbatch)]
    torch.cuda.synchronize()
    s_dm, s_sink = torch.cuda.Stream(), torch.cuda.Stream()
    done_sink = [None, None]
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm, pkg.PcmSink(ns, alpha, gain, exact=True) as sink:
        dm.set_stream(s_dm.cuda_stream)
        sink.set_stream(s_sink.cuda_stream)
        for b in range(nbatch):
            buf = b & 1
            if done_sink[buf] is not None:
                s_dm.wait_event(done_sink[buf])                  # the audio buffer is free again once its previous sink pass ended
            n = dm.process_batch_device(iqs[b], audio[buf])
            ev = torch.cuda.Event()
            ev.record(s_dm)
            s_sink.wait_event(ev)                                # the sink starts when this batch's audio exists ...
            sink.process_batch_device(audio[buf], pcm[b], n)
            done_sink[buf] = torch.cuda.Event()
            done_sink[buf].record(s_sink)                        # ... and the demodulator does not wait for it
        s_dm.synchronize(); s_sink.synchronize()
        got = [p.cpu().numpy() for p in pcm]
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm2:
        ref_audio = [dm2.process_batch(iq.cpu().numpy()) for iq in iqs]
    for s in (0, 1, 77, 255):
        st = 0.0
        for b in range(nbatch):
            want, st = pkg.pcm_deemph_s16_host(ref_audio[b][s], alpha, gain, st)
            assert np.array_equal(got[b][s], want), (s, b)


@pytest.mark.parametrize("ns,n", [(1, 4800), (3, 1), (5, 255), (64, 257), (65, 1300), (256, 4800), (2, 30000)])
def test_blocked_scan_sink_within_one_lsb_of_the_exact_form(pkg, ns, n):
    """the default device sink is a blocked first-order scan (256 lanes per stream, the carries between chunks by a scan, every chunk
    re-walked from its true carry-in): PCM within 1 LSB of int16 of the exact one-lane chain — and different from it at all in fewer than one sample in a
    thousand —, the carried state within 2.5e-7 (audio of unit scale), across two calls; lengths below, at and above the 256 chunks and above one LDS segment (4864)."""
    alpha, gain = _params(pkg)
    rng = np.random.default_rng(ns * 77 + n)
    x = (rng.standard_normal((ns, 2 * n)) * 1.5).astype(np.float32)
    x[0, : min(6, 2 * n)] = [9.0, -9.0, 0.0, 1e-30, 0.5, -0.5][: min(6, 2 * n)]
    outs = {}
    for tag, exact in (("scan", False), ("exact", True)):
        with pkg.PcmSink(ns, alpha, gain, exact=exact) as sink:
            outs[tag] = (np.concatenate([sink.process_batch(x[:, :n]), sink.process_batch(x[:, n:])], axis=1), sink.state())
    (a, sa), (b, sb) = outs["scan"], outs["exact"]
    assert np.array_equal(a[:, 0::2], a[:, 1::2])                                  # L = R
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1, (int(d.max()), np.argwhere(d > 1)[:4])
    assert (d != 0).mean() <= 1e-3 + 2.0 / d.size, float((d != 0).mean())
    assert np.all(np.abs(sa - sb) <= 1e-6 * np.maximum(np.abs(sb), 0.25)), (sa, sb)      # (a few ulps of the audio's own scale)


def test_blocked_scan_sink_behind_the_demodulator(pkg):
    """The consumer loop's arrangement with the default sink: demodulator and sink on one stream, every batch's PCM within 1 LSB of host-sinking the same audio."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp = 256, 240000
    h, g = pkg.default_config(64)
    iq = torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", first_id=900)).cuda()
    audio = torch.zeros((ns, 4801), dtype=torch.float32, device="cuda")            # (a row stride that is no multiple of 4 floats: bench.py's)
    pcm = torch.zeros((2, ns, 9600), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    auds = []
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm, pkg.PcmSink(ns, alpha, gain) as sink:
        dm.set_stream(stream.cuda_stream)
        sink.set_stream(stream.cuda_stream)
        for k in range(2):
            n = dm.process_batch_device(iq, audio)
            sink.process_batch_device(audio, pcm[k], n)
            stream.synchronize()
            auds.append(audio[:, :n].cpu().numpy())
        got = pcm.cpu().numpy()
    for s in (0, 1, 100, 255):
        st = 0.0
        for k in range(2):
            want, st = pkg.pcm_deemph_s16_host(auds[k][s], alpha, gain, st)
            assert np.abs(got[k, s].astype(np.int32) - want.astype(np.int32)).max() <= 1, (s, k)


def test_consumer_on_its_own_stream_behind_wait_previous(pkg):
    """The fast consumer loop of INTEGRATION.md: overlapped calls, the device PCM sink on a stream OF ITS OWN ordered behind call k - 1 by sdrfm_wait_previous
    (the handle's stream is left alone, so call k + 1 is held back by nothing), three audio buffers in turn, the reuse of a buffer guarded on the host.
    PCM within 1 LSB of host-sinking the serial calls' audio, for every batch."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp, nb = 256, 48000, 9
    h, g = pkg.default_config(64)
    iq = torch.from_numpy(pkg.make_iq(ns, nb * nsamp, mode="fm", first_id=2500)).cuda()
    audio = [torch.zeros((ns, nsamp // 50), dtype=torch.float32, device="cuda") for _ in range(3)]
    pcm = [torch.zeros((ns, 2 * (nsamp // 50)), dtype=torch.int16, device="cuda") for _ in range(nb)]
    torch.cuda.synchronize()
    s_dm, s_sink = torch.cuda.Stream(), torch.cuda.Stream()
    consumed = [None, None, None]
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm, pkg.PcmSink(ns, alpha, gain) as sink:
        dm.set_stream(s_dm.cuda_stream)
        sink.set_stream(s_sink.cuda_stream)
        n = 0
        for k in range(nb):
            if consumed[k % 3] is not None:
                consumed[k % 3].synchronize()                      # (the host guards the reuse of audio[k % 3]: no wait enters the demodulator's queues)
            n = dm.process_batch_device(iq[:, 2 * k * nsamp:], audio[k % 3], nbytes=2 * nsamp, overlap=True)
            if k:
                assert "overlapped" in dm.kernel_name, dm.kernel_name
                dm.wait_previous(s_sink.cuda_stream)
                sink.process_batch_device(audio[(k - 1) % 3], pcm[k - 1], n)
                consumed[(k - 1) % 3] = torch.cuda.Event()
                consumed[(k - 1) % 3].record(s_sink)
        dm.flush()
        s_sink.wait_stream(s_dm)
        sink.process_batch_device(audio[(nb - 1) % 3], pcm[nb - 1], n)
        s_sink.synchronize(); s_dm.synchronize()
        got = [p.cpu().numpy() for p in pcm]
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as ref:
        host_iq = iq.cpu().numpy()
        ref_audio = [ref.process_batch(host_iq[:, 2 * k * nsamp:2 * (k + 1) * nsamp]) for k in range(nb)]
    for s in (0, 3, 130, 255):
        st = 0.0
        for k in range(nb):
            want, st = pkg.pcm_deemph_s16_host(ref_audio[k][s], alpha, gain, st)
            assert np.abs(got[k][s].astype(np.int32) - want.astype(np.int32)).max() <= 1, (s, k)


# ---- the sink's chain as the TAIL of the demodulator's launch (sdrfm_process_batch_pcm, csrc/sdrfm_sink_chain.h) ---------------------------------------------
def _host_chain(pkg, auds, alpha, gain, s):
    """PCM of stream s over the calls' audio, by the host routine carrying its state from call to call."""
    st, out = 0.0, []
    for a in auds:
        want, st = pkg.pcm_deemph_s16_host(a[s], alpha, gain, st)
        out.append(want)
    return out, st


@pytest.mark.parametrize("overlap", [False, True])
def test_pcm_chain_inside_the_demodulators_launch(pkg, overlap):
    """sdrfm_process_batch_pcm over nine consecutive pieces of a capture: the first call (the start of a stream: the generic kernel recomputes its first outputs
    behind design Q's launch) is followed by the sink's own kernel, every later one ends with the sink's chain in design Q's own launch ("+ pcm"); the PCM of
    every call within 1 LSB of host-sinking that call's audio with the state carried across all of them, the carried state within 1e-6."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp, nb = 256, 48000, 9
    h, g = pkg.default_config(64)
    iq = torch.from_numpy(pkg.make_iq(ns, nb * nsamp, mode="fm", first_id=3100)).cuda()
    na = nsamp // 50
    audio = [torch.zeros((ns, na), dtype=torch.float32, device="cuda") for _ in range(nb)]
    pcm = [torch.zeros((ns, 2 * na), dtype=torch.int16, device="cuda") for _ in range(nb)]
    torch.cuda.synchronize()
    names = []
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm, pkg.PcmSink(ns, alpha, gain) as sink:
        for k in range(nb):
            n = dm.process_batch_pcm_device(sink, iq[:, 2 * k * nsamp:], audio[k], pcm[k], nbytes=2 * nsamp, overlap=overlap)
            assert n == na
            names.append(dm.kernel_name)
        dm.synchronize()
        state = sink.state()
    assert "+ pcm" not in names[0], names[0]
    assert all("+ pcm" in x for x in names[1:]), names
    assert all(("overlapped" in x) == overlap for x in names[1:]), names
    auds = [a.cpu().numpy() for a in audio]
    got = [p.cpu().numpy() for p in pcm]
    for s in (0, 1, 77, 128, 255):
        want, st = _host_chain(pkg, auds, alpha, gain, s)
        for k in range(nb):
            d = np.abs(got[k][s].astype(np.int32) - want[k].astype(np.int32))
            assert d.max() <= 1, (s, k, int(d.max()))
            assert got[k][s][0::2].tobytes() == got[k][s][1::2].tobytes()          # L = R
        assert abs(state[s] - st) <= 1e-6 * max(abs(st), 0.25), (s, state[s], st)


def test_pcm_chain_rows_that_are_not_16_byte_aligned_and_odd_lengths(pkg):
    """Row layouts and lengths: PCM and audio rows 8-byte aligned or not (audio stride 961 floats: odd rows start on an odd word — the chain stores 8-byte pairs
    aligned in memory), a call of 968 outputs, and a long one (0.12 s: 5760 outputs, runs of 4.7 steps); nothing may be written past a row's samples."""
    import torch
    alpha, gain = _params(pkg)
    ns = 64
    h, g = pkg.default_config(64)
    for nsamp, astride, pstride in ((48000, 961, 2 * 961), (48400, 1008, 2100), (288000, 5760, 11520)):   # (design Q: whole numbers of 8 audio periods)
        na = nsamp // 50
        assert astride >= na and pstride >= 2 * na
        iq = torch.from_numpy(pkg.make_iq(ns, 3 * nsamp, mode="fm", first_id=3300)).cuda()
        audio = [torch.zeros((ns, astride), dtype=torch.float32, device="cuda") for _ in range(3)]
        pcm = [torch.full((ns, pstride), 12345, dtype=torch.int16, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm, pkg.PcmSink(ns, alpha, gain) as sink:
            for k in range(3):
                n = dm.process_batch_pcm_device(sink, iq[:, 2 * k * nsamp:], audio[k], pcm[k], nbytes=2 * nsamp, overlap=True)
                assert n == na
                assert ("+ pcm" in dm.kernel_name) == (k > 0), dm.kernel_name
            dm.synchronize()
        auds = [a[:, :na].cpu().numpy() for a in audio]
        got = [p.cpu().numpy() for p in pcm]
        for s in (0, 3, 63):
            want, _ = _host_chain(pkg, auds, alpha, gain, s)
            for k in range(3):
                d = np.abs(got[k][s][:2 * na].astype(np.int32) - want[k].astype(np.int32))
                assert d.max() <= 1, (nsamp, s, k, int(d.max()))
                assert (got[k][s][2 * na:] == 12345).all(), (nsamp, s, k)            # nothing written past the call's samples


def test_pcm_call_on_the_bit_exact_kernels_and_on_routed_streams(pkg):
    """A bit-exact handle's calls are followed by the sink's own kernel; a batch with routed noise-only streams goes out as one launch of both designs with the chain in
    design Q's workgroups and the sink's list kernel behind it for the routed streams; a sequence that switches between the styles carries the state through (the
    device-side order between calls: the tagged per-stream word)."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp, nb = 128, 48000, 6
    na = nsamp // 50
    h, g = pkg.default_config(64)
    iq_np = pkg.make_iq(ns, nb * nsamp, mode="fm", first_id=3500)
    noisy = np.arange(ns) % 8 == 3
    iq_np[noisy] = pkg.make_iq(int(noisy.sum()), nb * nsamp, mode="random", first_id=3600)
    iq = torch.from_numpy(iq_np).cuda()
    for bit_exact in (True, False):
        audio = [torch.zeros((ns, na), dtype=torch.float32, device="cuda") for _ in range(nb)]
        pcm = [torch.zeros((ns, 2 * na), dtype=torch.int16, device="cuda") for _ in range(nb)]
        torch.cuda.synchronize()
        names = []
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, bit_exact=bit_exact)) as dm, \
                pkg.PcmSink(ns, alpha, gain) as sink:
            for k in range(nb):
                if not bit_exact and k == 3:
                    dm.route(noisy.astype(np.uint8))                                # calls 3, 4: the noisy streams on design B inside design Q's launch
                if not bit_exact and k == 5:
                    dm.route(np.zeros(ns, np.uint8))
                n = dm.process_batch_pcm_device(sink, iq[:, 2 * k * nsamp:], audio[k], pcm[k], nbytes=2 * nsamp, overlap=True)
                names.append(dm.kernel_name)
            dm.synchronize()
        if bit_exact:
            assert not any("+ pcm" in x for x in names), names
        else:
            assert ["+ pcm" in x for x in names] == [False, True, True, True, True, True], names
            assert "in one launch" in names[3] and "in one launch" in names[4], names   # (the clean streams' chain in design Q's workgroups, the routed streams by the sink's list kernel)
        auds = [a.cpu().numpy() for a in audio]
        got = [p.cpu().numpy() for p in pcm]
        for s in (0, 3, 11, 127):
            want, _ = _host_chain(pkg, auds, alpha, gain, s)
            for k in range(nb):
                d = np.abs(got[k][s].astype(np.int32) - want[k].astype(np.int32))
                assert d.max() <= 1, (bit_exact, s, k, int(d.max()))


def test_pcm_call_argument_errors(pkg):
    import torch
    alpha, gain = _params(pkg)
    h, g = pkg.default_config(64)
    lib = pkg.load_library()
    iq = torch.zeros((4, 96000), dtype=torch.uint8, device="cuda")
    audio = torch.zeros((4, 960), dtype=torch.float32, device="cuda")
    pcm = torch.zeros((4, 1920), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    import ctypes as C
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=4)) as dm, pkg.PcmSink(3, alpha, gain) as other, pkg.PcmSink(4, alpha, gain) as sink:
        n = C.c_uint32()
        args = lambda k, pp, ps, fl: (dm._h, k._h, C.c_void_p(iq.data_ptr()), iq.stride(0), 96000, C.c_void_p(audio.data_ptr()), audio.stride(0),
                                      C.c_void_p(pp), ps, C.byref(n), fl)
        assert lib.sdrfm_process_batch_pcm(*args(other, pcm.data_ptr(), 1920, 1)) == pkg.lib.EINVAL     # a sink of another stream count
        assert lib.sdrfm_process_batch_pcm(*args(sink, pcm.data_ptr(), 1920, 2)) == pkg.lib.EINVAL      # SDRFM_F_OVERLAP asks for device buffers
        assert lib.sdrfm_process_batch_pcm(*args(sink, pcm.data_ptr() + 2, 1920, 1)) == pkg.lib.EINVAL  # rows are written as (L, R) words
        assert lib.sdrfm_process_batch_pcm(*args(sink, pcm.data_ptr(), 1919, 1)) == pkg.lib.EINVAL
        assert lib.sdrfm_process_batch_pcm(*args(sink, pcm.data_ptr(), 1918, 1)) == pkg.lib.ECAPACITY
        assert lib.sdrfm_process_batch_pcm(*args(sink, pcm.data_ptr(), 1920, 1)) == 0 and n.value == 960
        dm.synchronize()


def test_pcm_call_without_an_audio_buffer(pkg):
    """audio = NULL: the PCM is all the call leaves — the same PCM, bit for bit, as the calls that also store the audio (one PCM buffer per call; the first call
    goes through rows of the library's own, and so do the routed streams of a mixed call)."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp, nb = 256, 48000, 6
    na = nsamp // 50
    h, g = pkg.default_config(64)
    iq_np = pkg.make_iq(ns, nb * nsamp, mode="fm", first_id=3700)
    noisy = np.arange(ns) % 16 == 5
    iq_np[noisy] = pkg.make_iq(int(noisy.sum()), nb * nsamp, mode="random", first_id=3800)
    iq = torch.from_numpy(iq_np).cuda()
    res = []
    for with_audio in (True, False):
        audio = [torch.zeros((ns, na), dtype=torch.float32, device="cuda") for _ in range(nb)]
        pcm = [torch.zeros((ns, 2 * na), dtype=torch.int16, device="cuda") for _ in range(nb)]
        torch.cuda.synchronize()
        names = []
        with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm, pkg.PcmSink(ns, alpha, gain) as sink:
            for k in range(nb):
                if k == 4:
                    dm.route(noisy.astype(np.uint8))
                n = dm.process_batch_pcm_device(sink, iq[:, 2 * k * nsamp:], audio[k] if with_audio else None, pcm[k], nbytes=2 * nsamp, overlap=True)
                assert n == na
                names.append(dm.kernel_name)
            dm.synchronize()
        assert ["+ pcm" in x for x in names] == [False, True, True, True, True, True], names
        assert all("overlapped" in x for x in names[1:]), names
        res.append(torch.stack(pcm).cpu().numpy())
    assert res[0].tobytes() == res[1].tobytes()


def test_pcm_chain_in_runs_that_flush_twice(pkg):
    """Runs longer than the four audio stages design Q parks (0.133 s x 256 streams: 21 steps = 4.2 stages per run) store — and sink — their outputs in two
    goes: the second scan continues from the run's own state, the first 64 outputs are still finished at the run's end."""
    import torch
    alpha, gain = _params(pkg)
    ns, nsamp, nb = 256, 320000, 3
    na = nsamp // 50
    h, g = pkg.default_config(64)
    rows = pkg.make_iq(8, nb * nsamp, mode="fm", first_id=3900)
    iq = torch.from_numpy(rows).cuda().repeat(ns // 8, 1)                       # (stream s carries row s % 8)
    audio = [torch.zeros((ns, na), dtype=torch.float32, device="cuda") for _ in range(nb)]
    pcm = [torch.zeros((ns, 2 * na), dtype=torch.int16, device="cuda") for _ in range(nb)]
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm, pkg.PcmSink(ns, alpha, gain) as sink:
        for k in range(nb):
            assert dm.process_batch_pcm_device(sink, iq[:, 2 * k * nsamp:], audio[k], pcm[k], nbytes=2 * nsamp, overlap=True) == na
            assert ("+ pcm" in dm.kernel_name) == (k > 0), dm.kernel_name
        dm.synchronize()
        state = sink.state()
    auds = [a.cpu().numpy() for a in audio]
    got = [p.cpu().numpy() for p in pcm]
    for s in (0, 9, 255):
        want, st = _host_chain(pkg, auds, alpha, gain, s)
        for k in range(nb):
            d = np.abs(got[k][s].astype(np.int32) - want[k].astype(np.int32))
            assert d.max() <= 1, (s, k, int(d.max()), int(np.argmax(d)))
        assert abs(state[s] - st) <= 1e-6 * max(abs(st), 0.25), (s, state[s], st)


@pytest.mark.parametrize("fs,decim,adecim,taps", [(2.048e6, 8, 8, 64), (3.2e6, 16, 5, 64), (2.4e6, 10, 5, 16)])
def test_pcm_chain_at_the_other_front_end_rates(pkg, fs, decim, adecim, taps):
    """The kernels with the sink's chain exist for every shape design Q has an instance for: 2.048 MS/s / 8 / 8, 3.2 MS/s / 16 / 5, and 16 channel taps."""
    import torch
    alpha, gain = _params(pkg)
    ns, nb = 256, 4
    nsamp = decim * adecim * 8 * 120                                             # whole numbers of 8 audio periods: 960 outputs per call
    na = nsamp // (decim * adecim)
    h, g = pkg.default_config(taps, fs=fs, fir_decim=decim, audio_decim=adecim) if decim != 10 else pkg.default_config(taps)
    rows = pkg.make_iq(8, nb * nsamp, mode="fm", first_id=4100)
    iq = torch.from_numpy(rows).cuda().repeat(ns // 8, 1)
    audio = [torch.zeros((ns, na), dtype=torch.float32, device="cuda") for _ in range(nb)]
    pcm = [torch.zeros((ns, 2 * na), dtype=torch.int16, device="cuda") for _ in range(nb)]
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, fir_decim=decim, audio_decim=adecim, max_bytes_per_call=2 * nsamp)) as dm, \
            pkg.PcmSink(ns, alpha, gain) as sink:
        for k in range(nb):
            assert dm.process_batch_pcm_device(sink, iq[:, 2 * k * nsamp:], audio[k], pcm[k], nbytes=2 * nsamp, overlap=True) == na
            assert ("+ pcm" in dm.kernel_name) == (k > 0), dm.kernel_name
        dm.synchronize()
    auds = [a.cpu().numpy() for a in audio]
    got = [p.cpu().numpy() for p in pcm]
    for s in (0, 13, 255):
        want, _ = _host_chain(pkg, auds, alpha, gain, s)
        for k in range(nb):
            d = np.abs(got[k][s].astype(np.int32) - want[k].astype(np.int32))
            assert d.max() <= 1, (s, k, int(d.max()))


@pytest.mark.parametrize("taps", [16, 64])
def test_pcm_chain_for_one_dongle(pkg, taps):
    """BASELINE configs[1]'s shape through the PCM call: ONE stream, a second of IQ per call — hundreds of short runs of one stream, each finishing its first 64 outputs
    with its neighbour's state."""
    import torch
    alpha, gain = _params(pkg)
    nsamp, nb = 2400000, 3
    na = nsamp // 50
    h, g = pkg.default_config(taps)
    iq = torch.from_numpy(pkg.make_iq(1, nb * nsamp, mode="fm", first_id=4300)).cuda()
    audio = [torch.zeros((1, na), dtype=torch.float32, device="cuda") for _ in range(nb)]
    pcm = [torch.zeros((1, 2 * na), dtype=torch.int16, device="cuda") for _ in range(nb)]
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=1, max_bytes_per_call=2 * nsamp)) as dm, pkg.PcmSink(1, alpha, gain) as sink:
        for k in range(nb):
            assert dm.process_batch_pcm_device(sink, iq[:, 2 * k * nsamp:], audio[k], pcm[k], nbytes=2 * nsamp, overlap=True) == na
            assert ("+ pcm" in dm.kernel_name) == (k > 0), dm.kernel_name
        dm.synchronize()
        assert sink.synchronize_status() == 0
        state = sink.state()
    auds = [a.cpu().numpy() for a in audio]
    want, st = _host_chain(pkg, auds, alpha, gain, 0)
    for k in range(nb):
        d = np.abs(pcm[k][0].cpu().numpy().astype(np.int32) - want[k].astype(np.int32))
        assert d.max() <= 1, (k, int(d.max()), int(np.argmax(d)))
    assert abs(state[0] - st) <= 1e-6 * max(abs(st), 0.25)


@pytest.mark.parametrize("ns,nsamp", [(1, 131072), (4, 48000), (64, 240000)])
def test_pcm_call_on_host_buffers(pkg, ns, nsamp):
    """Without SDRFM_F_DEVICE_PTRS the call takes host buffers, synchronous like sdrfm_process_batch with host buffers — the reference superloop's two steps on one
    filled CommItf.buff (here: one dongle's DEFAULT_BUF_LENGTH, four dongles, a batch design Q serves): PCM within 1 LSB of the host routine over the audio of the
    same call, with and without an audio buffer, the state carried across calls."""
    import ctypes as C
    alpha, gain = _params(pkg)
    h, g = pkg.default_config(64)
    lib = pkg.load_library()
    nb = 3
    iq = pkg.make_iq(ns, nb * nsamp, mode="fm", first_id=4500)
    na_cap = nsamp // 50 + 1
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm, pkg.PcmSink(ns, alpha, gain) as sink, \
            pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dref:
        state = {s: 0.0 for s in range(ns)}
        for k in range(nb):
            chunk = np.ascontiguousarray(iq[:, 2 * k * nsamp:2 * (k + 1) * nsamp])
            audio = np.zeros((ns, na_cap), np.float32)
            pcm = np.full((ns, 2 * na_cap), 12345, np.int16)
            n = C.c_uint32()
            with_audio = k != 1
            rc = lib.sdrfm_process_batch_pcm(dm._h, sink._h, chunk.ctypes.data_as(C.c_void_p), chunk.strides[0], 2 * nsamp,
                                             audio.ctypes.data_as(C.c_void_p) if with_audio else None, na_cap, pcm.ctypes.data_as(C.c_void_p), 2 * na_cap,
                                             C.byref(n), 0)
            assert rc == 0, rc
            ref = dref.process_batch(chunk) if ns > 1 else dref.process(chunk[0])[None, :]
            assert n.value == ref.shape[1]
            if with_audio:
                assert np.array_equal(audio[:, :n.value], ref)      # (the audio the call leaves is sdrfm_process_batch's)
            for s in range(0, ns, max(1, ns // 4)):
                want, state[s] = pkg.pcm_deemph_s16_host(ref[s], alpha, gain, state[s])
                assert np.abs(pcm[s, :2 * n.value].astype(np.int32) - want.astype(np.int32)).max() <= 1, (k, s)
                assert (pcm[s, 2 * n.value:] == 12345).all()
        got_state = sink.state()
        for s in state:
            if s % max(1, ns // 4) == 0:
                assert abs(got_state[s] - state[s]) <= 1e-6 * max(abs(state[s]), 0.25), (s, got_state[s], state[s])
