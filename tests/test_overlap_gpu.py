"""SDRFM_F_OVERLAP (include/sdrfm.h): consecutive device-pointer calls that may run concurrently on the device.  An overlapped call warms
its streams up from the previous call's buffer instead of reading the carried state, so the audio must be the same, bit for bit, as
that of the same calls made one after the other; the state the overlapped calls leave must serve whatever call comes next."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handles(pkg, ns, T=64, **kw):
    h, g = pkg.default_config(T)
    return pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, **kw))


@pytest.mark.parametrize("ns,nsamp,T", [(256, 24000, 64), (64, 240000, 64), (256, 24000, 16), (3, 1200000, 64)])
def test_overlapped_calls_equal_serial_calls_bitwise(pkg, ns, nsamp, T):
    import torch
    nb = 6
    iqs = [torch.from_numpy(pkg.make_iq(ns, nsamp, mode=("fm", "random")[b & 1], first_id=40 * b + 1)).cuda() for b in range(nb)]
    A = nsamp // 50
    got = [torch.full((ns, A), float("nan"), dtype=torch.float32, device="cuda") for _ in range(nb)]
    want = [torch.full((ns, A), float("nan"), dtype=torch.float32, device="cuda") for _ in range(nb)]
    tail_iq = torch.from_numpy(pkg.make_iq(ns, 777, mode="fm", first_id=999)).cuda()      # an odd-sized call: generic kernel, carried state
    tail_got = torch.zeros((ns, 64), dtype=torch.float32, device="cuda")
    tail_want = torch.zeros((ns, 64), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with _handles(pkg, ns, T) as dm, _handles(pkg, ns, T) as ref:
        names = []
        for b in range(nb):
            assert dm.process_batch_device(iqs[b], got[b], overlap=True) == A
            names.append(dm.kernel_name)
        n_tail = dm.process_batch_device(tail_iq, tail_got)            # no flag: ordered behind the overlapped calls by the library
        dm.synchronize()
        for b in range(nb):
            assert ref.process_batch_device(iqs[b], want[b]) == A
        assert ref.process_batch_device(tail_iq, tail_want) == n_tail
        ref.synchronize()
        assert "fast-q" in ref.kernel_name or "generic" in ref.kernel_name
    assert "overlapped" not in names[0], names                        # nothing to warm up from: served as an ordinary call
    assert all("fast-q" in n and "overlapped" in n for n in names[1:]), names
    for b in range(nb):
        a, w = got[b].cpu().numpy(), want[b].cpu().numpy()
        assert np.isfinite(w).all()
        assert np.array_equal(a.view(np.uint32), w.view(np.uint32)), (b, int(np.argmax((a != w).any(axis=0))))
    assert n_tail > 0
    assert np.array_equal(tail_got.cpu().numpy()[:, :n_tail].view(np.uint32), tail_want.cpu().numpy()[:, :n_tail].view(np.uint32))


def test_overlapped_calls_against_the_oracle(pkg, oracle_mod, tol):
    """Not only equal to the serial calls: right.  Three overlapped calls on 16 streams against the oracle's one-shot run."""
    import torch
    from conftest import scaled_err
    ns, nsamp = 16, 240000
    h, g = pkg.default_config(64)
    host = [pkg.make_iq(ns, nsamp, mode="fm", first_id=70 + 16 * b) for b in range(3)]
    iqs = [torch.from_numpy(x).cuda() for x in host]
    out = [torch.zeros((ns, 4800), dtype=torch.float32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm:
        for b in range(3):
            dm.process_batch_device(iqs[b], out[b], overlap=True)
        assert "overlapped" in dm.kernel_name
        dm.flush()
        dm.synchronize()
    got = np.concatenate([o.cpu().numpy() for o in out], axis=1)
    for s in (0, 7, 15):
        want = oracle_mod.Oracle(h, g).process(np.concatenate([x[s] for x in host]))
        assert scaled_err(got[s], want) <= tol


def test_overlapped_call_waits_for_input_produced_on_the_handles_stream(pkg):
    """iq filled on the caller's stream right before the call (an H2D copy): the overlapped call runs on an internal stream but behind
    that copy; the caller's stream sees the audio after flush()."""
    import torch
    ns, nsamp, nb = 256, 24000, 8
    host = [torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", first_id=300 + b)).pin_memory() for b in range(nb)]
    ring = [torch.zeros((ns, 2 * nsamp), dtype=torch.uint8, device="cuda") for _ in range(3)]     # the previous buffer stays intact
    audio = [torch.zeros((ns, 480), dtype=torch.float32, device="cuda") for _ in range(2)]
    res = [torch.zeros((ns, 480), dtype=torch.float32, device="cuda") for _ in range(nb)]
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with _handles(pkg, ns) as dm, _handles(pkg, ns) as ref:
        dm.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            for b in range(nb):
                ring[b % 3].copy_(host[b], non_blocking=True)
                dm.process_batch_device(ring[b % 3], audio[b & 1], overlap=True)
                dm.flush()                                                # the copy below runs on `st`: behind the call
                res[b].copy_(audio[b & 1], non_blocking=True)
        st.synchronize()
        want = [ref.process_batch(h.numpy()) for h in host]
    for b in range(nb):
        assert np.array_equal(res[b].cpu().numpy().view(np.uint32), want[b].view(np.uint32)), b


def test_overlap_flag_on_a_bit_exact_handle_and_flag_errors(pkg):
    import torch
    ns, nsamp = 256, 24000
    iqs = [torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", first_id=5 + b)).cuda() for b in range(3)]
    a = [torch.zeros((ns, 480), dtype=torch.float32, device="cuda") for _ in range(3)]
    w = [torch.zeros((ns, 480), dtype=torch.float32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    with _handles(pkg, ns, bit_exact=True) as dm, _handles(pkg, ns, bit_exact=True) as ref:
        for b in range(3):
            dm.process_batch_device(iqs[b], a[b], overlap=True)      # not served by the matrix-pipe kernel: as if the flag were absent
            assert "overlapped" not in dm.kernel_name and "fast-q" not in dm.kernel_name
            ref.process_batch_device(iqs[b], w[b])
        dm.synchronize(); ref.synchronize()
        for b in range(3):
            assert np.array_equal(a[b].cpu().numpy().view(np.uint32), w[b].cpu().numpy().view(np.uint32))
        lib = pkg.load_library()
        n = C.c_uint32()
        host = np.zeros((ns, 2 * nsamp), np.uint8)
        out = np.zeros((ns, 480), np.float32)
        rc = lib.sdrfm_process_batch(dm._h, host.ctypes.data_as(C.c_void_p), host.strides[0], 2 * nsamp, out.ctypes.data_as(C.c_void_p), 480,
                                     C.byref(n), 2)                   # SDRFM_F_OVERLAP without SDRFM_F_DEVICE_PTRS
        assert rc == 16                                                # SDRFM_EINVAL


def test_pipeline_with_the_pcm_sink_on_the_handles_stream(pkg):
    """The loop INTEGRATION.md shows: call k overlapped, then the consumer of call k-1 (the device PCM sink, on the handle's stream) behind
    sdrfm_flush_previous; two audio buffers in turn.  PCM within 1 LSB of the host sink over the serial calls' audio."""
    import torch
    lib = pkg.load_library()
    alpha, gain = lib.sdrfm_pcm_alpha(48000.0, 75e-6), np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3))
    ns, nsamp, nb = 256, 24000, 7
    iqs = [torch.from_numpy(pkg.make_iq(ns, nsamp, mode="fm", first_id=900 + 3 * b)).cuda() for b in range(nb)]
    audio = [torch.zeros((ns, 480), dtype=torch.float32, device="cuda") for _ in range(2)]
    pcm = [torch.zeros((ns, 960), dtype=torch.int16, device="cuda") for _ in range(nb)]
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with _handles(pkg, ns) as dm, pkg.PcmSink(ns, alpha, gain) as sink, _handles(pkg, ns) as ref:
        dm.set_stream(st.cuda_stream); sink.set_stream(st.cuda_stream)
        for k in range(nb):
            n = dm.process_batch_device(iqs[k], audio[k & 1], overlap=True)
            if k:
                dm.flush(keep_last=True)
                sink.process_batch_device(audio[(k - 1) & 1], pcm[k - 1], n)
        dm.flush()
        sink.process_batch_device(audio[(nb - 1) & 1], pcm[nb - 1], n)
        st.synchronize()
        want_audio = [ref.process_batch(x.cpu().numpy()) for x in iqs]
    for s in (0, 100, 255):
        state = 0.0
        for k in range(nb):
            want, state = pkg.pcm_deemph_s16_host(want_audio[k][s], alpha, gain, state)
            assert np.abs(pcm[k].cpu().numpy()[s].astype(np.int32) - want.astype(np.int32)).max() <= 1, (s, k)   # (the sink's default form: a blocked scan, 1 LSB)


def test_broken_promises_fall_back_to_serial_calls(pkg, oracle_mod, tol):
    """SDRFM_F_OVERLAP asks the caller to keep the previous call's input intact and to alternate the audio buffers.  Where the library can
    see that the promise does not hold — this call's rows overlap the previous call's (ONE buffer re-used for every call), or the audio
    buffer is the one the previous overlapped call is still writing — the call runs as if the flag were absent, and the audio is right."""
    import torch
    from conftest import scaled_err
    ns, nsamp = 256, 24000
    h, g = pkg.default_config(64)
    host = [pkg.make_iq(ns, nsamp, mode="fm", first_id=500 + 300 * b) for b in range(4)]
    one_buf = torch.zeros((ns, 2 * nsamp), dtype=torch.uint8, device="cuda")
    two_bufs = [torch.zeros_like(one_buf) for _ in range(2)]
    audio = [torch.zeros((ns, nsamp // 50), dtype=torch.float32, device="cuda") for _ in range(4)]
    tmp = torch.zeros_like(audio[0])
    torch.cuda.synchronize()
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns)) as dm:
        names = []
        for b in range(2):                                            # (a) one input buffer for every call, refilled in between
            dm.synchronize()
            one_buf.copy_(torch.from_numpy(host[b])); torch.cuda.synchronize()
            dm.process_batch_device(one_buf, audio[b], overlap=True)
            names.append(dm.kernel_name)
        dm.synchronize()
        for b in (2, 3):                                              # (b) two input buffers, but the SAME audio buffer twice in a row
            two_bufs[b & 1].copy_(torch.from_numpy(host[b])); torch.cuda.synchronize()
            dm.process_batch_device(two_bufs[b & 1], tmp, overlap=True)
            names.append(dm.kernel_name)
            dm.synchronize()
            audio[b].copy_(tmp); torch.cuda.synchronize()
    assert all(n.startswith("fast-q") for n in names), names
    assert "overlapped" not in names[1], names                        # same rows as the previous call's: serial
    assert "overlapped" in names[2], names                            # a fresh buffer behind call 1: the promise holds
    assert "overlapped" not in names[3], names                        # the audio buffer call 2 may still be writing: serial
    got = np.concatenate([a.cpu().numpy() for a in audio], axis=1)
    for s in (0, 100, 255):
        want = oracle_mod.Oracle(h, g).process(np.concatenate([x[s] for x in host]))
        assert scaled_err(got[s], want) <= tol
