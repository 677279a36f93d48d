"""The plain-C front end (examples/replay_main.c) driving libsdrfm.so through the reference's hand-off cadence."""
import os
import subprocess

import numpy as np
import pytest

from conftest import scaled_err, TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_matches_oracle(pkg, oracle_mod, tmp_path):
    exe = os.path.join(ROOT, "examples", "replay_main")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    h, g_ = pkg.default_config(64)
    iq = pkg.make_iq(1, 120000, mode="fm", first_id=21)[0]
    (tmp_path / "iq.u8").write_bytes(iq.tobytes())
    (tmp_path / "h.f32").write_bytes(h.tobytes())
    (tmp_path / "g.f32").write_bytes(g_.tobytes())
    for buff_size in (512, 127 * 512):
        out = tmp_path / ("audio_%d.f32" % buff_size)
        r = subprocess.run([exe, str(tmp_path / "iq.u8"), str(out), str(tmp_path / "h.f32"), str(tmp_path / "g.f32"),
                            str(buff_size)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = np.fromfile(out, dtype=np.float32)
        want = oracle_mod.Oracle(h, g_).process(iq)
        assert got.size == want.size == 2400
        assert scaled_err(got, want) <= TOL


def test_c_views_program_matches_oracles(pkg, oracle_mod, tmp_path):
    """examples/views_main.c (strict C99): spectrum view of a capture and the 16-channel WBFM path fed in 65 024-byte pieces."""
    exe = os.path.join(ROOT, "examples", "views_main")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
    g_ = (pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0).astype(np.float32)
    iq = pkg.make_iq(1, 200000, mode="fm", fs=3.2e6, first_id=23)[0]
    (tmp_path / "iq.u8").write_bytes(iq.tobytes())
    (tmp_path / "p.f32").write_bytes(p.astype(np.float32).tobytes())
    (tmp_path / "g.f32").write_bytes(g_.tobytes())
    buff = 127 * 512
    r = subprocess.run([exe, str(tmp_path / "iq.u8"), str(tmp_path / "spec.f32"), str(tmp_path / "wbfm.f32"), str(tmp_path / "p.f32"),
                        str(tmp_path / "g.f32"), "1024", str(buff)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    spec = np.fromfile(tmp_path / "spec.f32", dtype=np.float32)
    want_spec, frames = oracle_mod.SpectrumOracle(1024).process(iq)
    assert frames == 195 and np.array_equal(spec.view(np.uint32), want_spec.view(np.uint32))
    want = oracle_mod.WbfmOracle(p, g_).process(iq)                       # [16][n_total]; the C program wrote per-hand-off blocks
    raw = np.fromfile(tmp_path / "wbfm.f32", dtype=np.float32)
    o = oracle_mod.WbfmOracle(p, g_)
    pos, off, band0 = 0, 0, []
    while pos < iq.size:
        n = o.process(iq[pos:pos + buff]).shape[1]
        blk = raw[off:off + 16 * n].reshape(16, n)
        band0.append(blk[0])
        off += 16 * n
        pos += buff
    assert off == raw.size
    got0 = np.concatenate(band0)
    assert got0.size == want.shape[1]
    assert scaled_err(got0, want[0]) <= TOL


@pytest.mark.parametrize("transport", ["rccl", "peer-copy", "rccl-overlap"])
def test_c_multi_gpu_host_on_the_devices_present(pkg, oracle_mod, tmp_path, transport):
    """examples/multi_gpu_main.c: ONE C process, one sdrfm_t per device, contiguous stream shards (sdrfm_shard_range), fan-out of the IQ
    batch and fan-in of the audio over RCCL send / recv (or peer copies).  Runs on however many devices the box has (the GPU box:
    one, so the collectives degenerate to the root's own block; N > 1 is unmeasured on hardware) — every stream against the oracle."""
    import json
    import torch
    exe = os.path.join(ROOT, "examples", "multi_gpu_main")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(exe):
        pytest.skip("examples/multi_gpu_main did not build on this box (RCCL development files missing)")
    h, g_ = pkg.default_config(64)
    ns, nsamp = 160, 24000                                      # 19 steps per stream: the matrix-pipe kernel serves it
    rows = np.concatenate([pkg.make_iq(6, nsamp, mode="fm", first_id=61), pkg.make_iq(2, nsamp, mode="random", first_id=67)])
    iq = np.tile(rows, (ns // 8, 1))
    (tmp_path / "iq.u8").write_bytes(iq.tobytes())
    (tmp_path / "h.f32").write_bytes(h.tobytes())
    (tmp_path / "g.f32").write_bytes(g_.tobytes())
    out = tmp_path / "audio.f32"
    reps = 3 if transport == "rccl-overlap" else 2
    cmd = [exe, str(tmp_path / "iq.u8"), str(tmp_path / "h.f32"), str(tmp_path / "g.f32"), str(ns), str(2 * nsamp), "0", str(out), "--reps", str(reps)]
    if transport == "peer-copy":
        cmd.append("--peer-copy")
    if transport == "rccl-overlap":                             # SDRFM_F_OVERLAP calls, two buffers per device, the fan-in one call behind
        cmd.append("--overlap")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    info = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert info["n_gpus"] == torch.cuda.device_count() and info["n_streams"] == ns and info["n_audio"] == 480
    assert sum(c for _, c in info["shards"]) == ns and info["kernel"].startswith("fast-q")
    assert ("overlapped" in info["kernel"]) == (transport == "rccl-overlap"), info
    got = np.fromfile(out, dtype=np.float32).reshape(ns, 480)
    # the repetitions ran on carried state: the last call's audio is what was written (streams continue: feed the oracle as often)
    for s_ in range(8):
        o = oracle_mod.Oracle(h, g_)
        for _ in range(reps - 1):
            o.process(iq[s_])
        assert scaled_err(got[s_], o.process(iq[s_])) <= TOL, s_
    for rep in range(1, ns // 8):
        assert np.array_equal(got[8 * rep:8 * rep + 8].view(np.uint32), got[:8].view(np.uint32)), rep


def test_c_pipeline_host_overlapped_equals_serial_and_the_host_sink(pkg, tmp_path):
    """examples/pipeline_main.c: a capture ring of three device buffers filled on the stream, SDRFM_F_OVERLAP calls, the device PCM sink
    as the consumer of call k-1 behind sdrfm_flush_previous while call k runs.  The PCM must equal the same program's --serial output
    bit for bit, and — within the 1 LSB the device sink's blocked scan is held to — the host sink run over the Python wrapper's serial audio; so must the PCM
    of --one-call (sdrfm_process_batch_pcm: no consumer, no audio buffer — the sink inside the demodulator's own launch)."""
    import json
    exe = os.path.join(ROOT, "examples", "pipeline_main")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(exe):
        pytest.skip("examples/pipeline_main did not build on this box")
    h, g_ = pkg.default_config(64)
    ns, nsamp, n_calls = 256, 24000, 9
    batches = [pkg.make_iq(ns, nsamp, mode="fm", first_id=1000 + 5 * k) for k in range(n_calls)]
    (tmp_path / "iq.u8").write_bytes(b"".join(b.tobytes() for b in batches))
    (tmp_path / "h.f32").write_bytes(h.tobytes())
    (tmp_path / "g.f32").write_bytes(g_.tobytes())
    outs = {}
    for mode in ("overlapped", "serial", "one-call"):
        out = tmp_path / ("pcm_%s.s16" % mode)
        cmd = [exe, str(tmp_path / "iq.u8"), str(tmp_path / "h.f32"), str(tmp_path / "g.f32"), str(ns), str(2 * nsamp), str(n_calls), str(out)]
        if mode != "overlapped":
            cmd.append("--" + mode)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        info = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert info["mode"] == mode and info["n_audio"] == 480 and info["kernel"].startswith("fast-q")
        assert ("overlapped" in info["kernel"]) == (mode != "serial")
        assert ("+ pcm" in info["kernel"]) == (mode == "one-call")                           # (sdrfm_process_batch_pcm: the sink inside the demodulator's launch)
        outs[mode] = np.fromfile(out, dtype=np.int16).reshape(n_calls, ns, 960)
    assert np.array_equal(outs["overlapped"], outs["serial"])
    lib = pkg.load_library()
    alpha, gain = lib.sdrfm_pcm_alpha(48000.0, 75e-6), np.float32(16688.0)                   # the program's constants
    with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g_, n_streams=ns)) as dm:
        audio = [dm.process_batch(b) for b in batches]
    for s_ in (0, 131, 255):
        st = 0.0
        for k in range(n_calls):
            want, st = pkg.pcm_deemph_s16_host(audio[k][s_], alpha, gain, st)
            # (the program uses the device sink's default form, the blocked scan: within 1 LSB of the host routine's exact chain)
            assert np.abs(outs["overlapped"][k, s_].astype(np.int32) - want.astype(np.int32)).max() <= 1, (s_, k)
            assert np.abs(outs["one-call"][k, s_].astype(np.int32) - want.astype(np.int32)).max() <= 1, (s_, k)


def test_c_consumer_loop_host_runs_all_its_forms(pkg):
    """examples/consumer_loop_main.c: the consumer loop of INTEGRATION.md section 3 from a plain-C host — overlapped calls alone, the device PCM sink on the handle's
    stream behind sdrfm_flush_previous, and the sink on its own stream behind sdrfm_wait_previous with the audio buffers guarded on the host.  A short run of each
    form completes and reports sane figures (the timings themselves are profiles/r06_sink.txt's business)."""
    import json
    exe = os.path.join(ROOT, "examples", "consumer_loop_main")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(exe):
        pytest.skip("examples/consumer_loop_main did not build on this box")
    r = subprocess.run([exe, "256", "2", "40", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    info = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert info["kernel"].startswith("fast-q") and "overlapped" in info["kernel"] and info["kernel"].endswith("+ pcm"), info
    for form in ("calls", "simple", "fast", "fused", "fused_pcm_only"):
        us = info[form]["us_per_call_regions"]
        assert len(us) == 2 and all(5.0 < x < 5000.0 for x in us), (form, us)
