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
