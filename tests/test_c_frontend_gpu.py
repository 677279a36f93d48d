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
