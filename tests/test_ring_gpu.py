"""Pinned-ring streaming front end (sdrfm_ring_*): the reference's declared-but-unused 15 x 262144-byte buffer scheme
(usbh_rtlsdr.h:277-278) driven without blocking, audio in order and identical to the synchronous path."""
import ctypes as C

import numpy as np
import pytest

from conftest import scaled_err, TOL

pytestmark = pytest.mark.gpu
BUSY = 1


def test_ring_feeds_in_order_and_matches_oracle(pkg, oracle_mod):
    lib = pkg.load_library()
    h, g = pkg.default_config(64)
    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, max_bytes_per_call=262144))
    ring = C.c_void_p()
    assert lib.sdrfm_ring_create(dm._h, 15, 16 * 32 * 512, C.byref(ring)) == 0     # DEFAULT_BUF_NUMBER x DEFAULT_BUF_LENGTH
    iq = pkg.make_iq(1, 131072 * 40 + 12345 * 1, mode="fm", first_id=61)[0]
    iq = iq[: iq.size & ~1]
    out, pos, busy_submit = [], 0, 0
    buf = np.empty(262144 // 2 // 10 // 5 + 8, np.float32)
    n = C.c_uint32()
    reuse = np.empty(262144, np.uint8)                                             # ONE caller buffer, reused like CommItf.buff
    while pos < iq.size or True:
        if pos < iq.size:
            m = min(262144, iq.size - pos)
            reuse[:m] = iq[pos:pos + m]
            st = lib.sdrfm_ring_submit(ring, reuse.ctypes.data, m)
            if st == 0:
                pos += m
                reuse[:] = 0xEE                                                     # slot owns a copy: clobbering is harmless
            else:
                assert st == BUSY
                busy_submit += 1
                assert lib.sdrfm_ring_collect(ring, buf.ctypes.data, buf.size, C.byref(n), 1) == 0
                out.append(buf[: n.value].copy())
            continue
        st = lib.sdrfm_ring_collect(ring, buf.ctypes.data, buf.size, C.byref(n), 1)
        if st == BUSY:
            break                                                                  # ring drained
        assert st == 0
        out.append(buf[: n.value].copy())
    got = np.concatenate(out)
    want = oracle_mod.Oracle(h, g).process(iq)
    assert busy_submit > 0                                                         # 41 buffers through 15 slots
    assert got.size == want.size
    assert scaled_err(got, want) <= TOL
    # non-blocking collect on an empty ring and argument checks
    assert lib.sdrfm_ring_collect(ring, buf.ctypes.data, buf.size, C.byref(n), 0) == BUSY
    assert lib.sdrfm_ring_submit(ring, reuse.ctypes.data, 7) == 17
    assert lib.sdrfm_ring_submit(ring, reuse.ctypes.data, 262146) == 18
    lib.sdrfm_ring_destroy(ring)
    dm.close()
