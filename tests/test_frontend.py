"""The replay front end honours the reference FSM's hand-off contract (usbh_rtlsdr.c:1058-1101)."""
import numpy as np
import pytest


def test_urb_size_limits(pkg):
    data = np.arange(10, dtype=np.uint8)
    pkg.ReplayFrontEnd(data, 512)
    pkg.ReplayFrontEnd(data, 127 * 512)            # largest 512-multiple below the uint16_t limit
    for bad in (0, 100, 513, 128 * 512):
        with pytest.raises(ValueError):
            pkg.ReplayFrontEnd(data, bad)


def test_state_cadence_and_reassembly(pkg):
    data = (np.arange(5000) % 251).astype(np.uint8)
    fe = pkg.ReplayFrontEnd(data, 1024, wait_polls=3)
    seen, states = [], []
    while not fe.exhausted:
        st = fe.process()
        states.append(int(st))
        if st == pkg.XferState.COMPLETE:
            seen.append(fe.buff[: fe.last_xfer_size].copy())
    assert np.array_equal(np.concatenate(seen), data)
    assert [len(s) for s in seen] == [1024, 1024, 1024, 1024, 904]
    # START -> WAIT (x3 polls) -> COMPLETE -> START ...
    assert states[:6] == [1, 1, 1, 2, 0, 1]


def test_buffer_is_reused_so_consumer_must_not_keep_it(pkg):
    data = np.arange(2048, dtype=np.uint16).astype(np.uint8)
    fe = pkg.ReplayFrontEnd(data, 512)
    kept = fe.run(lambda buf, n: buf[:n])          # keeps VIEWS of the single buffer: all clobbered afterwards
    assert all(np.all(k == 0xEE) for k in kept)
    fe = pkg.ReplayFrontEnd(data, 512)
    copies = fe.run(lambda buf, n: buf[:n].copy())
    assert np.array_equal(np.concatenate(copies), data)
