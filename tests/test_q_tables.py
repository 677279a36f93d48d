"""CPU tests of design Q's host side (csrc/qtaps.c): the channel taps as i8 matrix-pipe operand tables.

The tables are exercised by a numpy emulation of exactly what the kernel does with them (csrc/sdrfm_q.hip): per block of 8
outputs the 320-byte window XOR 0x80 as i8, three K-chunks of 128 bytes against the three digit tables in the SPARSE operand
layout of v_smfmac_i32_16x16x128_i8 (lane l: row l & 15, K group l >> 4; slot 8 h + 2 q + j = the j-th kept value of dense
positions K = 32 (l >> 4) + 16 h + 4 q .. + 3, kept positions {0, 2} in an I row and {1, 3} in a Q row — the layout
tools/ubench/ubench14.hip found on the device), exact i32 sums, the fp32 recombination — against the oracle's y (its fp32 fmaf
chain): parity 1e-5 on the audio follows (tools/q_emulate.py, gate (a) of VERDICT r02 item 1).
"""
import ctypes as C

import numpy as np
import pytest


def _build(pkg, h, D=10):
    lib = pkg.load_library()
    h = np.ascontiguousarray(h, dtype=np.float32)
    A = np.zeros(((D + 3) // 4, 3, 64, 16), dtype=np.int8)
    q, cst, c0 = C.c_float(), C.c_float(), C.c_uint32()
    rc = lib.sdrfm_q_build(h.ctypes.data, h.size, D, A.ctypes.data, C.byref(q), C.byref(cst), C.byref(c0))
    return rc, A, np.float32(q.value), np.float32(cst.value), c0.value


def _dense(A):
    """[chunk][digit][row 16][dense K 128] from the sparse operand tables (zeros where the 2:4 pattern keeps nothing)."""
    nsc = A.shape[0]
    dense = np.zeros((nsc, 3, 16, 128), dtype=np.int64)
    for lane in range(64):
        row, gA, comp = lane & 15, lane >> 4, lane & 1
        for s in range(16):
            h2, q4, j = s >> 3, (s >> 1) & 3, s & 1
            dense[:, :, row, 32 * gA + 16 * h2 + 4 * q4 + comp + 2 * j] = A[:, :, lane, s]
    return dense


def _emulate_blocks(A, q, cst, iq, D=10):
    """y[m] for every output whose 2-block window lies inside the chunk: the kernel's data flow, block by block."""
    blkb = 16 * D
    nblk = iq.size // blkb
    x = (iq.astype(np.int16) - 128).astype(np.int64)                       # byte XOR 0x80 as i8
    dense = _dense(A)
    nsc = dense.shape[0]
    out = np.zeros((nblk, 8, 2), dtype=np.float32)
    for j in range(1, nblk):
        win = np.zeros(128 * nsc, dtype=np.int64)                         # 32 D bytes: the block before, then the block; then don't-cares
        win[:2 * blkb] = x[(j - 1) * blkb:(j + 1) * blkb]
        win[2 * blkb:] = 77                                                # (the kernel reads whatever follows: the tables hold no tap there)
        S = np.zeros((3, 16), dtype=np.int64)
        for c in range(nsc):
            S += dense[c] @ win[128 * c:128 * c + 128]
        assert np.abs(S).max() < 2 ** 31
        s01 = (S[0] + 256 * S[1]).astype(np.float32)
        y = (s01.astype(np.float64) * q + (S[2].astype(np.float32).astype(np.float64) * (np.float32(65536.0) * q) + cst).astype(np.float32)).astype(np.float32)
        out[j] = y.reshape(8, 2)                                           # row 2 o + comp
    return out


@pytest.mark.parametrize("T", [16, 32, 64, 90])
def test_tables_reproduce_the_fir_through_the_mfma_layout(pkg, oracle_mod, T):
    h, g = pkg.default_config(T if T != 90 else 64)
    if T == 90:
        h = pkg.lowpass_taps(90, 0.04)
    rc, A, q, cst, c0 = _build(pkg, h)
    assert rc == 0
    assert c0 == {16: 1, 32: 0, 64: 0, 90: 0}[T]                           # first 128-byte K-chunk holding a tap
    assert not A[:c0].any()                                               # chunks before the first tap hold zeros: no MFMA needed
    for mode in ("fm", "random"):
        iq = pkg.make_iq(1, 16000, mode=mode, first_id=3)[0]
        o = oracle_mod.Oracle(h, g)
        o.process(iq)
        yo, _ = o.last_stage()
        ye = _emulate_blocks(A, q, cst, iq).reshape(-1, 2)
        m0 = 16                                                            # skip the outputs touching the zero history / block 0
        err = np.abs(ye[m0:yo.shape[0]].astype(np.float64) - yo[m0:]).max()
        assert err <= 2e-4, (mode, err)                                    # |y| ~ 100: 2e-6 relative; the audio test is in tools/q_emulate.py


@pytest.mark.parametrize("T,D,Da", [(64, 8, 8), (16, 8, 8), (64, 16, 5), (16, 16, 5)])
def test_tables_of_the_other_front_end_rates(pkg, oracle_mod, T, D, Da):
    """The 2.048 MS/s (D = 8) and 3.2 MS/s (D = 16) instances of design Q: blocks of 128 / 256 bytes, two / four K-chunks per window."""
    h, g = pkg.default_config(T, fir_decim=D, audio_taps=32, audio_decim=Da)
    rc, A, q, cst, c0 = _build(pkg, h, D)
    assert rc == 0 and A.shape[0] == (D + 3) // 4
    for mode in ("fm", "random"):
        iq = pkg.make_iq(1, 16 * D * 100, mode=mode, first_id=3)[0]
        o = oracle_mod.Oracle(h, g, D, Da)
        o.process(iq)
        yo, _ = o.last_stage()
        ye = _emulate_blocks(A, q, cst, iq, D).reshape(-1, 2)
        err = np.abs(ye[16:yo.shape[0]].astype(np.float64) - yo[16:]).max()
        assert err <= 2e-4, (mode, err)


def test_digits_are_balanced_and_exact(pkg):
    h, _ = pkg.default_config(64)
    rc, A, q, cst, c0 = _build(pkg, h)
    assert rc == 0
    dense = _dense(A)
    # read the taps back from row 0 (I of output 0): window sample w meets tap k = 89 - w, window byte 2 w
    H = np.zeros(64, dtype=np.int64)
    for k in range(64):
        c, kb = divmod(2 * (89 - k), 128)
        H[k] = sum(int(dense[c, t, 0, kb]) * 256 ** t for t in range(3))
    assert np.abs(H).max() <= 127 * 65793
    assert np.abs(H.astype(np.float64) * float(q) - h.astype(np.float64)).max() <= 0.5 * float(q) * 1.0000001
    assert cst == np.float32(0.5 * h.astype(np.float64).sum())
    # exactly 2:4 sparse by construction: an I row holds nothing at odd K, a Q row nothing at even K, and the Q row is the I row one byte later
    assert not dense[:, :, 0::2, 1::2].any() and not dense[:, :, 1::2, 0::2].any()
    assert np.array_equal(dense[:, :, 1, 1::2], dense[:, :, 0, 0::2])


def test_rejects_what_it_cannot_serve(pkg):
    h, _ = pkg.default_config(64)
    assert _build(pkg, h, D=9)[0] != 0                                     # odd decimation
    assert _build(pkg, np.zeros(8, np.float32))[0] != 0                    # all-zero taps
    assert _build(pkg, np.ones(91, np.float32))[0] != 0                    # T > 9 D
    bad = h.copy(); bad[3] = np.inf
    assert _build(pkg, bad)[0] != 0
