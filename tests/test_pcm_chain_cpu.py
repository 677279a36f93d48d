"""CPU: the arithmetic of the PCM sink inside the demodulator's launch (csrc/sdrfm_sink_chain.h) restated in numpy (tools/pcm_chain_emulate.py), held to the host
routine sdrfm_pcm_deemph_s16 (csrc/pcm_sink.c) — the bit-exact definition of the sink — without a GPU."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
emu = importlib.import_module("pcm_chain_emulate")


def _audio(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 48000.0
    return (1.3 * np.sin(2 * np.pi * 1000 * t + 0.3) + 0.6 * np.sin(2 * np.pi * 3100 * t) + 0.3 * np.sin(2 * np.pi * 12000 * t) +
            0.05 * rng.standard_normal(n)).astype(np.float32)


@pytest.mark.parametrize("tau", [75e-6, 50e-6])
@pytest.mark.parametrize("run_len", [76, 400, 800, 1500])
def test_runs_sunk_on_their_own_are_within_one_lsb_of_the_exact_chain(pkg, tau, run_len):
    """Runs of 76 outputs (the shortest the host allows: 12 owned quads), 400 (BASELINE configs[2]), 800 (configs[3]'s share: two flushes per run) and 1500 (three)."""
    alpha = float(pkg.load_library().sdrfm_pcm_alpha(48000.0, tau))
    gain = np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3))
    state = 0.0
    emu_state = 0.0
    for call in range(3):
        x = _audio(4800, 10 * call + int(tau * 1e6))
        want, state = pkg.pcm_deemph_s16_host(x, alpha, gain, state)
        got, emu_state = emu.chain_emulate(x, alpha, gain, run_len=run_len, state0=emu_state)
        d = np.abs(got - want[0::2].astype(np.int32))
        assert d.max() <= 1, (call, int(d.max()), int(np.argmax(d)))
        assert (d > 0).mean() < 0.01                              # different only where y * gain sits on a rounding boundary
        assert abs(emu_state - state) <= 1e-6 * max(abs(state), 0.25), (emu_state, state)


def test_full_scale_and_clipping(pkg):
    alpha = float(pkg.load_library().sdrfm_pcm_alpha(48000.0, 75e-6))
    gain = np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3))
    x = (3.0 * _audio(4800, 3)).astype(np.float32)                  # drives the sink into its clamp
    want, _ = pkg.pcm_deemph_s16_host(x, alpha, gain)
    got, _ = emu.chain_emulate(x, alpha, gain)
    assert want.max() == 32767 and want.min() == -32768
    assert np.abs(got - want[0::2].astype(np.int32)).max() <= 1


def test_why_the_scheme_needs_a_short_memory(pkg):
    """(1 - alpha)^64 must be below rounding for a run's own end state to stand for the true one: at SDRFM_CHAIN_MIN_ALPHA it is 5e-8; at alpha = 0.05 (a 400 us time
    constant) a run's first 64 outputs are not all its predecessor reaches and the scheme is off by whole LSBs — which is why such a sink is served by the stand-alone
    kernel instead (sdrfm_sink_chain_params answers 1)."""
    assert (1.0 - 0.231) ** 64 <= 5.1e-8
    gain = np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3))
    x = (_audio(4800, 9) + np.float32(0.8)).astype(np.float32)     # (a DC offset: a state that matters)
    want, _ = pkg.pcm_deemph_s16_host(x, 0.05, gain)
    got, _ = emu.chain_emulate(x, 0.05, gain, run_len=100)
    assert np.abs(got - want[0::2].astype(np.int32)).max() > 4
