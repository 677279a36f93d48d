"""Fixtures for design Q's conditioning guard (csrc/sdrfm_q.hip): the two soak cases of round 3 in which the matrix-pipe kernel WITHOUT a
guard left the 1e-5 tolerance (profiles/r03b_fuzz_q.txt), cut down to the row and the stretch that matters.  Both are rows of uniform random
bytes (SURVEY.md 8d's worst-case class).

  q_guard_branch_cut_T64.npz   tools/fuzz_q.py seed 7 case 1033, row 4: the BASELINE 64 taps, two calls (16400 + 15200 samples) with a reset
                               before the second; in call 2 one discriminator input sits on the branch cut (oracle: d = 3.1415925, re = -521,
                               im = +1.0e-4): an unguarded design Q answers -pi, seven audio samples are 0.78 = 2 pi g[k] off.
  q_guard_deep_fade_T32.npz    seed 7 case 217, row 1, samples 500000 .. 532000 of 3.9 M: the 32-tap default filter; output 1804 of the cut
                               (51804 of the case) has |y| = 0.0035 against a median of 22: an unguarded design Q is 1.16e-5 off in one audio sample.

usage: make_golden_q_guard.py case_7_1033.npz case_7_217.npz   (the soak's dumps; not kept in the repo — 15 MB).  Expected audio = the oracle's.
       make_golden_q_guard.py --regenerate [--check]       from a clean checkout: the taps, bytes, call sizes and resets are read from the committed fixtures
                                                            themselves and the expected audio is re-derived from them with the oracle (--check: compare with
                                                            what the fixtures hold instead of rewriting them; exit status 1 on a difference)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Oracle  # noqa: E402


def expected(h, g, row, sizes, reset_before):
    orc, pos, out = Oracle(h, g), 0, []
    for ci, n in enumerate(sizes):
        if ci in reset_before:
            orc.reset()
        out.append(orc.process(row[2 * pos:2 * (pos + n)]).astype(np.float32))
        pos += n
    return np.concatenate(out)


def main(a, b):
    z = np.load(a)
    row, sizes = z["rows"][4], [int(x) for x in z["sizes"]]
    np.savez_compressed(os.path.join(HERE, "q_guard_branch_cut_T64.npz"), h=z["h"], g=z["g"], iq=row, sizes=np.array(sizes), reset_before=np.array([1]),
                        audio=expected(z["h"], z["g"], row, sizes, {1}))
    z = np.load(b)
    row, sizes = z["rows"][1][2 * 500000:2 * 532000], [16000, 16000]
    np.savez_compressed(os.path.join(HERE, "q_guard_deep_fade_T32.npz"), h=z["h"], g=z["g"], iq=row, sizes=np.array(sizes), reset_before=np.array([], dtype=np.int64),
                        audio=expected(z["h"], z["g"], row, sizes, set()))


def regenerate(check):
    ok = True
    for name in ("q_guard_branch_cut_T64.npz", "q_guard_deep_fade_T32.npz"):
        path = os.path.join(HERE, name)
        z = np.load(path)
        audio = expected(z["h"], z["g"], z["iq"], [int(x) for x in z["sizes"]], {int(x) for x in z["reset_before"]})
        same = np.array_equal(audio.view(np.uint32), z["audio"].view(np.uint32))
        print("%s: %d audio samples, %s the committed ones" % (name, audio.size, "bit-identical to" if same else "DIFFERENT from"))
        ok = ok and same
        if not check:
            np.savez_compressed(path, h=z["h"], g=z["g"], iq=z["iq"], sizes=z["sizes"], reset_before=z["reset_before"], audio=audio)
    return ok


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--regenerate":
        sys.exit(0 if regenerate("--check" in sys.argv[2:]) else 1)
    main(*sys.argv[1:3])
