#!/usr/bin/env python3
"""Pattern sets for tools/ldsbank/ldsbank: `python patterns.py <set> > patterns.txt` also writes patterns.labels (one label per line).
sets: calib (strides, broadcasts: what the bank model is), chain <pad...> (the access patterns of k_spectrum_chain<10> under a padding)."""
import sys
OPS = {"r64": 0, "w64": 1, "r128": 2, "w128": 3, "r32": 4, "w32": 5, "r2x64": 6, "w2x32": 7}
out, labels = [], []
def add(label, op, addrs):
    assert len(addrs) == 64 and max(addrs) < 40960 - 16, (label, max(addrs))
    out.append("%d %s" % (OPS[op], " ".join(str(int(a)) for a in addrs))); labels.append(label)
def brev(n, b): return int(format(n, "0%db" % b)[::-1], 2)

def calib():
    for op, w in (("r64", 8), ("w64", 8), ("r32", 4), ("w32", 4), ("r128", 16), ("w128", 16)):
        for st in (1, 2, 3, 4, 8, 16, 17, 32, 33, 64):
            add("%s stride %d elements" % (op, st), op, [(l * st * w) % 32768 for l in range(64)])
        add("%s all lanes one address" % op, op, [0] * 64)
        add("%s lanes 0-31 consecutive, 32-63 the same again" % op, op, [(l % 32) * w for l in range(64)])
        add("%s lanes 0-15 consecutive, repeated 4 times" % op, op, [(l % 16) * w for l in range(64)])
        add("%s two halves 256 elements apart" % op, op, [((l % 32) + 256 * (l // 32)) * w for l in range(64)])
        add("%s quarters 256 elements apart" % op, op, [((l % 16) + 256 * (l // 16)) * w for l in range(64)])
        add("%s quarters 272 elements apart" % op, op, [((l % 16) + 272 * (l // 16)) * w for l in range(64)])
        add("%s quarters 264 elements apart" % op, op, [((l % 16) + 264 * (l // 16)) * w for l in range(64)])
        add("%s pairs: lane l -> element 16 (l/2) + (l%%2)" % op, op, [(16 * (l // 2) + (l % 2)) * w for l in range(64)])

def chain(pad, gmap=None, n=1024, full=False):
    """patterns of k_spectrum_chain<10> (float2 elements, 8 bytes); pad = index padding; gmap = lane -> first-pass group"""
    gmap = gmap or (lambda l: brev(l, 6))
    cs = range(16) if full else (0, 1, 15)
    for c in cs:
        add("P1 write c=%d" % c, "w64", [8 * pad(16 * gmap(l) + c) for l in range(64)])
    base = lambda l: ((l >> 4) << 8) | (l & 15)
    for c in cs:
        add("P2 read c=%d" % c, "r64", [8 * pad(base(l) + 16 * c) for l in range(64)])
    for c in (cs if full else (0,)):
        add("P2 write c=%d" % c, "w64", [8 * pad(base(l) + 16 * c) for l in range(64)])
    for t in range(4):
        for j in (range(1 << t) if full else sorted(set((0, (1 << t) - 1)))):
            add("P2 twiddle t=%d j=%d" % (t, j), "r64", [8 * pad(((l & 15) + 16 * j) << (5 - t)) for l in range(64)])
    for i in (range(4) if full else (0, 3)):
        for c in (range(4) if full else (0, 3)):
            add("P3 read i=%d c=%d" % (i, c), "r64", [8 * pad(l + 64 * i + 256 * c) for l in range(64)])
    for i in (range(4) if full else (0,)):
        add("P3 twiddle stage 9 i=%d" % i, "r64", [8 * pad((l + 64 * i) << 1) for l in range(64)])
        for j in range(2):
            add("P3 twiddle stage 10 i=%d j=%d" % (i, j), "r64", [8 * pad(l + 64 * i + 256 * j) for l in range(64)])

def mfir():
    """the LDS accesses of k_mfir<0, 5, 10, 5> (csrc/sdrfm_q.hip), lane (n, g) = (l & 15, l >> 4): PRE = 160, ring 5120, d buffer behind it"""
    PRE, RING, DB0 = 160, 5120, 128
    db = PRE + RING
    for ringoff in (0, 2560):
        for c in (0, 5):
            add("window read piece c=%d, ring offset %d (ds_read_b128)" % (c, ringoff), "r128", [ringoff + 160 * (l & 15) + 16 * (l >> 4) + 64 * c for l in range(64)])
    for sigma in (0, 1):
        add("d write, lane (n, g) -> words 8 n + 2 g, + 1 (ds_write2_b32), sigma=%d" % sigma, "w2x32", [db + 4 * (DB0 + sigma + 8 * (l & 15) + 2 * (l >> 4)) for l in range(64)])
    add("d write as ds_write_b64", "w64", [db + 4 * (DB0 + 8 * (l & 15) + 2 * (l >> 4)) for l in range(64)])
    add("d write, padded layout: word m + 2 (m >> 5) (what a conflict-free layout would cost: not built)", "w2x32",
        [db + 4 * (DB0 + (8 * (l & 15) + 2 * (l >> 4)) + 2 * ((8 * (l & 15)) >> 5)) for l in range(64)])
    for phi in (0, 3):
        for i in (0, 9):
            add("K4 window read pair i=%d, phi=%d: word phi + 10 lane - 31 + 4 i (ds_read2_b64)" % (i, phi), "r2x64",
                [db + 4 * (DB0 + ((phi + 1) & 1) + phi + 10 * l - 31 + 4 * i - ((phi + 10 * l - 31) & 1) * 0) & ~7 for l in range(64)])
    add("parked audio write: 8 bytes per lane (ds_write_b64)", "w64", [db + 4 * (DB0 + 656) + 8 * l for l in range(64)])
    add("parked audio read at the flush (ds_read_b64)", "r64", [db + 4 * (DB0 + 656) + 8 * l for l in range(64)])
    add("history copy read (ds_read_b32, lanes 0-31)", "r32", [db + 4 * (DB0 + 640 - 32 + (l & 31)) for l in range(64)])
    add("history copy write (ds_write_b32, lanes 0-31)", "w32", [db + 4 * (DB0 - 32 + (l & 31)) for l in range(64)])
    add("repair list entry write (ds_write_b16 ~ ds_write_b32 at 2 l)", "w32", [db + 4 * (DB0 + 656 + 512) + 4 * (l // 2) for l in range(64)])

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "calib"
    if which == "calib":
        calib()
    elif which == "mfir":
        mfir()
    elif which in ("chain", "chainperm", "chainall", "chainpermall"):
        co = [int(x) for x in sys.argv[2:]] or [4, 1, 8, 1]          # pairs (shift, coefficient)
        # chainperm: first-pass group of a lane = bits (4, 0, 1, 2, 5, 3) of the lane number (group bit k = lane bit perm[k])
        perm = (4, 0, 1, 2, 5, 3)
        gm = (lambda l: sum(((l >> perm[k]) & 1) << k for k in range(6))) if "perm" in which else None
        chain(lambda i: i + sum(c * (i >> s) for s, c in zip(co[0::2], co[1::2])), gm, full=which.endswith("all"))
    print("\n".join(out))
    open("patterns.labels", "w").write("\n".join(labels) + "\n")
