#!/usr/bin/env python3
"""Per-wave stamps of design Q (tools/qbench, QBENCH_DUMP) grouped by where the wave ran: does a wave's finishing time depend on its
SIMD slot (issue priority by age), its CU, its XCC, or its place in the stream?   usage: stamps_by_slot.py dump.txt"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.uint64)
blk, word, t_in, t_first, t_loop, t_exit, wait, steps = a.T
xcc = (word & 0xff).astype(int); hw = (word >> 8).astype(np.int64)
wave_id = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
t0 = t_in.min()
us = lambda t: (t.astype(np.int64) - int(t0)) * 0.01
dur = us(t_loop) - us(t_first)
per_step = dur / steps
print("waves", len(blk), "kernel end %.2f us" % us(t_exit).max())
def table(name, key):
    print(name)
    for k in np.unique(key):
        m = key == k
        print("  %3d: n=%4d  first %.2f  loop_end mean %.2f  max %.2f  us/step %.3f" % (k, m.sum(), us(t_first)[m].mean(), us(t_loop)[m].mean(), us(t_loop)[m].max(), per_step[m].mean()))
table("by wave slot in the SIMD (HW_ID.wave_id)", wave_id)
table("by SIMD", simd)
table("by XCC", xcc)
table("by steps in the run", steps.astype(int))
order = np.argsort(us(t_entry := t_in))
# age rank within (xcc, se, sh, cu, simd)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
rank = np.zeros(len(blk), int)
for k in np.unique(key):
    idx = np.where(key == k)[0]
    rank[idx[np.argsort(t_in[idx], kind="stable")]] = np.arange(len(idx))
table("by age rank within the SIMD (0 = first to enter)", rank)
print("waves per SIMD: ", np.bincount(np.bincount(key)))
