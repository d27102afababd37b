#!/usr/bin/env python3
"""How much of the grouped AHC's time is lockstep waste: a wave's four frames iterate until the slowest is done.  Per-frame merge
counts (meta[0] - blocks) of the bench's 256 distinct frames -> sum over waves of the maximum of four consecutive frames against the
plain sum / 4 and against groups of four frames sorted by count.     python tools/ahc_lockstep_waste.py"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
g1, d1 = synth.make_batch("std", 0x5EED1000, 192, 640, 480); g2, d2 = synth.make_batch("lowtex", 0x5EED2000, 64, 640, 480)
d = np.concatenate([d1, d2]); g = np.concatenate([g1, g2])
ctx = hvo.Context(max_batch=256); ctx.batch_upload(g, d); ctx.batch_run(hvo.STAGE_PLANES)
L = hvo.lib(); L.hvo_debug_peac_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
it = []
for f in range(256):
    out = (ctypes.c_int * 16)(); L.hvo_debug_peac_stats(ctx.h, f, out); it.append(out[0] - 3072 + out[2])      # merges + extracted planes ~ live pops
it = np.array(it, float)
lock = it.reshape(-1, 4).max(1).sum(); ideal = it.sum() / 4; srt = np.sort(it).reshape(-1, 4).max(1).sum()
print("merges per frame: min %.0f median %.0f max %.0f" % (it.min(), np.median(it), it.max()))
print("wave iterations, 4 consecutive frames per wave: %.0f | no waste: %.0f (%.1f %% less) | frames sorted by count: %.0f (%.1f %% less)"
      % (lock, ideal, 100 * (1 - ideal / lock), srt, 100 * (1 - srt / lock)))
ctx.close()
