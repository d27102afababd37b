#!/usr/bin/env python3
"""Where k_lsd_grow's time goes in a resident batch: per-frame wall-clock ticks (100 MHz) of region growing, region2rect,
refine and everything else (seed scan, key lines, top-N), from the counters the product kernel keeps (hvo_debug_lsd_stats).
    python tools/lsd_phase_split.py [BATCH]"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g, d = synth.make_batch("std", 0x5EED1000, 48, 640, 480); g2, d2 = synth.make_batch("lowtex", 0x5EED2000, 16, 640, 480)
g = np.concatenate([g, g2]); d = np.concatenate([d, d2])
ctx = hvo.Context(max_batch=B); ctx.batch_upload(g, d, repeat=B // 64)
ctx.profile_enable(2)
for _ in range(2): ctx.batch_run(hvo.STAGE_LSD)
print({k: round(v, 2) for k, v in ctx.profile_last().items() if v > 0})
L = hvo.lib(); L.hvo_debug_lsd_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
acc = np.zeros(8)
for f in range(64):
    out = (ctypes.c_longlong * 8)(); L.hvo_debug_lsd_stats(ctx.h, f, out); acc += np.array(list(out), float)
acc /= 64
per = []
for f in range(64):
    out = (ctypes.c_longlong * 8)(); L.hvo_debug_lsd_stats(ctx.h, f, out); per.append((out[6] / 1e5, out[0], out[1]))
per.sort()
print("wave lifetime of the 64 distinct frames, ms (seeds, points): shortest", per[:3], "median", per[32], "longest", per[-4:])
tot = acc[6]
print("per frame: seeds %.0f points %.0f regions kept %.0f segments %.0f" % (acc[0], acc[1], acc[2], acc[7]))
print("ticks (100 MHz): total %.0f = %.2f ms | grow %.1f %% | region2rect %.1f %% | refine %.1f %% | rest (seed scan, key lines, top-N, line functions) %.1f %%"
      % (tot, tot / 1e5, 100 * acc[3] / tot, 100 * acc[4] / tot, 100 * acc[5] / tot, 100 * (tot - acc[3] - acc[4] - acc[5]) / tot))
ctx.close()
