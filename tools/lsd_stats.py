#!/usr/bin/env python3
"""Diagnostics: what a lone frame's k_lsd_grow spends its time on (the product library's own per-frame statistics: seeds, region points,
regions >= min_reg, 100 MHz wall-clock ticks in region_grow / region2rect / refine / whole kernel)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
ctx = hvo.Context()
L = hvo.lib(); L.hvo_debug_lsd_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
W = int(sys.argv[1]) if len(sys.argv) > 2 else 640; H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
for kind, seed in (("std", 0x5EED0002), ("std", 0x5EED1001), ("lowtex", 0x5EED2000)):
    g = synth.make_batch(kind, seed, 1, W, H)[0][0]
    ctx.extract_lsd(g); ctx.extract_lsd(g)
    out = (ctypes.c_longlong * 8)(); L.hvo_debug_lsd_stats(ctx.h, 0, out); s = list(out)
    print("%s: seeds %d, region points %d, regions >= min_reg %d, segments %d | grow %.2f ms, region2rect %.2f ms, refine %.2f ms, kernel %.2f ms"
          % (kind, s[0], s[1], s[2], s[7], s[3] / 1e5, s[4] / 1e5, s[5] / 1e5, s[6] / 1e5))
ctx.close()
