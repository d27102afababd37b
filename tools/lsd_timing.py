#!/usr/bin/env python3
"""Diagnostics: region-growing rounds and where their time goes (builds a -DHVO_LSD_TIMING libhvo.so in a temp dir)."""
import ctypes, importlib, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

tmp = tempfile.mkdtemp(prefix="hvo_timing_")
dst = os.path.join(tmp, "pkg", "csrc")
os.makedirs(os.path.join(tmp, "include"))
shutil.copytree(os.path.join(ge.PKG_DIR, "csrc"), dst, ignore=shutil.ignore_patterns("*.o", "*.so"))
for f in os.listdir(os.path.join(ROOT, "include")):
    shutil.copy(os.path.join(ROOT, "include", f), os.path.join(tmp, "include", f))
subprocess.check_call(["make", "-s", "-j8", "-C", dst, "DEFS=-DHVO_LSD_TIMING"])
hvo = ge.package(); hvo._LIBPATH = os.path.join(dst, "libhvo.so")
synth = importlib.import_module("hvo_amd.synth")
ctx = hvo.Context()
L = hvo.lib(); L.hvo_debug_lsd_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
for kind, seed in (("std", 0x5EED0002), ("std", 0x5EED1001)):
    g = synth.make_gray(kind, seed)
    ctx.extract_lsd(g); ctx.extract_lsd(g)
    out = (ctypes.c_longlong * 8)(); L.hvo_debug_lsd_stats(ctx.h, 0, out); s = list(out)
    print("%s: seeds %d points %d rounds %d (%.2f points/round) | gather %.0f ticks/round, add loop %.0f ticks/round = %.0f ticks/point | grow total %.2f ms"
          % (kind, s[0], s[1], s[2], s[1] / max(s[2], 1), s[4] / max(s[2], 1), s[5] / max(s[2], 1), s[5] / max(s[1], 1), s[3] / 1e5))
ctx.close(); shutil.rmtree(tmp, ignore_errors=True)
