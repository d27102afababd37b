#!/usr/bin/env python3
"""Diagnostics: where a round of k_peac_cluster_heads goes (clock64 ticks of wave 0 per phase; HVO_PEAC_TIMING build in a temp dir)."""
import ctypes, importlib, os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
NAMES = ["compaction check", "A heads", "B evaluate", "B edit loads + marks", "barrier 1", "decisions + queue masks", "(unused)", "D commit || E front merge", "-", "barrier 3"]

def main():
    tmp = tempfile.mkdtemp(prefix="hvo_timing_")
    dst = os.path.join(tmp, "pkg", "csrc"); os.makedirs(os.path.join(tmp, "include"))
    shutil.copytree(os.path.join(ge.PKG_DIR, "csrc"), dst, ignore=shutil.ignore_patterns("*.o", "*.so"))
    for f in os.listdir(os.path.join(ROOT, "include")): shutil.copy(os.path.join(ROOT, "include", f), os.path.join(tmp, "include", f))
    subprocess.check_call(["make", "-s", "-j8", "-C", dst, "DEFS=-DHVO_PEAC_TIMING " + " ".join(sys.argv[1:])])
    hvo = ge.package(); hvo._LIBPATH = os.path.join(dst, "libhvo.so")
    synth = importlib.import_module("hvo_amd.synth")
    g, d = synth.make_batch("std", 0x5EED1000, 1, 640, 480)
    ctx = hvo.Context(max_batch=1); ctx.batch_upload(g, d)
    L = hvo.lib(); out = (ctypes.c_ulonglong * 32)()
    ctx.batch_run(hvo.STAGE_PLANES); L.hvo_debug_peac_timing(out, 1)
    ctx.profile_enable(2); ctx.batch_run(hvo.STAGE_PLANES); print(ctx.profile_last())
    L.hvo_debug_peac_timing(out, 1)
    v = np.array(list(out), dtype=np.float64)
    tot = v[:10].sum()
    print("ticks total %.3e; rounds %d, pops %d (%.2f per round), %.0f ticks per round" % (tot, v[10], v[11], v[11] / max(v[10], 1), tot / max(v[10], 1)))
    for i in range(10): print("  %-26s %6.2f %%  %.0f ticks per round" % (NAMES[i], 100 * v[i] / tot, v[i] / max(v[10], 1)))
    print("queue wave after barrier 2: prefix %.0f, front merge (masks) %.0f, front stores %.0f ticks per round; from a round's start to barrier 1: %.0f (of which deferred updates + extraction %.0f)" % tuple(v[16 + i] / max(v[10], 1) for i in range(5)))
    ctx.close(); shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
