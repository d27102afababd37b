#!/usr/bin/env python3
"""Diagnostics: where k_peac_cluster spends its time (clock64 ticks per AHC phase).

Builds a -DHVO_PEAC_TIMING variant of libhvo.so in a temp directory (the in-tree product library is not
touched), runs the plane stage on a resident batch and prints the phase breakdown.
    python tools/peac_timing.py [--batch 4096]
"""
import argparse, ctypes, importlib, os, shutil, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

NAMES = ["round boundary: sync + next top's list", "stage 0 (partner list, queue lines) + queue update", "eval: loads", "eval: reduce+bcast", "eval: eigen-solve (+decision)", "stages 1-3 (member headers, lists, edits)", "list tails (entries beyond the first pass)", "record write",
         "#iters", "#eval passes", "#merge iters", "#nomerge iters", "init edges+lists", "heapify", "#waves"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--defs", default="")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="hvo_timing_")
    dst = os.path.join(tmp, "pkg", "csrc")
    os.makedirs(os.path.join(tmp, "include"))
    shutil.copytree(os.path.join(ge.PKG_DIR, "csrc"), dst, ignore=shutil.ignore_patterns("*.o", "*.so"))
    for f in os.listdir(os.path.join(ROOT, "include")):
        shutil.copy(os.path.join(ROOT, "include", f), os.path.join(tmp, "include", f))
    t0 = time.time()
    subprocess.check_call(["make", "-s", "-j8", "-C", dst, "DEFS=-DHVO_PEAC_TIMING " + args.defs])
    print("timing build: %.0f s" % (time.time() - t0), flush=True)
    hvo = ge.package()
    hvo._LIBPATH = os.path.join(dst, "libhvo.so")
    synth = importlib.import_module("hvo_amd.synth")
    g, d = synth.make_batch("std", 0x5EED1000, 16, 640, 480)
    reps = max(1, args.batch // 16)
    ctx = hvo.Context(max_batch=reps * 16)
    ctx.batch_upload(g, d, repeat=reps)
    L = hvo.lib()
    out = (ctypes.c_ulonglong * 32)()
    ctx.batch_run(hvo.STAGE_PLANES)
    L.hvo_debug_peac_timing(out, 1)
    ctx.profile_enable(2)
    ctx.batch_run(hvo.STAGE_PLANES)
    print(ctx.profile_last())
    L.hvo_debug_peac_timing(out, 1)
    v = np.array(list(out), dtype=np.float64)
    nw = max(v[14], 1)
    tot = v[:8].sum() + v[12] + v[13]
    print("waves %d, ticks per wave %.3e" % (nw, tot / nw))
    for i in list(range(8)) + [12, 13]:
        print("  %-18s %6.2f %%   %.3e ticks/wave" % (NAMES[i], 100 * v[i] / tot, v[i] / nw))
    for i in range(8, 12):
        print("  %-18s %.1f per wave" % (NAMES[i], v[i] / nw))
    nf = max(v[23], 1)
    ftot = v[16:20].sum()  # (election time excludes the sub-phases listed below)
    print("flood: frames %d, ticks per frame %.3e, queue entries per frame %.0f, rounds %.1f, rounds with ranked two-plane groups %.2f, rounds replayed serially %.2f" % (nf, ftot / nf, v[22] / nf, v[20] / nf, v[24] / nf, v[21] / nf))
    for i, nm in enumerate(["seeds", "state fetch + compaction", "distance + grouping", "apply + ordered append"]):
        print("  %-26s %6.2f %%" % (nm, 100 * v[16 + i] / ftot))
    print("  events per frame: %.0f (4 per queue entry), live (compacted) %.0f, passes per round %.2f"
          % (4 * v[22] / nf, v[30] / nf, v[31] / max(v[20], 1)))
    ctx.close()
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
