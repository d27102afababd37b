#!/usr/bin/env python3
"""Per-frame kernel time of ONE subsystem as a function of the resident batch.

A stage run alone allocates only its own plan, so far more frames fit than in the full front-end (25 MB per frame).  The
question this answers: would the serial-semantics kernels (AHC, flood, LSD growing) be faster per frame with more waves
resident, i.e. does a smaller per-frame footprint buy speed or only capacity?
    python tools/stage_batch_sweep.py planes 8192 16384 24576
"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
stage = {"planes": hvo.STAGE_PLANES, "lsd": hvo.STAGE_LSD, "orb": hvo.STAGE_ORB}[sys.argv[1]]
g, d = synth.make_batch("std", 0x5EED1000, 48, 640, 480)
g2, d2 = synth.make_batch("lowtex", 0x5EED2000, 16, 640, 480)
import numpy as np
g = np.concatenate([g, g2]); d = np.concatenate([d, d2])
for B in [int(x) for x in sys.argv[2:]]:
    ctx = hvo.Context(max_batch=B)
    ctx.batch_upload(g, d, repeat=B // 64)
    ctx.profile_enable(2)
    best = None
    for it in range(3):
        ctx.batch_run(stage)
        t = ctx.profile_last()
        if best is None or sum(t.values()) < sum(best.values()): best = t
    print("B=%6d" % B, " ".join("%s %.2f (%.2f us/frame)" % (k, v, v * 1e3 / B) for k, v in best.items() if v > 0), flush=True)
    ctx.close()
