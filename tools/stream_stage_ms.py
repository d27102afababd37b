#!/usr/bin/env python3
"""Diagnostics: per-chain time of ONE frame at a time through hvo_stream_* with the whole Frame constructor (each chain's events:
start of its first kernel to the end of its last, tail stages included)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
g, d, _ = synth.make_sequence("std", 0x5EED2000, 24)
full = hvo.STAGE_ORB | hvo.STAGE_LSD | hvo.STAGE_PLANES | hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS
for name, stages in (("front-end only", hvo.STAGE_ORB | hvo.STAGE_LSD | hvo.STAGE_PLANES), ("whole constructor", full)):
    st = hvo.Stream(depth=2, stages=stages, bf=40.0, seed=1)
    rows = []; wall = []
    for i in range(len(g)):
        t0 = time.perf_counter(); t = st.submit(g[i], d[i]); st.collect(t); wall.append((time.perf_counter() - t0) * 1e3)
        rows.append(st.stage_ms(t))
    st.close()
    med = lambda k: float(np.median([r[k] for r in rows[4:]]))
    print("%-18s orb %.2f  lsd %.2f  planes %.2f ms (medians of the chains' events); submit->collect wall p50 %.2f ms" % (name, med("orb"), med("lsd"), med("planes"), float(np.median(wall[4:]))))
