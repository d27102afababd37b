#!/usr/bin/env python3
"""Repeats the async line growing on 32 low-texture frames that exposed its rare duplicate-pixel race (see tests/test_lsd_gpu.py::test_lines_async_stolen_tags_stress):
    python tools/lsd_async_stress.py [reps=40] [stage mask=7] [crop HxW=479x638] [frames per batch=16]   (HVO_LSD_ASYNC=W forces the workers per frame)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
hvo = ge.package(); orc = ge.oracle(); synth = importlib.import_module("hvo_amd.synth")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 7
crop = (sys.argv[3] if len(sys.argv) > 3 else "479x638"); ch, cw = (int(v) for v in crop.split("x"))
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 16
chunks = []
for c0 in (16, 112):
    gray, depth = synth.make_batch("lowtex", 0xF00F4000 + c0, 16, 640, 480)
    gray = np.ascontiguousarray(gray[:, :ch, :cw]); depth = np.ascontiguousarray(depth[:, :ch, :cw])
    if NB < 16: gray = gray[16 - NB:]; depth = depth[16 - NB:]; c0 += 16 - NB
    chunks.append((c0, gray, depth, [orc.line_extract(g) for g in gray]))
ctx = hvo.Context(max_batch=NB)
nbad = 0
for rep in range(reps):
    for c0, gray, depth, ref in chunks:
        ctx.batch_upload(gray, depth); ctx.batch_run(mask); res = ctx.batch_download(hvo.STAGE_LSD)
        for b in range(NB):
            kl_o, d_o, fn_o = ref[b]; r = res[b]
            if len(r["kl"]) != len(kl_o) or not np.array_equal(r["ldesc"], d_o) or not np.array_equal(r["kl"]["num_pixels"], kl_o["num_pixels"]):
                nbad += 1
                same = len(r["kl"]) == len(kl_o)
                rows = np.flatnonzero((r["ldesc"] != d_o).any(axis=1)).tolist()[:6] if same else None
                npx = np.flatnonzero(r["kl"]["num_pixels"] != kl_o["num_pixels"]).tolist()[:6] if same else None
                so = {(float(a), float(bb), float(c), float(dd)) for a, bb, c, dd in zip(kl_o["sx"], kl_o["sy"], kl_o["ex"], kl_o["ey"])}
                sg = {(float(a), float(bb), float(c), float(dd)) for a, bb, c, dd in zip(r["kl"]["sx"], r["kl"]["sy"], r["kl"]["ex"], r["kl"]["ey"])}
                print("   only gpu", sorted(sg - so), "only oracle", sorted(so - sg))
                print("rep", rep, "frame", c0 + b, "status", r["status"], "n", len(r["kl"]), len(kl_o), "desc rows", rows, "npix rows", npx, flush=True)
print("env", {k: v for k, v in os.environ.items() if k.startswith("HVO_")}, "mask", mask, "reps", reps, "bad", nbad, flush=True)
ctx.close()
