#!/usr/bin/env python3
"""Builds profiles/rNN_sq_utilisation.json from a rocprofv3 --pmc pass with SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS of bench.py.     python tools/make_sq_json.py DIR FRAMES_PER_LAUNCH OUT.json"""
import csv, glob, json, os, sys
from collections import defaultdict


def sources_sha16():
    """as bench.py: sha256 of csrc/*.hip, *.inc, *.hpp (the sources the counters were measured on; run this right after the GPU pass)"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(root, "a-low-texture-robust-hybrid-feature-based-visual-odometry_amd", "csrc", "*"))):
        if p.endswith((".hip", ".inc", ".hpp")):
            hsh.update(os.path.basename(p).encode()); hsh.update(open(p, "rb").read())
    return hsh.hexdigest()[:16]

d, B, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = defaultdict(lambda: defaultdict(float)); steps = 0
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("__amd"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if k == "k_peac_blocks" and r["Counter_Name"] == "SQ_INSTS_VALU": steps += 1
out = {"note": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS of `python bench.py --steps 1 --warmup 0`; "
               "per-frame sums over all launches of a kernel.  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md): a wave64 "
               "VALU instruction keeps the SIMD's VALU busy for 1 quad-cycle, so sum(ACTIVE_INST_VALU) x frames/s / (simds x clock_hz / 4) is the VALU pipes' busy fraction; "
               "ACTIVE_INST_ANY sums over waves that can have instructions of different kinds in flight at once and is only meaningful relative to WAVE_CYCLES.",
       "frames_per_launch": B, "sources_sha16": sources_sha16(), "simds": 1024, "clock_hz": 2.4e9, "per_frame": {}}
tot = defaultdict(float)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_ACTIVE_INST_ANY"]):
    e = {c.lower(): round(v[c] / steps / B) for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    out["per_frame"][k] = e
    for c, x in e.items(): tot[c] += x
out["per_frame_total"] = dict(tot)
json.dump(out, open(out_path, "w"), indent=1)
print(dict(tot))
