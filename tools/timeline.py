#!/usr/bin/env python3
"""Kernel timeline of one overlapped step from a rocprofv3 --kernel-trace CSV directory.
    python tools/timeline.py DIR [step_index]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))
        if not r["Kernel_Name"].startswith("__amd")]
rows.sort()
starts = [s for s, e, k in rows if k == "k_peac_blocks"]
t0 = starts[step]; t1 = starts[step + 1] if len(starts) > step + 1 else None
for s, e, k in rows:
    if s >= t0 - 2e6 and (t1 is None or s < t1 - 2e6):
        print("%8.2f -> %8.2f  (%7.2f ms)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k))
