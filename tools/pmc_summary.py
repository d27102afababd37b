#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per dispatch
(largest-grid dispatches only, i.e. the batch launches) and mean duration.
    python tools/pmc_summary.py DIR [frames_per_launch]"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    d = sys.argv[1]
    frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"]),
                                              int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    names = sorted({c for k in acc for c in acc[k]})
    print("per FRAME values (counter / %g frames per launch); dur = ms per launch under PMC" % frames)
    print("%-28s %5s %9s " % ("kernel", "n", "dur_ms") + " ".join("%14s" % c[-14:] for c in names))
    rows = []
    for k, cs in acc.items():
        if k.startswith("__amd"): continue
        any_c = next(iter(cs.values()))
        gmax = max(g for g, _, _ in any_c)
        line = []
        n = 0; dur = 0
        for c in names:
            v = [x for g, x, _ in cs.get(c, []) if g == gmax]
            t = [x for g, _, x in cs.get(c, []) if g == gmax]
            n = len(v); dur = sum(t) / max(len(t), 1) / 1e6
            line.append(sum(v) / max(len(v), 1) / frames)
        rows.append((dur, k, n, line))
    for dur, k, n, line in sorted(rows, reverse=True):
        print("%-28s %5d %9.3f " % (k[:28], n, dur) + " ".join("%14.4g" % x for x in line))

if __name__ == "__main__":
    main()
