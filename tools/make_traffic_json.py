#!/usr/bin/env python3
"""Builds profiles/rNN_hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
    python tools/make_traffic_json.py DIR_FETCH DIR_WRITE FRAMES_PER_LAUNCH OUT.json"""
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

GROUPS = {"peac_cluster": ["k_peac_cluster", "k_peac_cluster_slots", "k_peac_cluster_heads", "k_peac_edges"], "lsd_grow": ["k_lsd_grow", "k_lsd_grow_dense"], "peac_refine": ["k_peac_blkmap", "k_peac_flood", "k_peac_final", "k_peac_relabel"],
          "orb_levels": ["k_orb_level"], "orb_fast_cells": ["k_fast_cells"], "lsd_gradient": ["k_lsd_resize_grad"], "lsd_pre": ["k_lsd_pre"], "lbd_desc": ["k_lbd_desc"], "orb_pyramid": ["k_resize", "k_resize_dw"],
          "orb_blur": ["k_blur7"], "lsd_blur_scale": ["k_lsd_blur"], "lbd_sobel": ["k_lbd_blur5", "k_lbd_sobel", "k_lbd_blur_sobel"],
          "orb_octree": ["k_octree"], "peac_blocks": ["k_peac_blocks"], "orb_orient": ["k_moments", "k_kpfinish"], "orb_brief": ["k_brief"]}
# kernels whose reads are 16 bytes per lane: gfx950's FETCH_SIZE tallies their 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM): doubled
WIDE_READS = {"k_orb_level", "k_moments", "k_brief"}


def per_step(d, ctr):
    """sum of the counter over all dispatches of a kernel, divided by the number of steps (= dispatches of k_peac_blocks)"""
    acc = defaultdict(float); steps = 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            acc[k] += float(r["Counter_Value"])
            if k == "k_peac_blocks": steps += 1
    return {k: v / max(steps, 1) for k, v in acc.items()}


def main():
    dfe, dwr, B, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe, wr = per_step(dfe, "FETCH_SIZE"), per_step(dwr, "WRITE_SIZE")
    j = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 0`; KiB per step "
                 "summed over a group's kernels / frames per launch, x1024.  Calibration on kernels with a known byte count and the same access "
                 "width: k_blur7 (dword loads/stores, 950532 B each way + halo/pitch) reads 1:1, k_lsd_blur writes 2400 KiB fp64 -> WRITE 2400 KiB; "
                 "the x2 FETCH_SIZE correction of MI355X_MICROARCH.md applies to 16 B/lane reads: applied to k_orb_level, k_moments, k_brief (tile rows / patch rows as dwordx4), the other kernels use <= 8 B/lane.",
         "frames_per_launch": B, "sources_sha16": sources_sha16(), "bytes_per_frame": {}}
    for g, ks in GROUPS.items():
        j["bytes_per_frame"][g] = {"fetch": round(sum(fe.get(k, 0) * (2 if k in WIDE_READS else 1) for k in ks) / B * 1024), "write": round(sum(wr.get(k, 0) for k in ks) / B * 1024)}
    json.dump(j, open(out, "w"), indent=1)
    tot = sum(v["fetch"] + v["write"] for v in j["bytes_per_frame"].values())
    print("total %.1f MB per frame" % (tot / 1e6)); print(json.dumps(j["bytes_per_frame"]))


if __name__ == "__main__":
    main()
