#!/usr/bin/env python3
"""Turns the raw output of tools/refresh_profiles.sh (merged back into gpurun_out/prof) into the committed
profiles/rNN_* summaries.     python tools/collect_profiles.py gpurun_out/prof r01"""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
py = sys.executable


def last_json_line(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def find(pattern):
    g = glob.glob(os.path.join(src, "**", pattern), recursive=True)
    if not g:
        raise SystemExit("missing " + pattern)
    return g[0]


d = last_json_line(os.path.join(src, "bench_default.json"))
B = d["config"]["frames_per_gpu"]
json.dump(d, open(os.path.join(P, tag + "_bench_default.json"), "w")); open(os.path.join(P, tag + "_bench_default.json"), "a").write("\n")
u = last_json_line(os.path.join(src, "bench_under_rocprof.json"))
json.dump(u, open(os.path.join(P, tag + "_bench_default_under_rocprof.json"), "w")); open(os.path.join(P, tag + "_bench_default_under_rocprof.json"), "a").write("\n")
shutil.copy(find("kt_kernel_stats.csv"), os.path.join(P, tag + "_bench_default_kernel_stats.csv"))

# per-kernel durations from the trace: n, mean, min (min = the serialised profiling steps, free of contention)
acc = defaultdict(list)
for r in csv.DictReader(open(find("kt_kernel_trace.csv"))):
    acc[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(P, tag + "_kernel_durations_ms.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of `python bench.py --steps 5 --warmup 2` (default: %d frames of 640x480 per step).\n" % B)
    f.write("# 7 steps with ORB || LSD || PEAC on concurrent streams (durations inflated by contention) followed by the 3\n")
    f.write("# serialised profiling steps; `min` is therefore the contention-free duration bench.py reports per kernel group.\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if k.startswith("__amd"): continue
        f.write("%-40s n=%3d mean=%9.3f min=%9.3f\n" % (k[:40], len(v), sum(v) / len(v), min(v)))


def summary(dirname):
    return subprocess.check_output([py, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, dirname), str(B)], text=True)


with open(os.path.join(P, tag + "_pmc_sq_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_* pass of `python bench.py --steps 1 --warmup 0` (%d frames per launch); tools/pmc_summary.py, largest-grid launch of each kernel\n" % B)
    f.write(summary("pmc_sq"))
with open(os.path.join(P, tag + "_pmc_hbm_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE pass (KiB per frame)\n" + summary("pmc_fetch"))
    f.write("# rocprofv3 --pmc WRITE_SIZE pass (KiB per frame)\n" + summary("pmc_write"))
subprocess.check_call([py, os.path.join(ROOT, "tools", "make_traffic_json.py"), os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"), str(B),
                       os.path.join(P, tag + "_hbm_traffic.json")])
subprocess.check_call([py, os.path.join(ROOT, "tools", "make_sq_json.py"), os.path.join(src, "pmc_sq"), str(B), os.path.join(P, tag + "_sq_utilisation.json")])
with open(os.path.join(P, tag + "_pcie_note.txt"), "w") as f:
    f.write("# tools/pcie_overlap.py BATCH CONTEXTS ROUNDS: upload + run + download of consecutive batches, C contexts on C host threads, one lock per\n"
            "# stage (last line: LOCKS=0, threads not coordinated); pinned host images and label slabs, int8 labels; leg_ms = upload / run / download\n")
    f.write(open(os.path.join(src, "pcie_note.txt")).read())
shutil.copy(os.path.join(src, "hbm_footprint.txt"), os.path.join(P, tag + "_hbm_footprint.txt"))
for name in ("bench_stream", "bench_big1280", "bench_batch256"):
    d2 = last_json_line(os.path.join(src, name + ".json"))
    json.dump(d2, open(os.path.join(P, tag + "_" + name + ".json"), "w")); open(os.path.join(P, tag + "_" + name + ".json"), "a").write("\n")
with open(os.path.join(P, tag + "_latency_small_batches.json"), "w") as f:
    json.dump({k: json.load(open(os.path.join(src, "latency_%s.json" % k))) for k in ("std", "lowtex")}, f, indent=1)
with open(os.path.join(P, tag + "_stream_scaling.txt"), "w") as f:
    f.write("# tools/stream_scaling.py: sustained frames/s of hvo_stream_* (no matching, labels not copied out) against the ring depth (depth - 1 frames in flight)\n")
    for k in ("q4", "torch_runtime", "q16"):
        pth = os.path.join(src, "stream_scaling_%s.json" % k)
        if os.path.exists(pth):
            j = json.load(open(pth))
            f.write("%-14s GPU_MAX_HW_QUEUES=%s torch_runtime=%s : %s\n" % (k, j.get("GPU_MAX_HW_QUEUES"), j.get("torch_runtime"), {a: b["fps"] for a, b in j.items() if a.startswith("depth")}))
    f.write("# 3 streams per frame (points / lines / planes).  Variants measured once (gpurun_out, 100 frames): lines behind ORB on one stream (2 streams per frame)\n"
            "# on the ROCm 7.2 runtime: 20.8 / 41.0 / 41.2 / 60.9 frames/s at depth 2 / 3 / 4 / 6 (the two streams serialise); GPU_MAX_HW_QUEUES=8: 95 frames/s at depth 4\n"
            "# but 3.6-7.6 frames/s at depth 6 (hardware queues oversubscribed).\n")
shutil.copy(os.path.join(src, "match_rate.txt"), os.path.join(P, tag + "_match_rate.txt"))
shutil.copy(os.path.join(src, "peac_timing_batch.txt"), os.path.join(P, tag + "_peac_cluster_phases_batch8192.txt"))
print("value", d["value"], "frames/s; under rocprof", u["value"])
for a, b in (("peac_heads_timing.txt", "_peac_cluster_heads_phases_1frame.txt"), ("lsd_stats.txt", "_lsd_grow_one_frame_stats.txt")):
    if os.path.exists(os.path.join(src, a)): shutil.copy(os.path.join(src, a), os.path.join(P, tag + b))
if os.path.exists(os.path.join(src, "latency_1280.json")):
    json.dump(json.load(open(os.path.join(src, "latency_1280.json"))), open(os.path.join(P, tag + "_latency_1280x960.json"), "w"), indent=1)
g1 = glob.glob(os.path.join(src, "kt1", "**", "one_kernel_stats.csv"), recursive=True)
if g1: shutil.copy(g1[0], os.path.join(P, tag + "_one_frame_kernel_stats.csv"))
