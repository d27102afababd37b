#!/usr/bin/env python3
"""bench.py -- RGB-D front-end throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--batch B] [--stages orb,lsd,planes]

One "step" = one pass of the whole front-end (ORB 1000 features + LSD/LBD lines + PEAC planes)
over one batch of B synthetic 640x480 RGB-D frames that is already resident in HBM
(BASELINE.json configs[1]).  Independent frames shard across ranks with no data-path collective
("weak" scaling: B frames per GPU); the only collectives are the timing barrier / max-reduce.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     HBM roofline of the dominant kernel group, from hipEvents recorded on the
                  kernels' own stream inside the timed region (hvo_profile_last)
  "cpu_baseline": the CPU oracle (kind "port") timed on this host on a bounded sample.
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)


def algorithmic_bytes(group, w, h, nkp, nlines):
    """ALGORITHMIC bytes per FRAME for each kernel group (DESIGN.md section 'Kernels and rooflines').
    S = sum of pyramid level areas."""
    ws, hs = [], []
    s = 1.0
    sc = np.float32(1.0)
    for l in range(8):
        inv = np.float32(1.0) / sc
        ws.append(int(np.rint(np.float32(w) * inv))); hs.append(int(np.rint(np.float32(h) * inv)))
        sc = sc * np.float32(1.2)
    areas = [a * b for a, b in zip(ws, hs)]
    S = sum(areas)
    sw, sh = int(round(w * 0.8)), int(round(h * 0.8))
    table = {
        "orb_pyramid": sum(areas[:-1]) + sum(areas[1:]),          # read levels 0..6, write 1..7
        "orb_fast_cells": S + 4 * 8 * nkp,                        # read every level once, emit candidates
        "orb_octree": 4 * 8 * nkp + 4 * nkp,
        "orb_orient": 749 * nkp + 28 * nkp,
        "orb_blur": 2 * S,                                        # read level, write blurred level
        "orb_brief": 512 * nkp + 60 * nkp,                        # 512 gathers + keypoint + descriptor
        "peac_blocks": 2 * w * h + 3072 * 160,                    # u16 depth in, block records out
        "peac_cluster": 3072 * 160 * 2,
        "peac_refine": 2 * w * h + 4 * w * h,                     # depth re-read + int32 labels out
        "lsd_blur_scale": w * h + sw * sh,
        "lsd_gradient": sw * sh + 12 * sw * sh,
        "lsd_grow": 13 * sw * sh,
        "lbd_sobel": w * h + w * h + 4 * w * h,
        "lbd_desc": 63 * 4 * 60 * nlines + 100 * nlines,
    }
    return table.get(group, 0), 4 * S + 60 * nkp


def hbm_traffic(group, B, width):
    """HBM bytes per launch of a kernel group from the PMC counters (FETCH_SIZE + WRITE_SIZE, collected in
    separate rocprofv3 --pmc passes of this same command and committed as profiles/r01_hbm_traffic.json --
    counters cannot be read from inside the process).  None when the committed measurement does not
    cover this configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            t = json.load(f)
        if t.get("frames_per_launch") != B or width != 640 or group not in t["bytes_per_frame"]:
            return None
        e = t["bytes_per_frame"][group]
        return int((e["fetch"] + e["write"]) * B)
    except (OSError, ValueError, KeyError):
        return None


def sq_utilisation(frames_per_s):
    """What the step spends on the SIMDs, from the committed SQ counters per frame (profiles/r01_sq_utilisation.json;
    SQ_* count quad-cycles summed over waves): the VALU pipes' busy fraction at the measured rate (VALU instructions of
    different waves of a SIMD cannot overlap, so this is a true pipe utilisation), the mean number of resident waves per
    SIMD, and the share of a wave's life with an instruction in flight.  The step is NOT HBM bound; it is a set of
    dependent-latency chains.  Extra keys of the roofline object; empty when the profile is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_sq_utilisation.json")) as f:
            t = json.load(f)
        tot = t["per_frame_total"]
        cap = t["simds"] * t["clock_hz"] / 4.0                     # SIMD quad-cycles per second
        return {"valu_busy_frac": round(tot["sq_active_inst_valu"] * frames_per_s / cap, 4),
                "resident_waves_per_simd": round(tot["sq_wave_cycles"] * frames_per_s / cap, 3),
                "wave_time_with_inst_active": round(tot["sq_active_inst_any"] / tot["sq_wave_cycles"], 4),
                "valu_quad_cycles_per_frame": int(tot["sq_active_inst_valu"])}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        return {}


def cpu_baseline(stages, gray, depth, budget_s=12.0):
    """time the CPU oracle (scalar port, 1 thread) on a bounded sample of the same workload"""
    orc = ge.oracle()
    orb = orc.Orb()
    n, t0 = 0, time.time()
    while True:
        g = gray[n % len(gray)]
        if "orb" in stages:
            orb.extract(g)
        if "planes" in stages:
            orc.peac(depth[n % len(depth)])
        if "lsd" in stages:
            orc.line_extract(g)
        n += 1
        el = time.time() - t0
        if el > budget_s or n >= 64:
            break
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same synthetic 640x480 workload (%s), oracle/liboracle.so, 1 thread, host has %d cores"
                      % (n, "+".join(stages), os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0,
                    help="frames resident per GPU per step (multiple of 16); 0 = the largest of 8192/4096/2048/1024 that fits in 85 %% of the free HBM")
    ap.add_argument("--stages", default="orb,lsd,planes")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (rehearsal of the N>1 path on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"     # where the two scalar reductions live
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)

    ge.build() if not os.path.exists(os.path.join(ge.PKG_DIR, "csrc", "libhvo.so")) else None
    hvo = ge.package()
    synth = importlib.import_module("hvo_amd.synth")
    stages = [s for s in args.stages.split(",") if s]
    mask = 0
    for s in stages:
        mask |= {"orb": hvo.STAGE_ORB, "lsd": hvo.STAGE_LSD, "planes": hvo.STAGE_PLANES}[s]

    B = args.batch
    if B <= 0:
        # 28.5 MB per resident 640x480 frame (profiles/r01_hbm_footprint.txt); more frames in flight = more latency hiding
        free_b, _ = torch.cuda.mem_get_info()
        per_frame = 28.5e6 * (args.width * args.height) / (640.0 * 480.0)
        B = next((c for c in (8192, 4096, 2048, 1024) if c * per_frame <= 0.85 * free_b), 512)
        if dist is not None:                 # every rank must run the same workload
            t = torch.tensor([B], dtype=torch.int64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            B = int(t.item())
    ndistinct = min(B, 16)
    g0, d0 = synth.make_batch("std", 0x5EED1000 + 1000 * rank, ndistinct, args.width, args.height)
    reps = max(1, B // ndistinct)
    B = reps * ndistinct                    # the 16 distinct frames are uploaded cyclically, by pointer

    s = args.width / 640.0
    ctx = hvo.Context(max_batch=B, device=local_rank, orb_nfeatures=1000 if args.width <= 640 else 2000,
                      fx=535.4 * s, fy=539.2 * s, cx=320.1 * s, cy=247.6 * s)
    ctx.batch_upload(g0, d0, repeat=reps)   # inputs resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.batch_run(mask)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.batch_run(mask)                 # enqueues every kernel (ORB || LSD || PEAC streams) and waits
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Per-kernel durations for the roofline: the same steps again with hipEvents on the kernels' stream
    # and the three subsystems serialised, so that a group's time is its own (with the streams
    # overlapped a group's event interval mostly measures waiting for CU slots held by the others).
    prof = {}
    psteps = max(1, min(args.steps, 3))
    ctx.profile_enable(2)
    for _ in range(psteps):
        ctx.batch_run(mask)
        for k, v in ctx.profile_last().items():
            prof[k] = prof.get(k, 0.0) + v
    ctx.profile_enable(0)

    res = ctx.batch_download(mask, n=64)     # a sample is enough for the workload statistics
    nkp = float(np.mean([len(r["kp"]) for r in res])) if "orb" in stages else 0.0
    nlines = float(np.mean([len(r["kl"]) for r in res])) if "lsd" in stages else 0.0
    nplanes = float(np.mean([len(r["planes"]) for r in res])) if "planes" in stages else 0.0
    bad = sum(1 for r in res if r["status"] != 0)

    if rank == 0:
        frames = world * B * args.steps
        value = frames / dt
        groups = {k: v / psteps for k, v in prof.items()}           # ms per launch group per step (serialised pass)
        dom = max(groups, key=groups.get) if groups else None
        roof = None
        if dom:
            per_frame, pass_bytes = algorithmic_bytes(dom, args.width, args.height, nkp, nlines)
            ach = per_frame * B / (groups[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": hbm_traffic(dom, B, args.width),
                    "bytes_per_launch": int(per_frame * B), "ms_per_launch": round(groups[dom], 4)}
            roof.update(sq_utilisation(value))
            orb_ms = sum(v for k, v in groups.items() if k in ("orb_pyramid", "orb_fast_cells", "orb_blur", "orb_brief", "orb_orient"))
            if orb_ms > 0:
                roof["orb_pyramid_brief_pass_GBps"] = round(pass_bytes * B / (orb_ms * 1e-3) / 1e9, 2)
        out = {
            "metric": "RGB-D frames/sec (640\u00d7480, 1k ORB + LSD + PEAC) at 1/2/4/8 GPUs", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u16 integer + f32/f64",
            "data": "synthetic",
            "config": {"workload": "%dx%d synthetic RGB-D, %d ORB + LSD lines + PEAC planes, %d frames per GPU per step"
                                   % (args.width, args.height, ctx.params.orb_nfeatures, B),
                       "stages": stages, "frames_per_gpu": B, "parallelism": "frames sharded, %d rank(s), no data-path collective" % world,
                       "mean_keypoints": round(nkp, 1), "mean_lines": round(nlines, 1), "mean_planes": round(nplanes, 2),
                       "frames_with_capacity_flags": bad},
            "kernel_ms_per_step_serialised": {k: round(v, 4) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
            "roofline": roof,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(stages, g0, d0)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
