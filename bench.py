#!/usr/bin/env python3
"""bench.py -- RGB-D front-end throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config std640|big1280|batch256] [--mode batch|stream]

mode batch (default): one "step" = one pass of the whole front-end (ORB 1000 features + LSD/LBD lines + PEAC planes)
over one batch of B synthetic RGB-D frames that is already resident in HBM (BASELINE.json configs[1]; 256 distinct
frames, 3/4 `std` + 1/4 `lowtex`, uploaded cyclically).  Independent frames shard across ranks with no data-path
collective ("weak" scaling: B frames per GPU); at N > 1 the result slabs are all-gathered once after the timed region
(device to device, reported as gather_ms).
mode stream (BASELINE.json configs[4]): one frame at a time through hvo_stream_* (pinned staging, `--depth` frames in
flight, SearchByProjection(Cur, Last) + line matching against the previous frame on the device); a "step" is a frame.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":       HBM roofline of the dominant kernel group, from hipEvents recorded on the kernels' own stream
                    (hvo_profile_last); "kernel_roofline" has the same figure for every kernel group
  "cpu_baseline":   the CPU oracle (kind "port") in the reference's own shape: one frame at a time, ORB || LSD || planes on
                    three threads (src/Frame.cc:210-215); "cpu_baseline_all_cores": frames spread over every host core
  "latency_ms", "pcie_inclusive_frames_per_s": single-frame / small-batch latency and the rate with upload and download.
`--gpus N` without a torchrun environment starts the N ranks itself (before anything touches the GPU).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
TRAFFIC_PROFILE = "profiles/r05_hbm_traffic.json"
SQ_PROFILE = "profiles/r05_sq_utilisation.json"
VALU_ISSUE_PROFILE = "profiles/r05_valu_salu_issue.txt"
VALU_ISSUE_NS = 1.77           # ns per wave64 instruction per SIMD, the integer / fp64 classes at 8 waves per SIMD (that profile)
BYTES_PER_FRAME_640 = 21.1e6      # resident footprint of one 640x480 frame incl. its share of the chunk scratch (profiles/r05_hbm_footprint.txt)
BYTES_PER_FRAME_1280 = 80.0e6     # 1280x960, 2000 ORB: 63 MB per frame + 32 GB of chunk scratch


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 20 batches, or 573 frames in stream mode)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="batch", choices=["batch", "stream"])
    ap.add_argument("--config", default="std640", choices=["std640", "big1280", "batch256"],
                    help="std640: BASELINE configs[1]; big1280: configs[2] (1280x960, 2000 ORB); batch256: configs[3] (256 frames over the ranks)")
    ap.add_argument("--batch", type=int, default=0, help="frames resident per GPU per step; 0 = per config (std640: the largest of 8192/4096/... that fits)")
    ap.add_argument("--stages", default="", help="comma list of orb,lsd,planes,tail (tail = isLineGood + vanishing points + ComputePlanes' tail + grids); default: orb,lsd,planes (batch), orb,lsd,planes,tail (stream: the whole Frame constructor)")
    ap.add_argument("--distinct", type=int, default=256, help="distinct synthetic frames (3/4 std, 1/4 lowtex)")
    ap.add_argument("--depth", type=int, default=4, help="stream mode: frames in flight")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency / PCIe-inclusive side measurements")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (rehearsal of the N>1 path on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child torchrun job (nothing has touched the GPU in
    this process) and relay rank 0's JSON line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            print(line)
    sys.exit(p.returncode)


# ---------------------------------------------------------------------------------------------------------------
# algorithmic bytes (SURVEY.md 8d for the pyramid+BRIEF pass)
# ---------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(w, h, nkp, nlines):
    import numpy as np
    ws, hs = [], []
    sc = np.float32(1.0)
    for _ in range(8):
        inv = np.float32(1.0) / sc
        ws.append(int(np.rint(np.float32(w) * inv))); hs.append(int(np.rint(np.float32(h) * inv)))
        sc = sc * np.float32(1.2)
    areas = [a * b for a, b in zip(ws, hs)]
    S = sum(areas)
    sw, sh = int(round(w * 0.8)), int(round(h * 0.8))
    nblk = (w // 10) * (h // 10)
    table = {
        "orb_pyramid": sum(areas[:-1]) + sum(areas[1:]),          # read levels 0..6, write 1..7
        "orb_fast_cells": S + 4 * 8 * nkp,                        # read every level once, emit candidates
        "orb_levels": sum(areas[:-1]) + sum(areas[1:]) + S + 4 * 8 * nkp + 2 * S,   # the fused pass: pyramid + FAST + blur of the same levels
        "orb_octree": 4 * 8 * nkp + 4 * nkp,
        "orb_orient": 749 * nkp + 28 * nkp,
        "orb_blur": 2 * S,                                        # read level, write blurred level
        "orb_brief": 512 * nkp + 60 * nkp,                        # 512 gathers + keypoint + descriptor
        "peac_blocks": 2 * w * h + nblk * 160,                    # u16 depth in, block records out
        "peac_cluster": nblk * 160 * 2,
        "peac_refine": 2 * w * h + 4 * w * h,                     # depth re-read + labels out (2WH + 4WH)
        "lsd_blur_scale": w * h + sw * sh,
        "lsd_gradient": sw * sh + 12 * sw * sh,                   # the unfused second kernel (HVO_LSD_PRE_SPLIT=1) ...
        "lsd_pre": w * h + 12 * sw * sh,                          # ... and the one kernel from the u8 image to the records (k_lsd_pre): both kernels' bytes less the scaled fp64 image between them
        "lsd_grow": 13 * sw * sh,
        "lbd_sobel": w * h + w * h + 4 * w * h,
        "lbd_desc": 63 * 4 * 60 * nlines + 100 * nlines,
        "lsd_cull": 200 * nlines,
    }
    return table, 4 * S + 60 * nkp


def load_json(rel):
    try:
        with open(os.path.join(ROOT, rel)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def sources_sha16():
    """sha256 of the kernel sources (csrc/*.hip, *.inc, *.hpp), first 16 hex digits: the replayed counter profiles carry the hash of the
    sources they were measured on, and the bench line says `*_stale: true` when a kernel has changed since"""
    import glob, hashlib
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "a-low-texture-robust-hybrid-feature-based-visual-odometry_amd", "csrc", "*"))):
        if f.endswith((".hip", ".inc", ".hpp")):
            hsh.update(os.path.basename(f).encode()); hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def profile_stale(rel):
    t = load_json(rel)
    return bool(t is None or t.get("sources_sha16") != sources_sha16())


def hbm_traffic(group, B, width):
    """HBM bytes per launch of a kernel group from the PMC counters (FETCH_SIZE + WRITE_SIZE, collected in separate
    rocprofv3 --pmc passes of this same command; the counters cannot be read from inside the process, so the figure is
    REPLAYED from the committed profile named in roofline.traffic_source).  None when that profile does not cover this
    configuration."""
    t = load_json(TRAFFIC_PROFILE)
    try:
        if not t or t.get("frames_per_launch") != B or width != 640 or group not in t["bytes_per_frame"]:
            return None
        e = t["bytes_per_frame"][group]
        return int((e["fetch"] + e["write"]) * B)
    except (KeyError, TypeError):
        return None


def sq_utilisation(frames_per_s):
    """VALU busy fraction / resident waves per SIMD at the measured rate, from the committed SQ counters per frame
    (replayed, see roofline.counters_source)."""
    t = load_json(SQ_PROFILE)
    try:
        tot = t["per_frame_total"]
        cap = t["simds"] * t["clock_hz"] / 4.0                     # SIMD quad-cycles per second
        # MEASURED issue cost (tools/microbench/valu_issue.hip -> VALU_ISSUE_PROFILE): at 2-8 waves per SIMD a SIMD issues one wave64
        # instruction per 1.77-1.9 ns for everything this front-end is made of (v_dot4 / v_perm / v_min3 / SDWA / DPP / v_pk_*_u16 /
        # v_mad / fp64: 4.2 cycles at 2.4 GHz) and per 1.0 ns for VOP2 add / logic / shift and fp32 add / fma only; the fraction below
        # prices every VALU instruction of the step at the former (an upper bound by the share of the latter, < 15 % of these kernels)
        return {"valu_issue_frac": round(tot["sq_insts_valu"] * frames_per_s * VALU_ISSUE_NS * 1e-9 / t["simds"], 4),
                "valu_issue_ns_per_wave_instruction": VALU_ISSUE_NS, "valu_issue_source": VALU_ISSUE_PROFILE,
                "resident_waves_per_simd": round(tot["sq_wave_cycles"] * frames_per_s / cap, 3),
                "wave_time_with_inst_active": round(tot["sq_active_inst_any"] / tot["sq_wave_cycles"], 4),
                "counters_source": SQ_PROFILE + " (replayed: PMC counters cannot be read in-process)"}
    except (KeyError, TypeError, ZeroDivisionError):
        return {}


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle, kind "port"): SURVEY.md 8d modes (i) and (ii)
# ---------------------------------------------------------------------------------------------------------------
def cpu_reference_shaped(ge, stages, gray, depth, budget_s=8.0, max_frames=96):
    """one frame at a time, ORB || LSD || planes on three threads, like the Frame constructor (src/Frame.cc:210-215)"""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    orc = ge.oracle(); orb = orc.Orb()
    jobs = []
    tail = "tail" in stages
    if "orb" in stages: jobs.append(lambda g, d: orb.extract(g))
    def lines_job(g, d):
        kl = orc.line_extract(g)[0]
        if tail and len(kl):                                  # the line thread also runs isLineGood and the vanishing points (src/Frame.cc:895-939, 328-337)
            orc.lines_3d(kl, d, seed=1); orc.vanishing_points(kl, seed=1)
        return kl
    def planes_job(g, d):
        lab, pl = orc.peac(d)
        if tail:                                              # ComputePlanes' tail (src/Frame.cc:2110-2212)
            orc.plane_clouds(d, lab, pl, dist_th=0.05); orc.surface_normals(d)
        return pl
    if "lsd" in stages: jobs.append(lines_job)
    if "planes" in stages: jobs.append(planes_job)
    lat = []
    with ThreadPoolExecutor(max(len(jobs), 1)) as pool:
        t_all = time.perf_counter()
        n = 0
        while n < max_frames and (time.perf_counter() - t_all) < budget_s:
            g, d = gray[n % len(gray)], depth[n % len(depth)]
            t0 = time.perf_counter()
            fs = [pool.submit(j, g, d) for j in jobs]
            for f in fs: f.result()
            lat.append(time.perf_counter() - t0)
            n += 1
        el = time.perf_counter() - t_all
    lat = np.array(lat) * 1e3
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": len(jobs), "kind": "port",
            "latency_ms_p50": round(float(np.percentile(lat, 50)), 3), "latency_ms_p99": round(float(np.percentile(lat, 99)), 3),
            "sample": "%d frames of the same synthetic workload (%s), one frame at a time with ORB || LSD || planes on %d threads as in src/Frame.cc:210-215, "
                      "oracle/liboracle.so; host has %d cores; the reference binary itself cannot run here (OpenCV 3.2 / PCL absent)"
                      % (n, "+".join(stages), len(jobs), os.cpu_count())}


def cpu_all_cores(ge, stages, gray, depth, per_thread=3):
    """independent frames spread over every host core (one oracle instance per thread; the C calls release the GIL)"""
    orc = ge.oracle()
    nthr = os.cpu_count() or 1
    done = [0] * nthr
    def work(t):
        orb = orc.Orb()
        for k in range(per_thread):
            i = (t * per_thread + k)
            g, d = gray[i % len(gray)], depth[i % len(depth)]
            if "orb" in stages: orb.extract(g)
            if "planes" in stages: orc.peac(d)
            if "lsd" in stages: orc.line_extract(g)
            done[t] += 1
    thr = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
    t0 = time.perf_counter()
    for t in thr: t.start()
    for t in thr: t.join()
    el = time.perf_counter() - t0
    n = sum(done)
    return {"value": round(n / el, 2), "unit": "frames/s", "cores": nthr, "kind": "port",
            "sample": "%d frames, %d per thread on %d threads (every host core), oracle/liboracle.so" % (n, per_thread, nthr)}


# ---------------------------------------------------------------------------------------------------------------
# self-certification: sampled frames of the timed batch against the oracle (the checker, after the timed region)
# ---------------------------------------------------------------------------------------------------------------
def parity_sample(ge, np, stages, res, g0, d0, kinds, nfeat, want=32, scale=1.0):
    """Compares `want` frames of the resident batch (both scene kinds; res[f] is frame f, whose input is g0[f], d0[f]) with the CPU
    oracle, stage by stage: key points (x, y, octave, response, size exact; angle 1e-4) and descriptor bytes; key lines (count, pixel
    counts, descriptor bytes exact; end points 1e-4); label image and plane supports exact.  Returns (checked, failures, detail)."""
    orc = ge.oracle(); orb = orc.Orb(nfeatures=nfeat)
    n = min(len(res), len(g0))
    step = max(1, n // want)
    idx = sorted({min(n - 1, step * k + (k % 4)) for k in range(want)})
    bad = []
    for f in idx:
        r = res[f]; why = []
        if "orb" in stages:
            kp, desc = orb.extract(g0[f])
            if len(kp) != len(r["kp"]) or not np.array_equal(desc, r["desc"]): why.append("orb")
            else:
                if any(not np.array_equal(kp[k], r["kp"][k]) for k in ("x", "y", "octave", "response", "size")): why.append("orb-fields")
                elif len(kp) and np.max(np.abs(kp["angle"] - r["kp"]["angle"])) > 1e-4: why.append("orb-angle")
        if "lsd" in stages:
            kl, ld, fn = orc.line_extract(g0[f])
            if len(kl) != len(r["kl"]) or not np.array_equal(ld, r["ldesc"]) or not np.array_equal(kl["num_pixels"], r["kl"]["num_pixels"]): why.append("lsd")
            elif len(kl) and max(np.max(np.abs(kl[k] - r["kl"][k])) for k in ("sx", "sy", "ex", "ey")) > 1e-4: why.append("lsd-ends")
        if "planes" in stages:
            lab, pl = orc.peac(d0[f], fx=535.4 * scale, fy=539.2 * scale, cx=320.1 * scale, cy=247.6 * scale)
            if not np.array_equal(lab, r["labels"]) or len(pl) != len(r["planes"]) or not np.array_equal(pl["n_points"], r["planes"]["n_points"]): why.append("planes")
        if why: bad.append("frame %d (%s): %s" % (f, kinds[f], ",".join(why)))
    return len(idx), len(bad), bad[:4], sum(1 for f in idx if kinds[f] == "lowtex")


# ---------------------------------------------------------------------------------------------------------------
# synthetic workload
# ---------------------------------------------------------------------------------------------------------------
def make_frames(synth, n, w, h, seed0):
    """n distinct frames, every fourth one `lowtex` (the paper's target regime), the rest `std`"""
    import numpy as np
    g = np.empty((n, h, w), np.uint8); d = np.empty((n, h, w), np.uint16)
    kinds = []
    for k in range(n):
        kind = "lowtex" if k % 4 == 3 else "std"
        g[k], d[k] = synth.make_frame(kind, seed0 + k, w, h)
        kinds.append(kind)
    return g, d, kinds


def geometry(cfg):
    return (1280, 960, 2000) if cfg == "big1280" else (640, 480, 1000)


def new_context(hvo, cfg, B, device):
    w, h, nfeat = geometry(cfg)
    s = w / 640.0
    return hvo.Context(max_batch=B, device=device, orb_nfeatures=nfeat, fx=535.4 * s, fy=539.2 * s, cx=320.1 * s, cy=247.6 * s)


def latency_probe(hvo, cfg, g, d, mask, device, sizes=(1, 32)):
    """ms per hvo_batch_run of small resident batches (the three subsystems overlapped), best of 5"""
    out = {}
    for B in sizes:
        ctx = new_context(hvo, cfg, B, device)
        try:
            reps = (B + len(g) - 1) // len(g)
            ctx.batch_upload(g[:B] if B <= len(g) else g, d[:B] if B <= len(d) else d, repeat=1 if B <= len(g) else reps)
            for _ in range(2): ctx.batch_run(mask)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); ctx.batch_run(mask); best = min(best, time.perf_counter() - t0)
            out["B%d" % B] = round(best * 1e3, 3)
        finally:
            ctx.close()
    return out


def pcie_inclusive(hvo, cfg, g, d, mask, device, B=2048, nctx=3, rounds=4):
    """upload + run + download of consecutive batches, overlapped (hvo.BatchPipeline): `nctx` contexts on `nctx` host threads,
    one lock per stage, so one batch is on the link while another is on the GPU and a third is coming back.  Host images and
    the label slabs are page-locked (hvo_pin_host); the label image crosses PCIe and is handed over as int8
    (hvo_frame_out.labels8); result arrays are reused from batch to batch."""
    reps = max(1, B // len(g))
    n = reps * len(g)
    pipe = hvo.BatchPipeline(nctx=nctx, batch=n, stages=mask, make_context=lambda b: new_context(hvo, cfg, b, device))
    hvo.pin(g); hvo.pin(d)
    try:
        pipe.run(g, d, repeat=reps, rounds=1)                         # warm-up: plans, pinned staging, result slabs
        t0 = time.perf_counter()
        frames = pipe.run(g, d, repeat=reps, rounds=rounds)
        el = time.perf_counter() - t0
    finally:
        pipe.close()
        hvo.unpin(g); hvo.unpin(d)
    return round(frames / el, 1), "%d contexts x %d frames, %d rounds each" % (nctx, n, rounds)


def end_to_end(hvo, np, ctx, g, d, mask, B, rounds=4):
    """consecutive batches INCLUDING PCIe at the resident batch's efficiency: one context, double-buffered -- while batch k runs, batch
    k + 1's images go up into staging slabs and batch k - 1's results (records + int8 label image, one packed slab) come down
    (hvo_batch_stage_upload / _commit_staged / _results_async, include/hvo.h).  Host images and the result slab are page-locked."""
    reps = max(1, B // len(g)); n = reps * len(g)
    hvo.pin(g); hvo.pin(d)
    sb = ctx.slab_layout(labels=True)[3]
    host = [np.empty(n * sb, np.uint8), np.empty(n * sb, np.uint8)]
    for a in host: hvo.pin(a)
    try:
        fr = ctx.batch_stage_upload(g, d, repeat=reps); ctx.batch_commit_staged()
        ctx.batch_run(mask); ctx.batch_results_async(n, host[0]); ctx.batch_results_wait()            # warm-up: staging slabs, result slab
        ctx.batch_stage_upload(None, None, frames_in=fr); ctx.batch_commit_staged()
        t0 = time.perf_counter()
        for k in range(rounds):
            ctx.batch_stage_upload(None, None, frames_in=fr)          # batch k + 1 goes up ...
            ctx.batch_run(mask)                                        # ... while batch k runs (and batch k - 1 comes down)
            ctx.batch_results_async(n, host[k & 1])
            ctx.batch_commit_staged()
        ctx.batch_results_wait()
        el = time.perf_counter() - t0
        # what came down is what a download gives: the header of the last slab against the batch's own counts
        hdr = host[(rounds - 1) & 1][:16].view(np.int32)
        ok = int(hdr[0]) > 0 and int(hdr[3]) == 0
        # ... and its PAYLOAD: the first 32 frames of the last slab (records and the int8 label image, as they crossed PCIe through the pack +
        # async-copy path) byte for byte against hvo_batch_download of the same images run once more (every batch of this loop is the same
        # images, and the pipeline is deterministic)
        import importlib
        hd = importlib.import_module("hvo_amd.dist")
        kc, lc, pc, sb2, _ = ctx.slab_layout(labels=True)
        ctx.batch_run(mask)
        m = min(32, n)
        ref = ctx.batch_download(mask, n=m)
        hh, ww = g.shape[1], g.shape[2]
        back = hd.unpack_results(hvo, host[(rounds - 1) & 1][: m * sb2].reshape(m, sb2), kc, lc, pc, label_shape=(hh, ww))
        nbad = 0
        for a, b in zip(back, ref):
            for key in ("kp", "desc", "kl", "ldesc", "linefn", "planes", "labels"):
                if key in b and not np.array_equal(a[key], b[key]): nbad += 1
        ok = ok and nbad == 0 and len(back) == m
    finally:
        for a in host: hvo.unpin(a)
        hvo.unpin(g); hvo.unpin(d)
    return round(rounds * n / el, 1), ok, "%d rounds of %d frames, one context, inputs and results double-buffered; %.1f MB up and %.1f MB down per frame; results_ok = %d frames of the last slab byte-equal to hvo_batch_download (records + label image)" % (rounds, n, (g[0].nbytes + d[0].nbytes) / 1e6, sb / 1e6, m)


# ---------------------------------------------------------------------------------------------------------------
# stream mode
# ---------------------------------------------------------------------------------------------------------------
def stream_queries(np, last, shift, sf, bf, th=15):
    kp = last["kp_un"]; z = last["zdepth"]
    sel = np.nonzero(z > 0)[0].astype(np.int32)
    u = (kp["x"][sel] - np.float32(shift[0])).astype(np.float32); v = (kp["y"][sel] - np.float32(shift[1])).astype(np.float32)
    octv = kp["octave"][sel]
    rad = (np.float32(th) * sf[octv]).astype(np.float32)
    ur = (u - np.float32(bf) / z[sel]).astype(np.float32)
    return sel, u, v, rad, (octv - 1).astype(np.int32), (octv + 1).astype(np.int32), ur, np.ones(len(sel), np.uint8)


def run_stream(hvo, np, g, d, off, mask, depth, device, nframes, paced_hz=0.0):
    """frames through hvo_stream_* with up to depth - 1 in flight (one slot keeps the previous frame resident for the
    matching); every frame is collected and matched against its predecessor (SearchByProjection(Cur, Last) +
    LSDmatcher::match).  paced_hz = 0: the next frame is submitted as soon as a slot is free (sustained rate); paced_hz > 0:
    frames arrive on a camera clock and a frame is collected as soon as it is complete (latency under a 30 fps feed).
    Returns (seconds, per-frame latency ms from submit to results + matches in hand, matches per frame)."""
    bf = 40.0
    st = hvo.Stream(width=g.shape[2], height=g.shape[1], depth=depth, stages=mask, bf=bf, device=device)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    lat = []; nm = [0, 0]
    try:
        tick = {}; tsub = {}; state = {"last": None, "done": 0}
        inflight = max(1, depth - 1)
        t_start = time.perf_counter()

        def finish(i):                                    # collect frame i, match it against frame i - 1
            cur = st.collect(tick[i], labels=True)
            last = state["last"]
            if last is not None and (mask & 1):
                shift = off[i % len(off)] - off[(i - 1) % len(off)] if i % len(off) else (0, 0)
                sel, u, v, rad, lmin, lmax, ur, blocks = stream_queries(np, last, shift, sf, bf)
                n, _, _ = st.search_by_projection(tick[i], tick[i - 1], sel, u, v, rad, lmin, lmax, ur, blocks)
                nm[0] += n
            if last is not None and (mask & 2):
                n, _ = st.match_lines(tick[i - 1], tick[i], hvo.LINE_MATCH_NNR, nnratio=0.95)
                nm[1] += n
            lat.append((time.perf_counter() - tsub[i]) * 1e3)
            state["last"] = cur; state["done"] = i + 1

        for k in range(nframes):
            if paced_hz > 0:
                due = t_start + k / paced_hz
                while time.perf_counter() < due:
                    if state["done"] < k and st.poll(tick[state["done"]]): finish(state["done"])
                    else: time.sleep(0.0002)
            while k - state["done"] >= inflight:          # every slot busy: the oldest frame has to leave first
                finish(state["done"])
            tsub[k] = time.perf_counter()
            tick[k] = st.submit(g[k % len(g)], d[k % len(d)])
        while state["done"] < nframes:
            finish(state["done"])
        el = time.perf_counter() - t_start
    finally:
        st.close()
    return el, np.array(lat), nm[0] / max(nframes - 1, 1), nm[1] / max(nframes - 1, 1)


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        spawn_ranks(args)                                            # does not return
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, env_world))
        sys.exit(2)

    import numpy as np
    import torch
    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"     # where the scalar reductions live
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)

    if not os.path.exists(os.path.join(ge.PKG_DIR, "csrc", "libhvo.so")):
        ge.build()
    hvo = ge.package()
    synth = importlib.import_module("hvo_amd.synth")
    hdist = importlib.import_module("hvo_amd.dist")
    stages = [s for s in (args.stages or ("orb,lsd,planes,tail" if args.mode == "stream" else "orb,lsd,planes")).split(",") if s]
    mask = 0
    for s in stages:
        mask |= {"orb": hvo.STAGE_ORB, "lsd": hvo.STAGE_LSD, "planes": hvo.STAGE_PLANES,
                 "tail": hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS}[s]
    w, h, nfeat = geometry(args.config)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ======================================================= stream mode ==========================================
    if args.mode == "stream":
        nframes = args.steps if args.steps > 0 else 573          # length of Examples/RGB-D/associations/fr1_desk.txt
        ndist = min(nframes + args.warmup, 573)
        data_kind = "synthetic"
        tum_dir = os.environ.get("HVO_TUM_DIR")
        if tum_dir and os.path.isdir(tum_dir):
            # BASELINE configs[4] on the real sequence where it exists (e.g. rgbd_dataset_freiburg1_desk with its association file):
            # frames in file order, the tracker's pose prediction replaced by "no drift" (shift 0) for SearchByProjection's queries
            tum = importlib.import_module("hvo_amd.tum")
            g, d = tum.load_sequence(tum_dir, limit=ndist, assoc=os.environ.get("HVO_TUM_ASSOC"))
            off = np.zeros((len(g), 2), np.float32); h, w = g.shape[1:]; data_kind = "TUM RGB-D sequence " + os.path.basename(os.path.normpath(tum_dir))
        else:
            g, d, off = synth.make_sequence("std", 0x5EED3000 + 1000 * rank, ndist, w, h)
        run_stream(hvo, np, g, d, off, mask, args.depth, local_rank, max(args.warmup, 2))          # warm-up: plans, pinned buffers
        barrier()
        el, lat, mpts, mlines = run_stream(hvo, np, g, d, off, mask, args.depth, local_rank, nframes)
        barrier()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        if rank == 0:
            out = {
                "metric": "RGB-D frames/sec (640×480, 1k ORB + LSD + PEAC) at 1/2/4/8 GPUs", "value": round(world * nframes / el, 2), "unit": "frames/s",
                "n_gpus": world, "steps": nframes, "warmup": args.warmup, "ms_per_step": round(el / nframes * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u16 integer + f32/f64", "data": data_kind,
                "config": {"workload": "stream-%d: %dx%d synthetic RGB-D sequence (smooth <= 4 px/frame drift), one frame at a time through hvo_stream_* "
                                       "(upload + %d ORB + LSD + PEAC%s + download + SearchByProjection(Cur,Last) + line match per frame), %d frames in flight"
                                       % (nframes, w, h, nfeat, " + isLineGood + vanishing points + plane clouds / refit / surface normals + grids" if "tail" in stages else "", args.depth),
                           "mode": "stream", "stages": stages, "depth": args.depth, "pcie_inclusive": True,
                           "frame_constructor": "whole (src/Frame.cc:205-233: ORB || LSD(+isLineGood, vanishing points) || planes(+clouds, refit, surface normals), undistort, stereo, grids)" if "tail" in stages else "extraction + undistort + stereo",
                           "mean_point_matches": round(mpts, 1), "mean_line_matches": round(mlines, 1)},
                "latency_ms": {"pipelined_p50": round(float(np.percentile(lat, 50)), 3), "pipelined_p99": round(float(np.percentile(lat, 99)), 3)},
                "roofline": None,
            }
            if not args.no_extras:
                # single-frame latency: nothing else in flight (submit, collect, match, then the next frame)
                _, lat1, _, _ = run_stream(hvo, np, g, d, off, mask, 2, local_rank, min(nframes, 64))
                out["latency_ms"].update(single_frame_p50=round(float(np.percentile(lat1, 50)), 3), single_frame_p99=round(float(np.percentile(lat1, 99)), 3))
                # which pipeline depth keeps up with a 30 fps camera (frames submitted on a 33.3 ms clock)
                sustain = {}
                for dep in (2, 3, 4):
                    e2, l2, _, _ = run_stream(hvo, np, g, d, off, mask, dep, local_rank, min(nframes, 90), paced_hz=30.0)
                    sustain["depth%d" % dep] = {"fps": round(min(nframes, 90) / e2, 2), "latency_ms_p50": round(float(np.percentile(l2, 50)), 3),
                                               "latency_ms_p99": round(float(np.percentile(l2, 99)), 3)}
                out["paced_30fps"] = sustain
                ok = [int(k[5:]) for k, v in sustain.items() if v["fps"] >= 29.5 and v["latency_ms_p99"] < 1000.0 / 30.0 * (int(k[5:]) - 1) + 5.0]
                out["pipeline_depth_sustaining_30fps"] = min(ok) if ok else None
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_reference_shaped(ge, stages, g, d)
                out["latency_ms"]["cpu_reference_shaped_3_threads_p50"] = out["cpu_baseline"]["latency_ms_p50"]
            print(json.dumps(out))
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    # ======================================================= batch mode ===========================================
    B = args.batch
    per_frame = BYTES_PER_FRAME_640 if w <= 640 else BYTES_PER_FRAME_1280
    if B <= 0:
        if args.config == "batch256":
            lo_, hi_ = hdist.shard_range(256, world, rank)           # BASELINE configs[3]: 256 frames over the ranks, contiguous shards (ragged when 256 % N != 0)
            B = max(1, hi_ - lo_)
        else:
            free_b, _ = torch.cuda.mem_get_info()
            B = next((c for c in (8192, 6144, 4096, 3072, 2048, 1024, 512, 256) if c * per_frame <= 0.85 * free_b), 128)
        if dist is not None and args.config != "batch256":      # every rank must run the same workload (batch256: the shards of 256 frames)
            t = torch.tensor([B], dtype=torch.int64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            B = int(t.item())
    ndistinct = max(1, min(B, args.distinct))
    g0, d0, kinds = make_frames(synth, ndistinct, w, h, 0x5EED1000 + 100000 * rank)
    reps = max(1, B // ndistinct)
    B = reps * ndistinct                    # the distinct frames are uploaded cyclically, by pointer
    steps = args.steps if args.steps > 0 else 20

    ctx = new_context(hvo, args.config, B, local_rank)
    ctx.batch_upload(g0, d0, repeat=reps)   # inputs resident in HBM before the timed region

    for _ in range(args.warmup):
        ctx.batch_run(mask)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
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
    psteps = max(1, min(steps, 3))
    ctx.profile_enable(2)
    for _ in range(psteps):
        ctx.batch_run(mask)
        for k, v in ctx.profile_last().items():
            prof[k] = prof.get(k, 0.0) + v
    ctx.profile_enable(0)

    # N > 1: the one collective of the path -- all_gather of the device-resident result slabs (RCCL over xGMI)
    gather = None
    if dist is not None:
        barrier()
        tg = time.perf_counter()
        ragged = args.config == "batch256" and args.batch <= 0
        ranks_seen, slab_bytes = hdist.gather_device_slabs(ctx, B, red_dev, n_frames=256 if ragged else None)
        barrier()
        gather = {"gather_ms": round((time.perf_counter() - tg) * 1e3, 3), "ranks_seen": ranks_seen, "slab_bytes_per_frame": slab_bytes,
                  "frames_gathered": 256 if ragged else world * B}

    res = ctx.batch_download(mask, n=min(B, 256))     # a sample is enough for the workload statistics
    nkp = float(np.mean([len(r["kp"]) for r in res])) if "orb" in stages else 0.0
    nlines = float(np.mean([len(r["kl"]) for r in res])) if "lsd" in stages else 0.0
    nplanes = float(np.mean([len(r["planes"]) for r in res])) if "planes" in stages else 0.0
    bad = sum(1 for r in res if r["status"] != 0)
    e2e = None
    if rank == 0 and world == 1 and not args.no_extras and args.config != "big1280" and mask == hvo.STAGE_ALL:
        e2e = end_to_end(hvo, np, ctx, g0, d0, mask, B)
    # the same step on `std` frames only (SURVEY.md 8d item 4 specifies 256 `std` frames; the headline's mix holds a quarter of `lowtex`
    # frames, whose line stage is five times cheaper): the mix's std frames + as many new ones as it has lowtex frames, same batch size
    std_only = None
    if rank == 0 and world == 1 and not args.no_extras and args.config == "std640":
        keep = [k for k in range(ndistinct) if kinds[k] == "std"]
        gs = np.empty_like(g0); ds = np.empty_like(d0)
        gs[: len(keep)] = g0[keep]; ds[: len(keep)] = d0[keep]
        for k in range(len(keep), ndistinct):
            gs[k], ds[k] = synth.make_frame("std", 0x5EED9000 + 100000 * rank + k, w, h)
        ctx.batch_upload(gs, ds, repeat=reps)
        ctx.batch_run(mask)
        ssteps = max(1, min(steps, 5))
        t1 = time.perf_counter()
        for _ in range(ssteps):
            ctx.batch_run(mask)
        dts = time.perf_counter() - t1
        rs = ctx.batch_download(mask, n=min(B, 64))
        std_only = {"value": round(B * ssteps / dts, 2), "ms_per_step": round(dts / ssteps * 1e3, 4), "steps": ssteps, "frames_per_gpu": B,
                    "mean_keypoints": round(float(np.mean([len(r["kp"]) for r in rs])), 1) if "orb" in stages else 0.0,
                    "mean_lines": round(float(np.mean([len(r["kl"]) for r in rs])), 1) if "lsd" in stages else 0.0,
                    "scene_mix": {"std": ndistinct}}
    ctx.close()
    parity = parity_sample(ge, np, stages, res, g0, d0, kinds, nfeat, scale=w / 640.0) if rank == 0 else None

    if rank == 0:
        frames = (256 if (args.config == "batch256" and args.batch <= 0) else world * B) * steps      # (batch256: the ranks' shards of 256 frames may differ by one)
        value = frames / dt
        groups = {k: v / psteps for k, v in prof.items()}           # ms per launch group per step (serialised pass)
        if "lsd_gradient" in groups and "lsd_blur_scale" not in groups: groups["lsd_pre"] = groups.pop("lsd_gradient")      # the fused preamble reports under the second kernel's name
        table, pass_bytes = algorithmic_bytes(w, h, nkp, nlines)
        kroof = {}
        for k, ms in groups.items():
            pf = table.get(k, 0)
            gbps = pf * B / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            kroof[k] = {"ms": round(ms, 4), "alg_bytes_per_frame": int(pf), "GBps": round(gbps, 2), "frac": round(gbps / HBM_PEAK_GBS, 5)}
        dom = max(groups, key=groups.get) if groups else None
        roof = None
        if dom:
            tr = hbm_traffic(dom, B, w)
            roof = {"bound": "hbm", "kernel": dom, "achieved": kroof[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kroof[dom]["frac"], "traffic": tr,
                    "traffic_source": (TRAFFIC_PROFILE + " (replayed: separate rocprofv3 --pmc passes of this command)") if tr is not None else None,
                    "traffic_stale": profile_stale(TRAFFIC_PROFILE) if tr is not None else None,      # true: a kernel source changed after that profile was taken
                    "bytes_per_launch": int(table.get(dom, 0) * B), "ms_per_launch": round(groups[dom], 4),
                    "limiter": "dependent-latency chain (serial semantics), not HBM bandwidth: see valu_issue_frac / DESIGN.md section 4"}
            roof.update(sq_utilisation(value))
            roof["counters_stale"] = profile_stale(SQ_PROFILE)
            vif = roof.get("valu_issue_frac")
            if vif is not None:
                # what limits the STEP (all kernels overlapped): the vector ALUs' issue slots once they are more than half taken; the dominant
                # KERNEL alone is a dependent chain (its own HBM fraction is `frac`)
                roof["step_limiter"] = ("valu-issue: %.0f %% of the SIMDs' issue cycles carry a VALU instruction (%.2f ns per wave instruction per SIMD, measured)" % (100 * vif, VALU_ISSUE_NS)) if vif >= 0.5 \
                    else "latency: the serial-semantics kernels wait on dependent memory round trips (VALU issue %.0f %%)" % (100 * vif)
            orb_ms = sum(v for k, v in groups.items() if k in ("orb_pyramid", "orb_fast_cells", "orb_blur", "orb_levels", "orb_brief", "orb_orient", "orb_describe"))
            if orb_ms > 0:
                roof["orb_pyramid_brief_pass_GBps"] = round(pass_bytes * B / (orb_ms * 1e-3) / 1e9, 2)
                roof["orb_pyramid_brief_pass_frac"] = round(pass_bytes * B / (orb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                # ... and with the quadtree, which sits between FAST and the orientation on the same stream
                oq = orb_ms + groups.get("orb_octree", 0.0)
                roof["orb_pyramid_brief_pass_with_quadtree_frac"] = round(pass_bytes * B / (oq * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out = {
            "metric": "RGB-D frames/sec (640×480, 1k ORB + LSD + PEAC) at 1/2/4/8 GPUs", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u16 integer + f32/f64",
            "data": "synthetic",
            "config": {"workload": "%dx%d synthetic RGB-D, %d ORB (quota; %.0f realised per frame: 3/4 std + 1/4 lowtex scenes) + LSD lines + PEAC planes, %d frames per GPU per step (%s)"
                                   % (w, h, nfeat, nkp, B, args.config),
                       "stages": stages, "frames_per_gpu": B, "parallelism": "frames sharded, %d rank(s), no data-path collective" % world,
                       "distinct_frames": ndistinct, "scene_mix": {k: kinds.count(k) for k in sorted(set(kinds))},
                       "mean_keypoints": round(nkp, 1), "mean_lines": round(nlines, 1), "mean_planes": round(nplanes, 2),
                       "frames_with_capacity_flags": bad, "resident_bytes_per_frame_estimate": int(per_frame)},
            "kernel_ms_per_step_serialised": {k: round(v, 4) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
            "kernel_roofline": kroof,
            "roofline": roof,
            "parity_checked_frames": parity[0], "parity_failures": parity[1], "parity_lowtex_frames": parity[3],
            "parity_note": "frames of the timed resident batch compared with the CPU oracle after the timed region, every stage" + ("; FAILED: " + "; ".join(parity[2]) if parity[1] else ""),
        }
        if gather:
            out["gather"] = gather
        if world == 1 and not args.no_extras:
            out["latency_ms"] = latency_probe(hvo, args.config, g0, d0, mask, local_rank)
            if std_only is not None:
                out["value_std_only"] = std_only["value"]; out["std_only"] = std_only
            if e2e is not None:
                # the end-to-end figure: host images in, host results (records + int8 label image) out, consecutive batches
                out["value_end_to_end"] = e2e[0]
                out["end_to_end_frac_of_resident"] = round(e2e[0] / value, 3)
                out["end_to_end_results_ok"] = e2e[1]
                out["end_to_end_note"] = e2e[2]
                out["pcie_inclusive_frames_per_s"] = e2e[0]             # (the name of rounds 1-3)
                out["pcie_inclusive_frac_of_resident"] = round(e2e[0] / value, 3)
            elif args.config != "big1280":
                rate, what = pcie_inclusive(hvo, args.config, g0, d0, mask, local_rank, B=min(2048, max(B, 256)))
                out["pcie_inclusive_frames_per_s"] = rate
                out["pcie_inclusive_frac_of_resident"] = round(rate / value, 3)
                out["pcie_inclusive_note"] = what + ": upload + run + download of consecutive batches overlapped (one lock per stage), pinned host buffers, int8 labels on the wire"
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_reference_shaped(ge, stages, g0, d0)
            out["cpu_baseline_all_cores"] = cpu_all_cores(ge, stages, g0, d0)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
