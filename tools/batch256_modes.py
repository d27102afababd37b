#!/usr/bin/env python3
"""batch256 on one GPU, run to run: BASELINE configs[3]'s 256 frames as one resident batch, R fresh contexts x K steps each, the step time
and the serialised per-kernel times of every context -- what differs between the fast and the slow mode (26.3 / 28.3 ms)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import __graft_entry__ as ge

def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
    g, d, _ = B.make_frames(synth, 256, 640, 480, 0x5EED1000)
    rows = []
    for r in range(R):
        ctx = B.new_context(hvo, "batch256", 256, 0)
        ctx.batch_upload(g, d)
        for _ in range(3): ctx.batch_run(7)
        ts = []
        for _ in range(K):
            t0 = time.perf_counter(); ctx.batch_run(7); ts.append((time.perf_counter() - t0) * 1e3)
        ctx.profile_enable(1); ctx.batch_run(7); ov = {k: round(v, 2) for k, v in ctx.profile_last().items()}; ctx.profile_enable(0)
        rows.append(dict(ctx=r, ms_min=round(min(ts), 2), ms_med=round(float(np.median(ts)), 2), ms_max=round(max(ts), 2), steps=[round(t, 1) for t in ts], overlapped_kernel_ms=ov))
        print(json.dumps(rows[-1]), flush=True)
        ctx.close()

if __name__ == "__main__":
    main()
