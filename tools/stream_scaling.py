#!/usr/bin/env python3
"""Sustained rate of the streamed mode against the number of frames in flight (and, through the environment, against
GPU_MAX_HW_QUEUES: the HIP runtime maps streams onto that many hardware queues per priority level, default 4).
    GPU_MAX_HW_QUEUES=16 python tools/stream_scaling.py [frames]"""
import importlib, json, os, sys, time
import numpy as np
if os.environ.get("IMPORT_TORCH"):
    import torch; torch.cuda.init()         # torch's bundled HIP runtime instead of /opt/rocm's (whichever loads first serves both)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
    g, d, off = synth.make_sequence("std", 0x5EED3000, 64)
    out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default"), "torch_runtime": bool(os.environ.get("IMPORT_TORCH")),
           "lsd_own_stream": os.environ.get("HVO_STREAM_LSD_OWN", "0")}
    for depth in [int(x) for x in os.environ.get("DEPTHS", "2,3,4,6,8,12,16").split(",")]:
        st = hvo.Stream(depth=depth, stages=hvo.STAGE_ALL, bf=40.0)
        inflight = depth - 1
        tick = [st.submit(g[k % 64], d[k % 64]) for k in range(inflight)]
        for k in range(inflight): st.collect(tick[k], labels=False)      # warm-up round
        tick = [st.submit(g[k % 64], d[k % 64]) for k in range(inflight)]
        t0 = time.perf_counter(); thost = 0.0
        for i in range(n):
            st.collect(tick[i], labels=False)
            h0 = time.perf_counter()
            tick.append(st.submit(g[(i + inflight) % 64], d[(i + inflight) % 64]))
            thost += time.perf_counter() - h0
        el = time.perf_counter() - t0
        for k in range(n, len(tick)): st.collect(tick[k], labels=False)
        ms = st.stage_ms(tick[-1])
        st.close()
        out["depth%d" % depth] = {"fps": round(n / el, 1), "submit_ms": round(thost / n * 1e3, 3), "last_frame_stage_ms": ms}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
