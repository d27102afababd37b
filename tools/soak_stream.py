#!/usr/bin/env python3
"""Differential soak of the streamed mode: sequences through hvo_stream_* (three frames in flight, every stage of the Frame constructor),
every frame against the CPU oracle (test infrastructure).      python tools/soak_stream.py [--frames 256] [--kind lowtex] [--seed 0x...] [--tail]"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256); ap.add_argument("--kind", default="std")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xB0B00000); ap.add_argument("--seq", type=int, default=32, help="frames per generated sequence")
    ap.add_argument("--tail", action="store_true")
    args = ap.parse_args()
    hvo = ge.package(); orc = ge.oracle(); synth = importlib.import_module("hvo_amd.synth")
    from test_stream_gpu import check_frame, BF
    from test_tail_gpu import check_tail
    stages = hvo.STAGE_ALL | ((hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS) if args.tail else 0)
    seed_t = 4242
    st = hvo.Stream(depth=4, stages=stages, bf=BF, seed=seed_t) if args.tail else hvo.Stream(depth=4, stages=stages, bf=BF)
    orb = orc.Orb(); bad = []; done = 0; t0 = time.time()
    try:
        while done < args.frames:
            n = min(args.seq, args.frames - done)
            g, d, _ = synth.make_sequence(args.kind, args.seed + done, n)
            tick = [st.submit(g[i], d[i]) for i in range(min(3, n))]
            for i in range(n):
                r = st.collect(tick[i])
                if i + 3 < n: tick.append(st.submit(g[i + 3], d[i + 3]))
                try:
                    assert r["status"] == 0
                    check_frame(r, g[i], d[i], orc, orb)
                    if args.tail: check_tail(r, d[i], orc, seed_t + tick[i], (0.0, 640.0, 0.0, 480.0))
                except AssertionError as e:
                    bad.append((done + i, str(e)[:60]))
            done += n
            print("frames %d done, %d differ, %.0f s" % (done, len(bad), time.time() - t0), flush=True)
    finally:
        st.close()
    print("RESULT stream kind=%s seed=%#x frames=%d tail=%s differing=%d %s" % (args.kind, args.seed, args.frames, args.tail, len(bad), bad[:8]))


if __name__ == "__main__":
    main()
