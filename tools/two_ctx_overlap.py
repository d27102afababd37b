#!/usr/bin/env python3
"""Does the chip have idle issue slots at the end of a step that a second, independent batch could fill?  One context with B frames against
two contexts with B/2 frames each driven from two host threads (the library calls release the GIL).  Diagnostic:
    python tools/two_ctx_overlap.py [B] [steps]"""
import importlib, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    g = np.empty((64, 480, 640), np.uint8); d = np.empty((64, 480, 640), np.uint16)
    for k in range(64): g[k], d[k] = synth.make_frame("lowtex" if k % 4 == 3 else "std", 0x5EED1000 + k)
    out = {}
    for parts in (1, 2, 1, 2):
        n = B // parts
        ctxs = [hvo.Context(max_batch=n) for _ in range(parts)]
        for c in ctxs: c.batch_upload(g, d, repeat=n // 64); c.batch_run(7)
        bar = threading.Barrier(parts)
        def work(c, stagger):
            bar.wait()
            for _ in range(K): c.batch_run(7)
        ts = [threading.Thread(target=work, args=(c, i)) for i, c in enumerate(ctxs)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
        out.setdefault("parts%d" % parts, []).append(round(B * K / dt, 1))
        print(parts, "context(s):", round(B * K / dt, 1), "frames/s", flush=True)
        for c in ctxs: c.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
