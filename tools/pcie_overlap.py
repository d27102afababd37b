#!/usr/bin/env python3
"""PCIe-inclusive rate with upload / run / download of consecutive batches overlapped: C contexts on C host threads, each looping
upload -> run -> download over its own resident batch (what bench.py reports as pcie_inclusive_frames_per_s), with the time
each leg took inside the threads.      python tools/pcie_overlap.py [batch] [contexts] [rounds]"""
import importlib, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B

def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import __graft_entry__ as ge
    hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
    g, d, _ = B.make_frames(synth, 256, 640, 480, 0x5EED1000)
    pinned = os.environ.get("PIN", "1") != "0"
    if pinned: hvo.pin(g); hvo.pin(d)
    reps = max(1, batch // 256); n = reps * 256
    ctxs = [B.new_context(hvo, "std640", n, 0) for _ in range(nctx)]
    legs = [[0.0, 0.0, 0.0] for _ in ctxs]
    use_locks = os.environ.get("LOCKS", "1") != "0"
    L = [threading.Lock() for _ in range(3)]
    class NoLock:
        def __enter__(self): pass
        def __exit__(self, *a): pass
    lk = L if use_locks else [NoLock()] * 3
    def loop(i, k, rec):
        c = ctxs[i]
        for _ in range(k):
            with lk[0]:
                t0 = time.perf_counter(); c.batch_upload(g, d, repeat=reps); t1 = time.perf_counter()
            with lk[1]:
                t1b = time.perf_counter(); c.batch_run(7); t2 = time.perf_counter()
            with lk[2]:
                t2b = time.perf_counter(); c.batch_download(7, reuse=True, labels8=True, pinned=pinned); t3 = time.perf_counter()
            if rec: legs[i][0] += t1 - t0; legs[i][1] += t2 - t1b; legs[i][2] += t3 - t2b
    for i in range(nctx): loop(i, 1, False)
    thr = [threading.Thread(target=loop, args=(i, rounds, True)) for i in range(nctx)]
    t0 = time.perf_counter()
    for t in thr: t.start()
    for t in thr: t.join()
    el = time.perf_counter() - t0
    for c in ctxs: c.close()
    if pinned: hvo.unpin(g); hvo.unpin(d)
    print(json.dumps({"batch": n, "contexts": nctx, "rounds": rounds, "frames_per_s": round(nctx * rounds * n / el, 1),
                      "leg_ms_per_batch": [[round(x / rounds * 1e3, 1) for x in l] for l in legs]}))

if __name__ == "__main__":
    main()
