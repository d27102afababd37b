#!/usr/bin/env python3
"""Device memory after repeated create / use / destroy of contexts and streams (every lazily allocated slab must come back):
    python tools/leak_check.py [rounds=25]"""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def free_bytes(hip):
    f = ctypes.c_size_t(0); t = ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
    hip = ctypes.CDLL("libamdhip64.so")
    g, d = synth.make_batch("std", 0x5EEDB000, 6)
    def once():
        ctx = hvo.Context(max_batch=6)
        ctx.extract_orb(g[0]); ctx.extract_lsd(g[0]); ctx.extract_lsd(g[0], culled=True); ctx.compute_planes(d[0])
        ctx.batch_upload(g, d); ctx.set_tail_params(seed=3)
        full = hvo.STAGE_ALL | hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS
        ctx.batch_run(full); res = ctx.batch_download(hvo.STAGE_ALL); ctx.batch_download_tail(full, res)
        ctx.close()
        st = hvo.Stream(depth=3, stages=full, bf=40.0, seed=5)
        t = [st.submit(g[i], d[i]) for i in range(3)]
        for x in t: st.collect(x)
        st.close()
    once(); once()
    f0 = free_bytes(hip); lo = f0
    for r in range(rounds):
        once(); lo = min(lo, free_bytes(hip))
    f1 = free_bytes(hip)
    print("RESULT leak check: free device memory %.1f MB before, %.1f MB after %d rounds (lowest %.1f): %s" % (f0 / 1e6, f1 / 1e6, rounds, lo / 1e6, "no growth" if f0 - f1 < 8e6 else "LEAK of %.1f MB" % ((f0 - f1) / 1e6)))


if __name__ == "__main__":
    main()
