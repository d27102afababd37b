#!/usr/bin/env python3
"""Hamming matching rate (instruction-bound, reported as Gpopc/s): knn-2 and full distance matrix of nq x nt 256-bit
descriptors with the descriptors resident in HBM (device-resident kernels timed with hipEvents through hvo_debug_match_rate),
and the host-array entry points (staging + PCIe included) beside them.     python tools/match_rate.py [nq] [nt]"""
import ctypes as C, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    hvo = ge.package(); L = hvo.lib()
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    ctx = hvo.Context()
    out = {"nq": nq, "nt": nt, "popcounts_per_call": nq * nt * 8}
    L.hvo_debug_match_rate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    for kind, name in ((0, "knn2"), (1, "matrix")):
        ms = C.c_float(0)
        rc = L.hvo_debug_match_rate(ctx.h, q.ctypes.data, nq, t.ctypes.data, nt, kind, 50, C.byref(ms))
        assert rc == 0
        out[name + "_resident_us"] = round(ms.value * 1e3, 2)
        out[name + "_resident_Gpopc_per_s"] = round(nq * nt * 8 / (ms.value * 1e-3) / 1e9, 1)
    for name, fn in (("knn2", lambda: ctx.hamming_knn2(q, t)), ("matrix", lambda: ctx.hamming_matrix(q, t))):
        fn(); t0 = time.perf_counter()
        for _ in range(20): fn()
        el = (time.perf_counter() - t0) / 20
        out[name + "_host_arrays_us"] = round(el * 1e6, 1)
    ctx.close()
    print(json.dumps(out))

if __name__ == "__main__":
    main()
