#!/usr/bin/env python3
"""Per-stage latency of small resident batches (B = 1, 4, 32, 64) and of the single-frame entry points, with the
CPU oracle's per-stage time on the same host beside it.  Diagnostic tool (not part of the bench contract)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
    kind = sys.argv[1] if len(sys.argv) > 1 else "std"
    W = int(sys.argv[2]) if len(sys.argv) > 3 else 640; H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
    sizes = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (1, 4, 32, 64)
    out = {}
    for B in sizes:
        g, d = synth.make_batch(kind, 0x5EED1000, B, W, H)
        ctx = hvo.Context(max_batch=B)
        ctx.batch_upload(g, d)
        row = {}
        for name, mask in (("orb", 1), ("lsd", 2), ("planes", 4), ("all", 7)):
            for _ in range(2): ctx.batch_run(mask)
            t0 = time.perf_counter(); n = 5
            for _ in range(n): ctx.batch_run(mask)
            row[name + "_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
        ctx.profile_enable(2); ctx.batch_run(7); row["kernels_ms"] = {k: round(v, 3) for k, v in ctx.profile_last().items()}; ctx.profile_enable(0)
        out["B%d" % B] = row
        if B == 1:
            for _ in range(2): ctx.extract_orb(g[0]); ctx.extract_lsd(g[0]); ctx.compute_planes(d[0])      # (first calls load code objects)
            t0 = time.perf_counter()
            for _ in range(5): ctx.extract_orb(g[0])
            row["extract_orb_call_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
            t0 = time.perf_counter()
            for _ in range(5): ctx.extract_lsd(g[0])
            row["extract_lsd_call_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
            t0 = time.perf_counter()
            for _ in range(5): ctx.compute_planes(d[0])
            row["compute_planes_call_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        ctx.close()
    orc = ge.oracle(); orb = orc.Orb()
    g, d = synth.make_batch(kind, 0x5EED1000, 4, W, H)
    cpu = {}
    for name, fn in (("orb", lambda i: orb.extract(g[i])), ("lsd", lambda i: orc.line_extract(g[i])), ("planes", lambda i: orc.peac(d[i]))):
        fn(0); t0 = time.perf_counter()
        for i in range(4): fn(i)
        cpu[name + "_ms"] = round((time.perf_counter() - t0) / 4 * 1e3, 3)
    out["cpu_oracle_1thread"] = cpu; out["host_cores"] = os.cpu_count(); out["kind"] = kind
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
