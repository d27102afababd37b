#!/usr/bin/env python3
"""One resident frame through hvo_batch_run N times (for rocprofv3 --kernel-trace --stats: per-kernel time of the latency case)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
kind = sys.argv[1] if len(sys.argv) > 1 else "std"
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 7
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g, d = synth.make_batch(kind, 0x5EED1000, B, 640, 480)
ctx = hvo.Context(max_batch=B); ctx.batch_upload(g, d)
for _ in range(20): ctx.batch_run(mask)
ctx.close()
