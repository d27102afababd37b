#!/usr/bin/env python3
"""ms per kernel group of one stage mask on a resident batch (profile mode 2: groups timed with events, one after another).
    python tools/stage_kernels.py [mask=2] [B=4096]"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = np.empty((64, 480, 640), np.uint8); d = np.empty((64, 480, 640), np.uint16)
for k in range(64): g[k], d[k] = synth.make_frame("lowtex" if k % 4 == 3 else "std", 0x5EED1000 + k)
ctx = hvo.Context(max_batch=B)
ctx.batch_upload(g, d, repeat=B // 64)
for _ in range(2): ctx.batch_run(mask)
ctx.profile_enable(2)
acc = {}
for _ in range(3):
    ctx.batch_run(mask)
    for k, v in ctx.profile_last().items(): acc[k] = acc.get(k, 0) + v / 3
print(json.dumps({k: round(v, 3) for k, v in acc.items()}))
ctx.close()
