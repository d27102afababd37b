#!/usr/bin/env python3
"""HBM footprint of a resident batch: free memory before / after creating the context and uploading."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
g, d = synth.make_batch("std", 1, 16, W, H)
torch.cuda.init()
f0, tot = torch.cuda.mem_get_info()
for B in ((1024, 2048) if W > 640 else (1024, 4096, 8192)):
    ctx = hvo.Context(max_batch=B, orb_nfeatures=2000 if W > 640 else 1000); ctx.batch_upload(g, d, repeat=B // 16); ctx.batch_run()
    f1, _ = torch.cuda.mem_get_info()
    print("batch %d: %.1f GB resident, %.2f MB per frame (device total %.0f GB)" % (B, (f0 - f1) / 1e9, (f0 - f1) / B / 1e6, tot / 1e9))
    ctx.close()
