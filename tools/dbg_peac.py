import sys, importlib, numpy as np
sys.path.insert(0, '/root/repo'); import __graft_entry__ as ge
hvo = ge.package(); orc = ge.oracle(); synth = importlib.import_module("hvo_amd.synth")
gray, depth = synth.make_batch("std", 0x5EED1000, 4)
ref = [orc.peac(depth[b]) for b in range(4)]
print("oracle", [r[1]["n_points"].tolist() for r in ref])
if len(sys.argv) > 1:   # run a big-geometry context first (like the test order)
    d2 = synth.make_depth(3, 1280, 960); c2 = hvo.Context(); c2.compute_planes(d2); c2.close()
for B in (4, 2):
    ctx = hvo.Context(max_batch=B)
    for rep in range(3):
        ctx.batch_upload(gray[:B], depth[:B]); ctx.batch_run(hvo.STAGE_PLANES); res = ctx.batch_download(hvo.STAGE_PLANES)
        print("B", B, "rep", rep, [r["planes"]["n_points"].tolist() for r in res], [int((res[b]["labels"] != ref[b][0]).sum()) for b in range(B)], [r["status"] for r in res])
    for b in range(B):
        l, p = ctx.compute_planes(depth[b]); print(" single", b, p["n_points"].tolist(), int((l != ref[b][0]).sum()))
    ctx.close()
