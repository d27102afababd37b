import importlib, sys, time
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
g, d = synth.make_batch("std", 0x5EED1000, 1, 640, 480)
ctx = hvo.Context(max_batch=1); ctx.batch_upload(g, d)
for _ in range(3): ctx.batch_run(4)
ctx.profile_enable(2); ctx.batch_run(4); print(ctx.profile_last()); print(ctx.peac_stats(0))
