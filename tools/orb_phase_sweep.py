#!/usr/bin/env python3
"""(needs a timing build: make -C .../csrc DEFS=-DHVO_TIMING_KNOBS -- the phase-skip mask is compiled out of the product library) ms per 8192-frame launch group of the ORB stage under HVO_LT_SKIP / HVO_ORB_TPW settings (timing experiments for orb_level.hip;
results with a skip mask are not valid outputs).  usage: python tools/orb_phase_sweep.py [batch]"""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, sys, json
sys.path.insert(0, %r)
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
B = %d
import numpy as np
g = np.empty((64, 480, 640), np.uint8); d = np.empty((64, 480, 640), np.uint16)
for k in range(64):
    g[k], d[k] = synth.make_frame("lowtex" if k %% 4 == 3 else "std", 0x5EED1000 + k)
ctx = hvo.Context(max_batch=B)
ctx.batch_upload(g, d, repeat=B // 64)
for _ in range(2): ctx.batch_run(hvo.STAGE_ORB)
ctx.profile_enable(2)
acc = {}
for _ in range(3):
    ctx.batch_run(hvo.STAGE_ORB)
    for k, v in ctx.profile_last().items(): acc[k] = acc.get(k, 0) + v / 3
print(json.dumps(acc))
ctx.close()
"""
def run(env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, B)], env=e, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else {"error": p.stderr[-300:]}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
import itertools
envs = [{}] + [{"HVO_ORB_NW": str(nw), "HVO_ORB_TPW": str(t)} for nw, t in ((1, 8), (1, 16), (1, 32), (2, 8), (2, 16), (4, 2), (4, 8))]
if len(sys.argv) > 2: envs = [dict(kv.split("=") for kv in a.split(",")) if a != "-" else {} for a in sys.argv[2:]]
for env in envs:
    r = run(env)
    print("%-44s %s" % (env, {k: round(v, 3) for k, v in r.items() if k.startswith("orb") or k == "error"}), flush=True)
