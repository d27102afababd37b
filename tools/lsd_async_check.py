#!/usr/bin/env python3
"""k_lsd_grow_async (lsd_async.inc) against the oracle and against the one-wave kernels: parity per frame, the kernel's own counters
(regions, speculated, valid, failed, swallowed, grown at the frontier) and the time of the growing kernel, for W = 0 (off), 4, 8, 16, 32.
Diagnostic tool:  python tools/lsd_async_check.py [W,W,...] [640x480|1280x960] [frames]"""
import ctypes, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    hvo = ge.package(); orc = ge.oracle(); synth = importlib.import_module("hvo_amd.synth")
    Ws = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4,8,16,32").split(",")]
    w, h = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "640x480").split("x"))
    nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    kinds = [("std", 0x5EED0002), ("std", 0x5EED1001), ("lowtex", 0x5EED0001), ("std", 9), ("std", 0x5EED1003), ("std", 77), ("lowtex", 5), ("std", 1234)][:nfr]
    frames = [synth.make_gray(k, s, w, h) for k, s in kinds]
    ref = [orc.line_extract(g) for g in frames]
    L = hvo.lib(); L.hvo_debug_lsd_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]; L.hvo_debug_lsd_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    out = {}
    for W in Ws:
        os.environ["HVO_LSD_ASYNC"] = str(W)
        for B in (1, len(frames)):
            ctx = hvo.Context(max_batch=B)
            g = np.stack(frames[:B]); ctx.batch_upload(g, np.zeros((B, h, w), np.uint16))
            ctx.batch_run(hvo.STAGE_LSD); res = ctx.batch_download(hvo.STAGE_LSD)
            bad = []
            for b in range(B):
                kl_o, d_o, fn_o = ref[b]
                ok = res[b]["status"] == 0 and len(res[b]["kl"]) == len(kl_o) and all(np.array_equal(res[b]["kl"][f], kl_o[f]) for f in kl_o.dtype.names) \
                    and np.array_equal(res[b]["ldesc"], d_o) and np.array_equal(res[b]["linefn"], fn_o)
                if not ok:
                    bad.append((b, res[b]["status"], len(res[b]["kl"]), len(kl_o)))
                    if len(res[b]["kl"]) == len(kl_o):
                        diff = {f: np.flatnonzero(res[b]["kl"][f] != kl_o[f])[:6].tolist() for f in kl_o.dtype.names if not np.array_equal(res[b]["kl"][f], kl_o[f])}
                        print("   frame %d differs in" % b, diff, "desc rows", np.flatnonzero((res[b]["ldesc"] != d_o).any(axis=1))[:6].tolist(), flush=True)
                        i = next(iter(diff.values()))[0] if diff else 0
                        print("   e.g. line", i, {f: (res[b]["kl"][f][i].item(), kl_o[f][i].item()) for f in ("sx", "sy", "ex", "ey", "response", "class_id")}, flush=True)
            st = (ctypes.c_longlong * 8)(); L.hvo_debug_lsd_stats(ctx.h, 0, st)
            ac = (ctypes.c_uint * 256)(); actl = list(ac) if W and L.hvo_debug_lsd_async(ctx.h, 0, ac) == 0 and not print('   ctl:', dict(zip(('lock','F','G','done','nseg','flags','abort'), list(ac)[:7])), 'ticks disp/grow/wait/head/tail/front, turns, gap', list(ac)[16:24], 'xcd, foreign', list(ac)[32:34], flush=True) else None
            ctx.profile_enable(2)
            ts = []
            for _ in range(3): ctx.batch_run(hvo.STAGE_LSD); ts.append(ctx.profile_last().get("lsd_grow", -1))
            ctx.profile_enable(0)
            out["W%d_B%d" % (W, B)] = dict(parity_failures=bad, lsd_grow_ms=[round(t, 3) for t in ts], frame0_counters=list(st))
            print("W=%d B=%d: parity %s, lsd_grow %s ms, frame 0 counters %s" % (W, B, "OK" if not bad else "FAIL %s" % bad, [round(t, 3) for t in ts], list(st)), flush=True)
            ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
