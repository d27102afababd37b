#!/usr/bin/env python3
"""Differential soak: N fresh synthetic frames through the resident-batch HIP path, every stage compared with the CPU
oracle (test infrastructure) frame by frame.      python tools/soak.py [--frames 512] [--kind std] [--seed 0xC0FFEE00]
Prints the indices of frames that differ in any bit-exact quantity (keypoints, descriptors, key-line structure,
LBD bytes, plane labels) or beyond 1e-4 in a float one."""
import argparse, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--kind", default="std")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xC0FFEE00)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--crop", default="", help="HxW: crop the generated frames to this size (odd geometries)")
    ap.add_argument("--tail", action="store_true", help="also the rest of the Frame constructor (3-D lines, vanishing points, plane clouds / refit / normals, grids) against the oracle")
    args = ap.parse_args()
    hvo = ge.package(); orc = ge.oracle()
    synth = importlib.import_module("hvo_amd.synth")
    o = orc.Orb()
    ctx = hvo.Context(max_batch=args.chunk)
    mask = hvo.STAGE_ALL
    if args.tail:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from test_tail_gpu import check_tail
        mask = hvo.STAGE_ALL | hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS
        ctx.set_tail_params(seed=1000)
    bad, t0 = [], time.time()
    for c0 in range(0, args.frames, args.chunk):
        n = min(args.chunk, args.frames - c0)
        gray, depth = synth.make_batch(args.kind, args.seed + c0, n, args.width, args.height)
        if args.crop:
            ch, cw = (int(v) for v in args.crop.lower().split("x"))
            gray = np.ascontiguousarray(gray[:, :ch, :cw]); depth = np.ascontiguousarray(depth[:, :ch, :cw])
        ctx.batch_upload(gray, depth); ctx.batch_run(mask); res = ctx.batch_download(hvo.STAGE_ALL)
        if args.tail: ctx.batch_download_tail(mask, res)
        for b in range(n):
            why = []
            if res[b]["status"] != 0: why.append("status %d" % res[b]["status"])
            kl_o, d_o, fn_o = orc.line_extract(gray[b])
            if len(res[b]["kl"]) != len(kl_o) or not np.array_equal(res[b]["ldesc"], d_o) or not np.array_equal(res[b]["kl"]["num_pixels"], kl_o["num_pixels"]): why.append("lines")
            elif not all(np.allclose(res[b]["kl"][f], kl_o[f], rtol=0, atol=1e-4) for f in ("sx", "sy", "ex", "ey", "angle", "length")): why.append("line floats")
            kp_o, dd_o = o.extract(gray[b])
            if len(res[b]["kp"]) != len(kp_o) or not np.array_equal(res[b]["desc"], dd_o) or not np.array_equal(res[b]["kp"]["x"], kp_o["x"]) or not np.array_equal(res[b]["kp"]["y"], kp_o["y"]): why.append("orb")
            lo, po = orc.peac(depth[b])
            if not np.array_equal(res[b]["labels"], lo) or len(res[b]["planes"]) != len(po): why.append("planes")
            elif len(po) and not np.allclose(res[b]["planes"]["normal"], po["normal"], rtol=1e-9, atol=1e-12): why.append("plane floats")
            if args.tail and not why:
                try: check_tail(res[b], depth[b], orc, 1000 + b, (0.0, float(gray.shape[2]), 0.0, float(gray.shape[1])))
                except AssertionError as e: why.append("tail: %s" % (str(e)[:80] or "assert"))
            if why: bad.append((c0 + b, why))
        print("frames %d..%d done, %d differ so far, %.0f s" % (c0, c0 + n - 1, len(bad), time.time() - t0), flush=True)
    ctx.close()
    print("RESULT kind=%s seed=%#x %dx%d%s frames=%d differing=%d %s" % (args.kind, args.seed, args.width, args.height, (" crop " + args.crop) if args.crop else "", args.frames, len(bad), bad[:10]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
