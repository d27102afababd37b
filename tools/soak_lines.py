#!/usr/bin/env python3
"""Differential soak of LSD + LBD (+ culling) on the periodic images of tools/soak_orb.py: exactly horizontal / vertical / diagonal edges (angle
ties in the growing, symmetric LBD bands whose comparisons are near-equalities -- where the one-ulp sqrt of round 4 showed), lone frames (async
growing) and batches (one-wave kernels).      python tools/soak_lines.py [images=200] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge
from soak_orb import image


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x11E5
    hvo = ge.package(); orc = ge.oracle()
    from test_lsd_gpu import check
    rng = np.random.default_rng(seed)
    imgs = [image(rng, 640, 480) for _ in range(n)]
    ref = [orc.line_extract(g) for g in imgs]
    bad = []
    def cmp(tag, i, r):
        try: check(r[0], r[1], r[2], *ref[i])
        except AssertionError as e: bad.append((tag, i, str(e)[:60]))
    ctx = hvo.Context()
    for i, g in enumerate(imgs):
        try: cmp("lone", i, ctx.extract_lsd(g))
        except Exception as e: bad.append(("lone", i, str(e)[:70]))
        try:                                               # Frame::cullingLine on top (merging of near-collinear segments, second LBD): periodic edges merge a lot
            ck_o, cd_o, cf_o = orc.cull_lines(g, *[ref[i][k] for k in (0, 2)])
            ck_g, cd_g, cf_g = ctx.extract_lsd(g, culled=True)
            check(ck_g, cd_g, cf_g, ck_o, cd_o, cf_o)
        except Exception as e: bad.append(("culled", i, str(e)[:70] or "assert"))
    ctx.close()
    for B in (6, 16, 48):
        ctx = hvo.Context(max_batch=B)
        for c0 in range(0, n, B):
            m = min(B, n - c0)
            ctx.batch_upload(np.stack(imgs[c0:c0 + m]), np.zeros((m, 480, 640), np.uint16)); ctx.batch_run(hvo.STAGE_LSD); res = ctx.batch_download(hvo.STAGE_LSD)
            for b in range(m):
                if res[b]["status"] != 0: bad.append(("batch%d status %d" % (B, res[b]["status"]), c0 + b, ""))
                else: cmp("batch%d" % B, c0 + b, (res[b]["kl"], res[b]["ldesc"], res[b]["linefn"]))
        ctx.close()
    nl = [len(r[0]) for r in ref]
    print("RESULT lines soak images=%d seed=%#x lines per image min/mean/max %d/%.0f/%d differing=%d %s" % (n, seed, min(nl), float(np.mean(nl)), max(nl), len(bad), bad[:8]))


if __name__ == "__main__":
    main()
