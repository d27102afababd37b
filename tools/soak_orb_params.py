#!/usr/bin/env python3
"""ORB under other extractor parameters than the TUM yaml's (ORBextractor's constructor takes them: src/ORBextractor.cc:408-468): scale factor,
number of levels, the two FAST thresholds, the quota -- level geometry, quotas per level, cell grids and tile plans all follow from them.
Against the CPU oracle (test infrastructure).   python tools/soak_orb_params.py [images per setting=6]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    hvo = ge.package(); orc = ge.oracle(); synth = importlib.import_module("hvo_amd.synth")
    from test_orb_gpu import check_orb
    from soak_textures import frame
    rng = np.random.default_rng(0x0A8A)
    imgs = [synth.make_gray("std", 0x5EED9000 + i) for i in range(per // 2)] + [frame(rng)[0] for _ in range(per - per // 2)]
    bad = []; ran = 0
    for sf in (1.1, 1.2, 1.33, 1.5, 2.0):
        for nl in (1, 3, 5, 8):
            for (ini, mn) in ((20, 7), (12, 5), (40, 20)):
                for nf in (500, 1500):
                    for fused in ("1", "0"):
                        os.environ["HVO_ORB_FUSED"] = fused
                        try:
                            ctx = hvo.Context(orb_nfeatures=nf, orb_scale_factor=sf, orb_nlevels=nl, orb_ini_th_fast=ini, orb_min_th_fast=mn)
                        except Exception as e:
                            bad.append((sf, nl, ini, mn, nf, fused, "create: " + str(e)[:40])); continue
                        o = orc.Orb(nfeatures=nf, scale_factor=sf, nlevels=nl, ini_th=ini, min_th=mn)
                        try:
                            for i, g in enumerate(imgs if fused == "1" else imgs[:2]):
                                kpo, do = o.extract(g)
                                try:
                                    kpg, dg = ctx.extract_orb(g); check_orb(kpg, dg, kpo, do); ran += 1
                                except Exception as e:
                                    bad.append((sf, nl, ini, mn, nf, fused, i, str(e)[:50]))
                        finally:
                            ctx.close()
    from collections import Counter
    # a pyramid whose smallest level has no 30-pixel FAST cell between its borders (nCols = width / 30 = 0: a division by zero in the reference,
    # src/ORBextractor.cc:782-785) is rejected with HVO_ERR_UNSUPPORTED: expected, counted apart
    rej = [b for b in bad if "status -4" in b[-1] and 640.0 / b[0] ** (b[1] - 1) - 32 < 30]
    bad = [b for b in bad if b not in rej]
    print("rejected as unsupported (degenerate pyramids):", dict(Counter((b[0], b[1]) for b in rej)))
    print("RESULT orb params soak: %d extractions over 5 scale factors x 4 level counts x 3 threshold pairs x 2 quotas x both paths, differing=%d %s" % (ran, len(bad), bad[:8]))


if __name__ == "__main__":
    main()
