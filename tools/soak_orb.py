#!/usr/bin/env python3
"""Differential soak of the ORB pass on images built to produce TIES: periodic patterns (checkerboards, stripes, dot lattices) whose corners have
equal FAST scores and symmetric quadtree splits, flat regions, saturated blocks, a few grey levels only, plus odd sizes and quotas.  The HIP
path (fused level kernel and the separate kernels) against the CPU oracle (test infrastructure).   python tools/soak_orb.py [images=200] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge


def image(rng, w, h):
    y, x = np.mgrid[0:h, 0:w]
    kind = int(rng.integers(0, 6))
    p = int(rng.choice([5, 8, 13, 16, 31, 40])); lo, hi = int(rng.integers(0, 100)), int(rng.integers(140, 256))
    if kind == 0: g = np.where(((x // p) + (y // p)) % 2 == 0, lo, hi)                         # checkerboard
    elif kind == 1: g = np.where((x // p) % 2 == 0, lo, hi)                                    # stripes (no corners but at the borders)
    elif kind == 2: g = np.where(((x % p) < 3) & ((y % p) < 3), hi, lo)                        # dot lattice
    elif kind == 3: g = np.where(((x + y) // p + (x - y + 4096) // p) % 2 == 0, lo, hi)        # diagonal checkerboard
    elif kind == 4: g = rng.choice(np.array([lo, (lo + hi) // 2, hi]), size=(h // 4 + 1, w // 4 + 1)).repeat(4, 0).repeat(4, 1)[:h, :w]   # 4x4 blocks of three grey levels
    else: g = np.full((h, w), lo); g[h // 4: 3 * h // 4, w // 4: 3 * w // 4] = hi            # one rectangle
    g = g.astype(np.float64)
    if rng.uniform() < 0.3: g += rng.normal(0, float(rng.choice([1.0, 6.0])), g.shape)
    if rng.uniform() < 0.3: g[: h // 3] = g[: h // 3] * 0.2 + 100                               # a low-contrast band (the minThFAST fallback)
    return np.clip(np.round(g), 0, 255).astype(np.uint8)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x0B
    hvo = ge.package(); orc = ge.oracle()
    from test_orb_gpu import check_orb
    rng = np.random.default_rng(seed)
    bad = []; nk = []
    sizes = [(640, 480), (640, 480), (501, 397), (322, 240), (704, 200), (638, 479)]
    ctxs = {}
    for i in range(n):
        w, h = sizes[int(rng.integers(0, len(sizes)))]
        nf = int(rng.choice([1000, 1000, 300, 2500]))
        g = image(rng, w, h)
        kpo, do = orc.Orb(nfeatures=nf).extract(g)
        nk.append(len(kpo))
        for fused in ("1", "0"):
            key = (nf, fused)
            os.environ["HVO_ORB_FUSED"] = fused                    # (read when a context builds its plan for a geometry)
            if key not in ctxs: ctxs[key] = hvo.Context(orb_nfeatures=nf)
            try:
                kpg, dg = ctxs[key].extract_orb(g)
                check_orb(kpg, dg, kpo, do)
            except (AssertionError, Exception) as e:
                bad.append((i, "fused=" + fused, (w, h, nf), str(e)[:50]))
        if (i + 1) % 50 == 0: print("image", i + 1, "bad", len(bad), flush=True)
    for c in ctxs.values(): c.close()
    print("RESULT orb soak images=%d seed=%#x key points min/mean/max %d/%.0f/%d differing=%d %s" % (n, seed, min(nk), float(np.mean(nk)), max(nk), len(bad), bad[:8]))


if __name__ == "__main__":
    main()
