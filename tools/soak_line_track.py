#!/usr/bin/env python3
"""Differential soak of the line tracker's two calls (csrc/line_track.inc) on inputs built to collide: key lines drawn on a coarse lattice of
positions and a handful of directions (windows hold many candidates, cosines sit at the 0.96 / cos 10 / cos 20 degree gates, lengths at the 0.75
ratio), descriptors from a small pool with a few flipped bits (equal distances: the visit order of GetFeaturesInAreaForLine decides), queries
that compete for the same current lines.  Every result against the CPU oracle (test infrastructure).
    python tools/soak_line_track.py [trials=200] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def descs(rng, n, pool, flips):
    d = pool[rng.integers(0, len(pool), n)].copy()
    for i in range(n):
        for _ in range(int(rng.integers(0, flips + 1))):
            b = int(rng.integers(0, 256)); d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def lines(rng, n, dt, step, ndir):
    kl = np.zeros(n, dt)
    cx = (rng.integers(2, int(640 / step) - 2, n) * step).astype(np.float32); cy = (rng.integers(2, int(480 / step) - 2, n) * step).astype(np.float32)
    ang = rng.integers(0, ndir, n) * (np.pi / ndir) + rng.choice([0.0, 0.0, 0.17, 0.28, 0.35], n)          # ~10, 16, 20 degrees off a lattice direction
    half = rng.choice([12.0, 16.0, 21.3, 30.0, 45.0], n).astype(np.float32) / 2
    dx = (np.cos(ang) * half).astype(np.float32); dy = (np.sin(ang) * half).astype(np.float32)
    kl["sx"] = cx - dx; kl["sy"] = cy - dy; kl["ex"] = cx + dx; kl["ey"] = cy + dy
    for a, b in (("sox", "sx"), ("soy", "sy"), ("eox", "ex"), ("eoy", "ey")): kl[a] = kl[b]
    kl["length"] = np.sqrt((kl["ex"] - kl["sx"]).astype(np.float64) ** 2 + (kl["ey"] - kl["sy"]).astype(np.float64) ** 2).astype(np.float32)
    kl["pt_x"] = (kl["sx"] + kl["ex"]) / 2; kl["pt_y"] = (kl["sy"] + kl["ey"]) / 2; kl["class_id"] = np.arange(n)
    sx, sy, ex, ey = (kl[k].astype(np.float64) for k in ("sx", "sy", "ex", "ey"))
    l0 = sy - ey; l1 = ex - sx; l2 = sx * ey - sy * ex; nrm = np.sqrt(l0 * l0 + l1 * l1); nrm[nrm == 0] = 1
    return kl, np.stack([l0 / nrm, l1 / nrm, l2 / nrm], axis=1)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x11AE
    hvo = ge.package(); orc = ge.oracle()
    rng = np.random.default_rng(seed)
    ctx = hvo.Context()
    bounds = np.array([0.0, 640.0, 0.0, 480.0], np.float32)
    bad = []
    tot_geom = tot_sbp = 0
    for t in range(trials):
        pool = rng.integers(0, 256, (int(rng.integers(2, 12)), 32), dtype=np.uint8)
        flips = int(rng.integers(0, 3)); step = float(rng.choice([9.0, 17.0, 31.0])); ndir = int(rng.choice([2, 4, 9]))
        n1 = int(rng.integers(1, 220)); n2 = int(rng.integers(0, 220)) if t % 9 else int(rng.integers(0, 3))
        kl1, _ = lines(rng, n1, hvo.KEYLINE_DT, step, ndir); kl2, fn2 = lines(rng, n2, hvo.KEYLINE_DT, step, ndir)
        if n2 and t % 5 == 0: kl2["sx"][rng.integers(0, n2, max(1, n2 // 10))] = 0          # SearchByGeomNApearance's startPointX == 0 rule
        d1 = descs(rng, n1, pool, flips); d2 = descs(rng, n2, pool, flips)
        hm = None if t % 3 == 0 else (rng.uniform(size=n1) < 0.8).astype(np.uint8)
        th = float(rng.choice([0.7, 0.9, 1.0, 1.05]))
        g = ctx.match_lines_geom(d1, kl1, d2, kl2, bounds, desc_th=th, last_has_mapline=hm)
        o = orc.lines_geom_match(d1, kl1, d2, kl2, bounds, desc_th=th, last_has_mapline=hm)
        if not (g[0] == o[0] and np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2])): bad.append((t, "geom"))
        tot_geom += o[0]
        if n2 == 0: continue
        cs, ci = ctx.assign_lines_to_grid(kl2, bounds)
        nq = int(rng.integers(1, 260))
        src = rng.integers(0, n2, nq)
        j = rng.integers(-4, 5, (nq, 4)).astype(np.float32) * np.float32(rng.choice([0.5, 1.0, 2.5]))
        q = np.stack([kl2["sx"][src], kl2["sy"][src], kl2["ex"][src], kl2["ey"][src]], axis=1).astype(np.float32) + j
        if nq > 4: q[0] = q[0][[0, 1, 0, 1]]; q[1] = (-300, -300, -200, -250); q[2] = (900, 100, 1000, 130)
        qi = rng.integers(0, n1, nq); qkl = kl1[qi]; qd = descs(rng, nq, pool, flips)
        blocks = (rng.uniform(size=nq) < float(rng.choice([0.0, 0.5, 1.0]))).astype(np.uint8)
        occ = (rng.uniform(size=n2) < float(rng.choice([0.0, 0.15]))).astype(np.uint8)
        r = float(rng.choice([2.0, 7.0, 15.0, 40.0]))
        g = ctx.search_lines_by_projection(q, qkl, qd, blocks, kl2, fn2, d2, occ, cs, ci, bounds, r)
        o = orc.search_lines_by_projection(q, qkl, qd, blocks, kl2, fn2, d2, occ, cs, ci, bounds, r)
        if not (g[0] == o[0] and np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2])): bad.append((t, "sbp"))
        tot_sbp += o[0]
    ctx.close()
    print("RESULT trials=%d seed=%#x geom_accepted=%d sbp_matches=%d differing=%d %s" % (trials, seed, tot_geom, tot_sbp, len(bad), bad[:8]))


if __name__ == "__main__":
    main()
