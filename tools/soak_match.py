#!/usr/bin/env python3
"""Differential soak of the matching entry points on TIE-HEAVY inputs: descriptors drawn from a small pool with a few flipped bits, so that equal
distances (first minimum wins, lowest train index on ties, the ratio tests' equalities) occur in every call; key points on a coarse lattice so
that search windows hold many candidates and queries compete for targets.  Every result against the CPU oracle (test infrastructure).
    python tools/soak_match.py [trials=300] [seed]"""
import importlib, os, sys
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


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xC0DE
    hvo = ge.package(); orc = ge.oracle()
    rng = np.random.default_rng(seed)
    ctx = hvo.Context()
    kpt = ctx.extract_orb(np.zeros((0, 0), np.uint8))[0].dtype
    bounds = (0.0, 0.0, 640.0, 480.0)
    bad = []
    def chk(name, t, ok):
        if not ok: bad.append((t, name))
    for t in range(trials):
        pool = rng.integers(0, 256, (int(rng.integers(2, 24)), 32), dtype=np.uint8)
        n1 = int(rng.integers(0, 260)) if t % 7 else int(rng.integers(0, 3)); n2 = int(rng.integers(0, 260)) if t % 11 else int(rng.integers(0, 3))
        flips = int(rng.integers(0, 4))
        d1 = descs(rng, n1, pool, flips); d2 = descs(rng, n2, pool, flips)
        if n1 and n2:
            chk("hamming_matrix", t, np.array_equal(ctx.hamming_matrix(d1, d2), orc.hamming_matrix(d1, d2)))
            if n2 >= 2:
                ig, dg = ctx.hamming_knn2(d1, d2); io, do = orc.hamming_knn2(d1, d2)
                chk("knn2", t, np.array_equal(ig, io) and np.array_equal(dg, do))
        nnr = float(rng.choice([0.6, 0.8, 0.95, 1.0]))
        ng, mg = ctx.match_nnr(d1, d2, nnr); no, mo = orc.match_nnr(d1, d2, nnr)
        chk("match_nnr", t, ng == no and np.array_equal(mg, mo))
        TH = float(rng.choice([10.0, 50.0, 100.0])); nr = float(rng.choice([0.7, 0.9, 1.0]))
        for mutual in (False, True):
            ng, mg = ctx.frame_bf_match(d1, d2, TH, nr, mutual); no, mo = orc.frame_bf_match(d1, d2, TH, nr, mutual)
            chk("bf_match mutual=%s" % mutual, t, ng == no and np.array_equal(mg, mo))
        # guided searches: targets on a lattice, queries around them
        nt = max(n2, 1); nq = n1
        t_kp = np.zeros(nt, kpt)
        step = float(rng.choice([4.0, 9.0, 23.0]))
        t_kp["x"] = (rng.integers(0, int(640 / step), nt) * step).astype(np.float32); t_kp["y"] = (rng.integers(0, int(480 / step), nt) * step).astype(np.float32)
        t_kp["octave"] = rng.integers(0, 8, nt); t_kp["angle"] = (rng.integers(0, 36, nt) * 10).astype(np.float32)
        t_desc = descs(rng, nt, pool, flips)
        src = rng.integers(0, nt, nq) if nq else np.zeros(0, np.int64)
        q_u = (t_kp["x"][src] + rng.integers(-6, 7, nq)).astype(np.float32); q_v = (t_kp["y"][src] + rng.integers(-6, 7, nq)).astype(np.float32)
        if nq: q_u[rng.uniform(size=nq) < 0.05] = np.float32(-5.0)                       # out of bounds
        lvl = t_kp["octave"][src].astype(np.int32) + rng.integers(-1, 2, nq).astype(np.int32)
        q_radius = (np.float32(rng.choice([3.0, 7.0, 15.0])) * (np.float32(1.2) ** np.clip(lvl, 0, 7).astype(np.float32))).astype(np.float32)
        q_min = (lvl - 1).astype(np.int32); q_max = (lvl + 1).astype(np.int32)
        q_ur = np.where(rng.uniform(size=nq) < 0.7, q_u - 40.0 / rng.uniform(1, 4, nq), -1).astype(np.float32)
        q_ang = (rng.integers(0, 36, nq) * 10).astype(np.float32)
        q_blocks = (rng.uniform(size=nq) < 0.85).astype(np.uint8)
        t_ur = np.where(rng.uniform(size=nt) < 0.7, t_kp["x"] - 40.0 / rng.uniform(1, 4, nt), -1).astype(np.float32)
        t_occ = (rng.uniform(size=nt) < float(rng.choice([0.0, 0.2]))).astype(np.uint8)
        q_desc = descs(rng, nq, pool, flips)
        th_high = int(rng.choice([50, 100]))
        for co in (True, False):
            a = (q_desc, q_u, q_v, q_radius, q_min, q_max, q_ur, q_ang, q_blocks, t_kp, t_ur, t_occ, t_desc, bounds)
            no, io, do = orc.search_by_projection(*a, th_high=th_high, check_orientation=co)
            ng, ig, dg = ctx.search_by_projection(*a, th_high=th_high, check_orientation=co)
            chk("sbp orient=%s" % co, t, ng == no and np.array_equal(ig, io) and np.array_equal(dg[ig >= 0], do[io >= 0]))
        a = (q_desc, q_u, q_v, q_radius, q_min, q_max, q_ur, q_blocks, t_kp, t_ur, t_occ, t_desc, bounds)
        ratio = float(rng.choice([0.6, 0.8, 1.0]))
        no, io, do = orc.search_by_projection_map(*a, th_high=th_high, nn_ratio=ratio)
        ng, ig, dg = ctx.search_by_projection_map(*a, th_high=th_high, nn_ratio=ratio)
        chk("sbp_map", t, ng == no and np.array_equal(ig, io) and np.array_equal(dg[ig >= 0], do[io >= 0]))
        if (t + 1) % 50 == 0: print("trial", t + 1, "bad", len(bad), flush=True)
    ctx.close()
    print("RESULT match soak trials=%d seed=%#x differing=%d %s" % (trials, seed, len(bad), bad[:10]))


if __name__ == "__main__":
    main()
