#!/usr/bin/env python3
"""Regenerates tests/golden/frontend_v1.npz.

PROVENANCE: the reference ships no tests, golden vectors or fixtures for this path and cannot be built
or run here (OpenCV 3.2 + contrib, Eigen, PCL absent) -- see DESIGN.md section 2, "parity unpinned".
These vectors are therefore outputs of THIS repository's CPU oracle (oracle/*.c) on seeded synthetic
inputs, not outputs of the reference.  They pin the oracle against drift (a change of the restated
arithmetic must be deliberate: re-run this script and review the diff) and give the GPU box a check that
does not depend on the oracle library being rebuilt identically.

    python tests/golden/make_golden.py          # rewrites frontend_v1.npz next to this file
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_oracle, load_synth  # noqa: E402

SEEDS = (0x5EED0002, 0x5EED1001)


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def build():
    orc, synth = load_oracle(), load_synth()
    orc.lib()
    out = {"seeds": np.array(SEEDS, np.uint64)}
    orb = orc.Orb()
    for i, seed in enumerate(SEEDS):
        g = synth.make_gray("std", seed)
        d = synth.make_depth(seed)
        kp, desc = orb.extract(g)
        out["orb%d_n" % i] = np.int64(len(kp))
        out["orb%d_kp_head" % i] = kp[:64].copy()                       # first 64 key points, all fields
        out["orb%d_desc_head" % i] = desc[:64].copy()
        out["orb%d_xy_octave_sha" % i] = sha(np.stack([kp["x"], kp["y"], kp["octave"].astype(np.float32)]))
        out["orb%d_desc_sha" % i] = sha(desc)
        kl, ldesc, fn = orc.line_extract(g)
        out["lsd%d_n" % i] = np.int64(len(kl))
        out["lsd%d_endpoints" % i] = np.stack([kl["sx"], kl["sy"], kl["ex"], kl["ey"]], 1).copy()
        out["lsd%d_desc_sha" % i] = sha(ldesc)
        out["lsd%d_linefn_head" % i] = fn[:16].copy()
        lab, pl = orc.peac(d)
        out["peac%d_planes" % i] = pl.copy()
        out["peac%d_label_hist" % i] = np.bincount((lab + 1).ravel(), minlength=8)[:8].astype(np.int64)
        out["peac%d_label_sha" % i] = sha(lab)
    # Hamming / kNN-2 on the two ORB descriptor sets
    g0 = synth.make_gray("std", SEEDS[0]); g1 = synth.make_gray("std", SEEDS[1])
    _, d0 = orb.extract(g0); _, d1 = orb.extract(g1)
    idx, dist = orc.hamming_knn2(d0[:256], d1)
    out["knn2_idx"] = idx.copy(); out["knn2_dist"] = dist.copy()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "frontend_v1.npz"), **build())
    print("wrote", os.path.join(HERE, "frontend_v1.npz"), os.path.getsize(os.path.join(HERE, "frontend_v1.npz")), "bytes")
