"""The C++ host-side mirror of the reference interfaces (include/hvo.hpp) compiles with g++ and, on
the GPU box, produces the same results as the Python path / the oracle."""
import os
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, PKG_DIR


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_mirror_compiles():
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-fsyntax-only", os.path.join(ROOT, "examples", "frontend_demo.cpp")])
    # and the C ABI header is plain C
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "hvo.h")])


@pytest.mark.gpu
def test_cpp_demo_matches_oracle(tmp_path, orc, synth):
    csrc = os.path.join(PKG_DIR, "csrc")
    exe = str(tmp_path / "demo")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "frontend_demo.cpp"),
                           "-L" + csrc, "-lhvo", "-Wl,-rpath," + csrc, "-o", exe])
    g, d = synth.make_frame("std", 0x5EED0002)
    g.tofile(tmp_path / "g.u8"); d.tofile(tmp_path / "d.u16")
    out = subprocess.check_output([exe, str(tmp_path / "g.u8"), str(tmp_path / "d.u16")]).decode().split()
    kp_o, d_o = orc.Orb().extract(g)
    kl_o, ld_o, _ = orc.line_extract(g)
    lab_o, pl_o = orc.peac(d)
    nm, _ = orc.match_nnr(d_o, d_o, 0.9)
    # "grid N hash linegrid N hash kp N desc H lines N ldesc H planes N labels H matches N"
    b = orc.image_bounds(640, 480, 535.4, 539.2, 320.1, 247.6, [0] * 5)
    _, gi = orc.assign_features_to_grid(kp_o, b)
    _, li, _ = orc.assign_lines_to_grid(kl_o, b)
    assert int(out[1]) == len(gi) and int(out[2], 16) == _fnv(gi.tobytes())
    assert int(out[4]) == len(li) and int(out[5], 16) == _fnv(li.tobytes())
    out = out[6:]
    assert int(out[1]) == len(kp_o) and int(out[3], 16) == _fnv(d_o.tobytes())
    assert int(out[5]) == len(kl_o) and int(out[7], 16) == _fnv(ld_o.tobytes())
    assert int(out[9]) == len(pl_o) and int(out[11], 16) == _fnv(lab_o.tobytes())
    assert int(out[13]) == nm
    # second line: "l3d GOOD vp N0 N1 N2 N3 best B clouds VALID cloudpts N normals N stream NKP HASH"
    out = out[14:]
    l3 = orc.lines_3d(kl_o, d, seed=7)
    vp = orc.vanishing_points(kl_o, seed=7)
    pcs, cloud = orc.plane_clouds(d, lab_o, pl_o, dist_th=0.05)
    sn = orc.surface_normals(d)
    assert int(out[1]) == int(l3["good"].sum())
    assert [int(x) for x in out[3:7]] == np.bincount(vp["vp_idx"], minlength=4).tolist()
    assert int(out[10]) == int(pcs["valid"].sum()) and int(out[12]) == len(cloud) and int(out[14]) == len(sn)
    assert int(out[16]) == len(kp_o) and int(out[17], 16) == _fnv(d_o.tobytes())
    # third line: the same frame through hvo::FrameStream with every stage of the Frame constructor (seed 7 + ticket 0)
    # "tail l3d GOOD vpbest B clouds VALID cloudpts N normals N ptitems N lnitems N"
    out = out[18:]
    assert out[0] == "tail" and int(out[2]) == int(l3["good"].sum()) and int(out[6]) == int(pcs["valid"].sum()) and int(out[8]) == len(cloud) and int(out[10]) == len(sn)
    assert int(out[12]) == len(gi) and int(out[14]) == len(li)
    # fourth line (round 5): "linetrack geom N HASH sbp N HASH" -- LSDmatcher::SearchByGeomNApearance / SearchByProjection of the frame against itself
    out = out[15:]
    fn_o = orc.line_extract(g)[2]
    ng, m12, _ = orc.lines_geom_match(ld_o, kl_o, ld_o, kl_o, b, desc_th=0.9)
    cs, ci, _ = orc.assign_lines_to_grid(kl_o, b)
    q = np.stack([kl_o["sx"] + np.float32(1.5), kl_o["sy"] - np.float32(0.5), kl_o["ex"] + np.float32(1.5), kl_o["ey"] - np.float32(0.5)], axis=1).astype(np.float32)
    ns, mi, _ = orc.search_lines_by_projection(q, kl_o, ld_o, np.ones(len(kl_o), np.uint8), kl_o, fn_o, ld_o, np.zeros(len(kl_o), np.uint8), cs, ci, b, 15.0)
    assert out[0] == "linetrack" and int(out[2]) == ng and int(out[3], 16) == _fnv(m12.astype(np.int32).tobytes())
    assert int(out[5]) == ns and int(out[6], 16) == _fnv(mi.astype(np.int32).tobytes()) and ns > 50
