"""GPU parity of the guided ORB search (SearchByProjection core) and ComputeStereoFromRGBD."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BOUNDS = (0.0, 0.0, 640.0, 480.0)      # mnMinX, mnMinY, mnMaxX, mnMaxY for an undistorted 640x480 camera


def _scene(orc, synth, dx=3, dy=2, seed=0x5EED0003):
    g1 = synth.make_gray("std", seed)
    g2 = np.roll(np.roll(g1, dy, axis=0), dx, axis=1)           # "previous frame translated (+3,+2) px"
    o = orc.Orb()
    kp1, d1 = o.extract(g1)
    kp2, d2 = o.extract(g2)
    return kp1, d1, kp2, d2


@pytest.mark.parametrize("th,occupied_frac,dup", [(15, 0.0, False), (7, 0.2, False), (30, 0.0, True)])
def test_search_by_projection_parity(gpu_ctx, orc, synth, th, occupied_frac, dup):
    kp1, d1, kp2, d2 = _scene(orc, synth)
    rng = np.random.default_rng(4)
    n1 = len(kp1)
    scale = np.float32(1.2) ** kp1["octave"].astype(np.float32)
    q_u = kp1["x"] + 3 + rng.normal(0, 1.0, n1).astype(np.float32)      # projection of last-frame points into the current frame
    q_v = kp1["y"] + 2 + rng.normal(0, 1.0, n1).astype(np.float32)
    q_radius = (np.float32(th) * scale).astype(np.float32)
    q_min = (kp1["octave"] - 1).astype(np.int32); q_max = (kp1["octave"] + 1).astype(np.int32)
    q_ur = (q_u - 40.0 / rng.uniform(1, 4, n1)).astype(np.float32)
    q_blocks = (rng.uniform(size=n1) < 0.9).astype(np.uint8)
    t_uright = np.where(rng.uniform(size=len(kp2)) < 0.7, kp2["x"] - 40.0 / rng.uniform(1, 4, len(kp2)), -1).astype(np.float32)
    t_occ = (rng.uniform(size=len(kp2)) < occupied_frac).astype(np.uint8)
    qd = d1.copy()
    if dup:                      # many queries compete for the same features: exercises the sequential occupancy
        q_u[1::2] = q_u[0::2][: len(q_u[1::2])]; q_v[1::2] = q_v[0::2][: len(q_v[1::2])]
        qd[1::2] = qd[0::2][: len(qd[1::2])]
    args = (qd, q_u, q_v, q_radius, q_min, q_max, q_ur, kp1["angle"], q_blocks, kp2, t_uright, t_occ, d2, BOUNDS)
    no, io, do = orc.search_by_projection(*args, th_high=100, check_orientation=True)
    ng, ig, dg = gpu_ctx.search_by_projection(*args, th_high=100, check_orientation=True)
    assert no > 50
    assert ng == no and np.array_equal(ig, io)
    assert np.array_equal(dg[ig >= 0], do[io >= 0])
    no2, io2, _ = orc.search_by_projection(*args, th_high=100, check_orientation=False)
    ng2, ig2, _ = gpu_ctx.search_by_projection(*args, th_high=100, check_orientation=False)
    assert ng2 == no2 and np.array_equal(ig2, io2)


def test_search_by_projection_empty(gpu_ctx):
    n, idx, dist = gpu_ctx.search_by_projection(np.zeros((0, 32), np.uint8), [], [], [], [], [], [], [], [],
                                                np.zeros(0, gpu_ctx.extract_orb(np.zeros((0, 0), np.uint8))[0].dtype), [], [], np.zeros((0, 32), np.uint8), BOUNDS)
    assert n == 0 and len(idx) == 0


def test_stereo_from_rgbd(gpu_ctx, orc, synth):
    g, d = synth.make_frame("std", 0x5EED0002)
    kp, _ = orc.Orb().extract(g)
    bf = np.float32(40.0)
    uo, zo = orc.stereo_from_rgbd(kp, kp, d, float(np.float32(1.0) / np.float32(5000.0)), float(bf))
    ug, zg = gpu_ctx.stereo_from_rgbd(kp, kp, d, float(bf))
    assert np.array_equal(zg, zo) and np.array_equal(ug, uo)
    assert (zo > 0).mean() > 0.8 and np.all((zo == -1) | ((zo > 0) & (zo < 7)))     # gate 0 < d < 7 (Frame.cc:1955)


@pytest.mark.parametrize("th,occupied_frac,dup,ratio", [(4, 0.0, False, 0.8), (2, 0.3, False, 0.6), (8, 0.0, True, 0.9)])
def test_search_by_projection_map_parity(gpu_ctx, orc, synth, th, occupied_frac, dup, ratio):
    """local-map variant (ORBmatcher.cc:45-132): best / second best with the same-octave ratio test"""
    kp1, d1, kp2, d2 = _scene(orc, synth, seed=0x5EED0004)
    rng = np.random.default_rng(9)
    n1 = len(kp1)
    lvl = kp1["octave"].astype(np.int32)
    scale = np.float32(1.2) ** lvl.astype(np.float32)
    q_u = kp1["x"] + 3 + rng.normal(0, 1.5, n1).astype(np.float32)
    q_v = kp1["y"] + 2 + rng.normal(0, 1.5, n1).astype(np.float32)
    r = np.where(rng.uniform(size=n1) < 0.5, 2.5, 4.0).astype(np.float32)            # RadiusByViewingCos (:134-140)
    q_radius = (r * np.float32(th) * scale).astype(np.float32)
    q_min = (lvl - 1).astype(np.int32); q_max = lvl.copy()                            # GetFeaturesInArea(.., level-1, level)
    q_ur = (q_u - 40.0 / rng.uniform(1, 4, n1)).astype(np.float32)
    q_blocks = (rng.uniform(size=n1) < 0.8).astype(np.uint8)
    t_uright = np.where(rng.uniform(size=len(kp2)) < 0.7, kp2["x"] - 40.0 / rng.uniform(1, 4, len(kp2)), -1).astype(np.float32)
    t_occ = (rng.uniform(size=len(kp2)) < occupied_frac).astype(np.uint8)
    qd = d1.copy()
    if dup:
        q_u[1::2] = q_u[0::2][: len(q_u[1::2])]; q_v[1::2] = q_v[0::2][: len(q_v[1::2])]
        qd[1::2] = qd[0::2][: len(qd[1::2])]
    args = (qd, q_u, q_v, q_radius, q_min, q_max, q_ur, q_blocks, kp2, t_uright, t_occ, d2, BOUNDS)
    no, io, do = orc.search_by_projection_map(*args, th_high=100, nn_ratio=ratio)
    ng, ig, dg = gpu_ctx.search_by_projection_map(*args, th_high=100, nn_ratio=ratio)
    assert no > 30
    assert ng == no and np.array_equal(ig, io) and np.array_equal(dg[ig >= 0], do[io >= 0])
    # the ratio test must have rejected something the plain best-match rule accepts
    n_plain, _, _ = orc.search_by_projection_map(*args, th_high=100, nn_ratio=10.0)
    assert n_plain >= no


@pytest.mark.parametrize("th", [1.0, 3.0])
def test_search_by_projection_tracked(gpu_ctx, orc, synth, th):
    """SearchByProjection(F, vpMapPoints, th) from the tracker's own fields: RadiusByViewingCos, th and the level band are applied on the
    device (src/ORBmatcher.cc:55-70, 134-140); against the oracle's prologue + search core"""
    kp1, d1, kp2, d2 = _scene(orc, synth)
    rng = np.random.default_rng(11)
    n1 = len(kp1)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    px = (kp1["x"] + 3 + rng.normal(0, 1.0, n1)).astype(np.float32); py = (kp1["y"] + 2 + rng.normal(0, 1.0, n1)).astype(np.float32)
    pxr = (px - 40.0 / rng.uniform(1, 4, n1)).astype(np.float32)
    level = np.clip(kp1["octave"] + rng.integers(0, 2, n1), 0, 7).astype(np.int32)
    vcos = np.where(rng.uniform(size=n1) < 0.5, 0.9995, 0.99).astype(np.float32)      # both sides of RadiusByViewingCos' 0.998
    blocks = np.ones(n1, np.uint8)
    t_uright = np.where(rng.uniform(size=len(kp2)) < 0.7, kp2["x"] - 40.0 / rng.uniform(1, 4, len(kp2)), -1).astype(np.float32)
    t_occ = (rng.uniform(size=len(kp2)) < 0.1).astype(np.uint8)
    rad, lo, hi = orc.track_windows(level, vcos, th, sf)
    assert set(np.unique(np.round(rad / sf[level] / np.float32(th if th != 1.0 else 1.0), 3))) == {2.5, 4.0}
    no, io, do = orc.search_by_projection_map(d1, px, py, rad, lo, hi, pxr, blocks, kp2, t_uright, t_occ, d2, BOUNDS, th_high=100, nn_ratio=0.8)
    ng, ig, dg = gpu_ctx.search_by_projection_tracked(d1, px, py, pxr, level, vcos, blocks, th, kp2, t_uright, t_occ, d2, BOUNDS, th_high=100, nn_ratio=0.8)
    assert no > 50 and ng == no and np.array_equal(ig, io) and np.array_equal(dg[ig >= 0], do[io >= 0])
