"""Vanishing-point clustering of the key lines (reference src/Frame.cc:442-778): oracle known answers on CPU,
HIP path against the oracle on the GPU.  sin / cos / atan / acos / atan2 are the host's in the oracle and the device's in the
HIP path, so continuous values are compared with a tolerance; the discrete outputs (clusters) must agree."""
import numpy as np
import pytest


def _lines(orc, synth, kind, seed):
    g = synth.make_gray(kind, seed)
    r = orc.line_extract(g)
    return r[0] if isinstance(r, tuple) else r


def _box_lines(orc):
    """key lines of a synthetic Manhattan scene: three pencils of lines through three orthogonal vanishing points"""
    fx, fy, cx, cy = 535.4, 539.2, 320.1, 247.6
    R = np.linalg.qr(np.array([[0.8, 0.1, 0.5], [-0.2, 0.9, 0.3], [-0.4, -0.3, 0.85]]))[0]
    vps = [(R[0, k] / R[2, k] * fx + cx, R[1, k] / R[2, k] * fy + cy) for k in range(3)]
    rng = np.random.RandomState(5)
    kl = np.zeros(60, orc.KEYLINE_DT)
    for i in range(60):
        vx, vy = vps[i % 3]
        mx, my = rng.uniform(40, 600), rng.uniform(40, 440)
        d = np.array([vx - mx, vy - my]); d /= np.linalg.norm(d)
        L = rng.uniform(20, 60)
        kl["sx"][i], kl["sy"][i] = mx - d[0] * L, my - d[1] * L
        kl["ex"][i], kl["ey"][i] = mx + d[0] * L, my + d[1] * L
    return kl, R


def test_vp_constants_and_params(orc):
    L = orc.lib()
    assert L.orc_vp_iterations() == 105                         # 105 x 360 = 37 800 hypotheses (src/Frame.cc:541)
    kl = np.zeros(2, orc.KEYLINE_DT)
    kl["sx"], kl["sy"], kl["ex"], kl["ey"] = [0, 10], [0, 0], [10, 10], [0, 10]
    assert orc.vanishing_points(kl[:1]) is None                  # NL > 1 (src/Frame.cc:328)
    o = orc.vanishing_points(kl, seed=3)
    assert o["vp_idx"].shape == (2,) and o["grid"].shape == (90, 360)
    assert not o["grid"][0].any() and not o["grid"][:, 0].any() and not o["grid"][89].any() and not o["grid"][:, 359].any()


def test_vp_manhattan_scene(orc):
    """three pencils of 20 lines each: the best hypothesis is the scene's frame (up to order and sign) and the lines join
    the clusters of their own vanishing points"""
    kl, R = _box_lines(orc)
    o = orc.vanishing_points(kl, seed=11, th_angle=3.0 / 180 * np.pi)
    V = o["vps"]
    assert np.allclose(np.linalg.norm(V, axis=1), 1, atol=1e-12)
    assert abs(V[0] @ V[1]) < 1e-9 and abs(V[0] @ V[2]) < 1e-9
    M = np.abs(V @ R)                                            # |cos| between found and true directions
    assert (M.max(1) > np.cos(np.deg2rad(3))).all()
    owner = M.argmax(0)                                          # true direction k -> found cluster
    got = o["vp_idx"]
    assert (got < 3).mean() > 0.9
    own = np.array([owner[i % 3] for i in range(len(kl))])
    assert (got[got < 3] == own[got < 3]).mean() > 0.9           # (a line may point at a second vanishing point more closely)
    assert np.array_equal(orc.vanishing_points(kl, seed=11, th_angle=3.0 / 180 * np.pi)["vp_idx"], got)    # reproducible under a seed


def _parallel_lines(orc, n=3):
    """n exactly horizontal key lines: every pair meets at infinity (z == 0 exactly)"""
    kl = np.zeros(n, orc.KEYLINE_DT)
    for i in range(n):
        kl["sx"][i], kl["ex"][i] = 10.0 + i, 200.0 + 3 * i
        kl["sy"][i] = kl["ey"][i] = 40.0 + 25 * i
    return kl


def test_vp_all_pairs_at_infinity_terminates(orc):
    """the reference redraws such pairs for ever; the bounded rule gives zero hypotheses, score 0, no structure lines"""
    kl = _parallel_lines(orc)
    o = orc.vanishing_points(kl, seed=5, want_scores=True)
    assert o["score"] == 0 and o["best"] == 0 and not o["scores"].any() and not o["vps"].any()
    assert o["vp_idx"].tolist() == [3, 3, 3]
    assert np.array_equal(orc.vp_line2vps(kl, orc.vp_hypothesis(kl, 5, 0)), o["vp_idx"])


@pytest.mark.gpu
def test_vp_gpu_all_pairs_at_infinity(hvo, orc):
    kl = _parallel_lines(orc)
    ctx = hvo.Context()
    try:
        got = ctx.vanishing_points(kl, seed=5)
    finally:
        ctx.close()
    assert got["score"] == 0 and got["best"] == 0 and not np.asarray(got["vps"]).any()
    assert got["vp_idx"].tolist() == [3, 3, 3]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed,rs", [("std", 0x5EED0002, 7), ("std", 0x5EED1001, 1), ("lowtex", 0x5EED2000, 3)])
def test_vp_gpu_vs_oracle(hvo, orc, synth, kind, seed, rs):
    kl = _lines(orc, synth, kind, seed)
    ref = orc.vanishing_points(kl, seed=rs, want_scores=True)
    ctx = hvo.Context()
    try:
        got = ctx.vanishing_points(kl, seed=rs, want_grid=True)
    finally:
        ctx.close()
    assert got["n_hypotheses"] == 37800
    # the sphere grid: same cells, sums in the same order (values of sin / acos may differ in the last bit)
    assert np.allclose(got["grid"], ref["grid"], rtol=1e-10, atol=1e-9)
    assert abs(got["score"] - ref["score"]) <= 1e-9 * max(1.0, ref["score"])
    if got["best"] != ref["best"]:
        # two hypotheses may tie exactly (rotation j and j + 180 give the same triple): then the device's last-bit differences pick the twin
        assert abs(ref["scores"][got["best"]] - ref["score"]) <= 1e-9 * max(1.0, ref["score"])
        # ... and what the device reports must still be the oracle's hypothesis at that index and the oracle's clusters under it
        twin = orc.vp_hypothesis(kl, rs, got["best"])
        assert np.allclose(got["vps"], twin, atol=1e-12)
        assert np.array_equal(got["vp_idx"], orc.vp_line2vps(kl, twin))
    else:
        assert np.allclose(got["vps"], ref["vps"], atol=1e-12)
        assert np.array_equal(got["vp_idx"], ref["vp_idx"])


@pytest.mark.gpu
def test_vp_gpu_manhattan_and_edges(hvo, orc):
    kl, R = _box_lines(orc)
    th = 3.0 / 180 * np.pi
    ref = orc.vanishing_points(kl, seed=11, th_angle=th)
    ctx = hvo.Context()
    try:
        got = ctx.vanishing_points(kl, seed=11, th_angle=th)
        # many hypotheses of this scene sum the same three peak cells: which of the tied ones is "first above" depends on the
        # last bit of the grid sums, so the index may differ; the score, the frame it stands for and the clusters may not
        assert np.array_equal(got["vp_idx"], ref["vp_idx"])
        assert abs(got["score"] - ref["score"]) <= 1e-9 * ref["score"]
        assert (np.abs(got["vps"] @ R).max(1) > np.cos(np.deg2rad(3))).all()
        one = ctx.vanishing_points(kl[:1])                       # NL <= 1: the reference skips the path
        assert one["vp_idx"].tolist() == [3] and one["score"] == 0
        assert ctx.vanishing_points(kl[:0])["vp_idx"].shape == (0,)
        with pytest.raises(hvo.HvoError):
            ctx.vanishing_points(np.zeros(2000, hvo.KEYLINE_DT))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_vp_gpu_one_crowded_cell(hvo, orc):
    """all lines through ONE vanishing point: every pair of the 240 lines (28 680 pairs) meets in the same few sphere cells -- the cells
    whose pairs the whole workgroup orders (a lone lane's insertion sort is quadratic there); the grid must still be the
    oracle's sums in pair order"""
    fx, fy, cx, cy = 535.4, 539.2, 320.1, 247.6
    rng = np.random.RandomState(9)
    n = 240
    kl = np.zeros(n, orc.KEYLINE_DT)
    vx, vy = 900.0, 120.0
    for i in range(n):
        mx, my = rng.uniform(40, 600), rng.uniform(40, 440)
        d = np.array([vx - mx, vy - my]); d /= np.linalg.norm(d)
        L = rng.uniform(15, 50)
        kl["sx"][i], kl["sy"][i] = mx - d[0] * L, my - d[1] * L
        kl["ex"][i], kl["ey"][i] = mx + d[0] * L, my + d[1] * L
    ref = orc.vanishing_points(kl, seed=4)
    assert (ref["grid"] > 0).sum() < 200 and ref["grid"].max() > 1000          # a few crowded cells
    ctx = hvo.Context(lsd_nfeatures=n)
    try:
        got = ctx.vanishing_points(kl, seed=4, want_grid=True)
    finally:
        ctx.close()
    assert np.allclose(got["grid"], ref["grid"], rtol=1e-10, atol=1e-9)
    assert np.array_equal(got["vp_idx"], ref["vp_idx"]) and abs(got["score"] - ref["score"]) <= 1e-9 * max(1.0, ref["score"])
