"""Frame post-processing of the front-end's outputs: UndistortKeyPoints, ComputeImageBounds,
AssignFeaturesToGrid, AssignFeaturesToGridForLine -- oracle known answers (CPU) and HIP parity (GPU)."""
import numpy as np
import pytest

TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989)           # Examples/RGB-D/TUM1.yaml
TUM1_DIST = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
TUM3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6)                               # Examples/RGB-D/TUM3.yaml (k = 0)


def _features(orc, synth, seed=0x5EED0002):
    g = synth.make_gray("std", seed)
    kp, _ = orc.Orb().extract(g)
    kl, _, _ = orc.line_extract(g)
    return kp, kl


def _brute_grid(kp, b):
    winv = np.float32(64) / (np.float32(b[1]) - np.float32(b[0])); hinv = np.float32(48) / (np.float32(b[3]) - np.float32(b[2]))
    cells = [[] for _ in range(64 * 48)]
    for i, k in enumerate(kp):
        fx = (np.float32(k["x"]) - np.float32(b[0])) * winv; fy = (np.float32(k["y"]) - np.float32(b[2])) * hinv
        px = int(np.floor(abs(fx) + np.float32(0.5)) * np.sign(fx)); py = int(np.floor(abs(fy) + np.float32(0.5)) * np.sign(fy))   # roundf
        if 0 <= px < 64 and 0 <= py < 48:
            cells[px * 48 + py].append(i)
    return cells


def test_oracle_undistort_known_answers(orc, synth):
    kp, _ = _features(orc, synth)
    # zero distortion: key points are copied, bounds are the image (Frame.cc:1703-1707, 1754-1760)
    same = orc.undistort_keypoints(kp, **TUM3, dist5=[0, 0, 0, 0, 0])
    assert np.array_equal(same, kp)
    assert orc.image_bounds(640, 480, **TUM3, dist5=[0] * 5).tolist() == [0.0, 640.0, 0.0, 480.0]
    # TUM1: re-applying the forward model to the undistorted point must give the pixel back
    un = orc.undistort_keypoints(kp, **TUM1, dist5=TUM1_DIST)
    k1, k2, p1, p2, k3 = TUM1_DIST
    x = (un["x"].astype(np.float64) - TUM1["cx"]) / TUM1["fx"]; y = (un["y"].astype(np.float64) - TUM1["cy"]) / TUM1["fy"]
    r2 = x * x + y * y; rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    assert np.abs(xd * TUM1["fx"] + TUM1["cx"] - kp["x"]).max() < 0.05 and np.abs(yd * TUM1["fy"] + TUM1["cy"] - kp["y"]).max() < 0.05
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(un[f], kp[f])
    b = orc.image_bounds(640, 480, **TUM1, dist5=TUM1_DIST)
    assert 0 < b[0] < 30 and 610 < b[1] < 640 and 0 < b[2] < 30 and 450 < b[3] < 480


def test_oracle_grid_against_brute_force(orc, synth):
    kp, kl = _features(orc, synth)
    b = orc.image_bounds(640, 480, **TUM3, dist5=[0] * 5)
    start, items = orc.assign_features_to_grid(kp, b)
    cells = _brute_grid(kp, b)
    assert start[0] == 0 and start[-1] == len(items) == sum(len(c) for c in cells)
    for c in range(64 * 48):
        assert items[start[c]:start[c + 1]].tolist() == cells[c]
    # lines: every line is in the cells of both its end points' columns range, indices ascend inside a cell
    ls, li, n = orc.assign_lines_to_grid(kl, b)
    assert n == len(li) and n >= len(kl)
    for c in range(64 * 48):
        seg = li[ls[c]:ls[c + 1]]
        assert np.all(np.diff(seg) > 0)
    # an axis-parallel segment visits exactly the cells between its end points
    one = np.zeros(1, kl.dtype); one["sx"], one["sy"], one["ex"], one["ey"] = 15.0, 105.0, 95.0, 105.0
    ls, li, n = orc.assign_lines_to_grid(one, b)
    assert n == 9 and [c for c in range(64 * 48) if ls[c + 1] > ls[c]] == [gx * 48 + 10 for gx in range(1, 10)]
    assert orc.assign_lines_to_grid(kl, b, cap=3)[2] == -1            # capacity reported


@pytest.mark.gpu
@pytest.mark.parametrize("cam,dist", [(TUM1, TUM1_DIST), (TUM3, [0, 0, 0, 0, 0])])
def test_frame_post_parity(hvo, orc, synth, cam, dist):
    kp, kl = _features(orc, synth, 0x5EED1001)
    ctx = hvo.Context(**cam)
    try:
        un_g = ctx.undistort_keypoints(kp, dist); un_o = orc.undistort_keypoints(kp, **cam, dist5=dist)
        for f in kp.dtype.names:
            assert np.array_equal(un_g[f], un_o[f]), f                       # fp64 iteration, bit-exact after the float cast
        b_g = ctx.image_bounds(640, 480, dist); b_o = orc.image_bounds(640, 480, **cam, dist5=dist)
        assert np.array_equal(b_g, b_o)
        s_g, i_g = ctx.assign_features_to_grid(un_g, b_g); s_o, i_o = orc.assign_features_to_grid(un_o, b_o)
        assert np.array_equal(s_g, s_o) and np.array_equal(i_g, i_o)
        ls_g, li_g = ctx.assign_lines_to_grid(kl, b_g); ls_o, li_o, n_o = orc.assign_lines_to_grid(kl, b_o)
        assert np.array_equal(ls_g, ls_o) and np.array_equal(li_g, li_o) and len(li_g) == n_o
    finally:
        ctx.close()


@pytest.mark.gpu
def test_frame_post_edge_cases(hvo, gpu_ctx):
    b = np.array([0, 640, 0, 480], np.float32)
    s, i = gpu_ctx.assign_features_to_grid(np.zeros(0, hvo.KEYPOINT_DT), b)
    assert len(i) == 0 and not s.any()
    s, i = gpu_ctx.assign_lines_to_grid(np.zeros(0, hvo.KEYLINE_DT), b)
    assert len(i) == 0 and not s.any()
    kp = np.zeros(3, hvo.KEYPOINT_DT); kp["x"] = [-5.0, 639.9, 10.0]; kp["y"] = [10.0, 10.0, 479.9]      # out of the grid after round()
    s, i = gpu_ctx.assign_features_to_grid(kp, b)
    assert len(i) == 0
    kl = np.zeros(1, hvo.KEYLINE_DT); kl["sx"], kl["sy"], kl["ex"], kl["ey"] = 0.0, 0.0, 639.0, 479.0
    with pytest.raises(hvo.HvoError):
        gpu_ctx.assign_lines_to_grid(kl, b, cap=2)                                                      # HVO_ERR_CAPACITY
    with pytest.raises(hvo.HvoError):
        gpu_ctx.assign_features_to_grid(kp, np.array([0, 0, 0, 480], np.float32))                        # empty bounds


# ---------------- Frame::cullingLine ----------------
def test_oracle_culling_known_answers(orc, synth):
    # cv::LineIterator count with clipping: inside -> max(|dx|,|dy|)+1; the image diagonal extended beyond the corners is
    # clipped to the image; a segment entirely left of the image has no pixels
    assert orc.line_iterator_count_clipped(640, 480, 10, 10, 20, 40) == 31
    assert orc.line_iterator_count_clipped(640, 480, -10, 5, -3, 80) == 0
    assert 0 < orc.line_iterator_count_clipped(640, 480, -10, -10, 700, 500) <= 640
    # an end point in (w-1, w) passes checkLineExtremes (LSDDetector_custom.cpp:76-102) and rounds to w: the segment
    # (642, 279) - (623, 300) is clipped at x = 641 -> (641, 280) - (623, 300): max(18, 20) + 1 pixels, not 22 and not 0
    assert orc.line_iterator_count_clipped(642, 400, np.float32(641.5019), np.float32(278.88806), np.float32(622.81824), np.float32(300.35803)) == 21
    g = np.full((480, 640), 128, np.uint8)
    kl = np.zeros(3, orc.KEYLINE_DT)
    # two collinear overlapping horizontal segments and one far away vertical one
    kl["sx"], kl["sy"], kl["ex"], kl["ey"] = [100, 150, 400], [200, 200.5, 50], [200, 300, 400], [200, 200.5, 300]
    fn = np.zeros((3, 3))
    for i in range(3):
        sx, sy, ex, ey = (float(kl[k][i]) for k in ("sx", "sy", "ex", "ey"))
        l = np.array([sy - ey, ex - sx, sx * ey - sy * ex]); fn[i] = l / np.hypot(l[0], l[1])
    out, desc, fo = orc.cull_lines(g, kl, fn)
    assert len(out) == 2                                            # the two horizontal ones merged
    assert out["class_id"].tolist() == [0, 1] and out["response"][0] >= out["response"][1]
    horiz = out[np.abs(out["sy"] - 200) < 2][0]
    assert abs(min(horiz["sx"], horiz["ex"]) - 100) < 1 and abs(max(horiz["sx"], horiz["ex"]) - 300) < 1   # spans both
    vert = out[np.abs(out["sx"] - 400) < 1][0]
    assert vert["num_pixels"] == 251 and np.isclose(vert["length"], 250)
    assert np.allclose(fo[:, 0] ** 2 + fo[:, 1] ** 2, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed", [("std", 0x5EED0002), ("std", 0x5EED1001), ("lowtex", 0x5EED0001)])
def test_culling_parity(gpu_ctx, orc, synth, kind, seed):
    g = synth.make_gray(kind, seed)
    kl_o, d_o, fn_o = orc.line_extract(g)
    ck_o, cd_o, cf_o = orc.cull_lines(g, kl_o, fn_o)
    ck_g, cd_g, cf_g = gpu_ctx.extract_lsd(g, culled=True)
    assert len(ck_g) == len(ck_o) and len(ck_o) <= len(kl_o)
    for f in ck_o.dtype.names:
        assert np.array_equal(ck_g[f], ck_o[f]), f
    assert np.array_equal(cd_g, cd_o)
    assert np.allclose(cf_g, cf_o, rtol=1e-12, atol=1e-12)
    # the plain extractor output is unchanged by the extra stage
    kl_g, d_g, _ = gpu_ctx.extract_lsd(g)
    assert np.array_equal(kl_g["sx"], kl_o["sx"]) and np.array_equal(d_g, d_o)


@pytest.mark.gpu
def test_culling_batch_stage_and_params(hvo, orc, synth):
    gray, depth = synth.make_batch("std", 0x5EED1000, 3)
    ctx = hvo.Context(max_batch=3)
    try:
        ctx.set_line_culling(7.5, 3.0, 20.0)                          # the commented-out alternative at Frame.cc:935
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_LSD_CULL)
        res = ctx.batch_download(hvo.STAGE_LSD_CULL)
    finally:
        ctx.close()
    for b in range(3):
        kl_o, _, fn_o = orc.line_extract(gray[b])
        ck_o, cd_o, _ = orc.cull_lines(gray[b], kl_o, fn_o, 7.5, 3.0, 20.0)
        assert res[b]["status"] == 0 and len(res[b]["kl"]) == len(ck_o)
        assert np.array_equal(res[b]["kl"]["sx"], ck_o["sx"]) and np.array_equal(res[b]["ldesc"], cd_o)


@pytest.mark.gpu
def test_culling_parity_1280(hvo, orc, synth):
    g = synth.make_gray("std", 0x5EED0003, 1280, 960)
    kl_o, d_o, fn_o = orc.line_extract(g)
    ck_o, cd_o, cf_o = orc.cull_lines(g, kl_o, fn_o)
    ctx = hvo.Context()
    try:
        ck_g, cd_g, cf_g = ctx.extract_lsd(g, culled=True)
    finally:
        ctx.close()
    assert len(ck_g) == len(ck_o)
    for f in ck_o.dtype.names:
        assert np.array_equal(ck_g[f], ck_o[f]), f
    assert np.array_equal(cd_g, cd_o) and np.allclose(cf_g, cf_o, rtol=1e-12, atol=1e-12)
