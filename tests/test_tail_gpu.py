"""The rest of the Frame constructor as pipeline stages (HVO_STAGE_LINES3D / _VP / _PLANE_TAIL / _GRIDS; reference src/Frame.cc:205-233,
328-337, 832-872, 934-939, 2110-2274): they run on what the front-end left in HBM -- culled key lines, raw depth, int8 labels, planes,
undistorted key points -- in the streamed mode (hvo_stream_*) and for a resident batch (hvo_batch_run).  Every field is compared with the
CPU oracle applied to the same frame's primary results (which the other tests compare with the oracle themselves)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BF = 40.0


def check_tail(r, d, orc, seed, bounds):
    kl = r["kl"]
    # Frame::isLineGood: every field of every line, bit for bit
    o3 = orc.lines_3d(kl, d, seed=seed)
    assert r["lines3d"].tobytes() == o3.tobytes(), [f for f in o3.dtype.names if not np.array_equal(r["lines3d"][f], o3[f])]
    # vanishing points: sums in the reference's order (sin / acos are the device's: 1e-9); an exact tie may pick the twin hypothesis
    vo = orc.vanishing_points(kl, seed=seed, want_scores=True)
    vg = r["vp"]
    if vo is None:
        assert vg["n_hypotheses"] == 0 and (vg["vp_idx"] == 3).all()
    else:
        assert vg["n_hypotheses"] == 37800
        assert abs(vg["score"] - vo["score"]) <= 1e-9 * max(1.0, vo["score"])
        if vg["best"] != vo["best"]:
            assert abs(vo["scores"][vg["best"]] - vo["score"]) <= 1e-9 * max(1.0, vo["score"])
            twin = orc.vp_hypothesis(kl, seed, vg["best"])
            assert np.allclose(vg["vps"], twin, atol=1e-12) and np.array_equal(vg["vp_idx"], orc.vp_line2vps(kl, twin))
        else:
            assert np.allclose(vg["vps"], vo["vps"], atol=1e-12) and np.array_equal(vg["vp_idx"], vo["vp_idx"])
    # ComputePlanes' tail: voxel clouds exact, refit at 1e-5 (float libm inside pcl::eigen33), normals bit for bit
    po, co = orc.plane_clouds(d, r["labels"], r["planes"], dist_th=0.05)
    pg, cg = r["plane_clouds"], r["cloud_xyz"]
    assert len(pg) == len(po)
    for f in ("valid", "gate_ok", "first", "n_points", "n_pixels"):
        assert np.array_equal(pg[f], po[f]), f
    assert np.array_equal(cg, co)
    assert np.allclose(pg["coef"], po["coef"], rtol=0, atol=1e-5)
    so = orc.surface_normals(d)
    assert r["normals"].tobytes() == so.tobytes()
    # the two 64 x 48 grids
    ps, pi = orc.assign_features_to_grid(r["kp_un"] if "kp_un" in r else r["kp"], bounds)
    assert np.array_equal(r["pt_grid"][0], ps) and np.array_equal(r["pt_grid"][1], pi)
    ls, li, ln = orc.assign_lines_to_grid(kl, bounds)
    assert np.array_equal(r["ln_grid"][0], ls) and np.array_equal(r["ln_grid"][1], li)
    assert r["tail_status"] == 0


def test_stream_whole_frame_constructor(hvo, orc, synth):
    """16 frames through hvo_stream_* with every stage of the Frame constructor; primary results and the tail against the oracle"""
    from test_stream_gpu import check_frame
    n = 16
    g, d, _ = synth.make_sequence("std", 0x5EED2000, n)
    g2, d2, _ = synth.make_sequence("lowtex", 0x5EED2300, 4)
    g = np.concatenate([g, g2]); d = np.concatenate([d, d2]); n = len(g)
    seed = 77
    st = hvo.Stream(depth=3, stages=hvo.STAGE_ORB | hvo.STAGE_LSD | hvo.STAGE_PLANES | hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS, bf=BF, seed=seed)
    orb = orc.Orb()
    try:
        tick = [st.submit(g[0], d[0]), st.submit(g[1], d[1])]
        for i in range(n):
            if i + 2 < n:
                tick.append(st.submit(g[i + 2], d[i + 2]))
            r = st.collect(tick[i])
            assert r["status"] == 0
            check_frame(r, g[i], d[i], orc, orb)
            check_tail(r, d[i], orc, seed + tick[i], (0.0, 640.0, 0.0, 480.0))
            ms = st.stage_ms(tick[i])
            assert ms["orb"] > 0 and ms["lsd"] > 0 and ms["planes"] > 0
    finally:
        st.close()


def test_stream_tail_on_culled_lines(hvo, orc, synth):
    """HVO_STAGE_LSD_CULL: the tail runs on the merged lines (what Frame::ExtractLSD hands to isLineGood, src/Frame.cc:934-939)"""
    g, d, _ = synth.make_sequence("std", 0x5EED2400, 3)
    st = hvo.Stream(depth=2, stages=hvo.STAGE_LSD_CULL | hvo.STAGE_LINES3D | hvo.STAGE_VP, bf=BF, seed=5)
    try:
        for i in range(3):
            t = st.submit(g[i], d[i])
            r = st.collect(t)
            o3 = orc.lines_3d(r["kl"], d[i], seed=5 + t)
            assert len(r["kl"]) > 10 and r["lines3d"].tobytes() == o3.tobytes()
            vo = orc.vanishing_points(r["kl"], seed=5 + t)
            assert abs(r["vp"]["score"] - vo["score"]) <= 1e-9 * max(1.0, vo["score"])
    finally:
        st.close()


def test_stream_tail_needs_its_producers(hvo):
    with pytest.raises(hvo.HvoError):
        hvo.Stream(depth=2, stages=hvo.STAGE_ORB | hvo.STAGE_VP)                 # vanishing points without lines
    with pytest.raises(hvo.HvoError):
        hvo.Stream(depth=2, stages=hvo.STAGE_LSD | hvo.STAGE_PLANE_TAIL)         # plane tail without planes


def test_batch_whole_frame_constructor(hvo, orc, synth):
    g, d = synth.make_batch("std", 0x5EED1000, 3)
    g2, d2 = synth.make_batch("lowtex", 0x5EED2000, 1)
    g = np.concatenate([g, g2]); d = np.concatenate([d, d2])
    ctx = hvo.Context(max_batch=4)
    try:
        ctx.batch_upload(g, d)
        ctx.set_tail_params(seed=9)
        full = hvo.STAGE_ALL | hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS
        ctx.batch_run(full)
        res = ctx.batch_download(hvo.STAGE_ALL)
        ctx.batch_download_tail(full, res)
        for f, r in enumerate(res):
            assert len(r["kl"]) > 5 and len(r["planes"]) >= 2
            check_tail(r, d[f], orc, 9 + f, (0.0, 640.0, 0.0, 480.0))
    finally:
        ctx.close()
