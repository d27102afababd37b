"""BASELINE.json configs[2] as written -- 1280x960 RGB-D, 2000 ORB, full line / plane extraction, Hamming match to the previous frame --
at the same bar as the 640x480 suites: a multi-frame batch with low-texture frames through every stage and the Frame tail, a streamed
sequence with guided search and line matching, and the projection prologue with 2000 features.  That geometry selects other kernels
than 640x480 does (in-kernel initGraph edges, no LDS-resident AHC queue, overlap policy 7, 1024-frame ORB chunks, the async line
growing with lists that overflow to the frontier), so one `std` frame per stage was not enough."""
import numpy as np
import pytest

from test_stream_gpu import check_frame
from test_tail_gpu import check_tail

pytestmark = pytest.mark.gpu
W, H, NF = 1280, 960, 2000
BF = 80.0                                                  # the TUM3 baseline at twice the resolution
SF = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)


class Orb2000:
    def __init__(self, orc): self.o = orc.Orb(nfeatures=NF)
    def extract(self, g): return self.o.extract(g)


def test_batch_1280_every_stage_and_tail(hvo, orc, synth):
    """six frames, two of them low-texture: ORB (2000), LSD + LBD, culling is exercised elsewhere, PEAC, and the Frame tail on the batch path"""
    g, d = synth.make_batch("std", 0x5EED4000, 4, W, H)
    g2, d2 = synth.make_batch("lowtex", 0x5EED4100, 2, W, H)
    g = np.concatenate([g[:2], g2[:1], g[2:], g2[1:]]); d = np.concatenate([d[:2], d2[:1], d[2:], d2[1:]])
    ctx = hvo.Context(max_batch=6, orb_nfeatures=NF)
    orb = Orb2000(orc)
    try:
        ctx.batch_upload(g, d)
        ctx.set_tail_params(seed=21)
        full = hvo.STAGE_ALL | hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS
        ctx.batch_run(full)
        res = ctx.batch_download(hvo.STAGE_ALL)
        ctx.batch_download_tail(full, res)
        from test_lsd_gpu import check as check_lines
        from test_peac_gpu import check as check_planes
        from test_orb_gpu import check_orb
        for f, r in enumerate(res):
            assert r["status"] == 0
            kpo, desco = orb.extract(g[f]); check_orb(r["kp"], r["desc"], kpo, desco)
            klo, ldo, fno = orc.line_extract(g[f]); check_lines(r["kl"], r["ldesc"], r["linefn"], klo, ldo, fno)
            lo, po = orc.peac(d[f]); check_planes(r["labels"], r["planes"], lo, po)
            check_tail(r, d[f], orc, 21 + f, (0.0, float(W), 0.0, float(H)))
        assert len(res[0]["kp"]) > 1500 and len(res[2]["kp"]) < len(res[0]["kp"])           # the low-texture frame misses the quota
    finally:
        ctx.close()


def test_stream_1280_with_matching(hvo, orc, synth):
    """eight streamed frames at 1280x960 / 2000 ORB, three in flight (ring of four: the previous frame stays matchable): every frame against the oracle, SearchByProjection(Cur, Last) with the
    projection prologue on the device against the oracle's prologue + core, line matching against the oracle"""
    n = 8
    g, d, off = synth.make_sequence("std", 0x5EED4200, n, W, H)
    cam = (535.4 * 2, 539.2 * 2, 320.1 * 2, 247.6 * 2, BF, BF / (535.4 * 2))
    st = hvo.Stream(width=W, height=H, depth=4, stages=hvo.STAGE_ALL, bf=BF, orb_nfeatures=NF)
    orb = Orb2000(orc)
    try:
        tick = [st.submit(g[0], d[0]), st.submit(g[1], d[1])]
        res = {}
        total = 0
        for i in range(n):
            if i + 2 < n: tick.append(st.submit(g[i + 2], d[i + 2]))
            res[i] = st.collect(tick[i])
            assert res[i]["status"] == 0
            kpo, desco = orb.extract(g[i])
            assert len(res[i]["kp"]) == len(kpo) and np.array_equal(res[i]["desc"], desco) and np.array_equal(res[i]["kp"]["x"], kpo["x"])
            klo, ldo, fno = orc.line_extract(g[i])
            assert np.array_equal(res[i]["ldesc"], ldo) and np.array_equal(res[i]["kl"]["num_pixels"], klo["num_pixels"])
            lo, po = orc.peac(d[i])
            assert np.array_equal(res[i]["labels"], lo) and len(res[i]["planes"]) == len(po)
            if i > 0:
                cur, last = res[i], res[i - 1]
                z = last["zdepth"]; sel = np.flatnonzero(z > 0).astype(np.int32)
                kpl = last["kp_un"][sel]
                X = np.stack([(kpl["x"] - np.float32(cam[2])) * z[sel] / np.float32(cam[0]), (kpl["y"] - np.float32(cam[3])) * z[sel] / np.float32(cam[1]), z[sel]], axis=1).astype(np.float32)
                shift = (off[i] - off[i - 1]).astype(np.float32)
                # the drift of the window as a camera translation at the mean depth (a pose guess, as a motion model gives one)
                zm = float(np.median(z[sel]))
                Tcw = np.concatenate([np.eye(3, dtype=np.float32), np.array([[-shift[0] * zm / cam[0]], [-shift[1] * zm / cam[1]], [0.0]], np.float32)], axis=1)
                Tlw = np.concatenate([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)], axis=1)
                blocks = (sel % 3 != 0).astype(np.uint8)
                q = orc.project_last(Tcw, Tlw, X, last["kp_un"]["octave"][sel], cam, False, 15.0, SF, (0.0, 0.0, float(W), float(H)))
                no, mio, mdo = orc.search_by_projection(last["desc"][sel], q["u"], q["v"], q["radius"], q["min_level"], q["max_level"], q["ur"],
                                                        last["kp_un"]["angle"][sel], blocks, cur["kp_un"], cur["uright"], np.zeros(len(cur["kp"]), np.uint8),
                                                        cur["desc"], (0.0, 0.0, float(W), float(H)))
                ng, mi, md = st.project_last(tick[i], tick[i - 1], cam, Tcw, Tlw, sel, X, blocks, 15.0)
                assert ng == no and np.array_equal(mi, mio) and np.array_equal(md[mi >= 0], mdo[mio >= 0])
                total += ng
                nl, ml = st.match_lines(tick[i - 1], tick[i], hvo.LINE_MATCH_NNR, nnratio=0.95)
                nlo, mlo = orc.match_nnr(last["ldesc"], cur["ldesc"], 0.95)
                assert nl == nlo and np.array_equal(ml, mlo)
                del res[i - 1]
        assert total > 300 * (n - 1), total
    finally:
        st.close()


def test_guided_search_2000_features(hvo, orc, synth):
    """both guided searches with 2000 features per frame (SBP's ranked candidates, the sequential occupancy pass and the rotation histogram
    at four times the 640x480 load)"""
    g1 = synth.make_gray("std", 0x5EED4300, W, H)
    g2 = np.roll(np.roll(g1, 4, axis=0), 6, axis=1)
    o = orc.Orb(nfeatures=NF)
    kp1, d1 = o.extract(g1); kp2, d2 = o.extract(g2)
    assert len(kp1) > 1800
    rng = np.random.default_rng(5)
    n1 = len(kp1)
    bounds = (0.0, 0.0, float(W), float(H))
    q_u = (kp1["x"] + 6 + rng.normal(0, 1.5, n1)).astype(np.float32); q_v = (kp1["y"] + 4 + rng.normal(0, 1.5, n1)).astype(np.float32)
    q_ur = (q_u - BF / rng.uniform(1, 4, n1)).astype(np.float32)
    t_ur = np.where(rng.uniform(size=len(kp2)) < 0.7, kp2["x"] - BF / rng.uniform(1, 4, len(kp2)), -1).astype(np.float32)
    t_occ = (rng.uniform(size=len(kp2)) < 0.1).astype(np.uint8)
    blocks = (rng.uniform(size=n1) < 0.9).astype(np.uint8)
    ctx = hvo.Context(orb_nfeatures=NF)
    try:
        rad = (np.float32(15) * SF[kp1["octave"]]).astype(np.float32)
        args = (d1, q_u, q_v, rad, (kp1["octave"] - 1).astype(np.int32), (kp1["octave"] + 1).astype(np.int32), q_ur, kp1["angle"], blocks, kp2, t_ur, t_occ, d2, bounds)
        no, io, do = orc.search_by_projection(*args, th_high=100, check_orientation=True)
        ng, ig, dg = ctx.search_by_projection(*args, th_high=100, check_orientation=True)
        assert no > 800 and ng == no and np.array_equal(ig, io) and np.array_equal(dg[ig >= 0], do[io >= 0])
        level = np.clip(kp1["octave"], 0, 7).astype(np.int32); vcos = np.where(rng.uniform(size=n1) < 0.5, 0.9995, 0.99).astype(np.float32)
        r2, lo, hi = orc.track_windows(level, vcos, 3.0, SF)
        no2, io2, do2 = orc.search_by_projection_map(d1, q_u, q_v, r2, lo, hi, q_ur, blocks, kp2, t_ur, t_occ, d2, bounds, th_high=100, nn_ratio=0.8)
        ng2, ig2, dg2 = ctx.search_by_projection_tracked(d1, q_u, q_v, q_ur, level, vcos, blocks, 3.0, kp2, t_ur, t_occ, d2, bounds, th_high=100, nn_ratio=0.8)
        assert no2 > 500 and ng2 == no2 and np.array_equal(ig2, io2) and np.array_equal(dg2[ig2 >= 0], do2[io2 >= 0])
    finally:
        ctx.close()
