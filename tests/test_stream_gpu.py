"""Streamed-sequence mode (BASELINE.json config 5; reference call shape src/Tracking.cc:262, 2299, 2396): one frame at a time
through hvo_stream_*, several frames in flight, frame-to-frame matching on the device-resident results of the previous frame.
Every frame's extraction and every frame pair's matching is compared with the CPU oracle, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BF = 40.0           # TUM3.yaml Camera.bf
TH = 15             # TrackWithMotionModel's window for RGB-D (src/Tracking.cc:2390-2394)


def queries_from_last(last, shift, scale_factors):
    """what TrackWithMotionModel hands to SearchByProjection for the map points of the last frame (src/ORBmatcher.cc:1376-1420):
    here the 'pose' is the known image-plane drift of the synthetic sequence"""
    kp = last["kp_un"]; z = last["zdepth"]
    sel = np.nonzero(z > 0)[0].astype(np.int32)                          # features that have a map point (depth available)
    u = kp["x"][sel] - np.float32(shift[0]); v = kp["y"][sel] - np.float32(shift[1])
    octv = kp["octave"][sel]
    radius = (np.float32(TH) * scale_factors[octv]).astype(np.float32)
    ur = (u - np.float32(BF) / z[sel]).astype(np.float32)
    blocks = (sel % 3 != 0).astype(np.uint8)                             # map points with observations claim their feature
    return sel, u.astype(np.float32), v.astype(np.float32), radius, (octv - 1).astype(np.int32), (octv + 1).astype(np.int32), ur, blocks


def check_frame(r, g, d, orc, orb):
    kpo, desco = orb.extract(g)
    assert len(r["kp"]) == len(kpo) and np.array_equal(r["desc"], desco)
    for f in ("x", "y", "octave", "response", "size", "angle"):
        assert np.array_equal(r["kp"][f], kpo[f]), f
    # whole key-line records, line functions and plane records, as the single-call tests compare them
    klo, ldo, fno = orc.line_extract(g)
    from test_lsd_gpu import check as check_lines
    from test_peac_gpu import check as check_planes
    check_lines(r["kl"], r["ldesc"], r["linefn"], klo, ldo, fno)
    lo, po = orc.peac(d)
    check_planes(r["labels"], r["planes"], lo, po)
    # TUM3: no distortion -> UndistortKeyPoints copies (src/Frame.cc:1703-1707); ComputeStereoFromRGBD vs the oracle
    assert np.array_equal(r["kp_un"], r["kp"])
    uro, zo = orc.stereo_from_rgbd(kpo, kpo, d, float(np.float32(1.0) / np.float32(5000.0)), BF)
    assert np.array_equal(r["uright"], uro) and np.array_equal(r["zdepth"], zo)


def test_stream_64_frames_match_oracle(hvo, orc, synth):
    n = 64
    g, d, off = synth.make_sequence("std", 0x5EED2000, n)
    st = hvo.Stream(depth=4, stages=hvo.STAGE_ALL, bf=BF)
    orb = orc.Orb()
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)     # mvScaleFactor (ORBextractor.cc:413-419)
    try:
        assert tuple(st.bounds) == (0.0, 640.0, 0.0, 480.0)
        tick = [st.submit(g[0], d[0]), st.submit(g[1], d[1])]          # two frames in flight before the first collect
        res = {}
        total_pt = total_ln = 0
        for i in range(n):
            if i + 2 < n:
                tick.append(st.submit(g[i + 2], d[i + 2]))
            res[i] = st.collect(tick[i])
            assert res[i]["status"] == 0
            check_frame(res[i], g[i], d[i], orc, orb)
            if i > 0:
                cur, last = res[i], res[i - 1]
                shift = off[i] - off[i - 1]
                sel, u, v, rad, lmin, lmax, ur, blocks = queries_from_last(last, shift, sf)
                nm, mi, md = st.search_by_projection(tick[i], tick[i - 1], sel, u, v, rad, lmin, lmax, ur, blocks)
                mio, mdo = orc.search_by_projection(last["desc"][sel], u, v, rad, lmin, lmax, ur, last["kp_un"]["angle"][sel], blocks,
                                                    cur["kp_un"], cur["uright"], np.zeros(len(cur["kp"]), np.uint8), cur["desc"],
                                                    (0.0, 0.0, 640.0, 480.0))[1:]
                assert np.array_equal(mi, mio) and np.array_equal(md[mi >= 0], mdo[mio >= 0])
                assert nm == int((mio >= 0).sum())
                total_pt += nm
                # LSDmatcher::match(LastFrame.mLdesc, CurrentFrame.mLdesc, 0.95, matches_12) (src/LSDmatcher.cpp:42)
                nl, ml = st.match_lines(tick[i - 1], tick[i], hvo.LINE_MATCH_NNR, nnratio=0.95)
                no, mo = orc.match_nnr(last["ldesc"], cur["ldesc"], 0.95)
                assert nl == no and np.array_equal(ml, mo)
                total_ln += nl
                del res[i - 1]
        assert total_pt > 100 * (n - 1) and total_ln > 10 * (n - 1), (total_pt, total_ln)
    finally:
        st.close()


def test_stream_slot_reuse_and_order(hvo, synth):
    """a slot is refused until its frame was collected; tickets can be collected out of order; stale tickets are rejected"""
    g, d, _ = synth.make_sequence("lowtex", 0x5EED2100, 4)
    st = hvo.Stream(depth=2, stages=hvo.STAGE_ORB | hvo.STAGE_PLANES, bf=BF)
    try:
        t0 = st.submit(g[0], d[0]); t1 = st.submit(g[1], d[1])
        with pytest.raises(hvo.HvoError) as e:
            st.submit(g[2], d[2])
        assert e.value.status == -7
        r1 = st.collect(t1); r0 = st.collect(t0)
        assert len(r0["planes"]) >= 3 and len(r1["planes"]) >= 3 and "kl" not in r0
        t2 = st.submit(g[2], d[2])
        with pytest.raises(hvo.HvoError):
            st.collect(t0)                                           # its slot now belongs to t2
        ms = None
        r2 = st.collect(t2); ms = st.stage_ms(t2)
        assert ms["planes"] > 0 and ms["orb"] > 0 and ms["lsd"] == 0
        a = hvo.Context()
        try:
            lg, pg = a.compute_planes(d[2])
        finally:
            a.close()
        assert np.array_equal(r2["labels"], lg) and np.array_equal(r2["planes"]["n_points"], pg["n_points"])
    finally:
        st.close()


def test_stream_lines_bf_and_double(hvo, orc, synth):
    g, d, _ = synth.make_sequence("std", 0x5EED2200, 3)
    st = hvo.Stream(depth=3, stages=hvo.STAGE_LSD, bf=0.0)
    try:
        t = [st.submit(g[k]) for k in range(3)]
        r = [st.collect(x) for x in t]
        for a, b in ((0, 1), (1, 2), (2, 0)):
            n1, m1 = st.match_lines(t[a], t[b], hvo.LINE_MATCH_BF, th=50.0, nnratio=0.9)
            no, mo = orc.frame_bf_match(r[a]["ldesc"], r[b]["ldesc"], 50.0, 0.9)
            assert n1 == no and np.array_equal(m1, mo)
            n2, m2 = st.match_lines(t[a], t[b], hvo.LINE_MATCH_DOUBLE, th=50.0, nnratio=0.9)
            no, mo = orc.frame_bf_match(r[a]["ldesc"], r[b]["ldesc"], 50.0, 0.9, mutual=True)
            assert n2 == no and np.array_equal(m2, mo)
    finally:
        st.close()


@pytest.mark.parametrize("tz,mono", [(0.0, False), (0.25, False), (-0.25, False), (0.25, True)])
def test_stream_project_last(hvo, orc, synth, tz, mono):
    """ORBmatcher::SearchByProjection(Cur, Last) WHOLE on the device: the projection prologue (src/ORBmatcher.cc:1364-1405: x3Dc, invzc, u, v,
    bounds, bForward / bBackward, radius, octave band, ur) + the search core, against the oracle's prologue feeding the oracle's core.
    The map points are the last frame's features back-projected with their depth; the current pose is a small rotation + translation
    (tz beyond mb = 0.08 switches the octave band: forward, backward, none; mono switches it off)."""
    g, d, off = synth.make_sequence("std", 0x5EED2100, 3)
    cam = (535.4, 539.2, 320.1, 247.6, 40.0, 40.0 / 535.4)                  # TUM3 fx fy cx cy, mbf, mb = mbf / fx
    st = hvo.Stream(depth=3, stages=hvo.STAGE_ORB | hvo.STAGE_PLANES, bf=cam[4])
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    try:
        t0, t1 = st.submit(g[0], d[0]), st.submit(g[1], d[1])
        last, cur = st.collect(t0), st.collect(t1)
        z = last["zdepth"]; sel = np.flatnonzero(z > 0).astype(np.int32)
        assert len(sel) > 300
        kpl = last["kp_un"][sel]
        X = np.stack([(kpl["x"] - np.float32(cam[2])) * z[sel] / np.float32(cam[0]), (kpl["y"] - np.float32(cam[3])) * z[sel] / np.float32(cam[1]), z[sel]], axis=1).astype(np.float32)
        X[::17, 2] *= -1                                                     # a few points behind the camera (invzc < 0)
        a = np.float32(0.01)
        Rcw = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        Tcw = np.concatenate([Rcw, np.array([[0.01], [-0.005], [tz]], np.float32)], axis=1).astype(np.float32)
        Tlw = np.concatenate([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)], axis=1)
        blocks = np.ones(len(sel), np.uint8)
        th = 15.0
        q = orc.project_last(Tcw, Tlw, X, last["kp_un"]["octave"][sel], cam, mono, th, sf, (0.0, 0.0, 640.0, 480.0))
        assert tuple(q["fwd_bwd"]) == ((1, 0) if (tz < -0.08 and not mono) else (0, 1) if (tz > 0.08 and not mono) else (0, 0))   # tlc = -Rcw^T tcw
        searched = q["radius"] > 0
        assert 100 < searched.sum() < len(sel)                               # some fail the tests (behind the camera, out of bounds)
        no, mio, mdo = orc.search_by_projection(last["desc"][sel], q["u"], q["v"], q["radius"], q["min_level"], q["max_level"], q["ur"],
                                                last["kp_un"]["angle"][sel], blocks, cur["kp_un"], cur["uright"], np.zeros(len(cur["kp"]), np.uint8),
                                                cur["desc"], (0.0, 0.0, 640.0, 480.0))
        ng, mi, md, uv = st.project_last(t1, t0, cam, Tcw, Tlw, sel, X, blocks, th, mono=mono, want_uv=True)
        assert np.array_equal(uv[:, 0], q["u"]) and np.array_equal(uv[:, 1], q["v"])         # the projections bit for bit
        assert ng == no and np.array_equal(mi, mio) and np.array_equal(md[mi >= 0], mdo[mio >= 0])
        assert no > 50
    finally:
        st.close()


def _write_png(path, arr):
    """8-bit RGB or 16-bit grey PNG, filter 0 rows (the loader's decoder is exercised filter by filter in test_tum_loader.py)"""
    import struct, zlib
    h, w = arr.shape[:2]
    if arr.dtype == np.uint16:
        raw = arr.astype(">u2").tobytes(); row = w * 2; depth, ctype = 16, 0
    else:
        raw = arr.tobytes(); row = w * 3; depth, ctype = 8, 2
    body = b"".join(b"\x00" + raw[y * row:(y + 1) * row] for y in range(h))
    def chunk(t, b): return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(body, 1)) + chunk(b"IEND", b""))


def test_tum_directory_through_the_stream_bench(hvo, orc, synth, tmp_path):
    """BASELINE configs[4] end to end on a TUM-format directory (generated: rgb/*.png, depth/*.png, an association file in the format of
    Examples/RGB-D/associations/fr1_desk.txt): the loader's frames equal the generated ones, `bench.py --mode stream` with HVO_TUM_DIR
    runs the whole constructor + matching on them, and two of its frames through hvo_stream_* equal the oracle."""
    import importlib, json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 6
    g, d, off = synth.make_sequence("std", 0x5EED2200, n)
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines = ["# color images and depth maps", "# timestamp filename timestamp filename"]
    for k in range(n):
        t = 1305031453.359684 + k / 30.0
        _write_png(tmp_path / "rgb" / ("%.6f.png" % t), np.repeat(g[k][:, :, None], 3, axis=2))     # R = G = B: the grey conversion's weights sum to 1
        _write_png(tmp_path / "depth" / ("%.6f.png" % (t + 0.01)), d[k])
        lines.append("%.6f rgb/%.6f.png %.6f depth/%.6f.png" % (t, t, t + 0.01, t + 0.01))
    (tmp_path / "associations.txt").write_text("\n".join(lines) + "\n")
    tum = importlib.import_module("hvo_amd.tum")
    lg, ld = tum.load_sequence(str(tmp_path))
    assert np.array_equal(lg, g) and np.array_equal(ld, d)
    env = dict(os.environ, HVO_TUM_DIR=str(tmp_path))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "stream", "--steps", "4", "--warmup", "2"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "TUM RGB-D sequence" in line["data"] and line["value"] > 0 and line["config"]["workload"]
    st = hvo.Stream(depth=2, stages=hvo.STAGE_ALL, bf=BF)
    try:
        orb = orc.Orb()
        for k in (0, 5):
            r = st.collect(st.submit(lg[k], ld[k]))
            check_frame(r, g[k], d[k], orc, orb)
    finally:
        st.close()


def test_double_buffered_batches(hvo, orc, synth):
    """hvo_batch_stage_upload / _commit_staged / _results_async: while a batch runs, the next one's images go up into staging slabs and
    the last one's results come down from one packed slab.  Three different batches through one context: every slab equals what
    hvo_batch_download gives for that batch, and the staged images never disturb the batch that is running."""
    import importlib
    hd = importlib.import_module("hvo_amd.dist")
    batches = [synth.make_batch("std", 0x5EED5000 + 16 * k, 3) for k in range(3)]
    ctx = hvo.Context(max_batch=3)
    try:
        ctx.batch_stage_upload(*batches[0]); ctx.batch_commit_staged()
        kc, lc, pc, sb, lo = None, None, None, None, None
        host = [None, None]
        got = []
        for k in range(3):
            if k + 1 < 3: ctx.batch_stage_upload(*batches[k + 1])          # goes up while batch k runs
            ctx.batch_run(hvo.STAGE_ALL)
            ref = ctx.batch_download(hvo.STAGE_ALL)
            if sb is None:
                kc, lc, pc, sb, lo = ctx.slab_layout(labels=True)
                host = [hvo.pin(np.zeros(3 * sb, np.uint8)), hvo.pin(np.zeros(3 * sb, np.uint8))]
            ctx.batch_results_async(3, host[k & 1])
            if k + 1 < 3: ctx.batch_commit_staged()
            ctx.batch_results_wait()
            back = hd.unpack_results(hvo, host[k & 1].reshape(3, sb), kc, lc, pc, label_shape=(480, 640))
            for a, b in zip(back, ref):
                for key in ("kp", "desc", "kl", "ldesc", "linefn", "planes", "labels"):
                    assert np.array_equal(a[key], b[key]), (k, key)
            got.append(ref)
        for h_ in host: hvo.unpin(h_)
        o = orc.Orb()
        for k in range(3):                                                  # and each batch is ITS images' result
            kp_o, d_o = o.extract(batches[k][0][1])
            assert np.array_equal(got[k][1]["desc"], d_o)
            lab_o, _ = orc.peac(batches[k][1][2])
            assert np.array_equal(got[k][2]["labels"], lab_o)
    finally:
        ctx.close()


def test_three_contexts_on_three_threads(hvo, orc, synth):
    """The reference runs ORB, LSD and the planes of a frame on three threads (src/Frame.cc:210-215); include/hvo.h allows the same with one context
    per thread.  Three contexts, three threads, 12 frames each way round: every result against the oracle (the lone-frame kernels -- async line
    growing, multi-head AHC -- of different contexts share the device)."""
    import threading
    from test_lsd_gpu import check as check_lines
    from test_peac_gpu import check as check_planes
    from test_orb_gpu import check_orb
    g, d = synth.make_batch("std", 0x5EED7000, 8); g2, d2 = synth.make_batch("lowtex", 0x5EED7100, 4)
    g = np.concatenate([g, g2]); d = np.concatenate([d, d2]); n = len(g)
    orb = orc.Orb()
    ref = [(orb.extract(g[i]), orc.line_extract(g[i]), orc.peac(d[i])) for i in range(n)]
    ctxs = [hvo.Context() for _ in range(3)]
    out = [[None] * n for _ in range(3)]; err = []
    def work(k):
        try:
            for rep in range(2):
                for i in range(n):
                    j = (i + 4 * k) % n                      # the threads are on different frames at any time
                    out[k][j] = ctxs[k].extract_orb(g[j]) if k == 0 else ctxs[k].extract_lsd(g[j]) if k == 1 else ctxs[k].compute_planes(d[j])
        except Exception as e:                               # (an assertion in a thread would be lost)
            err.append((k, repr(e)))
    # contexts earlier tests dropped without close() (the operator mirrors own theirs) are finalised HERE, on this thread, and no collection
    # runs inside the worker threads: a finaliser is an hvo_destroy -- stream / event destruction and device-synchronising frees -- at a point
    # of the interpreter's choosing, which is not part of what this test is about
    import gc
    gc.collect(); gc.disable()
    try:
        th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
        for t in th: t.start()
        for t in th: t.join()
        gc.enable()
        assert not err, err
        for i in range(n):
            check_orb(out[0][i][0], out[0][i][1], *ref[i][0])
            check_lines(out[1][i][0], out[1][i][1], out[1][i][2], *ref[i][1])
            check_planes(out[2][i][0], out[2][i][1], *ref[i][2])
    finally:
        gc.enable()
        for c in ctxs: c.close()


def test_two_contexts_both_growing_lines(hvo, orc, synth):
    """two threads, two contexts, both in hvo_extract_lsd at once: two async growing kernels share the device (each frame's workers must still
    sit on one XCD -- the kernel checks, a worker elsewhere counts itself out)"""
    import threading
    from test_lsd_gpu import check as check_lines
    g, _ = synth.make_batch("lowtex", 0x5EED7200, 6); g2, _ = synth.make_batch("std", 0x5EED7300, 6)
    g = np.concatenate([g, g2]); n = len(g)
    ref = [orc.line_extract(x) for x in g]
    ctxs = [hvo.Context() for _ in range(2)]
    out = [[None] * n for _ in range(2)]; err = []
    def work(k):
        try:
            for rep in range(3):
                for i in range(n):
                    j = (i + 5 * k) % n
                    out[k][j] = ctxs[k].extract_lsd(g[j])
        except Exception as e:
            err.append((k, repr(e)))
    try:
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in th: t.start()
        for t in th: t.join()
        assert not err, err
        for k in range(2):
            for i in range(n): check_lines(out[k][i][0], out[k][i][1], out[k][i][2], *ref[i])
    finally:
        for c in ctxs: c.close()
