"""The line tracker's own two matching calls (round 5): LSDmatcher::SearchByGeomNApearance (reference src/LSDmatcher.cpp:36-108) and
LSDmatcher::SearchByProjection(Cur, Last, th) (561-662) over Frame::GetFeaturesInAreaForLine (src/Frame.cc:1557-1627) -- host-array forms and
the forms on two resident frames of a stream, against the oracle (oracle/match.c), through the C ABI.  Integer results: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOUNDS = np.array([0.0, 640.0, 0.0, 480.0], np.float32)          # mnMinX, mnMaxX, mnMinY, mnMaxY (TUM3: no distortion)


def xorshift_bytes(seed, n):
    s = np.uint64(seed | 1); out = np.empty(n, np.uint8)
    for i in range(n):
        s ^= (s << np.uint64(13)) & np.uint64(0xFFFFFFFFFFFFFFFF); s ^= s >> np.uint64(7); s ^= (s << np.uint64(17)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        out[i] = np.uint8(int(s) & 0xFF)
    return out


def shifted_queries(kl, shift, jitter_seed=0):
    """what TrackWithMotionModel hands the guided line search: the last frame's lines projected into the current image -- here the known
    image-plane drift of the synthetic sequence, plus a sub-pixel jitter so that sample points do not sit on cell borders by construction"""
    rng = np.random.RandomState(jitter_seed)
    j = (rng.rand(len(kl), 4).astype(np.float32) - np.float32(0.5)) * np.float32(1.5)
    q = np.stack([kl["sx"] - np.float32(shift[0]), kl["sy"] - np.float32(shift[1]), kl["ex"] - np.float32(shift[0]), kl["ey"] - np.float32(shift[1])], axis=1).astype(np.float32)
    return (q + j).astype(np.float32)


def test_lines_geom_match_host_arrays(hvo, orc, synth):
    g, d, off = synth.make_sequence("std", 0x5EED6100, 4)
    ctx = hvo.Context()
    try:
        fr = [ctx.extract_lsd(g[k]) for k in range(4)]
        rng = np.random.RandomState(7)
        for a, b in ((0, 1), (1, 2), (2, 3), (3, 0), (0, 3)):
            kl1, d1, _ = fr[a]; kl2, d2, _ = fr[b]
            for hm in (None, (rng.rand(len(kl1)) < 0.7).astype(np.uint8)):
                for th in (0.9, 0.75, 1.0):
                    n, m, acc = ctx.match_lines_geom(d1, kl1, d2, kl2, BOUNDS, desc_th=th, last_has_mapline=hm)
                    no, mo, acco = orc.lines_geom_match(d1, kl1, d2, kl2, BOUNDS, desc_th=th, last_has_mapline=hm)
                    assert n == no and np.array_equal(m, mo) and np.array_equal(acc, acco), (a, b, th)
            assert no > 10                                            # the gates are exercised on real matches
        # a frame matched against a far-away crop: descriptor matches exist, most fail the position gate
        n, m, acc = ctx.match_lines_geom(fr[0][1], fr[0][0], fr[0][1][::-1].copy(), fr[0][0][::-1].copy(), BOUNDS, desc_th=1.01)
        no, mo, acco = orc.lines_geom_match(fr[0][1], fr[0][0], fr[0][1][::-1].copy(), fr[0][0][::-1].copy(), BOUNDS, desc_th=1.01)
        assert n == no and np.array_equal(m, mo) and np.array_equal(acc, acco)
        # degenerate inputs: no current lines, one current line (knnMatch(k = 2) has no second neighbour), a current line at x == 0
        kl1, d1, _ = fr[0]
        for k in (0, 1):
            n, m, acc = ctx.match_lines_geom(d1, kl1, d1[:k], kl1[:k], BOUNDS)
            assert n == 0 and np.all(m == -1) and not acc.any()
        kz = fr[1][0].copy(); kz["sx"][::2] = 0
        n, m, acc = ctx.match_lines_geom(d1, kl1, fr[1][1], kz, BOUNDS)
        no, mo, acco = orc.lines_geom_match(d1, kl1, fr[1][1], kz, BOUNDS)
        assert n == no and np.array_equal(m, mo) and np.array_equal(acc, acco) and (mo[~acco.astype(bool)] >= 0).any()
    finally:
        ctx.close()


@pytest.mark.parametrize("th", [3.0, 15.0, 40.0])
def test_search_lines_by_projection_host_arrays(hvo, orc, synth, th):
    g, d, off = synth.make_sequence("std", 0x5EED6200, 3)
    ctx = hvo.Context()
    try:
        fr = [ctx.extract_lsd(g[k]) for k in range(3)]
        for a, b in ((0, 1), (1, 2), (0, 2)):
            klq, dq, _ = fr[a]; klt, dt, fnt = fr[b]
            cs, ci = ctx.assign_lines_to_grid(klt, BOUNDS)
            cso, cio, _ = orc.assign_lines_to_grid(klt, BOUNDS)
            assert np.array_equal(cs, cso) and np.array_equal(ci, cio)
            shift = (off[b] - off[a]).astype(np.float32)
            q = shifted_queries(klq, shift, jitter_seed=a * 3 + b)
            q[5] = q[5][[0, 1, 0, 1]]                                    # a zero-length projection: its direction is NaN and passes, as in the reference
            q[6] = (-500.0, -500.0, -400.0, -450.0)                     # outside every window
            q[7] = (700.0, 100.0, 900.0, 100.0)
            blocks = (np.arange(len(klq)) % 3 != 0).astype(np.uint8)
            occ = (np.arange(len(klt)) % 7 == 0).astype(np.uint8)
            n, mi, md = ctx.search_lines_by_projection(q, klq, dq, blocks, klt, fnt, dt, occ, cs, ci, BOUNDS, th)
            no, mio, mdo = orc.search_lines_by_projection(q, klq, dq, blocks, klt, fnt, dt, occ, cs, ci, BOUNDS, th)
            assert n == no and np.array_equal(mi, mio) and np.array_equal(md, mdo), (a, b, th)
            if th >= 15.0: assert no > 20
            # every current line claimed already: nothing is found
            n, mi, md = ctx.search_lines_by_projection(q, klq, dq, blocks, klt, fnt, dt, np.ones(len(klt), np.uint8), cs, ci, BOUNDS, th)
            assert n == 0 and np.all(mi == -1)
    finally:
        ctx.close()


def test_search_lines_by_projection_ties_and_claims(hvo, orc, synth):
    """descriptors drawn from a pool of FOUR: every window holds equal distances, so the order of GetFeaturesInAreaForLine's visits decides, and
    queries compete for the same current lines (q_blocks set: a claimed line is passed over by later queries)"""
    g, d, off = synth.make_sequence("std", 0x5EED6300, 2)
    ctx = hvo.Context()
    try:
        klq, _, _ = ctx.extract_lsd(g[0]); klt, _, fnt = ctx.extract_lsd(g[1])
        pool = xorshift_bytes(0x1234, 4 * 32).reshape(4, 32)
        pool[1] = pool[0]; pool[1, 0] ^= 1                              # distance 1 from pool[0]
        dq = pool[np.arange(len(klq)) % 4].copy(); dt = pool[(np.arange(len(klt)) * 3) % 4].copy()
        cs, ci = ctx.assign_lines_to_grid(klt, BOUNDS)
        q = shifted_queries(klq, (off[1] - off[0]).astype(np.float32), jitter_seed=5)
        q = np.concatenate([q, q[::2]]); klq2 = np.concatenate([klq, klq[::2]]); dq2 = np.concatenate([dq, dq[::2]])     # repeated queries: the second finds its line taken
        for blocks in (np.ones(len(q), np.uint8), np.zeros(len(q), np.uint8), (np.arange(len(q)) % 2).astype(np.uint8)):
            n, mi, md = ctx.search_lines_by_projection(q, klq2, dq2, blocks, klt, fnt, dt, np.zeros(len(klt), np.uint8), cs, ci, BOUNDS, 25.0)
            no, mio, mdo = orc.search_lines_by_projection(q, klq2, dq2, blocks, klt, fnt, dt, np.zeros(len(klt), np.uint8), cs, ci, BOUNDS, 25.0)
            assert n == no and np.array_equal(mi, mio) and np.array_equal(md, mdo)
        assert no > 20
    finally:
        ctx.close()


@pytest.mark.parametrize("w,h", [(640, 480), (1280, 960)])
def test_stream_line_tracker_calls(hvo, orc, synth, w, h):
    """both calls between resident frames of a streamed sequence (culled lines + the line grid of HVO_STAGE_GRIDS), at both geometries of BASELINE's configs"""
    n = 4
    g, d, off = synth.make_sequence("std", 0x5EED6400, n, w=w, h=h)
    kw = dict(fx=535.4 * w / 640, fy=539.2 * h / 480, cx=320.1 * w / 640, cy=247.6 * h / 480) if w != 640 else {}
    st = hvo.Stream(depth=4, stages=hvo.STAGE_LSD | hvo.STAGE_LSD_CULL | hvo.STAGE_ORB | hvo.STAGE_GRIDS, bf=0.0, width=w, height=h, **kw)
    try:
        b4 = np.array(st.bounds, np.float32)
        t = [st.submit(g[k]) for k in range(n)]
        r = [st.collect(x) for x in t]
        for a, b in ((0, 1), (1, 2), (2, 3), (0, 3)):
            kl1, d1 = r[a]["kl"], r[a]["ldesc"]; kl2, d2, fn2 = r[b]["kl"], r[b]["ldesc"], r[b]["linefn"]
            hm = (np.arange(len(kl1)) % 5 != 0).astype(np.uint8)
            ng, m, acc = st.match_lines_geom(t[b], t[a], desc_th=0.9, last_has_mapline=hm)
            no, mo, acco = orc.lines_geom_match(d1, kl1, d2, kl2, b4, desc_th=0.9, last_has_mapline=hm)
            assert ng == no and np.array_equal(m, mo) and np.array_equal(acc, acco), (a, b)
            cs, ci, _ = orc.assign_lines_to_grid(kl2, b4)
            assert np.array_equal(r[b]["ln_grid"][0], cs) and np.array_equal(r[b]["ln_grid"][1], ci)
            qi = np.nonzero(np.arange(len(kl1)) % 4 != 1)[0].astype(np.int32)
            q = shifted_queries(kl1, (off[b] - off[a]).astype(np.float32), jitter_seed=11)[qi]
            blocks = (qi % 3 != 0).astype(np.uint8); occ = (np.arange(len(kl2)) % 9 == 0).astype(np.uint8)
            for th in (15.0 * w / 640, 40.0):
                ns, mi, md = st.search_lines_by_projection(t[b], t[a], qi, q, th, q_blocks=blocks, t_occupied=occ)
                no, mio, mdo = orc.search_lines_by_projection(q, kl1[qi], d1[qi], blocks, kl2, fn2, d2, occ, cs, ci, b4, th)
                assert ns == no and np.array_equal(mi, mio) and np.array_equal(md, mdo), (a, b, th)
            assert no > 10
            # pML->GetDescriptor() differing from the frame's own descriptor: q_desc given
            qd = d1[qi][::-1].copy()
            ns, mi, md = st.search_lines_by_projection(t[b], t[a], qi, q, 40.0, q_blocks=blocks, t_occupied=None, q_desc=qd)
            no, mio, mdo = orc.search_lines_by_projection(q, kl1[qi], qd, blocks, kl2, fn2, d2, np.zeros(len(kl2), np.uint8), cs, ci, b4, 40.0)
            assert ns == no and np.array_equal(mi, mio) and np.array_equal(md, mdo)
    finally:
        st.close()
