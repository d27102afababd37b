"""The alternative readings of cv::GaussianBlur / cv::LineSegmentDetector in the PRODUCT (hvo_set_readings, csrc/readings.hip; round 5): with the
oracle's switch (orc_set_reading) and the library's flipped on both sides the results are equal -- every combination, ORB and lines, single
calls, a batch with the culled lines, and a streamed frame -- and flipping back restores the defaults."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def readings(orc):
    def set_both(ctx, blur_float, lsd_8u):
        orc.set_reading("blur_float", int(blur_float)); orc.set_reading("lsd_8u", int(lsd_8u))
        ctx.set_readings(blur_float=blur_float, lsd_8u=lsd_8u)
    yield set_both
    for name in orc.READINGS:
        orc.set_reading(name, 0)


@pytest.mark.parametrize("blur_float,lsd_8u", [(True, False), (False, True), (True, True)])
def test_readings_single_calls(hvo, orc, synth, readings, blur_float, lsd_8u):
    from test_lsd_gpu import check as check_lines
    from test_orb_gpu import check_orb
    ctx = hvo.Context()
    try:
        for kind, seed, hh, ww in (("std", 0x5EED8100, 480, 640), ("lowtex", 0x5EED8101, 480, 640), ("std", 0x5EED8102, 397, 501)):
            g = np.ascontiguousarray(synth.make_gray(kind, seed)[:hh, :ww])
            readings(ctx, False, False)
            kp0, d0 = ctx.extract_orb(g); kl0, ld0, fn0 = ctx.extract_lsd(g)
            readings(ctx, blur_float, lsd_8u)
            kp1, d1 = ctx.extract_orb(g)
            check_orb(kp1, d1, *orc.Orb().extract(g))
            kl1, ld1, fn1 = ctx.extract_lsd(g)
            check_lines(kl1, ld1, fn1, *orc.line_extract(g))
            klc, ldc, fnc = ctx.extract_lsd(g, culled=True)
            klo, ldo, fno = orc.line_extract(g)
            klco, ldco, fnco = orc.cull_lines(g, klo, fno)
            assert len(klc) == len(klco) and np.array_equal(ldc, ldco)
            # what each reading reaches: the float blur moves descriptors only; the u8 detector moves the lines themselves
            assert np.array_equal(kp1, kp0)
            if blur_float: assert not np.array_equal(d1, d0) or len(d0) == 0
            else: assert np.array_equal(d1, d0)
            if lsd_8u and kind == "std": assert len(kl1) != len(kl0) or not np.array_equal(kl1["sx"], kl0["sx"])
            if not lsd_8u: assert np.array_equal(kl1["sx"], kl0["sx"]) and (not np.array_equal(ld1, ld0) or len(ld0) == 0)
        readings(ctx, False, False)                                # and back: the defaults did not move
        kp2, d2 = ctx.extract_orb(g); kl2, ld2, fn2 = ctx.extract_lsd(g)
        assert np.array_equal(d2, d0) and np.array_equal(ld2, ld0) and np.array_equal(kl2, kl0)
    finally:
        ctx.close()


def test_readings_batch_and_unfused_orb(hvo, orc, synth, readings, monkeypatch):
    """a small resident batch (chunked scratch, the unfused ORB kernels forced: their blur is replaced too) under both readings"""
    from test_lsd_gpu import check as check_lines
    monkeypatch.setenv("HVO_ORB_FUSED", "0"); monkeypatch.setenv("HVO_ORB_CHUNK", "2"); monkeypatch.setenv("HVO_LSD_CHUNK", "2")
    g, d = synth.make_batch("std", 0x5EED8200, 5)
    ctx = hvo.Context(max_batch=5)
    try:
        readings(ctx, True, True)
        ctx.batch_upload(g, d)
        ctx.batch_run(hvo.STAGE_ORB | hvo.STAGE_LSD)
        res = ctx.batch_download(hvo.STAGE_ORB | hvo.STAGE_LSD)
        o = orc.Orb()
        for b in range(5):
            assert np.array_equal(res[b]["desc"], o.extract(g[b])[1])
            check_lines(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], *orc.line_extract(g[b]))
    finally:
        ctx.close()


def test_readings_streamed(hvo, orc, synth, readings):
    g, d, _ = synth.make_sequence("std", 0x5EED8300, 2)
    st = hvo.Stream(depth=2, stages=hvo.STAGE_ORB | hvo.STAGE_LSD, bf=0.0)
    try:
        orc.set_reading("blur_float", 1); orc.set_reading("lsd_8u", 1)
        st.set_readings(blur_float=True, lsd_8u=True)
        r = st.collect(st.submit(g[0]))
        assert np.array_equal(r["desc"], orc.Orb().extract(g[0])[1])
        klo, ldo, _ = orc.line_extract(g[0])
        assert len(r["kl"]) == len(klo) and np.array_equal(r["ldesc"], ldo)
    finally:
        st.close()
