"""GPU parity of LINEextractor::operator() (LSD detect + top-N + LBD + line functions) vs the oracle.
Bar: key-line structure (count, order, class ids, pixel counts) and descriptor bytes bit-exact;
float geometry within 1e-4 (north_star tolerance) -- expected to be bit-equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def check(kl_g, d_g, fn_g, kl_o, d_o, fn_o):
    assert len(kl_g) == len(kl_o), (len(kl_g), len(kl_o))
    for f in ("class_id", "octave", "num_pixels"):
        assert np.array_equal(kl_g[f], kl_o[f]), f
    for f in ("angle", "pt_x", "pt_y", "response", "sx", "sy", "ex", "ey", "sox", "soy", "eox", "eoy", "length"):
        assert np.allclose(kl_g[f], kl_o[f], rtol=0, atol=TOL), f
    assert np.allclose(kl_g["size"], kl_o["size"], rtol=1e-6, atol=1e-2)
    assert np.array_equal(d_g, d_o), int((d_g != d_o).sum())
    assert np.allclose(fn_g, fn_o, rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize("kind,seed", [("std", 0x5EED0002), ("lowtex", 0x5EED0001), ("std", 0x5EED1001), ("std", 9)])
def test_lines_parity_640(gpu_ctx, orc, synth, kind, seed):
    g = synth.make_gray(kind, seed)
    kl_o, d_o, fn_o = orc.line_extract(g)
    kl_g, d_g, fn_g = gpu_ctx.extract_lsd(g)
    assert len(kl_o) > 10
    check(kl_g, d_g, fn_g, kl_o, d_o, fn_o)


@pytest.mark.parametrize("dense", ["0", "1"])
def test_lines_both_grow_kernels(hvo, orc, synth, monkeypatch, dense):
    """k_lsd_grow (five waves per SIMD, small batches) and k_lsd_grow_dense (eight, batches that fill the wave slots) are the
    same body under two register budgets; HVO_LSD_DENSE forces either on a small batch"""
    monkeypatch.setenv("HVO_LSD_DENSE", dense)
    g = np.stack([synth.make_gray(k, s) for k, s in (("std", 0x5EED0002), ("lowtex", 0x5EED0001), ("std", 0x5EED1003))])
    ctx = hvo.Context(max_batch=3)
    try:
        ctx.batch_upload(g, np.zeros((3, 480, 640), np.uint16))
        ctx.batch_run(hvo.STAGE_LSD)
        res = ctx.batch_download(hvo.STAGE_LSD)
    finally:
        ctx.close()
    for b in range(3):
        kl_o, d_o, fn_o = orc.line_extract(g[b])
        check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], kl_o, d_o, fn_o)


@pytest.mark.parametrize("dense,frac,chunk", [("0", None, None), ("1", None, "2"), ("0", "0.01", None)])
def test_lines_compact_records(hvo, orc, synth, monkeypatch, dense, frac, chunk):
    """Plans of large resident batches keep a 32-byte record only for the pixels that have a gradient angle (k_lsd_prefix / _bases / _compact;
    the growing kernels reach them through the defined mask's prefix counts).  HVO_LSD_COMPACT forces the layout on a small batch: both growing
    kernels, chunks smaller than the batch, an odd geometry, and a pool far too small (HVO_LSD_COMPACT_FRAC: lsd_run must grow it and come again)."""
    monkeypatch.setenv("HVO_LSD_COMPACT", "1")
    monkeypatch.setenv("HVO_LSD_DENSE", dense)
    if frac: monkeypatch.setenv("HVO_LSD_COMPACT_FRAC", frac)
    if chunk: monkeypatch.setenv("HVO_LSD_CHUNK", chunk)
    for hh, ww in ((480, 640), (397, 501)):
        g = np.stack([synth.make_gray(k, s)[:hh, :ww] for k, s in (("std", 0x5EED0002), ("lowtex", 0x5EED0001), ("std", 0x5EED1003), ("std", 9), ("std", 0x5EED1001))])
        g = np.ascontiguousarray(g)
        ctx = hvo.Context(max_batch=5)
        try:
            ctx.batch_upload(g, np.zeros((5, hh, ww), np.uint16))
            for _ in range(2):                                    # the second run starts from the pool the first one left
                ctx.batch_run(hvo.STAGE_LSD)
                res = ctx.batch_download(hvo.STAGE_LSD)
                for b in range(5):
                    kl_o, d_o, fn_o = orc.line_extract(g[b])
                    assert res[b]["status"] == 0
                    check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], kl_o, d_o, fn_o)
        finally:
            ctx.close()


@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("hh,ww", [(480, 640), (397, 501), (960, 1280)])
def test_lines_preamble_fused_and_split(hvo, orc, synth, monkeypatch, split, hh, ww):
    """k_lsd_pre (one kernel from the u8 image to the gradient records, streaming down bands of 192 scaled columns) and the pair
    k_lsd_blur + k_lsd_resize_grad with its fp64 image between them (HVO_LSD_PRE_SPLIT=1) are the same arithmetic in the same order:
    both against the oracle, on a batch (bands x segments x frames), an odd geometry (a partial last band and segment) and 1280x960"""
    monkeypatch.setenv("HVO_LSD_PRE_SPLIT", split)
    big = synth.make_gray("std", 12, 1280, 960) if ww > 704 else synth.make_gray("std", 11, 704, 480)
    g = np.ascontiguousarray(np.stack([big[:hh, :ww], np.flipud(big[:hh, :ww])]))
    ctx = hvo.Context(max_batch=2)
    try:
        ctx.batch_upload(g, np.zeros((2, hh, ww), np.uint16))
        for _ in range(2):
            ctx.batch_run(hvo.STAGE_LSD)
            res = ctx.batch_download(hvo.STAGE_LSD)
            for b in range(2):
                kl_o, d_o, fn_o = orc.line_extract(g[b])
                assert res[b]["status"] == 0 and len(kl_o) > 10
                check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], kl_o, d_o, fn_o)
    finally:
        ctx.close()


@pytest.mark.parametrize("hh,ww", [(397, 501), (479, 638), (400, 642)])
def test_lines_odd_geometry(hvo, orc, synth, hh, ww):
    """widths that are not a multiple of 4 (scalar tails of the LBD blur / Sobel strips, unaligned Sobel rows) and lines
    that end at the right / bottom border"""
    big = synth.make_gray("std", 11, 704, 480)
    g = np.ascontiguousarray(big[:hh, :ww])
    kl_o, d_o, fn_o = orc.line_extract(g)
    ctx = hvo.Context()
    try:
        kl_g, d_g, fn_g = ctx.extract_lsd(g)
    finally:
        ctx.close()
    assert len(kl_o) > 10
    check(kl_g, d_g, fn_g, kl_o, d_o, fn_o)


def test_lines_exactness_report(gpu_ctx, orc, synth):
    """how exact is 'within 1e-4'?  key-line floats are expected bit-equal on these frames"""
    g = synth.make_gray("std", 0x5EED0002)
    kl_o, d_o, fn_o = orc.line_extract(g)
    kl_g, d_g, fn_g = gpu_ctx.extract_lsd(g)
    same = all(np.array_equal(kl_g[f], kl_o[f]) for f in kl_o.dtype.names)
    assert same or np.allclose(kl_g["sx"], kl_o["sx"], atol=TOL)


def test_lines_flat_image(gpu_ctx):
    kl, d, fn = gpu_ctx.extract_lsd(np.full((480, 640), 90, np.uint8))
    assert len(kl) == 0


def test_lines_empty_image(gpu_ctx):
    kl, d, fn = gpu_ctx.extract_lsd(np.zeros((0, 0), np.uint8))     # LineExtractor.cpp:331-332
    assert len(kl) == 0


def test_lines_wrong_dtype(gpu_ctx, hvo):
    with pytest.raises(hvo.HvoError):
        gpu_ctx.extract_lsd(np.zeros((480, 640), np.float32))


def test_lines_few_lines_keep_detection_order(gpu_ctx, orc):
    """fewer than nLSDFeature lines: no sort, class_id = detection index (LineExtractor.cpp:351)"""
    g = np.full((480, 640), 60, np.uint8)
    g[100:300, 200:420] = 180
    g[350:420, 50:600] = 20
    kl_o, d_o, fn_o = orc.line_extract(g)
    kl_g, d_g, fn_g = gpu_ctx.extract_lsd(g)
    assert 4 <= len(kl_o) < 200
    assert np.array_equal(kl_o["class_id"], np.arange(len(kl_o)))
    check(kl_g, d_g, fn_g, kl_o, d_o, fn_o)


def test_lines_1280(hvo, orc, synth):
    g = synth.make_gray("std", 0x5EED0003, 1280, 960)
    kl_o, d_o, fn_o = orc.line_extract(g)
    ctx = hvo.Context()
    try:
        kl_g, d_g, fn_g = ctx.extract_lsd(g)
    finally:
        ctx.close()
    check(kl_g, d_g, fn_g, kl_o, d_o, fn_o)


def test_lines_batch_and_all_stages(hvo, orc, synth):
    gray, depth = synth.make_batch("std", 0x5EED1000, 3)
    ctx = hvo.Context(max_batch=3)
    try:
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_ALL)
        res = ctx.batch_download(hvo.STAGE_ALL)
    finally:
        ctx.close()
    o = orc.Orb()
    for b in range(3):
        assert res[b]["status"] == 0
        kl_o, d_o, fn_o = orc.line_extract(gray[b])
        check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], kl_o, d_o, fn_o)
        kp_o, dd_o = o.extract(gray[b])
        assert np.array_equal(res[b]["desc"], dd_o) and len(res[b]["kp"]) == len(kp_o)
        lo, po = orc.peac(depth[b])
        assert np.array_equal(res[b]["labels"], lo) and len(res[b]["planes"]) == len(po)


def test_all_stages_batch_odd_geometry(hvo, orc, synth):
    """the resident-batch path at a geometry where nothing is a multiple of anything: 3 frames of 501 x 397"""
    gray0, depth0 = synth.make_batch("std", 0x5EED1000, 3)
    gray = np.ascontiguousarray(gray0[:, :397, :501]); depth = np.ascontiguousarray(depth0[:, :397, :501])
    ctx = hvo.Context(max_batch=3)
    try:
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_ALL)
        res = ctx.batch_download(hvo.STAGE_ALL)
    finally:
        ctx.close()
    o = orc.Orb()
    for b in range(3):
        assert res[b]["status"] == 0
        kl_o, d_o, fn_o = orc.line_extract(gray[b])
        check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], kl_o, d_o, fn_o)
        kp_o, dd_o = o.extract(gray[b])
        assert np.array_equal(res[b]["desc"], dd_o) and len(res[b]["kp"]) == len(kp_o)
        for f in ("x", "y", "octave"):
            assert np.array_equal(res[b]["kp"][f], kp_o[f]), f
        lo, po = orc.peac(depth[b])
        assert np.array_equal(res[b]["labels"], lo) and len(res[b]["planes"]) == len(po)


def test_all_stages_soak(hvo, orc, synth):
    """40 more frames (24 'std', 16 'lowtex' scenes, fresh seeds) through the resident-batch path, every stage against the
    oracle: the speculative region growing, the flood replay paths and the quadtree ties are data dependent"""
    g1, d1 = synth.make_batch("std", 0xA11CE000, 24)
    g2, d2 = synth.make_batch("lowtex", 0xB0B0000, 16)
    gray = np.concatenate([g1, g2]); depth = np.concatenate([d1, d2])
    n = len(gray)
    ctx = hvo.Context(max_batch=n)
    try:
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_ALL)
        res = ctx.batch_download(hvo.STAGE_ALL)
    finally:
        ctx.close()
    o = orc.Orb()
    bad = []
    for b in range(n):
        ok = res[b]["status"] == 0
        kl_o, d_o, fn_o = orc.line_extract(gray[b])
        ok = ok and len(res[b]["kl"]) == len(kl_o) and np.array_equal(res[b]["ldesc"], d_o) and np.array_equal(res[b]["kl"]["num_pixels"], kl_o["num_pixels"]) \
            and np.allclose(res[b]["kl"]["sx"], kl_o["sx"], atol=TOL) and np.allclose(res[b]["kl"]["ey"], kl_o["ey"], atol=TOL)
        kp_o, dd_o = o.extract(gray[b])
        ok = ok and len(res[b]["kp"]) == len(kp_o) and np.array_equal(res[b]["desc"], dd_o) and np.array_equal(res[b]["kp"]["x"], kp_o["x"]) \
            and np.array_equal(res[b]["kp"]["y"], kp_o["y"])
        lo, po = orc.peac(depth[b])
        ok = ok and np.array_equal(res[b]["labels"], lo) and len(res[b]["planes"]) == len(po)
        if not ok:
            bad.append(b)
    assert not bad, bad


@pytest.mark.parametrize("workers,early", [("4", "1"), ("16", "0"), ("32", "1"), ("7", "0")])
def test_lines_async_growing(hvo, orc, synth, monkeypatch, workers, early):
    """k_lsd_grow_async (lsd_async.inc): W waves per frame grow regions side by side and commit them in seed order after validation --
    the committed sequence must be the sequential run's.  Forced onto a ragged 5-frame batch (odd geometry, a low-texture frame), with
    and without the early drop of void results, and checked to have speculated at all and to have kept a frame's workers on one XCD
    (frame 0's control block)."""
    import ctypes
    monkeypatch.setenv("HVO_LSD_ASYNC", workers); monkeypatch.setenv("HVO_LSD_ASYNC_EARLY", early)
    for hh, ww in ((480, 640), (397, 501)):
        g = np.ascontiguousarray(np.stack([synth.make_gray(k, s)[:hh, :ww] for k, s in (("std", 0x5EED0002), ("lowtex", 0x5EED0001), ("std", 0x5EED1003), ("std", 9), ("std", 0x5EED1001))]))
        ctx = hvo.Context(max_batch=5)
        try:
            ctx.batch_upload(g, np.zeros((5, hh, ww), np.uint16))
            for _ in range(2):                                    # the second run starts from the tags and lists the first one left
                ctx.batch_run(hvo.STAGE_LSD)
                res = ctx.batch_download(hvo.STAGE_LSD)
                for b in range(5):
                    kl_o, d_o, fn_o = orc.line_extract(g[b])
                    assert res[b]["status"] == 0
                    check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], kl_o, d_o, fn_o)
            L = hvo.lib(); L.hvo_debug_lsd_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            ac = (ctypes.c_uint * 256)()
            assert L.hvo_debug_lsd_async(ctx.h, 0, ac) == 0
            ctl = list(ac)
            assert ctl[3] == 1 and ctl[6] == 0                      # done, not aborted
            regions, spec, valid = ctl[8], ctl[9], ctl[10]
            assert regions > 100 and valid > regions // 2 and spec >= valid, (regions, spec, valid)      # most regions were committed from a speculative growth
            assert ctl[32] != 0 and ctl[33] == 0                    # one XCD, nobody elsewhere
        finally:
            ctx.close()


def test_lines_async_1280(hvo, orc, synth, monkeypatch):
    """the async growing at 1280x960 (5.4 k seeds, regions of up to thousands of points: private lists that overflow go to the frontier)"""
    monkeypatch.setenv("HVO_LSD_ASYNC", "32")
    g = synth.make_gray("std", 0x5EED0003, 1280, 960)
    kl_o, d_o, fn_o = orc.line_extract(g)
    ctx = hvo.Context()
    try:
        kl_g, d_g, fn_g = ctx.extract_lsd(g)
    finally:
        ctx.close()
    check(kl_g, d_g, fn_g, kl_o, d_o, fn_o)


def test_lbd_float_sqrt_is_correctly_rounded(hvo, orc, synth):
    """A 3-pixel vertical key line of this frame has a band deviation sqrt(0x1.b8a4d0p-5): `__fsqrt_rn` of this ROCm's headers is
    v_sqrt_f32 (0x1.dafbdap-3, one ulp low) and flipped one bit of the line's LBD descriptor; sqrtf() is correctly rounded
    (tools/microbench/sqrt_check.hip).  Found by tools/soak_suite.sh long, frame 26 of its second run."""
    g = synth.make_gray("lowtex", 0xE00E1000 + 26)
    kl_o, d_o, fn_o = orc.line_extract(g)
    ctx = hvo.Context()
    try:
        kl, dsc, fn = ctx.extract_lsd(g)
    finally:
        ctx.close()
    assert len(kl) == len(kl_o) and int(kl_o["num_pixels"][13]) == 3
    assert np.array_equal(dsc, d_o)


def test_lines_async_stolen_tags_stress(hvo, orc, synth, monkeypatch):
    """Low-texture frames on which round 4's first async kernel produced, once in ~100 runs, a degenerate extra segment at a seed: an older
    region took a pixel's owner tag from a growing region and let it go again (dropped as void), the pixel read as free and was added a
    second time; at the frontier a committed region's not-yet-removed tag did the same.  Membership is a bitmap of the worker's own now
    (lsd_async.inc), the frontier takes tags unconditionally.  8 workers per frame made it most frequent: 16 frames x 24 runs."""
    monkeypatch.setenv("HVO_LSD_ASYNC", "8")
    gray, _ = synth.make_batch("lowtex", 0xF00F4000 + 16, 16)
    gray = np.ascontiguousarray(gray[:, :479, :638])
    ref = [orc.line_extract(g) for g in gray]
    ctx = hvo.Context(max_batch=16)
    try:
        for rep in range(24):
            ctx.batch_upload(gray, np.zeros((16, 479, 638), np.uint16)); ctx.batch_run(hvo.STAGE_LSD); res = ctx.batch_download(hvo.STAGE_LSD)
            for b in range(16):
                kl_o, d_o, _ = ref[b]
                assert res[b]["status"] == 0 and len(res[b]["kl"]) == len(kl_o) and np.array_equal(res[b]["ldesc"], d_o), (rep, b, len(res[b]["kl"]), len(kl_o))
    finally:
        ctx.close()


def test_lines_more_segments_than_a_fixed_cap(hvo, orc):
    """A diagonal checkerboard of period 13 has ~6600 line segments before the 200 longest are kept (LineExtractor.cpp:351-360); a fixed capacity of 4096
    turned it into HVO_ERR_CAPACITY.  The capacity is what the scaled image can hold now (pixels / min_reg_size).  Lone frame (async growing) and a batch."""
    y, x = np.mgrid[0:480, 0:640]
    g = np.where(((x + y) // 13 + (x - y + 4096) // 13) % 2 == 0, 30, 220).astype(np.uint8)
    assert len(orc.lsd_detect(g)) > 4096
    kl_o, d_o, fn_o = orc.line_extract(g)
    ctx = hvo.Context(max_batch=3)
    try:
        kl, dsc, fn = ctx.extract_lsd(g)
        check(kl, dsc, fn, kl_o, d_o, fn_o)
        ctx.batch_upload(np.stack([g, g[::-1].copy(), g]), np.zeros((3, 480, 640), np.uint16)); ctx.batch_run(hvo.STAGE_LSD); res = ctx.batch_download(hvo.STAGE_LSD)
        assert all(r["status"] == 0 for r in res)
        check(res[0]["kl"], res[0]["ldesc"], res[0]["linefn"], kl_o, d_o, fn_o)
        check(res[2]["kl"], res[2]["ldesc"], res[2]["linefn"], kl_o, d_o, fn_o)
    finally:
        ctx.close()



@pytest.mark.parametrize("n,workers", [(1, "32"), (5, "8")])
def test_lines_async_gives_up_and_the_frame_is_grown_again(hvo, orc, synth, monkeypatch, n, workers):
    """VERDICT r4 (fail soft): a frame on which the async growing's bounded wait expires used to come back as HVO_ERR_CAPACITY with no lines.
    HVO_LSD_ASYNC_SPIN_MAX=0 (a test hook, read when the plan is built) makes every wait of 16 polls expire: the frames are formed again
    and grown by the one-wave kernel inside the same hvo_batch_run -- status 0, the oracle's lines, and the report says so."""
    monkeypatch.setenv("HVO_LSD_ASYNC", workers); monkeypatch.setenv("HVO_LSD_ASYNC_SPIN_MAX", "0")
    g = np.stack([synth.make_gray("std" if k != 1 else "lowtex", 0x5EED7300 + k) for k in range(n)])
    ctx = hvo.Context(max_batch=n)
    try:
        ctx.batch_upload(g, np.zeros((n, 480, 640), np.uint16))
        regrown_total = 0
        for _ in range(2):
            ctx.batch_run(hvo.STAGE_LSD)
            res = ctx.batch_download(hvo.STAGE_LSD)
            regrown, foreign, wpf = ctx.lsd_async_report()
            assert wpf == int(workers) and foreign == 0 and 0 <= regrown <= n
            regrown_total += regrown
            for b in range(n):
                assert res[b]["status"] == 0
                check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], *orc.line_extract(g[b]))
        assert regrown_total > 0                                    # the hook did force the fallback
    finally:
        ctx.close()
    # and without the hook nothing is grown twice
    monkeypatch.delenv("HVO_LSD_ASYNC_SPIN_MAX")
    ctx = hvo.Context(max_batch=n)
    try:
        ctx.batch_upload(g, np.zeros((n, 480, 640), np.uint16)); ctx.batch_run(hvo.STAGE_LSD)
        assert ctx.lsd_async_report()[0] == 0
    finally:
        ctx.close()


def test_lines_async_growing_on_a_mid_size_batch(hvo, orc, synth, monkeypatch):
    """round 5: the async growing's scratch is allocated per (frames, workers), so HVO_LSD_ASYNC = W works beyond 16 frames (it is not the
    default there: measured slower, profiles/r05_async_midsize_batches.txt).  40 frames x 3 workers, then the same context on 12 frames
    x 3 (smaller: the scratch is kept), every frame against the oracle."""
    monkeypatch.setenv("HVO_LSD_ASYNC", "3")
    g = np.stack([synth.make_gray("std" if k % 5 else "lowtex", 0x5EED7400 + k) for k in range(40)])
    ref = [orc.line_extract(g[b]) for b in range(40)]
    ctx = hvo.Context(max_batch=40)
    try:
        for n in (40, 12):
            ctx.batch_upload(g[:n], np.zeros((n, 480, 640), np.uint16))
            ctx.batch_run(hvo.STAGE_LSD)
            res = ctx.batch_download(hvo.STAGE_LSD)
            assert ctx.lsd_async_report()[2] == 3
            for b in range(n):
                assert res[b]["status"] == 0
                check(res[b]["kl"], res[b]["ldesc"], res[b]["linefn"], *ref[b])
    finally:
        ctx.close()


def test_culling_with_a_large_line_quota(hvo, orc, synth):
    """round 5: Frame::cullingLine holds its lines in dynamic LDS sized by the plan's quota (up to 2048; 512 and an error before).  nLSDFeature = 1500
    on a dense periodic pattern (thousands of segments before the quota): the culled lines against the oracle."""
    yy, xx = np.mgrid[0:480, 0:640]
    g = (((xx + yy) // 13 + (xx - yy) // 13) % 2 * 90 + 60).astype(np.uint8)             # a diagonal checkerboard of period 13
    g = np.clip(g.astype(np.int32) + (synth.make_gray("std", 0x5EED7500).astype(np.int32) - 120) // 8, 0, 255).astype(np.uint8)
    ctx = hvo.Context(lsd_nfeatures=1500)
    try:
        kl, ld, fn = ctx.extract_lsd(g, culled=True)
    finally:
        ctx.close()
    klo, ldo, fno = orc.line_extract(g, nfeatures=1500)
    assert len(klo) > 600                                          # beyond the old 512-line limit
    klc, ldc, fnc = orc.cull_lines(g, klo, fno)
    assert len(kl) == len(klc) and np.array_equal(ld, ldc)
    for k in ("sx", "sy", "ex", "ey"): assert np.array_equal(kl[k], klc[k]), k


def test_reduce_region_radius_closed_form(hvo):
    """reduce_region_radius (OpenCV 3.2 lsd.cpp) removes the points outside a radius with `swap(reg[i], reg[size - 1]); --size; --i`: the
    order it leaves behind feeds the ordered fp64 sums of region2rect.  csrc/lsd.hip evaluates that loop in closed form on 64 lanes
    (reduce_radius_wave); here: random lists of every length class (chunk boundaries of 64, the 4096-entry limit, nothing / everything
    removed) against the loop itself -- the kept prefix in the loop's order, the released set, the tail as a set."""
    import ctypes as C
    L = hvo.lib()
    L.hvo_debug_reduce_radius.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.hvo_debug_reduce_radius.restype = C.c_int
    rng = np.random.RandomState(0xC105ED)
    W, H = 512, 384
    def loop(reg, xc, yc, rad):
        reg = list(reg); n = len(reg); i = 0; rel = []
        while i < n:
            a = reg[i]; dx = (a & 0xFFFF) - xc; dy = (a >> 16) - yc
            if dx * dx + dy * dy > rad:
                rel.append(a); reg[i], reg[n - 1] = reg[n - 1], reg[i]; n -= 1
            else: i += 1
        return reg[:n], set(rel), set(reg[n:])
    sizes = [1, 2, 3, 63, 64, 65, 127, 128, 129, 500, 1000, 2047, 2048, 2049, 4095, 4096] + [int(x) for x in rng.randint(2, 4096, 40)]
    for n in sizes:
        for frac in (0.0, 0.1, 0.5, 0.9, 1.0, None):
            pts = rng.choice(W * H, n, replace=False)
            reg = ((pts // W) << 16 | (pts % W)).astype(np.int32)
            xc, yc = float(rng.randint(0, W)), float(rng.randint(0, H))
            d2 = ((reg & 0xFFFF) - xc) ** 2 + ((reg >> 16) - yc) ** 2
            if frac is None: rad = float(rng.choice(d2))                                   # ties at the boundary: `>` keeps them
            elif frac == 0.0: rad = -1.0
            elif frac == 1.0: rad = float(d2.max())
            else: rad = float(np.quantile(d2, frac))
            kept_o, rel_o, tail_o = loop(reg.tolist(), xc, yc, rad)
            out = reg.copy(); released = np.zeros(W * H, np.uint8); nk = C.c_int(0)
            assert L.hvo_debug_reduce_radius(out.ctypes.data_as(C.c_void_p), n, xc, yc, rad, W, H, released.ctypes.data_as(C.c_void_p), C.byref(nk)) == 0
            assert nk.value == len(kept_o), (n, frac)
            assert out[:nk.value].tolist() == kept_o, (n, frac)
            assert set(out[nk.value:].tolist()) == tail_o
            r = np.nonzero(released)[0]
            assert set(((r // W) << 16 | (r % W)).tolist()) == rel_o
