"""GPU parity of the PEAC/AHC plane extraction (hvo_compute_planes) vs the CPU oracle.
Bar: plane labels bit-exact; plane parameters are fp64 computed in the same operation order --
expected bit-equal, asserted to 1e-9 relative (well inside the 1e-4 of BASELINE.json)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def check(labels_g, planes_g, labels_o, planes_o):
    assert len(planes_g) == len(planes_o), (len(planes_g), len(planes_o))
    assert np.array_equal(planes_g["n_points"], planes_o["n_points"])
    assert np.array_equal(planes_g["rid"], planes_o["rid"])
    for f in ("normal", "center", "mse"):
        assert np.allclose(planes_g[f], planes_o[f], rtol=1e-9, atol=1e-12), f
    assert np.array_equal(labels_g, labels_o), int((labels_g != labels_o).sum())


@pytest.mark.parametrize("seed", [0x5EED0002, 0x5EED1000, 0x5EED1003, 77])
def test_peac_parity_640(gpu_ctx, orc, synth, seed):
    d = synth.make_depth(seed)
    lo, po = orc.peac(d)
    lg, pg = gpu_ctx.compute_planes(d)
    assert len(po) >= 3
    check(lg, pg, lo, po)


def test_peac_exact_plane(gpu_ctx, orc):
    """known answer: an exact plane gives one plane, mse ~ 0, n.c <= 0"""
    h, w = 480, 640
    j = np.arange(w)[None, :]; i = np.arange(h)[:, None]
    z = 2.0 / (0.1 * (j - 320.1) / 535.4 + 0.2 * (i - 247.6) / 539.2 + 1.0)
    d = np.rint(z * 5000).astype(np.uint16)
    lg, pg = gpu_ctx.compute_planes(d)
    lo, po = orc.peac(d)
    check(lg, pg, lo, po)
    assert len(pg) == 1 and pg["n_points"][0] == w * h and pg["mse"][0] < 1e-6
    n = pg["normal"][0]; c = pg["center"][0]
    assert np.dot(n, c) <= 0
    assert np.allclose(np.abs(n), np.array([0.1, 0.2, 1.0]) / np.linalg.norm([0.1, 0.2, 1.0]), atol=1e-3)
    assert np.all(lg == 0)


def test_peac_no_depth_gives_no_planes(gpu_ctx):
    lg, pg = gpu_ctx.compute_planes(np.zeros((480, 640), np.uint16))
    assert len(pg) == 0 and np.all(lg == -1)


def test_peac_holes_and_steps(gpu_ctx, orc):
    """two fronto-parallel planes at different depth + a band of missing data"""
    d = np.full((480, 640), 10000, np.uint16)
    d[:, 330:] = 14000
    d[200:230, :] = 0
    rng = np.random.default_rng(5)
    d = (d.astype(np.int32) + rng.integers(-20, 21, d.shape) * (d > 0)).astype(np.uint16)
    lo, po = orc.peac(d)
    lg, pg = gpu_ctx.compute_planes(d)
    assert len(po) >= 2
    check(lg, pg, lo, po)


def corner_depth(seed, noise, w=640, h=480, cu=320.0, cv=240.0):
    """three planes meeting in one image point (a box corner pointing at the camera) + integer depth noise"""
    j = (np.arange(w)[None, :] - cu) / 535.4; i = (np.arange(h)[:, None] - cv) / 539.2
    zs = [2.0 / (a * j + b * i + 1.0) for a, b in ((0.9, 0.5), (-0.9, 0.5), (0.0, -1.0))]
    z = np.minimum(np.minimum(zs[0], zs[1]), zs[2])
    rng = np.random.default_rng(seed)
    return (np.rint(z * 5000).astype(np.int32) + rng.integers(-noise, noise + 1, z.shape)).clip(1, 65535).astype(np.uint16)


@pytest.mark.parametrize("seed,noise,cu,cv", [(1, 0, 320.0, 240.0), (2, 6, 320.0, 240.0), (3, 25, 323.0, 236.0), (4, 60, 317.5, 243.5)])
def test_peac_three_plane_corner(gpu_ctx, orc, seed, noise, cu, cv):
    """floodFill where three planes race for the same pixels: rounds whose pixel groups have no closed form are replayed
    in rank order / serially (k_peac_flood); labels must still match the sequential loop bit for bit"""
    d = corner_depth(seed, noise, cu=cu, cv=cv)
    lo, po = orc.peac(d)
    lg, pg = gpu_ctx.compute_planes(d)
    assert len(po) >= 3
    check(lg, pg, lo, po)


@pytest.mark.parametrize("flood_t,epl", [(64, 1), (64, 2), (128, 1)])
def test_peac_three_plane_corner_flood_variants(hvo, orc, monkeypatch, flood_t, epl):
    """the one-wave flood variants large batches select, on the scenes that need the ranked and the serial replay"""
    monkeypatch.setenv("HVO_FLOOD_T", str(flood_t))
    monkeypatch.setenv("HVO_FLOOD_EPL", str(epl))
    ctx = hvo.Context()
    try:
        ranked = serial = 0
        for seed, noise, cu, cv in ((3, 25, 323.0, 236.0), (4, 60, 317.5, 243.5)):
            d = corner_depth(seed, noise, cu=cu, cv=cv)
            lo, po = orc.peac(d)
            lg, pg = ctx.compute_planes(d)
            check(lg, pg, lo, po)
            st = ctx.peac_stats(0)
            ranked += st["flood_ranked_rounds"]; serial += st["flood_serial_rounds"]
        assert ranked > 0 and serial > 0, (ranked, serial)
    finally:
        ctx.close()


def polyhedron_depth(seed, w=640, h=480):
    """random convex polyhedral surface (lower envelope of 3-6 planes through random image points), depth noise of a random
    amplitude, random rectangular holes and a few dropped pixels"""
    rng = np.random.default_rng(seed)
    j = (np.arange(w)[None, :] - 320.1) / 535.4; i = (np.arange(h)[:, None] - 247.6) / 539.2
    z = np.full((h, w), np.inf)
    for _ in range(int(rng.integers(3, 7))):
        a, b = rng.uniform(-1.2, 1.2, 2); z0 = rng.uniform(1.2, 3.5)
        ju, iv = rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3)
        den = a * (j - ju) + b * (i - iv) + 1.0
        zz = np.where(den > 0.2, z0 / np.maximum(den, 0.2), np.inf)
        z = np.minimum(z, zz)
    z = np.where(np.isfinite(z), z, 0.0).clip(0, 12.0)
    noise = int(rng.integers(0, 16))
    d = np.rint(z * 5000).astype(np.int64) + (rng.integers(-noise, noise + 1, z.shape) if noise else 0)
    d = np.where(z > 0, d.clip(1, 65535), 0)
    for _ in range(int(rng.integers(0, 4))):
        y0, x0 = int(rng.integers(0, h - 40)), int(rng.integers(0, w - 40))
        d[y0:y0 + int(rng.integers(5, 40)), x0:x0 + int(rng.integers(5, 40))] = 0
    d[rng.random(d.shape) < 0.0003] = 0
    return d.astype(np.uint16)


@pytest.mark.parametrize("flood_t", [256, 64])
def test_peac_random_polyhedra(hvo, orc, monkeypatch, flood_t):
    """randomised scenes with several plane intersections, holes and noise: labels and planes must match the sequential
    reference loop on every one of them (both flood layouts)"""
    monkeypatch.setenv("HVO_FLOOD_T", str(flood_t))
    seeds = list(range(100, 116))
    depth = np.stack([polyhedron_depth(s) for s in seeds])
    gray = np.zeros((len(seeds), 480, 640), np.uint8)
    ctx = hvo.Context(max_batch=len(seeds))
    try:
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_PLANES)
        nplanes = 0
        for f, s in enumerate(seeds):
            lo, po = orc.peac(depth[f])
            assert res[f]["status"] == 0, (s, res[f]["status"])
            check(res[f]["labels"], res[f]["planes"], lo, po)
            nplanes += len(po)
        assert nplanes >= 2 * len(seeds)      # the scenes are not degenerate
    finally:
        ctx.close()


def test_peac_flood_replay_paths_are_exercised(gpu_ctx, orc, synth):
    """the parity scenes must reach all three ways k_peac_flood resolves a round (closed form, ranked, serial replay)"""
    ranked = serial = rounds = 0
    scenes = [synth.make_depth(s) for s in (0x5EED0002, 0x5EED1000, 0x5EED1003, 77)] + [corner_depth(3, 25, cu=323.0, cv=236.0), corner_depth(4, 60, cu=317.5, cv=243.5)]
    for d in scenes:
        lo, po = orc.peac(d)
        lg, pg = gpu_ctx.compute_planes(d)
        check(lg, pg, lo, po)
        st = gpu_ctx.peac_stats(0)
        assert st["flags"] == 0
        rounds += st["flood_rounds"]; ranked += st["flood_ranked_rounds"]; serial += st["flood_serial_rounds"]
    assert rounds > 1000 and ranked > 0 and serial > 0, (rounds, ranked, serial)


@pytest.mark.parametrize("hh,ww", [(397, 501), (473, 638), (480, 335)])
def test_peac_odd_geometry(hvo, orc, synth, hh, ww):
    """image sizes that are not multiples of the 10x10 block (pixels outside the block grid stay -1 unless the flood fill
    reaches them) and of the flood / relabel tile sizes"""
    d = np.ascontiguousarray(synth.make_depth(0x5EED1000)[:hh, :ww])
    lo, po = orc.peac(d)
    ctx = hvo.Context()
    try:
        lg, pg = ctx.compute_planes(d)
    finally:
        ctx.close()
    assert len(po) >= 1
    check(lg, pg, lo, po)


def test_peac_1280(hvo, orc, synth):
    d = synth.make_depth(0x5EED0003, 1280, 960)
    K = synth.intrinsics(1280, 960)
    lo, po = orc.peac(d, fx=np.float32(K["fx"]), fy=np.float32(K["fy"]), cx=np.float32(K["cx"]), cy=np.float32(K["cy"]))
    ctx = hvo.Context(fx=K["fx"], fy=K["fy"], cx=K["cx"], cy=K["cy"])
    try:
        lg, pg = ctx.compute_planes(d)
    finally:
        ctx.close()
    check(lg, pg, lo, po)


def test_peac_wrong_dtype(gpu_ctx, hvo):
    with pytest.raises(hvo.HvoError):
        gpu_ctx.compute_planes(np.zeros((480, 640), np.float32))     # PlaneExtractor.cpp:34-38


def test_peac_batch(hvo, orc, synth):
    gray, depth = synth.make_batch("std", 0x5EED1000, 4)
    ctx = hvo.Context(max_batch=4)
    try:
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_PLANES)
    finally:
        ctx.close()
    for b in range(4):
        lo, po = orc.peac(depth[b])
        assert res[b]["status"] == 0
        check(res[b]["labels"], res[b]["planes"], lo, po)


@pytest.mark.parametrize("slots,lend,poolcap,edges", [("1", "1", None, None), ("0", "1", None, None), ("0", "0", None, None), ("1", "1", "22000", None), ("0", "1", "22000", None),
                                                     ("0", "0", "22000", None), ("1", "1", None, "0"), ("0", "1", None, "0")])
def test_peac_four_frames_per_wave_variants(hvo, orc, synth, monkeypatch, slots, lend, poolcap, edges):
    """the three forms of the four-frames-per-wave AHC -- k_peac_cluster_slots (round 5: the merged node keeps the slot of the longer
    neighbour list; default), ah_cluster_lend (idle lanes lent between the frames, a new record per merge) and ah_cluster_grouped --
    each also with a list pool small enough to be compacted on the way; 10 frames = two full waves and a half-empty one, with an exact
    plane (every candidate ties: labels decide), a three-plane corner and an empty frame among them"""
    monkeypatch.setenv("HVO_PEAC_GL", "16"); monkeypatch.setenv("HVO_PEAC_SLOTS", slots); monkeypatch.setenv("HVO_PEAC_LEND", lend)
    if poolcap: monkeypatch.setenv("HVO_PEAC_POOLCAP", poolcap)
    if edges: monkeypatch.setenv("HVO_PEAC_EDGES", edges)       # initGraph's edges inside the clustering kernel (what 1280x960 frames get)
    j = np.arange(640)[None, :]; i = np.arange(480)[:, None]
    exact = np.rint(2.0 / (0.1 * (j - 320.1) / 535.4 + 0.2 * (i - 247.6) / 539.2 + 1.0) * 5000).astype(np.uint16)
    depth = [synth.make_depth(s) for s in (0x5EED0002, 0x5EED1000, 0x5EED1003, 77, 0x5EED1001, 0x5EED1002, 0x5EED2001)] + [exact, corner_depth(3, 25, cu=323.0, cv=236.0), np.zeros((480, 640), np.uint16)]
    depth = np.stack(depth)
    ctx = hvo.Context(max_batch=len(depth))
    try:
        ctx.batch_upload(np.zeros((len(depth), 480, 640), np.uint8), depth)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_PLANES)
        stats = ctx.peac_stats(1)
    finally:
        ctx.close()
    for b in range(len(depth)):
        lo, po = orc.peac(depth[b])
        assert res[b]["status"] == 0
        check(res[b]["labels"], res[b]["planes"], lo, po)


@pytest.mark.parametrize("gl,flood_t", [(16, 64), (16, 128), (32, 128), (64, 256)])
def test_peac_batch_grouped_paths(hvo, orc, synth, monkeypatch, gl, flood_t):
    """the configurations large batches select (several frames per wave in lockstep, smaller flood
    blocks), forced on a small ragged batch: 6 frames = one full and one half-empty wave at 16 lanes"""
    monkeypatch.setenv("HVO_PEAC_GL", str(gl))
    monkeypatch.setenv("HVO_FLOOD_T", str(flood_t))
    depth = np.stack([synth.make_depth(s) for s in (0x5EED0002, 0x5EED1000, 0x5EED1003, 77, 0x5EED1001, 0x5EED1002)])
    # one exact plane (every candidate merge ties at mse ~ 0) rides along in the same wave
    j = np.arange(640)[None, :]; i = np.arange(480)[:, None]
    depth[4] = np.rint(2.0 / (0.1 * (j - 320.1) / 535.4 + 0.2 * (i - 247.6) / 539.2 + 1.0) * 5000).astype(np.uint16)
    gray = np.zeros((6, 480, 640), np.uint8)
    ctx = hvo.Context(max_batch=6)
    try:
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_PLANES)
    finally:
        ctx.close()
    for b in range(6):
        lo, po = orc.peac(depth[b])
        assert res[b]["status"] == 0
        check(res[b]["labels"], res[b]["planes"], lo, po)


def test_peac_large_batch_adaptive(hvo, orc, synth):
    """3072 resident frames (16 distinct, repeated): the batch size at which the grouped AHC and the small
    flood blocks are selected by themselves; a sample of the results must equal the oracle's"""
    gray, depth = synth.make_batch("std", 0x5EED1000, 16)
    ctx = hvo.Context(max_batch=3072)
    try:
        ctx.batch_upload(gray, depth, repeat=192)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_PLANES, n=40)
    finally:
        ctx.close()
    ref = [orc.peac(depth[b]) for b in range(16)]
    for b in range(40):
        lo, po = ref[b % 16]
        assert res[b]["status"] == 0
        check(res[b]["labels"], res[b]["planes"], lo, po)


@pytest.mark.parametrize("heads,poolcap,big", [("4", None, None), ("3", None, None), ("2", None, None), ("4", "22000", None), ("3", "22000", None), ("0", None, None),
                                               ("3", None, "1"), ("4", "22000", "1"), ("2", None, "1")])
def test_peac_queue_heads(hvo, orc, synth, monkeypatch, heads, poolcap, big):
    """k_peac_cluster_heads (the AHC of small batches: several queue heads per round, one wave each, the longest conflict-free
    prefix committed) against the oracle -- at four and two heads, with a list pool small enough to be compacted on the way,
    and HVO_PEAC_HEADS=0 = the one-wave kernel it replaces for a lone frame.  The scenes: textured rooms, an exact plane (every
    candidate ties at mse ~ 0: the tie rule and the created ids decide), a three-plane corner, no depth at all."""
    monkeypatch.setenv("HVO_PEAC_HEADS", heads)
    if poolcap: monkeypatch.setenv("HVO_PEAC_POOLCAP", poolcap)
    # big: the form for frames whose queue does not fit LDS (1280x960: keys in global memory, list headers in the node records), forced
    # onto 640x480 frames where every scene of this test exists
    if big: monkeypatch.setenv("HVO_PEAC_HEADS_BIG", big)
    j = np.arange(640)[None, :]; i = np.arange(480)[:, None]
    exact = np.rint(2.0 / (0.1 * (j - 320.1) / 535.4 + 0.2 * (i - 247.6) / 539.2 + 1.0) * 5000).astype(np.uint16)
    depth = [synth.make_depth(s) for s in (0x5EED0002, 0x5EED1000, 0x5EED1003, 77)] + [exact, corner_depth(3, 25, cu=323.0, cv=236.0), np.zeros((480, 640), np.uint16)]
    depth = np.stack(depth)
    ctx = hvo.Context(max_batch=len(depth))
    try:
        ctx.batch_upload(np.zeros((len(depth), 480, 640), np.uint8), depth)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_PLANES)
        stats = ctx.peac_stats(1)
        one = ctx.compute_planes(depth[1])                       # the single-frame entry point takes the same kernel
    finally:
        ctx.close()
    for b in range(len(depth)):
        lo, po = orc.peac(depth[b])
        assert res[b]["status"] == 0
        check(res[b]["labels"], res[b]["planes"], lo, po)
    lo, po = orc.peac(depth[1])
    check(one[0], one[1], lo, po)
    # the speculation must pay, not only be right: rounds per merge (tools/ahc_spec_sim.c: 0.47 at four heads, 0.64 at two).  Marks that
    # are never cleared, say, keep every result exact and commit one head per round.
    merges = stats["segments"] - 3072
    if heads == "4": assert 0 < stats["ahc_rounds"] < 0.55 * merges, stats
    if heads == "3": assert 0 < stats["ahc_rounds"] < 0.6 * merges, stats
    if heads == "2": assert 0 < stats["ahc_rounds"] < 0.75 * merges, stats


def test_peac_four_frames_per_wave_1280(hvo, orc, synth, monkeypatch):
    """the slot AHC at 1280x960 (12 288 blocks: initGraph's edges inside the kernel, 15-bit labels, a queue of 48 super-buckets), forced onto six frames"""
    monkeypatch.setenv("HVO_PEAC_GL", "16")
    w, h = 1280, 960
    g, d = synth.make_batch("std", 0x5EED4400, 5, w, h)
    g2, d2 = synth.make_batch("lowtex", 0x5EED4500, 1, w, h)
    d = np.concatenate([d, d2])
    ctx = hvo.Context(max_batch=len(d))
    try:
        ctx.batch_upload(np.zeros((len(d), h, w), np.uint8), d)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_PLANES)
    finally:
        ctx.close()
    for b in range(len(d)):
        lo, po = orc.peac(d[b])
        assert res[b]["status"] == 0
        check(res[b]["labels"], res[b]["planes"], lo, po)
