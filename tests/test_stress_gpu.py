"""Pathological inputs through the single-frame entry points, HIP vs oracle: checkerboards (every cell full of corners,
candidate lists at capacity), binary / uniform noise, thin line grids (more segments than are kept), saturated images,
depth images made of steps, stripes of missing data and full-range noise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H, W = 480, 640


def gray_images(synth):
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:H, 0:W]
    out = {}
    out["checker8"] = (((yy // 8 + xx // 8) & 1) * 255).astype(np.uint8)
    out["checker17_lowc"] = (100 + ((yy // 17 + xx // 17) & 1) * 40).astype(np.uint8)
    out["binary_noise"] = (rng.integers(0, 2, (H, W)) * 255).astype(np.uint8)
    out["uniform_noise"] = rng.integers(0, 256, (H, W)).astype(np.uint8)
    out["gradient"] = ((xx * 255) // (W - 1)).astype(np.uint8)
    st = np.zeros((H, W), np.uint8) + 60
    for k in range(0, W, 23): st[:, k:k + 2] = 220
    for k in range(0, H, 31): st[k:k + 1, :] = 20
    out["thin_lines"] = st
    d = np.zeros((H, W), np.int32) + 128
    for _ in range(60):
        a = rng.uniform(0, np.pi); off = rng.uniform(-300, 700)
        d += ((xx * np.cos(a) + yy * np.sin(a)) > off) * int(rng.integers(-30, 31))
    out["halfplanes"] = np.clip(d + rng.integers(-3, 4, (H, W)), 0, 255).astype(np.uint8)
    base = synth.make_gray("std", 5)
    out["std_noise40"] = np.clip(base.astype(np.int32) + rng.integers(-40, 41, (H, W)), 0, 255).astype(np.uint8)
    out["saturated"] = np.clip((base.astype(np.int32) - 100) * 6, 0, 255).astype(np.uint8)
    return out


def depth_images(synth):
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:H, 0:W]
    base = synth.make_depth(0x5EED1000).astype(np.int32)
    out = {}
    out["steps10"] = (8000 + (xx // 10) * 37 + (yy // 10) * 11).astype(np.uint16)                 # every block a different fronto-parallel patch
    out["stripes_missing"] = np.where((xx // 7) % 5 == 0, 0, base).astype(np.uint16)              # no block without a hole
    out["coarse_holes"] = np.where(((xx // 40 + yy // 40) % 3) == 0, 0, base).astype(np.uint16)
    out["heavy_noise"] = np.clip(base + rng.integers(-400, 401, (H, W)) * (base > 0), 1, 65535).astype(np.uint16)
    out["far_plane"] = np.full((H, W), 65535, np.uint16)
    out["two_depths_checker"] = np.where(((xx // 60 + yy // 60) & 1) == 0, 9000, 9600).astype(np.uint16)
    out["ramp_x"] = (5000 + xx * 20).astype(np.uint16)
    return out


def test_stress_gray(gpu_ctx, orc, synth):
    o = orc.Orb()
    bad = []
    for name, g in gray_images(synth).items():
        kp_g, d_g = gpu_ctx.extract_orb(g); kp_o, d_o = o.extract(g)
        ok = len(kp_g) == len(kp_o) and np.array_equal(d_g, d_o) and all(np.array_equal(kp_g[f], kp_o[f]) for f in ("x", "y", "octave")) \
            and np.allclose(kp_g["angle"], kp_o["angle"], rtol=0, atol=1e-4) and np.array_equal(kp_g["response"], kp_o["response"])
        if not ok: bad.append(name + ":orb")
        kl_o, dl_o, fn_o = orc.line_extract(g)
        kl_g, dl_g, fn_g = gpu_ctx.extract_lsd(g)
        ok = len(kl_g) == len(kl_o) and np.array_equal(dl_g, dl_o) and np.array_equal(kl_g["num_pixels"], kl_o["num_pixels"]) \
            and all(np.allclose(kl_g[f], kl_o[f], rtol=0, atol=1e-4) for f in ("sx", "sy", "ex", "ey", "angle", "length", "response"))
        if not ok: bad.append(name + ":lines")
    assert not bad, bad


def test_stress_depth(gpu_ctx, orc, synth):
    bad = []
    for name, d in depth_images(synth).items():
        lo, po = orc.peac(d)
        lg, pg = gpu_ctx.compute_planes(d)
        ok = np.array_equal(lg, lo) and len(pg) == len(po) and (len(po) == 0 or (np.array_equal(pg["n_points"], po["n_points"])
             and np.allclose(pg["normal"], po["normal"], rtol=1e-9, atol=1e-12) and np.allclose(pg["mse"], po["mse"], rtol=1e-9, atol=1e-12)))
        if not ok: bad.append((name, len(pg), len(po)))
    assert not bad, bad


@pytest.mark.parametrize("env", [{"HVO_SCHED": "5"}, {"HVO_SCHED": "2", "HVO_PRIO": "0,0,0"}, {"HVO_SCHED": "0", "HVO_FRAME_PERM": "0", "HVO_PEAC_PERM": "0", "HVO_PEAC_EDGES": "0", "HVO_LSD_LAT_LDS": "0"},
                                 {"HVO_SCHED": "5", "HVO_PEAC_GL": "16", "HVO_FLOOD_T": "64", "HVO_LSD_DENSE": "1", "HVO_ORB_BLUR_LATE": "1"}])
def test_overlap_policy_and_launch_order_do_not_change_results(hvo, orc, synth, monkeypatch, env):
    """which stream waits for which kernel (HVO_SCHED), the stream priorities and the order in which the serial kernels take their
    frames (hvo_frame_perm; 11 frames: not a power of two) are scheduling only: every configuration returns the oracle's results"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 11
    gray = np.stack([synth.make_gray("std" if i % 3 else "lowtex", 0x5EED3000 + i) for i in range(n)])
    depth = np.stack([synth.make_depth(0x5EED3000 + i) for i in range(n)])
    ctx = hvo.Context(max_batch=n)
    try:
        ctx.batch_upload(gray, depth)
        ctx.batch_run(hvo.STAGE_ALL)
        res = ctx.batch_download(hvo.STAGE_ALL)
    finally:
        ctx.close()
    o = orc.Orb()
    for b in (0, 4, 7, 10):
        assert res[b]["status"] == 0
        kp_o, dd_o = o.extract(gray[b])
        assert len(res[b]["kp"]) == len(kp_o) and np.array_equal(res[b]["desc"], dd_o)
        kl_o, d_o, fn_o = orc.line_extract(gray[b])
        assert len(res[b]["kl"]) == len(kl_o) and np.array_equal(res[b]["ldesc"], d_o)
        lo, po = orc.peac(depth[b])
        assert np.array_equal(res[b]["labels"], lo) and len(res[b]["planes"]) == len(po)
