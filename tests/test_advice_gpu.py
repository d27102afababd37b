"""Regression tests for the round-1 code-review findings: plan cache keys after a rejected geometry,
T_ang(P_INIT) with a depth_map_factor that puts z beyond z_near, batch downloads of stages that did not run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rejected_geometry_is_rejected_again(hvo):
    """a 100x100 image has pyramid levels below 38 pixels: HVO_ERR_UNSUPPORTED on the first call AND on the second
    (the plan key must not survive its half-built plan); the context stays usable for a supported geometry"""
    ctx = hvo.Context()
    try:
        g = np.full((100, 100), 90, np.uint8)
        for _ in range(2):
            with pytest.raises(hvo.HvoError) as e:
                ctx.extract_orb(g)
            assert e.value.status == -4
        d = np.full((16, 16), 5000, np.uint16)
        for _ in range(2):
            with pytest.raises(hvo.HvoError) as e:
                ctx.compute_planes(d)
            assert e.value.status == -4
        kp, desc = ctx.extract_orb(np.full((480, 640), 90, np.uint8))
        assert len(kp) == 0
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [0x5EED0002, 0x5EED1003])
def test_peac_parity_depth_in_millimetres(hvo, orc, synth, seed):
    """DepthMapFactor 1 (depth_map_factor = 1.0): z is in millimetres, beyond z_near = 500, so T_ang(P_INIT, z) and T_mse
    are evaluated per segment (AHCParamSet.hpp:88-121) instead of collapsing to cos 15 deg"""
    d = (synth.make_depth(seed).astype(np.float64) / 5.0).round().astype(np.uint16)       # 1/5000 m -> mm
    ctx = hvo.Context(depth_map_factor=1.0)
    try:
        lg, pg = ctx.compute_planes(d)
    finally:
        ctx.close()
    lo, po = orc.peac(d, depth_factor=1.0)
    assert len(po) >= 1
    assert len(pg) == len(po) and np.array_equal(pg["n_points"], po["n_points"])
    assert np.array_equal(lg, lo)
    assert np.allclose(pg["normal"], po["normal"], rtol=1e-9, atol=1e-12)


def test_batch_download_reports_only_stages_that_ran(hvo, synth):
    g, d = synth.make_batch("std", 0x5EED1000, 2)
    ctx = hvo.Context(max_batch=2)
    try:
        ctx.batch_upload(g, d)
        res = ctx.batch_download(hvo.STAGE_ALL)                       # nothing has run yet
        assert all(len(r["kp"]) == 0 and len(r["kl"]) == 0 and len(r["planes"]) == 0 and r["status"] == 0 for r in res)
        ctx.batch_run(hvo.STAGE_PLANES)
        res = ctx.batch_download(hvo.STAGE_ALL)
        assert all(len(r["kp"]) == 0 and len(r["kl"]) == 0 and len(r["planes"]) >= 3 and r["status"] == 0 for r in res)
        ctx.batch_run(hvo.STAGE_ORB)
        res = ctx.batch_download(hvo.STAGE_ALL)
        assert all(len(r["kp"]) > 500 and len(r["kl"]) == 0 and len(r["planes"]) >= 3 for r in res)
    finally:
        ctx.close()
