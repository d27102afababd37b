"""Manhattan::computeNormalsLPVO (reference src/Manhattan.cpp:237-393; SURVEY.md 8f.4, second half), the CV_32F reading: oracle known answers
on CPU, HIP path against the oracle bit for bit on the GPU (float / double arithmetic in the same order on both sides)."""
import numpy as np
import pytest


def plane_depth(a, b, c, d0, w=640, h=480, fx=535.4, fy=539.2, cx=320.1, cy=247.6):
    """depth image (1/5000 m units) of the plane a*x + b*y + c*z = d0 in camera coordinates"""
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    z = d0 / (a * (u - cx) / fx + b * (v - cy) / fy + c)
    return np.clip(np.round(z * 5000.0), 0, 65535).astype(np.uint16)


def test_lpvo_known_answers(orc):
    nrm0 = np.array([0.1, 0.2, 1.0]) / np.linalg.norm([0.1, 0.2, 1.0])
    d = plane_depth(0.1, 0.2, 1.0, 3.0)
    n, z, px = orc.normals_lpvo(d)
    # the sampling grid: v = 10, 25, ... < h - 1; u = 10, 25, ... < w - 1 (src/Manhattan.cpp:335-337); a full plane is valid everywhere
    assert len(n) == len(range(10, 479, 15)) * len(range(10, 639, 15))
    assert px[0].tolist() == [10, 10] and px[1].tolist() == [25, 10] and px[-1].tolist() == [625, 475]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
    assert np.abs(np.abs(n @ nrm0) - 1.0).max() < 2e-3                     # vVector x uVector: the plane's normal (up to depth quantisation)
    assert (n @ nrm0 < 0).all()                                            # ... pointing towards the camera
    assert np.allclose(z, d[px[:, 1], px[:, 0]] * np.float32(1.0 / 5000.0), atol=1e-6)
    # a hole: samples whose pixel or one of its four neighbours lacks depth are skipped (tangeMask), 10 x 10 boxes that touch the hole
    # average over fewer points (numPts)
    d2 = d.copy(); d2[100:140, 200:260] = 0
    n2, z2, px2 = orc.normals_lpvo(d2)
    inside = (px[:, 0] >= 199) & (px[:, 0] <= 260) & (px[:, 1] >= 99) & (px[:, 1] <= 140)
    assert len(n2) == len(n) - int(inside.sum())
    # out of range depth (> 7 m) is no depth
    d3 = d.copy(); d3[:, 320:] = 40000
    assert (orc.normals_lpvo(d3)[2][:, 0] < 320).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0x5EED0002, 0x5EED1003, 0x5EED2001])
def test_lpvo_parity(gpu_ctx, orc, synth, seed):
    d = synth.make_depth(seed)
    no, zo, po = orc.normals_lpvo(d)
    ng, zg, pg = gpu_ctx.normals_lpvo(d)
    assert len(no) > 500 and np.array_equal(pg, po)
    assert ng.tobytes() == no.tobytes() and zg.tobytes() == zo.tobytes()


@pytest.mark.gpu
def test_lpvo_edge_cases(gpu_ctx, orc, hvo):
    d = np.zeros((480, 640), np.uint16)
    assert len(gpu_ctx.normals_lpvo(d)[0]) == 0                            # no depth: no normals
    d = plane_depth(0.0, 0.0, 1.0, 1.5, w=322, h=247)                      # odd geometry, fronto-parallel plane
    no, zo, po = orc.normals_lpvo(d, cx=161.0, cy=123.0)
    ctx = hvo.Context(cx=161.0, cy=123.0)
    try:
        ng, zg, pg = ctx.normals_lpvo(d)
    finally:
        ctx.close()
    assert np.array_equal(pg, po) and ng.tobytes() == no.tobytes() and zg.tobytes() == zo.tobytes()
