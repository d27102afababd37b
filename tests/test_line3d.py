"""Frame::isLineGood (reference src/Frame.cc:1205-1322): 3-D lines from key lines and depth.
CPU: known answers of the oracle on a synthetic 3-D edge.  GPU (-m gpu): hvo_lines_3d vs the oracle, through the C ABI."""
import numpy as np
import pytest

FX, FY, CX, CY = 535.4, 539.2, 320.1, 247.6


def plane_depth(nx, ny, nz, d, w=640, h=480):
    j = np.arange(w)[None, :]; i = np.arange(h)[:, None]
    z = d / (nx * (j - CX) / FX + ny * (i - CY) / FY + nz)
    return np.rint(z * 5000).astype(np.uint16)


def keyline(orc, sx, sy, ex, ey):
    k = np.zeros(1, orc.KEYLINE_DT)
    k["sx"], k["sy"], k["ex"], k["ey"] = sx, sy, ex, ey
    return k


def test_oracle_known_answer_on_a_plane(orc):
    """a segment lying on a slanted plane: all 21 samples are inliers, A / B are the back-projections of the two extreme
    samples, line_eq is their unit difference (float), line_nor = A x B, and the 3-D points lie on the plane"""
    d = plane_depth(0.2, -0.1, 1.0, 2.0)
    kl = keyline(orc, 100.3, 120.7, 420.9, 300.2)
    o = orc.lines_3d(kl, d, seed=3)[0]
    assert o["good"] == 1 and o["n_samples"] == 21 and o["n_inliers"] == 21 and o["inlier_mask"] == (1 << 21) - 1
    for P in (o["A"], o["B"]):
        assert abs(0.2 * P[0] - 0.1 * P[1] + 1.0 * P[2] - 2.0) < 2e-3
    # the extreme samples are the end points of the segment (nearest pixel, truncated)
    ends = []
    for (x, y) in ((100.3, 120.7), (420.9, 300.2)):
        col, row = int(np.float32(x)), int(np.float32(y))
        z = np.float32(d[row, col]) * (np.float32(1.0) / np.float32(5000.0))
        f64 = np.float64
        ends.append(np.array([f64(np.float32(col) - np.float32(CX)) * f64(z) * f64(np.float32(1) / np.float32(FX)),
                              f64(np.float32(row) - np.float32(CY)) * f64(z) * f64(np.float32(1) / np.float32(FY)), f64(z)]))
    got = sorted([tuple(o["A"]), tuple(o["B"])]); want = sorted([tuple(e) for e in ends])
    assert np.allclose(got, want, rtol=0, atol=1e-12)
    u = (o["B"] - o["A"]) / np.linalg.norm(o["B"] - o["A"])
    assert np.allclose(o["line_eq"], u, atol=1e-6) and np.allclose(o["line_nor"], np.cross(o["A"], o["B"]), rtol=1e-12)


def test_oracle_rejects_lines_without_depth_and_short_ones(orc):
    d = plane_depth(0.0, 0.0, 1.0, 2.0)
    d[:, :200] = 0
    o = orc.lines_3d(keyline(orc, 10, 50, 150, 300), d, seed=1)[0]                 # entirely in the hole
    assert o["good"] == 0 and o["n_samples"] == 0 and tuple(o["line_eq"]) == (-1, -1, -1) and tuple(o["line_nor"]) == (-1, -1, -1)
    o = orc.lines_3d(keyline(orc, 300, 200, 302.5, 201), d, seed=1)[0]             # (int)len = 2 -> 3 samples < 5
    assert o["good"] == 0 and o["n_samples"] == 3
    o = orc.lines_3d(keyline(orc, 500, 100, 630, 400), d, seed=1)[0]               # samples at columns >= rows (480) are dropped (sic)
    assert o["n_samples"] == 0
    o = orc.lines_3d(keyline(orc, 300.5, 100.5, 300.5, 100.5), d, seed=1)[0]       # zero length
    assert o["good"] == 0 and o["n_samples"] == 0


def test_oracle_depth_edge_selects_the_dominant_side(orc):
    """a segment that crosses a depth step: RANSAC keeps the larger collinear subset; different seeds may pick another pair
    but the result stays a valid line; the same seed reproduces the same bytes"""
    d = plane_depth(0.0, 0.0, 1.0, 2.0)
    d[:, 400:] = plane_depth(0.0, 0.0, 1.0, 3.5)[:, 400:]
    kl = keyline(orc, 100.2, 200.4, 470.8, 230.1)
    a = orc.lines_3d(kl, d, seed=11)[0]; b = orc.lines_3d(kl, d, seed=11)[0]
    assert a.tobytes() == b.tobytes()
    assert a["good"] == 1 and 13 <= a["n_inliers"] <= 21
    assert abs(a["A"][2] - 2.0) < 1e-3 and abs(a["B"][2] - 2.0) < 1e-3


@pytest.mark.gpu
def test_lines_3d_parity(gpu_ctx, orc, synth):
    """every field of every line, bit for bit (the float line_eq included), on extracted lines of synthetic frames, several seeds"""
    total = 0
    for s, kind in ((0x5EED0002, "std"), (0x5EED1003, "std"), (0x5EED0001, "lowtex")):
        g, d = synth.make_frame(kind, s)
        kl, _, _ = orc.line_extract(g)
        for seed in (1, 0xC0FFEE):
            o = orc.lines_3d(kl, d, seed=seed)
            r = gpu_ctx.lines_3d(kl, d, seed=seed)
            assert r.tobytes() == o.tobytes(), [f for f in o.dtype.names if not np.array_equal(r[f], o[f])]
            total += int(o["good"].sum())
    assert total > 300


@pytest.mark.gpu
def test_lines_3d_edge_cases(gpu_ctx, orc):
    d = plane_depth(0.2, -0.1, 1.0, 2.0)
    d[100:140, :] = 0
    d[:, 400:] = plane_depth(0.0, 0.1, 1.0, 3.1)[:, 400:]
    rng = np.random.default_rng(9)
    kl = np.zeros(300, orc.KEYLINE_DT)
    kl["sx"] = rng.uniform(-20, 660, 300); kl["sy"] = rng.uniform(-20, 500, 300)
    kl["ex"] = kl["sx"] + rng.uniform(-200, 200, 300); kl["ey"] = kl["sy"] + rng.uniform(-200, 200, 300)
    kl["sx"][:10] = np.floor(kl["sx"][:10]); kl["sy"][:10] = np.floor(kl["sy"][:10])          # integer coordinates: the "boundary issue" branch
    kl["ex"][5:15] = kl["sx"][5:15]; kl["ey"][5:15] = kl["sy"][5:15] + 0.5                     # (int)len == 0
    o = orc.lines_3d(kl, d, seed=5); r = gpu_ctx.lines_3d(kl, d, seed=5)
    assert r.tobytes() == o.tobytes(), [f for f in o.dtype.names if not np.array_equal(r[f], o[f])]
    assert 50 < o["good"].sum() < 300
