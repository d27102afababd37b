"""Known-answer tests that pin the oracle to constants derivable from the reference text alone
(SURVEY.md section 8c).  The reference ships no tests/fixtures for this path, so these are the
only external pins -- the oracle header says "parity unpinned"."""
import numpy as np
import pytest


def test_umax_table(orc):
    # ORBextractor.cc:452-467
    assert orc.Orb().umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_features_per_level(orc):
    # ORBextractor.cc:432-444 with TUM3.yaml:41-54
    assert orc.Orb(1000, 1.2, 8).features_per_level().tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert orc.Orb(2000, 1.2, 8).features_per_level().tolist() == [434, 362, 302, 251, 209, 175, 145, 122]


def test_pattern_rows(orc):
    # first / last rows of bit_pattern_31_ (ORBextractor.cc:150,405)
    p = orc.pattern()
    assert p[:4].tolist() == [8, -3, 9, 5] and p[-4:].tolist() == [-1, -6, 0, -11]
    assert np.abs(p).max() <= 13 and len(p) == 1024


def test_pyramid_sizes_and_grid(orc, synth):
    o = orc.Orb()
    o.extract(synth.make_gray("lowtex", 1, 640, 480))
    sizes = [o.level(l).shape[::-1] for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert sum(w * h for w, h in sizes) == 950532
    grids = [o.grid(l)[:2] for l in range(8)]
    assert grids == [(20, 14), (16, 12), (13, 10), (11, 8), (9, 6), (7, 5), (6, 4), (4, 3)]
    assert sum(c * r for c, r in grids) == 815


def test_gaussian_taps(orc):
    # cv::GaussianBlur fixed-point taps (pre-3.4 CV_8U path): each sums to 257
    assert orc.gaussian_kernel_q8(7, 2.0).tolist() == [18, 34, 49, 55, 49, 34, 18]
    assert orc.gaussian_kernel_q8(5, 1.0).tolist() == [14, 63, 103, 63, 14]


def test_hamming_known(orc):
    z = np.zeros((1, 32), np.uint8); o = np.full((1, 32), 255, np.uint8)
    assert orc.hamming_matrix(z, o)[0, 0] == 256
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (50, 32), dtype=np.uint8); b = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    ref = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(orc.hamming_matrix(a, b), ref)      # SWAR trick == popcount
    idx, dist = orc.hamming_knn2(a, b)
    order = np.lexsort((np.arange(60)[None, :].repeat(50, 0), ref), axis=1)   # (dist, idx) ascending
    assert np.array_equal(idx, order[:, :2])
    assert np.array_equal(dist, np.take_along_axis(ref, order[:, :2], 1))


def test_fast_atan2(orc):
    # degrees in [0,360), polynomial error < 0.3 deg
    L = orc.lib()
    for y, x in [(0.0, 1.0), (1.0, 0.0), (1.0, 1.0), (-1.0, 1.0), (-1.0, -1.0), (3.0, -7.0), (0.0, 0.0)]:
        a = L.orc_fast_atan2(y, x)
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        assert 0.0 <= a < 360.0 and abs(a - ref) < 0.31


def test_cvround_half_even(orc):
    L = orc.lib()
    assert [L.orc_cvround_f(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5)] == [0, 2, 2, 0, -2]


def test_fast_on_synthetic_corner(orc):
    # a bright square corner on a dark background is a FAST-9 corner; flat image has none
    img = np.full((40, 40), 50, np.uint8)
    assert len(orc.fast9_16(img, 20)) == 0
    yy, xx = np.mgrid[0:20, 0:20]
    img[20:, 20:] = (200 - 2 * xx - 2 * yy).astype(np.uint8)   # strict NMS needs a unique maximum
    k = orc.fast9_16(img, 20)
    assert len(k) >= 1 and all(3 <= x < 37 and 3 <= y < 37 for x, y, _ in k)
    # score == largest threshold at which the point is still a corner (cornerScore semantics)
    x, y, s = k[0]
    assert len([1 for xx, yy, _ in orc.fast9_16(img, int(s)) if (xx, yy) == (x, y)]) == 1
    assert len([1 for xx, yy, _ in orc.fast9_16(img, int(s) + 1) if (xx, yy) == (x, y)]) == 0


def test_resize_identity_and_constant(orc):
    img = np.full((48, 64), 77, np.uint8)
    assert np.all(orc.resize_linear(img, 53, 40) == 77)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    assert np.array_equal(orc.resize_linear(img, 64, 48), img)


def test_blur_constant_saturation(orc):
    # taps sum to 257: a constant image c maps to round(c*257*257/65536); 255 saturates
    for c in (0, 1, 100, 200, 255):
        img = np.full((20, 24), c, np.uint8)
        out = orc.gaussian_blur(img, 7, 2.0)
        exp = min(255, int(np.floor(c * 257 * 257 / 65536 + 0.5)))
        assert np.all(out == exp), (c, out[0, 0], exp)


def test_orb_extract_runs_and_is_deterministic(orc, synth):
    g = synth.make_gray("std", 0x5EED0002)
    o = orc.Orb()
    kp1, d1 = o.extract(g)
    kp2, d2 = orc.Orb().extract(g)
    assert len(kp1) > 500 and np.array_equal(kp1, kp2) and np.array_equal(d1, d2)
    assert np.all(kp1["octave"][:-1] <= kp1["octave"][1:])     # level-major order
    assert np.all((kp1["angle"] >= 0) & (kp1["angle"] < 360))
    assert np.all(kp1["class_id"] == -1)
    # per-level quota: DistributeOctTree returns at most N+3 nodes
    nf = o.features_per_level()
    for l in range(8):
        assert o.level_count(l) <= nf[l] + 3


def test_orb_empty_image(orc):
    kp, d = orc.Orb().extract(np.full((480, 640), 128, np.uint8))
    assert len(kp) == 0


# ---------------- PEAC known answers ----------------
def _peac_fn(orc, name):
    import ctypes
    f = getattr(orc.lib(), name); f.restype = ctypes.c_double; f.argtypes = [ctypes.c_double]
    return f


def test_peac_thresholds(orc):
    # AHCParamSet.hpp:68-76,88-146 with the metres-vs-mm quirk (H4)
    assert abs(_peac_fn(orc, "orc_peac_T_mse_init")(3.0) - (1.6e-6 * 9 + 5) ** 2) < 1e-12
    assert abs(_peac_fn(orc, "orc_peac_T_ang_init")(3.0) - np.cos(np.radians(15.0))) < 1e-15
    assert abs(_peac_fn(orc, "orc_peac_T_ang_init")(499.0) - np.cos(np.radians(15.0))) < 1e-15
    assert abs(_peac_fn(orc, "orc_peac_T_dz")(2.0) - 0.1) < 1e-15


def test_peac_exact_plane_oracle(orc):
    h, w = 480, 640
    j = np.arange(w)[None, :]; i = np.arange(h)[:, None]
    z = 2.0 / (0.1 * (j - 320.1) / 535.4 + 0.2 * (i - 247.6) / 539.2 + 1.0)
    lab, pl = orc.peac(np.rint(z * 5000).astype(np.uint16))
    assert len(pl) == 1 and pl["n_points"][0] == w * h and pl["mse"][0] < 1e-6
    assert np.dot(pl["normal"][0], pl["center"][0]) <= 0 and np.all(lab == 0)


def test_eig33sym_against_numpy(orc):
    rng = np.random.default_rng(3)
    for _ in range(50):
        a = rng.normal(size=(3, 3)); K = a @ a.T * rng.uniform(1e-3, 1e3)
        s, V = orc.eig33sym(K)
        assert np.allclose(s, np.linalg.eigvalsh(K), rtol=1e-10, atol=1e-12)
        assert np.allclose(K @ V, V * s, atol=1e-9 * np.abs(K).max())


def test_eig33_smallest_against_jacobi_and_numpy(orc):
    """Stats::compute's solver (Laguerre + adjugate column) on covariances of noisy planar patches, from nearly exact planes
    to blobs: the eigenvalue is within 4 ulp of trace(K) of the Jacobi / LAPACK value (the backward-stable bound the
    reference's Eigen solver has) and the eigenvector satisfies K v = lambda v to the same tolerance"""
    rng = np.random.default_rng(5)
    worst = 0.0
    for t in range(400):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        ext = rng.uniform(0.02, 2.0); sig = ext * 10.0 ** rng.uniform(-6, -0.3)
        P = rng.uniform(-ext, ext, size=(int(rng.integers(100, 3000)), 3))
        P -= np.outer(P @ n, n); P += np.outer(rng.normal(scale=sig, size=len(P)), n); P += rng.uniform(-3, 3, size=3)
        s = P.sum(0); K = P.T @ P - np.outer(s, s) / len(P)
        l0, v = orc.eig33_smallest(K)
        sj, Vj = orc.eig33sym(K)
        tr = np.trace(K)
        assert abs(l0 - sj[0]) <= 9e-16 * tr and abs(l0 - np.linalg.eigvalsh(K)[0]) <= 2e-15 * tr
        assert abs(np.linalg.norm(v) - 1) < 1e-14
        assert np.linalg.norm(K @ v - l0 * v) <= 1e-14 * tr
        gap = sj[1] - sj[0]
        ang = np.linalg.norm(np.cross(v, Vj[:, 0]))
        assert ang <= 1e-13 * tr / max(gap, 1e-300) + 1e-12, (ang, gap / tr)
        worst = max(worst, abs(l0 - sj[0]) / tr)
    # exact plane: lambda0 = 0 up to rounding of K, and a multiple of the identity returns a unit vector
    l0, v = orc.eig33_smallest(np.diag([2.0, 3.0, 0.0]))
    assert l0 == 0.0 and np.allclose(np.abs(v), [0, 0, 1])
    l0, v = orc.eig33_smallest(np.eye(3) * 0.5)
    assert abs(l0 - 0.5) < 1e-15 and abs(np.linalg.norm(v) - 1) < 1e-15


def test_peac_synthetic_scene_finds_the_walls(orc, synth):
    lab, pl = orc.peac(synth.make_depth(0x5EED0002))
    assert len(pl) == 4 and np.all(np.diff(pl["n_points"]) <= 0)
    assert set(np.unique(lab)) == {-1, 0, 1, 2, 3}


# ---------------- LSD / LBD known answers ----------------
def test_lbd_combinations_and_weights(orc):
    comb = orc.lbd_combinations()
    # binary_descriptor_custom.cpp:74-107: the 32 band pairs, lexicographic except the far pairs 0-7,0-8,1-7,1-8
    assert comb[0].tolist() == [0, 1] and comb[5].tolist() == [0, 6] and comb[6].tolist() == [1, 2]
    assert comb[31].tolist() == [7, 8] and len({tuple(c) for c in comb}) == 32
    assert not any(tuple(c) in {(0, 7), (0, 8), (1, 7), (1, 8)} for c in comb.tolist())
    gL, gG = orc.lbd_weights()
    assert np.allclose(gL, np.exp(-(np.arange(21) - 10) ** 2 / 98.0), rtol=1e-15)       # sigma_l = 7 (int div)
    assert np.allclose(gG, np.exp(-(np.arange(63) - 31) ** 2 / 1922.0), rtol=1e-15)     # sigma_g = 31


def test_lsd_finds_a_synthetic_edge(orc):
    g = np.full((480, 640), 60, np.uint8)
    g[:, 300:] = 180                                   # one vertical step edge
    segs = orc.lsd_detect(g)
    assert len(segs) >= 1
    long = segs[np.argmax(np.hypot(segs[:, 2] - segs[:, 0], segs[:, 3] - segs[:, 1]))]
    assert abs(long[0] - 300) < 1.5 and abs(long[2] - 300) < 1.5 and abs(long[3] - long[1]) > 400


def test_line_extract_top_n_and_functions(orc, synth):
    g = synth.make_gray("std", 0x5EED0002)
    kl, desc, fn = orc.line_extract(g, nfeatures=200)
    assert len(kl) == 200 and np.array_equal(kl["class_id"], np.arange(200))
    assert np.all(np.diff(kl["response"]) <= 0)                    # sorted by response, descending
    assert np.allclose(kl["response"], kl["length"] / 640.0, rtol=1e-6)
    assert np.allclose(np.hypot(fn[:, 0], fn[:, 1]), 1.0, atol=1e-12)
    # both end points lie on the line function
    assert np.allclose(fn[:, 0] * kl["sx"] + fn[:, 1] * kl["sy"] + fn[:, 2], 0, atol=1e-6)
    assert np.allclose(fn[:, 0] * kl["ex"] + fn[:, 1] * kl["ey"] + fn[:, 2], 0, atol=1e-6)
    d2 = orc.lbd_compute(g, kl)
    assert np.array_equal(d2, desc)


def test_lbd_float_descriptor_is_unit_and_clipped(orc, synth):
    g = synth.make_gray("std", 0x5EED0002)
    kl, _, _ = orc.line_extract(g)
    _, f = orc.lbd_compute(g, kl, want_float=True)
    ok = ~np.isnan(f).any(axis=1)
    assert ok.mean() > 0.9
    assert np.allclose(np.linalg.norm(f[ok], axis=1), 1.0, atol=1e-5)


def test_division_by_a_kernel_constant_is_the_ieee_division(tmp_path):
    """csrc/hvo_internal.hpp hvo_div_const (five instructions for x = (j - cx) z / fx in k_peac_blocks and the flood) must give the bits of `/`:
    tools/microbench/div_const_check.c runs the same sequence with the host's fma on 15 divisors x 2e6 operands (9e8 in the committed run)"""
    import os, shutil, subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "div_const_check")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, os.path.join(root, "tools", "microbench", "div_const_check.c"), "-lm"])
    p = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "two corrections 0" in p.stdout, p.stdout
