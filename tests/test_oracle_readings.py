"""The two "(?)" readings of SURVEY.md Appendix A as oracle switches (oracle.h: ORC_READING_*).

The defaults are what the golden vectors and the HIP path implement; a box with the author's OpenCV 3.2 settles each
reading by flipping its flag and comparing.  These tests pin what flipping does (and that the default does not move).
CPU only."""
import numpy as np
import pytest


@pytest.fixture(autouse=True)
def _readings_off(orc):
    yield
    for name in orc.READINGS:
        orc.set_reading(name, 0)


def test_defaults_are_off(orc):
    assert not any(orc.get_reading(n) for n in orc.READINGS)


@pytest.mark.parametrize("ksize,sigma", [(7, 2.0), (5, 1.0)])
def test_blur_float_reading(orc, synth, ksize, sigma):
    """GaussianBlur served by IPP (float kernel, one rounding) against OpenCV's own CV_8U fixed-point path: the q8 kernels
    ([18 34 49 55 49 34 18], [14 63 103 63 14]) sum to 257, a gain of (257/256)^2, so the two differ by at most 2 grey
    levels, and never on black"""
    g = synth.make_gray("std", 0x5EED0002)
    fixed = orc.gaussian_blur(g, ksize, sigma)
    orc.set_reading("blur_float", 1)
    try:
        assert orc.get_reading("blur_float")
        flt = orc.gaussian_blur(g, ksize, sigma)
        z = orc.gaussian_blur(np.zeros((32, 48), np.uint8), ksize, sigma)
        c = orc.gaussian_blur(np.full((32, 48), 100, np.uint8), ksize, sigma)
    finally:
        orc.set_reading("blur_float", 0)
    d = fixed.astype(np.int32) - flt.astype(np.int32)
    assert np.abs(d).max() <= 2 and (d != 0).any()
    assert (d >= 0).mean() > 0.99                     # the fixed-point path is the brighter one (kernel sum 257)
    assert not z.any() and (c == 100).all()           # the float kernel is normalised
    assert np.array_equal(orc.gaussian_blur(g, ksize, sigma), fixed)       # switched off again: the default did not move


def test_blur_float_reading_reaches_orb_and_lbd(orc, synth):
    """the switch sits in the one blur both ORB (7x7) and LBD (5x5) call: descriptors change, key points do not (FAST and
    the orientation read the unblurred levels)"""
    g = synth.make_gray("std", 0x5EED0002)
    o = orc.Orb()
    k0, d0 = o.extract(g)
    orc.set_reading("blur_float", 1)
    try:
        k1, d1 = o.extract(g)
    finally:
        orc.set_reading("blur_float", 0)
    assert np.array_equal(k0, k1)
    ham = np.unpackbits(d0 ^ d1, axis=1).sum(1)
    assert ham.max() > 0 and np.median(ham) < 40      # a few of 256 tests flip per descriptor


def test_lsd_8u_reading(orc, synth):
    """LineSegmentDetector on CV_8U (blur and 0.8x resize rounded to bytes) against the CV_64F pipeline: the same scene
    yields mostly the same segments (mid points within 2 px for > 75 % of them), not the same floats"""
    g = synth.make_gray("std", 0x5EED0002)
    a = np.asarray(orc.lsd_detect(g))
    orc.set_reading("lsd_8u", 1)
    try:
        b = np.asarray(orc.lsd_detect(g))
    finally:
        orc.set_reading("lsd_8u", 0)
    assert len(a) > 50 and abs(len(a) - len(b)) < 0.2 * len(a)
    assert not (len(a) == len(b) and np.array_equal(a, b))
    mid = lambda s: np.stack([(s[:, 0] + s[:, 2]) / 2, (s[:, 1] + s[:, 3]) / 2], 1)
    d = np.sqrt(((mid(a)[:, None, :] - mid(b)[None, :, :]) ** 2).sum(-1)).min(1)
    assert (d < 2).mean() > 0.75
    assert np.array_equal(np.asarray(orc.lsd_detect(g)), a)
