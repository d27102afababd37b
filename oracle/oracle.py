"""ctypes loader for the CPU oracle (liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  Parity unpinned (see oracle.h).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KEYPOINT_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                        ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
KEYLINE_DT = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"),
                       ("pt_x", "<f4"), ("pt_y", "<f4"), ("response", "<f4"), ("size", "<f4"),
                       ("sx", "<f4"), ("sy", "<f4"), ("ex", "<f4"), ("ey", "<f4"),
                       ("sox", "<f4"), ("soy", "<f4"), ("eox", "<f4"), ("eoy", "<f4"),
                       ("length", "<f4"), ("num_pixels", "<i4")])
PLANE_DT = np.dtype([("normal", "<f8", 3), ("center", "<f8", 3), ("mse", "<f8"),
                     ("n_points", "<i4"), ("rid", "<i4")])
assert KEYPOINT_DT.itemsize == 28 and KEYLINE_DT.itemsize == 68 and PLANE_DT.itemsize == 64


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        L = _LIB
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cvround_f.argtypes = [C.c_float]
        L.orc_cvround_d.argtypes = [C.c_double]
        L.orc_orb_create.restype = C.c_void_p
        L.orc_orb_create.argtypes = [C.POINTER(OrbParams)]
        L.orc_orb_destroy.argtypes = [C.c_void_p]
        L.orc_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_orb_umax.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_orb_features_per_level.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_orb_scale.restype = C.c_float
        L.orc_orb_scale.argtypes = [C.c_void_p, C.c_int]
        L.orc_orb_pattern.restype = C.POINTER(C.c_int8)
        L.orc_orb_level.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.orc_orb_level_blurred.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.orc_orb_grid.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 4
        L.orc_orb_candidates.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_orb_level_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hamming_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_hamming_matrix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_match_nnr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gaussian_kernel_q8.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.orc_gaussian_blur_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.orc_sobel3_u8_s16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_score.argtypes = [C.c_void_p, C.c_int]
        L.orc_line_iterator_count.argtypes = [C.c_int, C.c_int] + [C.c_float] * 4
        _bind_optional(L)
    return _LIB


def _bind_optional(L):
    """bindings for the LSD/LBD/PEAC oracles (present once those files are built)"""
    if hasattr(L, "orc_peac_run"):
        L.orc_peac_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_int, C.POINTER(C.c_int)]
    if hasattr(L, "orc_lsd_detect"):
        L.orc_lsd_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                     C.POINTER(C.c_int)]
    if hasattr(L, "orc_line_extract"):
        L.orc_line_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    if hasattr(L, "orc_lbd_compute"):
        L.orc_lbd_compute.restype = None
        L.orc_lbd_compute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p]
        L.orc_lbd_weights.restype = None
        L.orc_lbd_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_lbd_combinations.restype = C.POINTER(C.c_int)


READINGS = {"blur_float": 0, "lsd_8u": 1}


def set_reading(name, on):
    """alternative reading of an un-vendored dependency (oracle.h: ORC_READING_*); process-global, default off"""
    lib().orc_set_reading(READINGS[name], int(bool(on)))


def get_reading(name):
    return bool(lib().orc_get_reading(READINGS[name]))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Orb:
    """ORBextractor oracle (src/ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.p = OrbParams(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.h = lib().orc_orb_create(C.byref(self.p))
        self.nlevels = nlevels
        self.cap = max(nfeatures * 2, 64)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_orb_destroy(self.h)
            self.h = None

    def extract(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        kp = np.zeros(self.cap, dtype=KEYPOINT_DT)
        desc = np.zeros((self.cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        rc = lib().orc_orb_extract(self.h, _p(gray), w, h, gray.strides[0], _p(kp), _p(desc), self.cap, C.byref(n))
        assert rc == 0
        return kp[: n.value].copy(), desc[: n.value].copy()

    def umax(self):
        a = np.zeros(16, dtype=np.int32); lib().orc_orb_umax(self.h, _p(a)); return a

    def features_per_level(self):
        a = np.zeros(self.nlevels, dtype=np.int32); lib().orc_orb_features_per_level(self.h, _p(a)); return a

    def level(self, l, blurred=False):
        w, h, s, d = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
        lib().orc_orb_level(self.h, l, C.byref(w), C.byref(h), C.byref(s), C.byref(d))
        if blurred:
            lib().orc_orb_level_blurred(self.h, l, C.byref(d), C.byref(s))
        buf = (C.c_uint8 * (s.value * h.value)).from_address(d.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h.value, s.value)[:, : w.value].copy()

    def grid(self, l):
        v = [C.c_int() for _ in range(4)]
        lib().orc_orb_grid(self.h, l, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def candidates(self, l):
        d = C.c_void_p()
        n = lib().orc_orb_candidates(self.h, l, C.byref(d))
        if n == 0:
            return np.zeros((0, 3), dtype=np.int32)
        buf = (C.c_int32 * (3 * n)).from_address(d.value)
        return np.frombuffer(buf, dtype=np.int32).reshape(n, 3).copy()

    def level_count(self, l):
        return lib().orc_orb_level_count(self.h, l)


def pattern():
    return np.ctypeslib.as_array(lib().orc_orb_pattern(), shape=(1024,)).copy()


def hamming_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((len(q), 2), np.int32); dist = np.zeros((len(q), 2), np.int32)
    lib().orc_hamming_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


def hamming_matrix(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    d = np.zeros((len(q), len(t)), np.uint16)
    lib().orc_hamming_matrix(_p(q), len(q), _p(t), len(t), _p(d))
    return d


def match_nnr(d1, d2, nnr):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    m = np.zeros(len(d1), np.int32)
    n = lib().orc_match_nnr(_p(d1), len(d1), _p(d2), len(d2), nnr, _p(m))
    return n, m


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), w, h, src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur(src, ksize, sigma):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    dst = np.zeros((h, w), np.uint8)
    lib().orc_gaussian_blur_u8(_p(src), w, h, src.strides[0], _p(dst), w, ksize, sigma)
    return dst


def gaussian_kernel_q8(ksize, sigma):
    k = np.zeros(ksize, np.int32)
    lib().orc_gaussian_kernel_q8(ksize, sigma, _p(k))
    return k


def sobel3(src, dx, dy):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    dst = np.zeros((h, w), np.int16)
    lib().orc_sobel3_u8_s16(_p(src), w, h, src.strides[0], _p(dst), w, dx, dy)
    return dst


def fast9_16(view, threshold, cap=4096):
    view = np.ascontiguousarray(view, np.uint8); h, w = view.shape
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast9_16(_p(view), view.strides[0], w, h, threshold, _p(out), cap)
    return out[:n].copy()


def peac(depth, fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_factor=None, cap=64):
    """PlaneDetection::readDepthImage + runPlaneDetection (src/PlaneExtractor.cpp:26-66)"""
    depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
    if depth_factor is None:
        depth_factor = float(np.float32(1.0) / np.float32(5000.0))
    labels = np.zeros((h, w), np.int32); planes = np.zeros(cap, PLANE_DT); n = C.c_int(0)
    L = lib()
    L.orc_peac_run(_p(depth), w, h, depth.strides[0], fx, fy, cx, cy, depth_factor, _p(labels), _p(planes), cap, C.byref(n))
    return labels, planes[: min(n.value, cap)].copy()


LINE3D_DT = np.dtype([("A", "<f8", 3), ("B", "<f8", 3), ("line_nor", "<f8", 3), ("line_eq", "<f4", 3), ("good", "<i4"),
                      ("n_samples", "<i4"), ("n_inliers", "<i4"), ("inlier_mask", "<u4"), ("pad", "<i4")])


def lines_3d(kl, depth, seed=1, fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_factor=None):
    """Frame::isLineGood (src/Frame.cc:1205-1322) -> LINE3D_DT array"""
    kl = np.ascontiguousarray(kl); depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
    if depth_factor is None:
        depth_factor = float(np.float32(1.0) / np.float32(5000.0))
    out = np.zeros(len(kl), LINE3D_DT)
    L = lib()
    L.orc_lines_3d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_uint32, C.c_void_p]
    L.orc_lines_3d(_p(kl), len(kl), _p(depth), w, h, depth.strides[0], fx, fy, cx, cy, depth_factor, seed, _p(out))
    return out


def vanishing_points(kl, seed=1, th_angle=None, fx=535.4, fy=539.2, cx=320.1, cy=247.6, want_scores=False):
    """Frame::getVPHypVia2Lines .. line2Vps (src/Frame.cc:442-778) -> dict(vps (3,3), best, score, vp_idx (n), grid (90,360)[, scores])"""
    kl = np.ascontiguousarray(kl); n = len(kl)
    if th_angle is None:
        th_angle = 1.0 / 180.0 * 3.1415926535897932384626433832795
    L = lib()
    L.orc_vanishing_points.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_uint32, C.c_double] + [C.c_void_p] * 6
    L.orc_vp_iterations.restype = C.c_int
    vps = np.zeros((3, 3)); best = C.c_int(0); score = C.c_double(0); idx = np.full(n, 3, np.int32); grid = np.zeros((90, 360))
    scores = np.zeros(L.orc_vp_iterations() * 360) if want_scores else None
    rc = L.orc_vanishing_points(_p(kl), n, fx, fy, cx, cy, seed, th_angle, _p(vps), C.byref(best), C.byref(score), _p(idx),
                                _p(scores) if want_scores else None, _p(grid))
    if rc != 0:
        return None
    out = dict(vps=vps, best=best.value, score=score.value, vp_idx=idx, grid=grid)
    if want_scores:
        out["scores"] = scores
    return out


def vp_line2vps(kl, vps, th_angle=None, fx=535.4, fy=539.2, cx=320.1, cy=247.6):
    """Frame::line2Vps (src/Frame.cc:708-778) for a given hypothesis triple -> vp_idx (n)"""
    kl = np.ascontiguousarray(kl); n = len(kl)
    if th_angle is None:
        th_angle = 1.0 / 180.0 * 3.1415926535897932384626433832795
    L = lib()
    L.orc_vp_line2vps.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_double, C.c_void_p]
    L.orc_vp_line2vps.restype = None
    vps = np.ascontiguousarray(vps, np.float64); idx = np.full(n, 3, np.int32)
    L.orc_vp_line2vps(_p(kl), n, fx, fy, cx, cy, _p(vps), th_angle, _p(idx))
    return idx


def vp_hypothesis(kl, seed, index, fx=535.4, cx=320.1, cy=247.6):
    """the hypothesis triple (3,3) the path draws at `index` (group index // 360, rotation index % 360)"""
    kl = np.ascontiguousarray(kl)
    L = lib()
    L.orc_vp_hypothesis.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 3 + [C.c_uint32, C.c_int, C.c_void_p]
    L.orc_vp_hypothesis.restype = None
    h = np.zeros((3, 3))
    L.orc_vp_hypothesis(_p(kl), len(kl), fx, cx, cy, seed, index, _p(h))
    return h


PLANE_CLOUD_DT = np.dtype([("coef", "<f4", 4), ("valid", "<i4"), ("gate_ok", "<i4"), ("first", "<i4"), ("n_points", "<i4"), ("n_pixels", "<i4"), ("n_inliers", "<i4")])
SURFACE_NORMAL_DT = np.dtype([("normal", "<f4", 3), ("position", "<f4", 3), ("frame_x", "<i4"), ("frame_y", "<i4")])


def plane_clouds(depth, labels, planes, dist_th=0.05, fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_factor=None, cap=200000):
    """the per-plane part of Frame::ComputePlanes' tail (src/Frame.cc:2110-2154, 2214-2274) -> (PLANE_CLOUD_DT array, cloud (n,3) f32)"""
    depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
    labels = np.ascontiguousarray(labels, np.int32); planes = np.ascontiguousarray(planes)
    if depth_factor is None:
        depth_factor = float(np.float32(1.0) / np.float32(5000.0))
    out = np.zeros(len(planes), PLANE_CLOUD_DT); cloud = np.zeros((cap, 3), np.float32)
    L = lib()
    L.orc_plane_clouds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p]
    n = L.orc_plane_clouds(_p(depth), w, h, depth.strides[0], fx, fy, cx, cy, depth_factor, _p(labels), _p(planes), len(planes), dist_th, _p(cloud), cap, _p(out))
    return out, cloud[: min(n, cap)].copy()


def sac_plane(xyz, threshold):
    xyz = np.ascontiguousarray(xyz, np.float32); coef = np.zeros(4, np.float32)
    L = lib()
    L.orc_sac_plane.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    n = L.orc_sac_plane(_p(xyz), len(xyz), threshold, _p(coef))
    return n, coef


def surface_normals(depth, fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_factor=None):
    """the 1/3-resolution cloud + integral-image normals of Frame::ComputePlanes (src/Frame.cc:2157-2212) -> SURFACE_NORMAL_DT array"""
    depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
    if depth_factor is None:
        depth_factor = float(np.float32(1.0) / np.float32(5000.0))
    cap = ((h + 2) // 3) * ((w + 2) // 3)
    out = np.zeros(cap, SURFACE_NORMAL_DT)
    L = lib()
    L.orc_surface_normals.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_void_p, C.c_int]
    n = L.orc_surface_normals(_p(depth), w, h, depth.strides[0], fx, fy, cx, cy, depth_factor, _p(out), cap)
    return out[:n].copy()


def normals_lpvo(depth, fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_factor=None):
    """Manhattan::computeNormalsLPVO (src/Manhattan.cpp:237-393), the CV_32F reading -> (normals (n,3) f64, depth (n) f32, pixel (n,2) i32 = (u, v))"""
    depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
    if depth_factor is None:
        depth_factor = float(np.float32(1.0) / np.float32(5000.0))
    cap = ((h + 14) // 15) * ((w + 14) // 15)
    nrm = np.zeros((cap, 3)); dz = np.zeros(cap, np.float32); px = np.zeros((cap, 2), np.int32)
    L = lib()
    L.orc_normals_lpvo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    n = L.orc_normals_lpvo(_p(depth), w, h, depth.strides[0], fx, fy, cx, cy, depth_factor, _p(nrm), _p(dz), _p(px), cap)
    return nrm[:n], dz[:n], px[:n]


def eig33_smallest(K):
    """the smallest eigenpair as Stats::compute uses it (orc_eig33_smallest) -> (lambda0, v)"""
    K = np.ascontiguousarray(K, np.float64); l = C.c_double(0); v = np.zeros(3)
    L = lib()
    L.orc_eig33_smallest.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]; L.orc_eig33_smallest.restype = None
    L.orc_eig33_smallest(_p(K), C.byref(l), _p(v))
    return l.value, v


def eig33sym(K):
    K = np.ascontiguousarray(K, np.float64); s = np.zeros(3); V = np.zeros((3, 3))
    L = lib()
    L.orc_eig33sym.argtypes = [C.c_void_p] * 3
    L.orc_eig33sym(_p(K), _p(s), _p(V))
    return s, V


def lsd_detect(gray, cap=16384):
    """cv::LineSegmentDetector::detect, default parameters -> (n,4) float32 segments"""
    gray = np.ascontiguousarray(gray, np.uint8); h, w = gray.shape
    segs = np.zeros((cap, 4), np.float32); n = C.c_int(0)
    lib().orc_lsd_detect(_p(gray), w, h, gray.strides[0], _p(segs), cap, C.byref(n))
    return segs[: min(n.value, cap)].copy()


def line_extract(gray, nfeatures=200):
    """LINEextractor::operator() -> (keylines, descriptors, line functions)"""
    gray = np.ascontiguousarray(gray, np.uint8); h, w = gray.shape
    cap = max(nfeatures, 1)
    kl = np.zeros(cap, KEYLINE_DT); desc = np.zeros((cap, 32), np.uint8); fn = np.zeros((cap, 3)); n = C.c_int(0)
    lib().orc_line_extract(_p(gray), w, h, gray.strides[0], nfeatures, _p(kl), _p(desc), _p(fn), cap, C.byref(n))
    return kl[: n.value].copy(), desc[: n.value].copy(), fn[: n.value].copy()


def lbd_compute(gray, keylines, want_float=False):
    gray = np.ascontiguousarray(gray, np.uint8); h, w = gray.shape
    kl = np.ascontiguousarray(keylines); n = len(kl)
    desc = np.zeros((n, 32), np.uint8); f = np.zeros((n, 72), np.float32)
    lib().orc_lbd_compute(_p(gray), w, h, gray.strides[0], _p(kl), n, _p(desc), _p(f))
    return (desc, f) if want_float else desc


def lbd_weights():
    a = np.zeros(21); b = np.zeros(63)
    lib().orc_lbd_weights(_p(a), _p(b))
    return a, b


def lbd_combinations():
    return np.ctypeslib.as_array(lib().orc_lbd_combinations(), shape=(32, 2)).copy()


def search_by_projection(q_desc, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_angle, q_blocks,
                         t_kp, t_uright, t_occupied, t_desc, bounds, th_high=100, check_orientation=True):
    """ORBmatcher::SearchByProjection(Cur, Last) core -> (nmatches, match_idx, match_dist)"""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    q_desc = np.ascontiguousarray(q_desc, np.uint8); t_desc = np.ascontiguousarray(t_desc, np.uint8)
    nq, nt = len(q_desc), len(t_desc)
    q_u, q_v, q_radius, q_ur, q_angle = map(f32, (q_u, q_v, q_radius, q_ur, q_angle))
    q_min_level = np.ascontiguousarray(q_min_level, np.int32); q_max_level = np.ascontiguousarray(q_max_level, np.int32)
    q_blocks = np.ascontiguousarray(q_blocks, np.uint8); t_occupied = np.ascontiguousarray(t_occupied, np.uint8)
    t_kp = np.ascontiguousarray(t_kp); t_uright = f32(t_uright)
    mi = np.zeros(nq, np.int32); md = np.zeros(nq, np.int32)
    L = lib()
    L.orc_search_by_projection.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_void_p] * 4 + [C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = L.orc_search_by_projection(_p(q_desc), nq, _p(q_u), _p(q_v), _p(q_radius), _p(q_min_level), _p(q_max_level), _p(q_ur), _p(q_angle),
                                   _p(q_blocks), _p(t_kp), _p(t_uright), _p(t_occupied), _p(t_desc), nt,
                                   bounds[0], bounds[1], bounds[2], bounds[3], th_high, 1 if check_orientation else 0, _p(mi), _p(md))
    return n, mi, md


def project_last(Tcw, Tlw, x3Dw, octave, cam, mono, th, scale_factors, bounds):
    """prologue of ORBmatcher::SearchByProjection(Cur, Last) (src/ORBmatcher.cc:1364-1405) -> dict of query arrays; cam = (fx, fy, cx, cy, mbf, mb)"""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    Tcw = f32(Tcw).reshape(12); Tlw = f32(Tlw).reshape(12); x3Dw = f32(x3Dw); octave = np.ascontiguousarray(octave, np.int32); sf = f32(scale_factors)
    n = len(octave)
    out = dict(u=np.zeros(n, np.float32), v=np.zeros(n, np.float32), radius=np.zeros(n, np.float32), min_level=np.zeros(n, np.int32),
               max_level=np.zeros(n, np.int32), ur=np.zeros(n, np.float32), fwd_bwd=np.zeros(2, np.int32))
    L = lib(); L.orc_project_last.restype = None
    L.orc_project_last.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 6 + [C.c_int, C.c_float, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p] * 7
    L.orc_project_last(_p(Tcw), _p(Tlw), n, _p(x3Dw), _p(octave), *[float(c) for c in cam], 1 if mono else 0, float(th), _p(sf),
                       bounds[0], bounds[1], bounds[2], bounds[3], _p(out["u"]), _p(out["v"]), _p(out["radius"]), _p(out["min_level"]), _p(out["max_level"]),
                       _p(out["ur"]), _p(out["fwd_bwd"]))
    return out


def track_windows(level, view_cos, th, scale_factors):
    """prologue of SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:55-70, 134-140) -> (radius, min_level, max_level)"""
    level = np.ascontiguousarray(level, np.int32); vc = np.ascontiguousarray(view_cos, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    n = len(level); r = np.zeros(n, np.float32); lo = np.zeros(n, np.int32); hi = np.zeros(n, np.int32)
    L = lib(); L.orc_track_windows.restype = None
    L.orc_track_windows.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_track_windows(n, _p(level), _p(vc), float(th), _p(sf), _p(r), _p(lo), _p(hi))
    return r, lo, hi


def stereo_from_rgbd(kp, kp_un, depth, depth_factor, bf):
    kp = np.ascontiguousarray(kp); kp_un = np.ascontiguousarray(kp_un); depth = np.ascontiguousarray(depth, np.uint16)
    h, w = depth.shape
    ur = np.zeros(len(kp), np.float32); z = np.zeros(len(kp), np.float32)
    L = lib()
    L.orc_stereo_from_rgbd.restype = None
    L.orc_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_stereo_from_rgbd(_p(kp), _p(kp_un), len(kp), _p(depth), w, h, depth.strides[0], depth_factor, bf, _p(ur), _p(z))
    return ur, z


# ---------------- Frame post-processing (frame.c) ----------------
def undistort_keypoints(kp, fx, fy, cx, cy, dist5):
    kp = np.ascontiguousarray(kp); out = np.zeros_like(kp)
    d = np.ascontiguousarray(dist5, np.float32)
    L = lib()
    L.orc_undistort_keypoints.restype = None
    L.orc_undistort_keypoints.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p]
    L.orc_undistort_keypoints(_p(kp), len(kp), fx, fy, cx, cy, _p(d), _p(out))
    return out


def image_bounds(w, h, fx, fy, cx, cy, dist5):
    d = np.ascontiguousarray(dist5, np.float32); b = np.zeros(4, np.float32)
    L = lib()
    L.orc_image_bounds.restype = None
    L.orc_image_bounds.argtypes = [C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p]
    L.orc_image_bounds(w, h, fx, fy, cx, cy, _p(d), _p(b))
    return b


def assign_features_to_grid(kp_un, bounds4):
    kp_un = np.ascontiguousarray(kp_un); b = np.ascontiguousarray(bounds4, np.float32)
    start = np.zeros(64 * 48 + 1, np.int32); items = np.zeros(max(len(kp_un), 1), np.int32)
    L = lib()
    L.orc_assign_features_to_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.orc_assign_features_to_grid(_p(kp_un), len(kp_un), _p(b), _p(start), _p(items))
    return start, items[:n]


def assign_lines_to_grid(kl, bounds4, cap=None):
    kl = np.ascontiguousarray(kl); b = np.ascontiguousarray(bounds4, np.float32)
    cap = cap if cap is not None else max(len(kl), 1) * 128
    start = np.zeros(64 * 48 + 1, np.int32); items = np.zeros(cap, np.int32)
    L = lib()
    L.orc_assign_lines_to_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    n = L.orc_assign_lines_to_grid(_p(kl), len(kl), _p(b), _p(start), _p(items), cap)
    return start, items[:max(n, 0)], n


def lines_geom_match(d_last, kl_last, d_cur, kl_cur, bounds4, desc_th=0.9, last_has_mapline=None):
    """LSDmatcher::SearchByGeomNApearance -> (lmatches, matches12, accepted)"""
    d_last = np.ascontiguousarray(d_last, np.uint8); d_cur = np.ascontiguousarray(d_cur, np.uint8)
    kl_last = np.ascontiguousarray(kl_last); kl_cur = np.ascontiguousarray(kl_cur); b = np.ascontiguousarray(bounds4, np.float32)
    n1, n2 = len(kl_last), len(kl_cur)
    hm = None if last_has_mapline is None else np.ascontiguousarray(last_has_mapline, np.uint8)
    m = np.zeros(max(n1, 1), np.int32); acc = np.zeros(max(n1, 1), np.uint8)
    L = lib()
    L.orc_lines_geom_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.orc_lines_geom_match(_p(d_last), _p(kl_last), None if hm is None else _p(hm), n1, _p(d_cur), _p(kl_cur), n2, desc_th, _p(b), _p(m), _p(acc))
    return n, m[:n1], acc[:n1]


def search_lines_by_projection(q_xyxy, q_kl, q_desc, q_blocks, t_kl, t_linefn, t_desc, t_occupied, cell_start, cell_items, bounds4, th):
    """LSDmatcher::SearchByProjection(Cur, Last, th) core -> (nmatches, match_idx, match_dist)"""
    q_xyxy = np.ascontiguousarray(q_xyxy, np.float32).reshape(-1, 4); nq = len(q_xyxy)
    q_kl = np.ascontiguousarray(q_kl); t_kl = np.ascontiguousarray(t_kl); nt = len(t_kl)
    q_desc = np.ascontiguousarray(q_desc, np.uint8); t_desc = np.ascontiguousarray(t_desc, np.uint8)
    q_blocks = np.ascontiguousarray(q_blocks, np.uint8); t_occupied = np.ascontiguousarray(t_occupied, np.uint8)
    t_linefn = np.ascontiguousarray(t_linefn, np.float64); b = np.ascontiguousarray(bounds4, np.float32)
    cs = np.ascontiguousarray(cell_start, np.int32); ci = np.ascontiguousarray(cell_items, np.int32)
    mi = np.zeros(max(nq, 1), np.int32); md = np.zeros(max(nq, 1), np.int32)
    L = lib()
    L.orc_search_lines_by_projection.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    n = L.orc_search_lines_by_projection(nq, _p(q_xyxy), _p(q_kl), _p(q_desc), _p(q_blocks), _p(t_kl), _p(t_linefn), _p(t_desc), _p(t_occupied), nt,
                                         _p(cs), _p(ci), _p(b), th, _p(mi), _p(md))
    return n, mi[:nq], md[:nq]


def cull_lines(gray, kl, fn, dis=5.0, angle=2.5, endpoint_dis=15.0):
    """Frame::cullingLine(im, 5, 2.5, 15, 30) (src/Frame.cc:934, 952-1116) -> (keylines, descriptors, line functions)"""
    gray = np.ascontiguousarray(gray, np.uint8); kl = np.ascontiguousarray(kl); fn = np.ascontiguousarray(fn, np.float64)
    h, w = gray.shape; n = len(kl)
    out = np.zeros(max(n, 1), KEYLINE_DT); desc = np.zeros((max(n, 1), 32), np.uint8); fo = np.zeros((max(n, 1), 3))
    L = lib()
    L.orc_cull_lines.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    m = L.orc_cull_lines(_p(gray), w, h, gray.strides[0], _p(kl), _p(fn), n, dis, angle, endpoint_dis, _p(out), _p(desc), _p(fo))
    return out[:m], desc[:m], fo[:m]


def line_iterator_count_clipped(w, h, x1, y1, x2, y2):
    L = lib()
    L.orc_line_iterator_count_clipped.argtypes = [C.c_int, C.c_int] + [C.c_float] * 4
    return L.orc_line_iterator_count_clipped(w, h, x1, y1, x2, y2)


def search_by_projection_map(q_desc, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_blocks,
                             t_kp, t_uright, t_occupied, t_desc, bounds, th_high=100, nn_ratio=0.8):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) core (src/ORBmatcher.cc:45-132) -> (nmatches, match_idx, match_dist)"""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    q_desc = np.ascontiguousarray(q_desc, np.uint8); t_desc = np.ascontiguousarray(t_desc, np.uint8)
    nq, nt = len(q_desc), len(t_desc)
    q_u, q_v, q_radius, q_ur = map(f32, (q_u, q_v, q_radius, q_ur))
    q_min_level = np.ascontiguousarray(q_min_level, np.int32); q_max_level = np.ascontiguousarray(q_max_level, np.int32)
    q_blocks = np.ascontiguousarray(q_blocks, np.uint8); t_occupied = np.ascontiguousarray(t_occupied, np.uint8)
    t_kp = np.ascontiguousarray(t_kp); t_uright = f32(t_uright)
    mi = np.zeros(nq, np.int32); md = np.zeros(nq, np.int32)
    L = lib()
    L.orc_search_by_projection_map.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p] * 4 + [C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    n = L.orc_search_by_projection_map(_p(q_desc), nq, _p(q_u), _p(q_v), _p(q_radius), _p(q_min_level), _p(q_max_level), _p(q_ur),
                                       _p(q_blocks), _p(t_kp), _p(t_uright), _p(t_occupied), _p(t_desc), nt,
                                       bounds[0], bounds[1], bounds[2], bounds[3], th_high, nn_ratio, _p(mi), _p(md))
    return n, mi, md


def frame_bf_match(d1, d2, TH=50.0, nnratio=0.9, mutual=False):
    """LSDmatcher::FrameBFMatch (mutual=False) / SearchDouble core (mutual=True) -> (nmatches, matches12)"""
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    m = np.full(max(len(d1), 1), -1, np.int32)
    L = lib()
    fn = L.orc_search_double if mutual else L.orc_frame_bf_match
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
    n = fn(_p(d1), len(d1), _p(d2), len(d2), TH, nnratio, _p(m))
    return n, m[: len(d1)]

