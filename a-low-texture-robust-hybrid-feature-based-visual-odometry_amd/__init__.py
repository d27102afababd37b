"""MI355X-native RGB-D front-end (ORB + LSD/LBD + PEAC planes + Hamming matching).

Thin ctypes binding over libhvo.so (csrc/, C ABI in include/hvo.h) plus host-side mirrors
of the reference's operator interfaces so parity tests read like the reference's call sites:

    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)(image)
        -> reference include/ORBextractor.h:53-61, src/ORBextractor.cc:1041
    LINEextractor(numOctaves, scale, nLSDFeature)(image)
        -> reference include/LineExtractor.h:186-193, src/LineExtractor.cpp:329
    PlaneDetection(K, depthMapFactor).run(depth_u16)
        -> reference include/PlaneExtractor.h:36-56, src/PlaneExtractor.cpp:26-66
    ORBmatcher.DescriptorDistance / LSDmatcher.match
        -> reference src/ORBmatcher.cc:1676, src/LSDmatcher.cpp:828

There is no CPU path: if libhvo.so is missing or no gfx950 device is present every call
raises HvoError.  This package never imports the oracle.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# HVO_LIB: developer knob, points the binding at an experimental build of the same library (A/B measurements)
_LIBPATH = os.environ.get("HVO_LIB") or os.path.join(_CSRC, "libhvo.so")
_LIB = None
# Load-order rule (multi-GPU path): torch's wheel carries its own HIP runtime; whichever of torch / libhvo.so is loaded first decides
# which libamdhip64 the process runs on, and a process that loaded libhvo.so first cannot initialise torch.cuda afterwards.  lib()
# records the order and torch_order_check() (called by every entry point that hands device memory to torch, dist.py) raises a clear
# error instead of the runtime's obscure one.
_LOADED_BEFORE_TORCH = False

HVO_OK = 0
STAGE_ORB, STAGE_LSD, STAGE_PLANES, STAGE_ALL = 1, 2, 4, 7
SLAB_LABELS = 1            # hvo_batch_pack_results_ex: the int8 label image at the end of every slab
STAGE_LSD_CULL = 8        # STAGE_LSD followed by Frame::cullingLine (merged lines replace the extractor's)
# the rest of the Frame constructor as pipeline stages (include/hvo.h): isLineGood, vanishing points, ComputePlanes' tail, the two grids
STAGE_LINES3D, STAGE_VP, STAGE_PLANE_TAIL, STAGE_GRIDS = 16, 32, 64, 128
STAGE_FRAME = 1 | 2 | 8 | 4 | 16 | 32 | 64 | 128
LINE_MATCH_NNR, LINE_MATCH_BF, LINE_MATCH_DOUBLE = 0, 1, 2

KEYPOINT_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                        ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
KEYLINE_DT = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"),
                       ("pt_x", "<f4"), ("pt_y", "<f4"), ("response", "<f4"), ("size", "<f4"),
                       ("sx", "<f4"), ("sy", "<f4"), ("ex", "<f4"), ("ey", "<f4"),
                       ("sox", "<f4"), ("soy", "<f4"), ("eox", "<f4"), ("eoy", "<f4"),
                       ("length", "<f4"), ("num_pixels", "<i4")])
PLANE_DT = np.dtype([("normal", "<f8", 3), ("center", "<f8", 3), ("mse", "<f8"),
                     ("n_points", "<i4"), ("rid", "<i4")])
class VpResult(C.Structure):
    _fields_ = [("vps", (C.c_double * 3) * 3), ("score", C.c_double), ("best", C.c_int32), ("n_hypotheses", C.c_int32)]


LINE3D_DT = np.dtype([("A", "<f8", 3), ("B", "<f8", 3), ("line_nor", "<f8", 3), ("line_eq", "<f4", 3), ("good", "<i4"),
                      ("n_samples", "<i4"), ("n_inliers", "<i4"), ("inlier_mask", "<u4"), ("pad", "<i4")])
PLANE_CLOUD_DT = np.dtype([("coef", "<f4", 4), ("valid", "<i4"), ("gate_ok", "<i4"), ("first", "<i4"), ("n_points", "<i4"), ("n_pixels", "<i4"), ("n_inliers", "<i4")])
SURFACE_NORMAL_DT = np.dtype([("normal", "<f4", 3), ("position", "<f4", 3), ("frame_x", "<i4"), ("frame_y", "<i4")])
assert KEYPOINT_DT.itemsize == 28 and KEYLINE_DT.itemsize == 68 and PLANE_DT.itemsize == 64 and LINE3D_DT.itemsize == 104
assert PLANE_CLOUD_DT.itemsize == 40 and SURFACE_NORMAL_DT.itemsize == 32

READING_BLUR_FLOAT, READING_LSD_8U = 1, 2

EXPORTS = [
    "hvo_abi_version", "hvo_default_params", "hvo_create", "hvo_destroy", "hvo_strerror", "hvo_last_error",
    "hvo_extract_orb", "hvo_extract_lsd", "hvo_compute_planes",
    "hvo_hamming_matrix", "hvo_hamming_knn2", "hvo_match_nnr", "hvo_match_lines_geom", "hvo_search_lines_by_projection", "hvo_stream_match_lines_geom", "hvo_stream_search_lines_by_projection", "hvo_search_by_projection", "hvo_stereo_from_rgbd",
    "hvo_undistort_keypoints", "hvo_image_bounds", "hvo_assign_features_to_grid", "hvo_assign_lines_to_grid",
    "hvo_extract_lsd_culled", "hvo_set_line_culling", "hvo_lines_3d", "hvo_vanishing_points", "hvo_plane_clouds", "hvo_surface_normals", "hvo_search_by_projection_map", "hvo_frame_bf_match", "hvo_search_double",
    "hvo_batch_upload", "hvo_batch_run", "hvo_batch_download", "hvo_extract_batch", "hvo_batch_slab_layout", "hvo_batch_pack_results", "hvo_batch_slab_layout_ex", "hvo_batch_pack_results_ex", "hvo_batch_stage_upload", "hvo_batch_commit_staged", "hvo_batch_results_async", "hvo_batch_results_wait",
    "hvo_profile_last", "hvo_profile_enable", "hvo_lsd_async_report", "hvo_set_readings", "hvo_stream_set_readings", "hvo_pin_host", "hvo_unpin_host",
    "hvo_stream_create", "hvo_stream_destroy", "hvo_stream_last_error", "hvo_stream_capacity", "hvo_stream_image_bounds",
    "hvo_stream_submit", "hvo_stream_poll", "hvo_stream_collect", "hvo_stream_stage_ms",
    "hvo_stream_search_by_projection", "hvo_stream_match_lines", "hvo_stream_project_last", "hvo_search_by_projection_tracked",
    "hvo_tail_capacity", "hvo_set_tail_params", "hvo_batch_download_tail", "hvo_stream_collect_tail", "hvo_normals_lpvo",
]


class HvoError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = _LIB.hvo_strerror(status).decode() if _LIB is not None else "libhvo.so unavailable"
        super().__init__("hvo status %d (%s) %s" % (status, msg, what))


class Params(C.Structure):
    _fields_ = [("orb_nfeatures", C.c_int32), ("orb_scale_factor", C.c_float), ("orb_nlevels", C.c_int32),
                ("orb_ini_th_fast", C.c_int32), ("orb_min_th_fast", C.c_int32),
                ("lsd_num_octaves", C.c_int32), ("lsd_scale", C.c_float), ("lsd_nfeatures", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("depth_map_factor", C.c_float), ("device", C.c_int32), ("max_batch", C.c_int32)]


class FrameIn(C.Structure):
    _fields_ = [("gray", C.c_void_p), ("gray_stride", C.c_int),
                ("depth", C.c_void_p), ("depth_stride", C.c_int)]


class FrameOut(C.Structure):
    _fields_ = [("kp", C.c_void_p), ("desc", C.c_void_p), ("kp_cap", C.c_int), ("n_kp", C.c_int),
                ("kl", C.c_void_p), ("ldesc", C.c_void_p), ("linefn", C.c_void_p), ("kl_cap", C.c_int), ("n_kl", C.c_int),
                ("labels", C.c_void_p), ("planes", C.c_void_p), ("pl_cap", C.c_int), ("n_planes", C.c_int),
                ("status", C.c_int), ("labels8", C.c_void_p)]


class StreamParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_int32), ("stages", C.c_uint32),
                ("dist5", C.c_float * 5), ("bf", C.c_float), ("seed", C.c_uint32), ("plane_dist_th", C.c_float), ("vp_th_angle", C.c_float)]


class FrameTail(C.Structure):
    _fields_ = [("lines3d", C.c_void_p), ("vp", C.c_void_p), ("vp_idx", C.c_void_p),
                ("plane_clouds", C.c_void_p), ("cloud_xyz", C.c_void_p), ("cloud_cap", C.c_int), ("n_cloud", C.c_int),
                ("normals", C.c_void_p), ("normals_cap", C.c_int), ("n_normals", C.c_int),
                ("pt_cell_start", C.c_void_p), ("pt_cell_items", C.c_void_p), ("pt_items_cap", C.c_int), ("n_pt_items", C.c_int),
                ("ln_cell_start", C.c_void_p), ("ln_cell_items", C.c_void_p), ("ln_items_cap", C.c_int), ("n_ln_items", C.c_int),
                ("status", C.c_int)]


def _tail_buffers(kp_cap, kl_cap, w, h):
    """numpy result arrays of one frame's tail stages + the FrameTail that points at them"""
    cc = C.c_int(0); nn = C.c_int(0); lc = C.c_int(0)
    lib().hvo_tail_capacity(kl_cap, w, h, C.byref(cc), C.byref(nn), C.byref(lc))
    b = dict(lines3d=np.zeros(kl_cap, LINE3D_DT), vp=VpResult(), vp_idx=np.full(kl_cap, 3, np.int32), plane_clouds=np.zeros(64, PLANE_CLOUD_DT),
             cloud_xyz=np.zeros((cc.value, 3), np.float32), normals=np.zeros(max(nn.value, 1), SURFACE_NORMAL_DT),
             pt_cell_start=np.zeros(64 * 48 + 1, np.int32), pt_cell_items=np.zeros(max(kp_cap, 1), np.int32),
             ln_cell_start=np.zeros(64 * 48 + 1, np.int32), ln_cell_items=np.zeros(max(lc.value, 1), np.int32))
    t = FrameTail()
    t.lines3d = b["lines3d"].ctypes.data; t.vp = C.addressof(b["vp"]); t.vp_idx = b["vp_idx"].ctypes.data
    t.plane_clouds = b["plane_clouds"].ctypes.data; t.cloud_xyz = b["cloud_xyz"].ctypes.data; t.cloud_cap = cc.value
    t.normals = b["normals"].ctypes.data; t.normals_cap = nn.value
    t.pt_cell_start = b["pt_cell_start"].ctypes.data; t.pt_cell_items = b["pt_cell_items"].ctypes.data; t.pt_items_cap = kp_cap
    t.ln_cell_start = b["ln_cell_start"].ctypes.data; t.ln_cell_items = b["ln_cell_items"].ctypes.data; t.ln_items_cap = lc.value
    return b, t


def _tail_result(b, t, n_kl, n_planes, stages):
    r = {"tail_status": t.status}
    if stages & STAGE_LINES3D: r["lines3d"] = b["lines3d"][:n_kl]
    if stages & STAGE_VP:
        v = b["vp"]
        r["vp"] = dict(vps=np.array([[v.vps[i][j] for j in range(3)] for i in range(3)]), score=v.score, best=v.best, n_hypotheses=v.n_hypotheses, vp_idx=b["vp_idx"][:n_kl])
    if stages & STAGE_PLANE_TAIL:
        r["plane_clouds"] = b["plane_clouds"][:n_planes]; r["cloud_xyz"] = b["cloud_xyz"][: t.n_cloud]; r["normals"] = b["normals"][: t.n_normals]
    if stages & STAGE_GRIDS:
        r["pt_grid"] = (b["pt_cell_start"], b["pt_cell_items"][: t.n_pt_items]); r["ln_grid"] = (b["ln_cell_start"], b["ln_cell_items"][: t.n_ln_items])
    return r


def build(force=False):
    """compile libhvo.so in-tree with hipcc --offload-arch=gfx950 (cross-compiles without a GPU)"""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", _CSRC])


def torch_order_check():
    """raise if libhvo.so was loaded into this process before torch (see _LOADED_BEFORE_TORCH): `import torch` must come first
    in every process that shares device memory between the two (dist.device_slabs, dist.gather_device_slabs, bench.py --gpus N)"""
    if _LIB is not None and _LOADED_BEFORE_TORCH:
        raise RuntimeError("libhvo.so was loaded before torch in this process: torch.cuda cannot be initialised on top of it. "
                           "Import torch (and call torch.cuda.init()) BEFORE the first hvo call in processes that use hvo_amd.dist.")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIBPATH):
            raise HvoError(-3, "libhvo.so not built (run __graft_entry__.build())")
        import sys
        global _LOADED_BEFORE_TORCH
        _LOADED_BEFORE_TORCH = "torch" not in sys.modules
        L = C.CDLL(_LIBPATH)
        L.hvo_strerror.restype = C.c_char_p
        L.hvo_strerror.argtypes = [C.c_int]
        L.hvo_last_error.restype = C.c_char_p
        L.hvo_last_error.argtypes = [C.c_void_p]
        L.hvo_default_params.argtypes = [C.POINTER(Params)]
        L.hvo_default_params.restype = None
        L.hvo_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.hvo_destroy.argtypes = [C.c_void_p]
        L.hvo_destroy.restype = None
        L.hvo_extract_orb.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.POINTER(C.c_int)]
        L.hvo_extract_lsd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.hvo_compute_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.POINTER(C.c_int)]
        L.hvo_hamming_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.hvo_hamming_knn2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.hvo_match_nnr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                    C.POINTER(C.c_int)]
        L.hvo_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_void_p] * 4 + [C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_frame_bf_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_search_double.argtypes = L.hvo_frame_bf_match.argtypes
        L.hvo_match_lines_geom.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_search_lines_by_projection.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_stream_match_lines_geom.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.hvo_stream_search_lines_by_projection.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_search_by_projection_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p] * 4 + [C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.hvo_extract_lsd_culled.argtypes = L.hvo_extract_lsd.argtypes
        L.hvo_set_line_culling.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.hvo_vanishing_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hvo_lines_3d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
        L.hvo_plane_clouds.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_surface_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.hvo_undistort_keypoints.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.hvo_image_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.hvo_assign_features_to_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_assign_lines_to_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.hvo_batch_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameIn), C.c_int, C.c_int]
        L.hvo_batch_run.argtypes = [C.c_void_p, C.c_uint]
        L.hvo_batch_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameOut)]
        L.hvo_extract_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameIn), C.POINTER(FrameOut), C.c_int, C.c_int, C.c_uint]
        L.hvo_batch_slab_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        L.hvo_batch_pack_results.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hvo_profile_last.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
        L.hvo_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.hvo_pin_host.argtypes = [C.c_void_p, C.c_size_t]; L.hvo_unpin_host.argtypes = [C.c_void_p]
        L.hvo_stream_create.argtypes = [C.POINTER(Params), C.POINTER(StreamParams), C.POINTER(C.c_void_p)]
        L.hvo_stream_destroy.argtypes = [C.c_void_p]; L.hvo_stream_destroy.restype = None
        L.hvo_stream_last_error.argtypes = [C.c_void_p]; L.hvo_stream_last_error.restype = C.c_char_p
        L.hvo_stream_capacity.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        L.hvo_stream_image_bounds.argtypes = [C.c_void_p, C.c_void_p]
        L.hvo_stream_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.hvo_stream_poll.argtypes = [C.c_void_p, C.c_int64]
        L.hvo_stream_collect.argtypes = [C.c_void_p, C.c_int64, C.POINTER(FrameOut), C.c_void_p, C.c_void_p, C.c_void_p]
        L.hvo_stream_stage_ms.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.hvo_stream_search_by_projection.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [C.c_void_p] * 10 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.hvo_stream_match_lines.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.hvo_normals_lpvo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.hvo_tail_capacity.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 3
        L.hvo_set_tail_params.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double]
        L.hvo_batch_download_tail.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameTail)]
        L.hvo_stream_collect_tail.argtypes = [C.c_void_p, C.c_int64, C.POINTER(FrameTail)]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pin(a):
    """page-lock a numpy array (hvo_pin_host); returns the array.  Unpin with unpin(a) before it is freed."""
    rc = lib().hvo_pin_host(a.ctypes.data, a.nbytes)
    if rc != HVO_OK:
        raise HvoError(rc, "hvo_pin_host")
    return a


def unpin(a):
    lib().hvo_unpin_host(a.ctypes.data)


def default_params(**kw):
    p = Params()
    lib().hvo_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown hvo_params field " + k)
        setattr(p, k, v)
    return p


class Context:
    """One hvo_ctx: one HIP stream + device slabs.  Not thread-safe (like ORBextractor)."""

    def __init__(self, params=None, **kw):
        self.params = params if params is not None else default_params(**kw)
        h = C.c_void_p()
        rc = lib().hvo_create(C.byref(self.params), C.byref(h))
        if rc != HVO_OK:
            raise HvoError(rc, "hvo_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            for a in getattr(self, "_pinned", []):
                unpin(a)
            self._pinned = []
            lib().hvo_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != HVO_OK:
            raise HvoError(rc, what + ": " + lib().hvo_last_error(self.h).decode())

    # ---- single frame ----
    def extract_orb(self, gray):
        if gray is None or gray.size == 0:
            return np.zeros(0, KEYPOINT_DT), np.zeros((0, 32), np.uint8)
        if gray.dtype != np.uint8 or gray.ndim != 2:
            raise HvoError(-6, "ORB input must be CV_8UC1 (ORBextractor.cc:1048)")
        gray = np.ascontiguousarray(gray)
        h, w = gray.shape
        cap = self.params.orb_nfeatures + 8 * self.params.orb_nlevels + 64
        kp = np.zeros(cap, KEYPOINT_DT); desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._chk(lib().hvo_extract_orb(self.h, _p(gray), w, h, gray.strides[0], _p(kp), _p(desc), cap, C.byref(n)), "extract_orb")
        return kp[: n.value].copy(), desc[: n.value].copy()

    def extract_lsd(self, gray, cap=None, culled=False):
        """LINEextractor::operator(); culled=True: Frame::ExtractLSD up to and including cullingLine (Frame.cc:895-934)"""
        if gray is None or gray.size == 0:
            return np.zeros(0, KEYLINE_DT), np.zeros((0, 32), np.uint8), np.zeros((0, 3))
        if gray.dtype != np.uint8 or gray.ndim != 2:
            raise HvoError(-6, "LSD input must be CV_8UC1 (LineExtractor.cpp:335)")
        gray = np.ascontiguousarray(gray)
        h, w = gray.shape
        cap = cap or max(self.params.lsd_nfeatures, 1)
        kl = np.zeros(cap, KEYLINE_DT); desc = np.zeros((cap, 32), np.uint8); fn = np.zeros((cap, 3), np.float64)
        n = C.c_int(0)
        fnc = lib().hvo_extract_lsd_culled if culled else lib().hvo_extract_lsd
        self._chk(fnc(self.h, _p(gray), w, h, gray.strides[0], _p(kl), _p(desc), _p(fn), cap, C.byref(n)), "extract_lsd")
        return kl[: n.value].copy(), desc[: n.value].copy(), fn[: n.value].copy()

    def set_line_culling(self, dis=5.0, angle_deg=2.5, endpoint_dis=15.0):
        """parameters of Frame::cullingLine (Frame.cc:934)"""
        self._chk(lib().hvo_set_line_culling(self.h, dis, angle_deg, endpoint_dis), "set_line_culling")

    def lines_3d(self, kl, depth, seed=1):
        """Frame::isLineGood (src/Frame.cc:1205-1322): mvLines3D / mvLineEq / mvLineNor of every key line -> LINE3D_DT array"""
        kl = np.ascontiguousarray(kl); depth = np.ascontiguousarray(depth, np.uint16)
        h, w = depth.shape
        out = np.zeros(len(kl), LINE3D_DT)
        self._chk(lib().hvo_lines_3d(self.h, _p(kl), len(kl), _p(depth), w, h, depth.strides[0], seed, _p(out)), "lines_3d")
        return out

    def vanishing_points(self, kl, seed=1, th_angle=None, want_grid=False):
        """Frame::getVPHypVia2Lines .. line2Vps (src/Frame.cc:442-778) -> dict(vps (3,3), best, score, n_hypotheses, vp_idx (n)[, grid (90,360)])"""
        kl = np.ascontiguousarray(kl); n = len(kl)
        if th_angle is None:
            th_angle = 1.0 / 180.0 * 3.1415926535897932384626433832795          # Frame.h:365
        res = VpResult(); idx = np.full(n, 3, np.int32); grid = np.zeros((90, 360)) if want_grid else None
        self._chk(lib().hvo_vanishing_points(self.h, _p(kl), n, seed, th_angle, C.byref(res), _p(idx), _p(grid) if want_grid else None), "vanishing_points")
        out = dict(vps=np.array([[res.vps[i][j] for j in range(3)] for i in range(3)]), best=res.best, score=res.score, n_hypotheses=res.n_hypotheses, vp_idx=idx)
        if want_grid:
            out["grid"] = grid
        return out

    def compute_planes(self, depth, cap=64):
        if depth.dtype != np.uint16 or depth.ndim != 2:
            raise HvoError(-6, "depth must be CV_16UC1 (PlaneExtractor.cpp:34-38)")
        depth = np.ascontiguousarray(depth)
        h, w = depth.shape
        labels = np.zeros((h, w), np.int32); planes = np.zeros(cap, PLANE_DT)
        n = C.c_int(0)
        self._chk(lib().hvo_compute_planes(self.h, _p(depth), w, h, depth.strides[0], _p(labels), _p(planes), cap, C.byref(n)), "compute_planes")
        return labels, planes[: n.value].copy()

    def plane_clouds(self, depth, labels, planes, dist_th=0.05, cap=200000):
        """the per-plane tail of Frame::ComputePlanes (src/Frame.cc:2110-2154, 2214-2274) -> (PLANE_CLOUD_DT array, cloud (n, 3) f32)"""
        depth = np.ascontiguousarray(depth, np.uint16); labels = np.ascontiguousarray(labels, np.int32); planes = np.ascontiguousarray(planes)
        h, w = depth.shape
        out = np.zeros(len(planes), PLANE_CLOUD_DT); cloud = np.zeros((cap, 3), np.float32); n = C.c_int(0)
        self._chk(lib().hvo_plane_clouds(self.h, _p(depth), w, h, depth.strides[0], _p(labels), _p(planes), len(planes), dist_th, _p(cloud), cap, _p(out), C.byref(n)), "plane_clouds")
        return out, cloud[: n.value].copy()

    def normals_lpvo(self, depth):
        """Manhattan::computeNormalsLPVO (src/Manhattan.cpp:237-393), the CV_32F reading -> (normals (n,3) f64, depth (n) f32, pixel (n,2) i32)"""
        depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
        cap = ((h + 14) // 15) * ((w + 14) // 15)
        nrm = np.zeros((cap, 3)); dz = np.zeros(cap, np.float32); px = np.zeros((cap, 2), np.int32); n = C.c_int(0)
        self._chk(lib().hvo_normals_lpvo(self.h, _p(depth), w, h, depth.strides[0], _p(nrm), _p(dz), _p(px), cap, C.byref(n)), "normals_lpvo")
        return nrm[: n.value], dz[: n.value], px[: n.value]

    def surface_normals(self, depth):
        """vSurfaceNormal of Frame::ComputePlanes (src/Frame.cc:2157-2212) -> SURFACE_NORMAL_DT array"""
        depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape
        cap = (((h + 2) // 3) // 2) * (((w + 2) // 3) // 2)
        out = np.zeros(max(cap, 1), SURFACE_NORMAL_DT); n = C.c_int(0)
        self._chk(lib().hvo_surface_normals(self.h, _p(depth), w, h, depth.strides[0], _p(out), cap, C.byref(n)), "surface_normals")
        return out[: n.value].copy()

    def peac_stats(self, frame=0):
        """diagnostics (not part of the reference interface): bookkeeping words of the last plane run of `frame`"""
        L = lib()
        L.hvo_debug_peac_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; L.hvo_debug_peac_stats.restype = C.c_int
        m = np.zeros(16, np.int32)
        self._chk(L.hvo_debug_peac_stats(self.h, frame, _p(m)), "peac_stats")
        return {"segments": int(m[0]), "coarse_planes": int(m[2]), "flags": int(m[3]), "planes": int(m[4]), "queue_entries": int(m[5]),
                "flood_rounds": int(m[8]), "flood_ranked_rounds": int(m[9]), "flood_serial_rounds": int(m[10]), "ahc_rounds": int(m[11]), "ahc_stops_no_head": int(m[12]), "ahc_stops_conflict": int(m[13]), "ahc_stops_key_order": int(m[14])}

    # ---- matching ----
    def hamming_matrix(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        d = np.zeros((len(q), len(t)), np.uint16)
        self._chk(lib().hvo_hamming_matrix(self.h, _p(q), len(q), _p(t), len(t), _p(d)), "hamming_matrix")
        return d

    def hamming_knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.zeros((len(q), 2), np.int32); dist = np.zeros((len(q), 2), np.int32)
        self._chk(lib().hvo_hamming_knn2(self.h, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist)), "hamming_knn2")
        return idx, dist

    def match_nnr(self, d1, d2, nnr):
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        m = np.full(len(d1), -1, np.int32); n = C.c_int(0)
        self._chk(lib().hvo_match_nnr(self.h, _p(d1), len(d1), _p(d2), len(d2), nnr, _p(m), C.byref(n)), "match_nnr")
        return n.value, m

    def search_by_projection(self, q_desc, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_angle, q_blocks,
                             t_kp, t_uright, t_occupied, t_desc, bounds, th_high=100, check_orientation=True):
        """ORBmatcher::SearchByProjection(Cur, Last) core (src/ORBmatcher.cc:1353-1497) -> (nmatches, idx, dist)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        q_desc = np.ascontiguousarray(q_desc, np.uint8); t_desc = np.ascontiguousarray(t_desc, np.uint8)
        nq, nt = len(q_desc), len(t_desc)
        q_u, q_v, q_radius, q_ur, q_angle, t_uright = map(f32, (q_u, q_v, q_radius, q_ur, q_angle, t_uright))
        q_min_level = np.ascontiguousarray(q_min_level, np.int32); q_max_level = np.ascontiguousarray(q_max_level, np.int32)
        q_blocks = np.ascontiguousarray(q_blocks, np.uint8); t_occupied = np.ascontiguousarray(t_occupied, np.uint8)
        t_kp = np.ascontiguousarray(t_kp)
        mi = np.zeros(nq, np.int32); md = np.zeros(nq, np.int32); n = C.c_int(0)
        self._chk(lib().hvo_search_by_projection(self.h, _p(q_desc), nq, _p(q_u), _p(q_v), _p(q_radius), _p(q_min_level), _p(q_max_level),
                                                 _p(q_ur), _p(q_angle), _p(q_blocks), _p(t_kp), _p(t_uright), _p(t_occupied), _p(t_desc), nt,
                                                 bounds[0], bounds[1], bounds[2], bounds[3], th_high, 1 if check_orientation else 0,
                                                 _p(mi), _p(md), C.byref(n)), "search_by_projection")
        return n.value, mi, md

    def frame_bf_match(self, d1, d2, TH=50.0, nnratio=0.9, mutual=False):
        """LSDmatcher::FrameBFMatch (LSDmatcher.cpp:942-966); mutual=True: SearchDouble's two-way check (902-939)"""
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        m = np.full(max(len(d1), 1), -1, np.int32); n = C.c_int(0)
        fn = lib().hvo_search_double if mutual else lib().hvo_frame_bf_match
        self._chk(fn(self.h, _p(d1), len(d1), _p(d2), len(d2), TH, nnratio, _p(m), C.byref(n)), "frame_bf_match")
        return n.value, m[: len(d1)]

    def match_lines_geom(self, d_last, kl_last, d_cur, kl_cur, bounds4, desc_th=0.9, last_has_mapline=None):
        """LSDmatcher::SearchByGeomNApearance (src/LSDmatcher.cpp:36-108) -> (lmatches, matches12, accepted)"""
        d_last = np.ascontiguousarray(d_last, np.uint8); d_cur = np.ascontiguousarray(d_cur, np.uint8)
        kl_last = np.ascontiguousarray(kl_last); kl_cur = np.ascontiguousarray(kl_cur); b = np.ascontiguousarray(bounds4, np.float32)
        n1, n2 = len(kl_last), len(kl_cur)
        hm = None if last_has_mapline is None else np.ascontiguousarray(last_has_mapline, np.uint8)
        m = np.zeros(max(n1, 1), np.int32); acc = np.zeros(max(n1, 1), np.uint8); n = C.c_int(0)
        self._chk(lib().hvo_match_lines_geom(self.h, _p(d_last), _p(kl_last), None if hm is None else _p(hm), n1, _p(d_cur), _p(kl_cur), n2, desc_th, _p(b),
                                             _p(m), _p(acc), C.byref(n)), "match_lines_geom")
        return n.value, m[:n1], acc[:n1]

    def search_lines_by_projection(self, q_xyxy, q_kl, q_desc, q_blocks, t_kl, t_linefn, t_desc, t_occupied, cell_start, cell_items, bounds4, th):
        """LSDmatcher::SearchByProjection(Cur, Last, th) core (src/LSDmatcher.cpp:561-662) -> (nmatches, match_idx, match_dist)"""
        q_xyxy = np.ascontiguousarray(q_xyxy, np.float32).reshape(-1, 4); nq = len(q_xyxy)
        q_kl = np.ascontiguousarray(q_kl); t_kl = np.ascontiguousarray(t_kl); nt = len(t_kl)
        q_desc = np.ascontiguousarray(q_desc, np.uint8); t_desc = np.ascontiguousarray(t_desc, np.uint8)
        q_blocks = np.ascontiguousarray(q_blocks, np.uint8); t_occupied = np.ascontiguousarray(t_occupied, np.uint8)
        t_linefn = np.ascontiguousarray(t_linefn, np.float64); b = np.ascontiguousarray(bounds4, np.float32)
        cs = np.ascontiguousarray(cell_start, np.int32); ci = np.ascontiguousarray(cell_items, np.int32)
        if len(ci) == 0: ci = np.zeros(1, np.int32)
        mi = np.zeros(max(nq, 1), np.int32); md = np.zeros(max(nq, 1), np.int32); n = C.c_int(0)
        self._chk(lib().hvo_search_lines_by_projection(self.h, nq, _p(q_xyxy), _p(q_kl), _p(q_desc), _p(q_blocks), _p(t_kl), _p(t_linefn), _p(t_desc), _p(t_occupied), nt,
                                                       _p(cs), _p(ci), _p(b), th, _p(mi), _p(md), C.byref(n)), "search_lines_by_projection")
        return n.value, mi[:nq], md[:nq]

    def set_readings(self, blur_float=False, lsd_8u=False):
        """the alternative readings of cv::GaussianBlur / cv::LineSegmentDetector (include/hvo.h HVO_READING_*); the next extraction uses them"""
        lib().hvo_set_readings.argtypes = [C.c_void_p, C.c_uint]
        self._chk(lib().hvo_set_readings(self.h, (READING_BLUR_FLOAT if blur_float else 0) | (READING_LSD_8U if lsd_8u else 0)), "set_readings")

    def lsd_async_report(self):
        """(frames grown again by the one-wave kernel, workers that sat on a foreign XCD, workers per frame) of the last async line growing"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        lib().hvo_lsd_async_report.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self._chk(lib().hvo_lsd_async_report(self.h, C.byref(a), C.byref(b), C.byref(c)), "lsd_async_report")
        return a.value, b.value, c.value

    def search_by_projection_map(self, q_desc, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_blocks,
                                 t_kp, t_uright, t_occupied, t_desc, bounds, th_high=100, nn_ratio=0.8):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th) core (src/ORBmatcher.cc:45-132) -> (nmatches, match_idx, match_dist)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        q_desc = np.ascontiguousarray(q_desc, np.uint8); t_desc = np.ascontiguousarray(t_desc, np.uint8)
        nq, nt = len(q_desc), len(t_desc)
        q_u, q_v, q_radius, q_ur = map(f32, (q_u, q_v, q_radius, q_ur))
        q_min_level = np.ascontiguousarray(q_min_level, np.int32); q_max_level = np.ascontiguousarray(q_max_level, np.int32)
        q_blocks = np.ascontiguousarray(q_blocks, np.uint8); t_occupied = np.ascontiguousarray(t_occupied, np.uint8)
        t_kp = np.ascontiguousarray(t_kp); t_uright = f32(t_uright)
        mi = np.zeros(nq, np.int32); md = np.zeros(nq, np.int32); n = C.c_int(0)
        self._chk(lib().hvo_search_by_projection_map(self.h, _p(q_desc), nq, _p(q_u), _p(q_v), _p(q_radius), _p(q_min_level), _p(q_max_level), _p(q_ur),
                                                     _p(q_blocks), _p(t_kp), _p(t_uright), _p(t_occupied), _p(t_desc), nt,
                                                     bounds[0], bounds[1], bounds[2], bounds[3], th_high, nn_ratio, _p(mi), _p(md), C.byref(n)),
                  "search_by_projection_map")
        return n.value, mi, md

    def search_by_projection_tracked(self, q_desc, proj_x, proj_y, proj_xr, level, view_cos, q_blocks, th,
                                     t_kp, t_uright, t_occupied, t_desc, bounds, th_high=100, nn_ratio=0.8):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th) from mTrackProjX/Y/XR, mnTrackScaleLevel, mTrackViewCos: the window
        prologue (src/ORBmatcher.cc:55-70, 134-140) runs on the device -> (nmatches, match_idx, match_dist)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        q_desc = np.ascontiguousarray(q_desc, np.uint8); t_desc = np.ascontiguousarray(t_desc, np.uint8)
        nq, nt = len(q_desc), len(t_desc)
        proj_x, proj_y, view_cos = map(f32, (proj_x, proj_y, view_cos)); proj_xr = f32(proj_xr) if proj_xr is not None else None
        level = np.ascontiguousarray(level, np.int32)
        q_blocks = np.ascontiguousarray(q_blocks, np.uint8); t_occupied = np.ascontiguousarray(t_occupied, np.uint8)
        t_kp = np.ascontiguousarray(t_kp); t_uright = f32(t_uright)
        mi = np.zeros(nq, np.int32); md = np.zeros(nq, np.int32); n = C.c_int(0)
        fn = lib().hvo_search_by_projection_tracked
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 4 + [C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float] + [C.c_void_p] * 3
        self._chk(fn(self.h, _p(q_desc), nq, _p(proj_x), _p(proj_y), _p(proj_xr) if proj_xr is not None else None, _p(level), _p(view_cos), _p(q_blocks), float(th),
                     _p(t_kp), _p(t_uright), _p(t_occupied), _p(t_desc), nt, bounds[0], bounds[1], bounds[2], bounds[3], th_high, nn_ratio,
                     _p(mi), _p(md), C.byref(n)), "search_by_projection_tracked")
        return n.value, mi, md

    def stereo_from_rgbd(self, kp, kp_un, depth, bf):
        """Frame::ComputeStereoFromRGBD (src/Frame.cc:1940-1961) -> (mvuRight, mvDepth)"""
        kp = np.ascontiguousarray(kp); kp_un = np.ascontiguousarray(kp_un); depth = np.ascontiguousarray(depth, np.uint16)
        h, w = depth.shape
        ur = np.zeros(len(kp), np.float32); z = np.zeros(len(kp), np.float32)
        self._chk(lib().hvo_stereo_from_rgbd(self.h, _p(kp), _p(kp_un), len(kp), _p(depth), w, h, depth.strides[0], bf, _p(ur), _p(z)), "stereo_from_rgbd")
        return ur, z

    # ---- Frame post-processing ----
    def undistort_keypoints(self, kp, dist5):
        """Frame::UndistortKeyPoints (src/Frame.cc:1701-1731); dist5 = (k1, k2, p1, p2, k3)"""
        kp = np.ascontiguousarray(kp); out = np.zeros_like(kp); d = np.ascontiguousarray(dist5, np.float32)
        assert d.shape == (5,)
        self._chk(lib().hvo_undistort_keypoints(self.h, _p(kp), len(kp), _p(d), _p(out)), "undistort_keypoints")
        return out

    def image_bounds(self, w, h, dist5):
        """Frame::ComputeImageBounds (src/Frame.cc:1733-1762) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)"""
        d = np.ascontiguousarray(dist5, np.float32); b = np.zeros(4, np.float32)
        self._chk(lib().hvo_image_bounds(self.h, w, h, _p(d), _p(b)), "image_bounds")
        return b

    def assign_features_to_grid(self, kp_un, bounds4):
        """Frame::AssignFeaturesToGrid (src/Frame.cc:832-847) as CSR (cell = col*48 + row)"""
        kp_un = np.ascontiguousarray(kp_un); b = np.ascontiguousarray(bounds4, np.float32)
        start = np.zeros(64 * 48 + 1, np.int32); items = np.zeros(max(len(kp_un), 1), np.int32); n = C.c_int(0)
        self._chk(lib().hvo_assign_features_to_grid(self.h, _p(kp_un), len(kp_un), _p(b), _p(start), _p(items), C.byref(n)), "assign_features_to_grid")
        return start, items[: n.value]

    def assign_lines_to_grid(self, kl, bounds4, cap=None):
        """Frame::AssignFeaturesToGridForLine (src/Frame.cc:849-872) as CSR"""
        kl = np.ascontiguousarray(kl); b = np.ascontiguousarray(bounds4, np.float32)
        cap = cap if cap is not None else max(len(kl), 1) * 128
        start = np.zeros(64 * 48 + 1, np.int32); items = np.zeros(max(cap, 1), np.int32); n = C.c_int(0)
        self._chk(lib().hvo_assign_lines_to_grid(self.h, _p(kl), len(kl), _p(b), _p(start), _p(items), cap, C.byref(n)), "assign_lines_to_grid")
        return start, items[: n.value]

    # ---- batch ----
    def batch_upload(self, gray, depth=None, repeat=1):
        """gray: (B,H,W) u8; depth: (B,H,W) u16 or None.  repeat > 1 uploads the B frames cyclically
        B*repeat times (frames are passed by pointer, host memory is not replicated)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        B0, h, w = gray.shape
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.uint16)
        B = B0 * repeat
        fi = (FrameIn * B)()
        for b in range(B):
            s = b % B0
            fi[b].gray = gray[s].ctypes.data; fi[b].gray_stride = gray.strides[1]
            if depth is not None:
                fi[b].depth = depth[s].ctypes.data; fi[b].depth_stride = depth.strides[1]
        self._chk(lib().hvo_batch_upload(self.h, B, fi, w, h), "batch_upload")
        self._B, self._w, self._h = B, w, h

    def _frames_in(self, gray, depth, repeat):
        gray = np.ascontiguousarray(gray, np.uint8)
        B0, h, w = gray.shape
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.uint16)
        B = B0 * repeat
        fi = (FrameIn * B)()
        for b in range(B):
            s = b % B0
            fi[b].gray = gray[s].ctypes.data; fi[b].gray_stride = gray.strides[1]
            if depth is not None:
                fi[b].depth = depth[s].ctypes.data; fi[b].depth_stride = depth.strides[1]
        return fi, B, w, h, (gray, depth)

    def batch_stage_upload(self, gray, depth=None, repeat=1, frames_in=None):
        """the NEXT batch's images into the staging slabs, enqueued on a copy stream of its own (returns at once; the host arrays must
        stay alive and unchanged until batch_commit_staged).  frames_in: a tuple from a previous call (the FrameIn table is reused)."""
        fr = frames_in if frames_in is not None else self._frames_in(gray, depth, repeat)
        fi, B, w, h, _keep = fr
        self._chk(lib().hvo_batch_stage_upload(self.h, B, fi, w, h), "batch_stage_upload")
        self._staged = fr
        return fr

    def batch_commit_staged(self):
        """wait for the staged upload and make it the resident batch (device-to-device)"""
        self._chk(lib().hvo_batch_commit_staged(self.h), "batch_commit_staged")
        _, self._B, self._w, self._h, _ = self._staged

    def batch_results_async(self, n, host_slabs, labels=True):
        """pack the first n frames' results (hvo_batch_pack_results_ex) and start ONE copy of the slabs into `host_slabs` (a page-locked
        uint8 array of n * slab_bytes); batch_results_wait() waits for it.  The resident batch may run again at once."""
        fn = lib().hvo_batch_results_async
        fn.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p]
        self._chk(fn(self.h, n, SLAB_LABELS if labels else 0, host_slabs.ctypes.data), "batch_results_async")

    def batch_results_wait(self):
        self._chk(lib().hvo_batch_results_wait(self.h), "batch_results_wait")

    def batch_run(self, stages=STAGE_ALL):
        self._chk(lib().hvo_batch_run(self.h, stages), "batch_run")

    def set_tail_params(self, seed=1, plane_dist_th=0.05, vp_th_angle=0.0):
        """seed (frame f draws with seed + f), Plane.DistanceThreshold and line2Vps' angle for the tail stages of batch_run"""
        self._chk(lib().hvo_set_tail_params(self.h, seed, plane_dist_th, vp_th_angle), "set_tail_params")

    def batch_download_tail(self, stages, results):
        """results of the tail stages (STAGE_LINES3D | STAGE_VP | STAGE_PLANE_TAIL | STAGE_GRIDS) of the resident batch, merged into the
        per-frame dicts `results` of batch_download (their line / plane counts size the arrays)"""
        n = len(results)
        kc, lc, _, _ = self.slab_layout()
        bufs = []; arr = (FrameTail * n)()
        for f in range(n):
            b, t = _tail_buffers(kc, lc, self._w, self._h)
            bufs.append(b); arr[f] = t
        self._chk(lib().hvo_batch_download_tail(self.h, n, arr), "batch_download_tail")
        for f in range(n):
            results[f].update(_tail_result(bufs[f], arr[f], len(results[f].get("kl", ())), len(results[f].get("planes", ())), stages))
        return results

    def batch_download(self, stages=STAGE_ALL, pl_cap=64, n=None, reuse=False, labels8=False, pinned=False):
        """results of the first n (default all) frames of the resident batch.  reuse=True keeps the host result arrays of the
        previous call with the same shape (a caller that consumes the results before the next download avoids re-faulting
        ~1.4 MB of fresh pages per frame)."""
        B, w, h = self._B, self._w, self._h
        B = B if n is None else min(B, n)
        kcap = self.params.orb_nfeatures + 8 * self.params.orb_nlevels + 64
        lcap = max(self.params.lsd_nfeatures, 1)
        key = (B, w, h, stages, pl_cap, labels8)
        if reuse and getattr(self, "_dl_key", None) == key:
            fo, res_proto = self._dl_fo, self._dl_res
            self._chk(lib().hvo_batch_download(self.h, B, fo), "batch_download")
            return self._dl_finish(fo, [dict(r) for r in res_proto])
        fo = (FrameOut * B)()
        res = [dict() for _ in range(B)]
        # one slab per output kind for the whole batch (per-frame results are views): 8 allocations, not 8 per frame
        if stages & STAGE_ORB:
            kp = np.zeros((B, kcap), KEYPOINT_DT); desc = np.zeros((B, kcap, 32), np.uint8)
            for b in range(B):
                res[b]["kp"] = kp[b]; res[b]["desc"] = desc[b]
                fo[b].kp = kp[b].ctypes.data; fo[b].desc = desc[b].ctypes.data; fo[b].kp_cap = kcap
        if stages & (STAGE_LSD | STAGE_LSD_CULL):
            kl = np.zeros((B, lcap), KEYLINE_DT); ldesc = np.zeros((B, lcap, 32), np.uint8); linefn = np.zeros((B, lcap, 3))
            for b in range(B):
                res[b]["kl"] = kl[b]; res[b]["ldesc"] = ldesc[b]; res[b]["linefn"] = linefn[b]
                fo[b].kl = kl[b].ctypes.data; fo[b].ldesc = ldesc[b].ctypes.data; fo[b].linefn = linefn[b].ctypes.data
                fo[b].kl_cap = lcap
        if stages & STAGE_PLANES:
            # labels are always written in full; labels8=True: as int8, the way they cross PCIe (no widening to CV_32S)
            labels = np.empty((B, h, w), np.int8 if labels8 else np.int32); planes = np.zeros((B, pl_cap), PLANE_DT)
            if pinned and reuse:
                pin(labels); self._pinned = getattr(self, "_pinned", []) + [labels]
            for b in range(B):
                res[b]["labels"] = labels[b]; res[b]["planes"] = planes[b]
                if labels8: fo[b].labels8 = labels[b].ctypes.data
                else: fo[b].labels = labels[b].ctypes.data
                fo[b].planes = planes[b].ctypes.data; fo[b].pl_cap = pl_cap
        if reuse:
            self._dl_key, self._dl_fo, self._dl_res = key, fo, [dict(r) for r in res]
        self._chk(lib().hvo_batch_download(self.h, B, fo), "batch_download")
        return self._dl_finish(fo, res)

    @staticmethod
    def _dl_finish(fo, res):
        for b, r in enumerate(res):
            r["status"] = fo[b].status
            if "kp" in r:
                r["kp"] = r["kp"][: fo[b].n_kp]; r["desc"] = r["desc"][: fo[b].n_kp]
            if "kl" in r:
                r["kl"] = r["kl"][: fo[b].n_kl]; r["ldesc"] = r["ldesc"][: fo[b].n_kl]; r["linefn"] = r["linefn"][: fo[b].n_kl]
            if "planes" in r:
                r["planes"] = r["planes"][: fo[b].n_planes]
        return res

    def slab_layout(self, labels=False):
        """(kp_cap, kl_cap, pl_cap, slab_bytes) of the resident batch's device result slabs; labels=True: with the int8 label image at
        the end of every slab (HVO_SLAB_LABELS) -> (kp_cap, kl_cap, pl_cap, slab_bytes, labels_off)"""
        a, b, c, d, e = C.c_int(0), C.c_int(0), C.c_int(0), C.c_size_t(0), C.c_size_t(0)
        fn = lib().hvo_batch_slab_layout_ex
        fn.argtypes = [C.c_void_p, C.c_uint] + [C.c_void_p] * 5
        self._chk(fn(self.h, SLAB_LABELS if labels else 0, C.byref(a), C.byref(b), C.byref(c), C.byref(e), C.byref(d)), "batch_slab_layout")
        return (a.value, b.value, c.value, d.value, e.value) if labels else (a.value, b.value, c.value, d.value)

    def pack_results(self, n, device_ptr, labels=False):
        """write the first n frames' result slabs to device memory at `device_ptr` (n * slab_bytes bytes)"""
        fn = lib().hvo_batch_pack_results_ex
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint]
        self._chk(fn(self.h, n, C.c_void_p(device_ptr), SLAB_LABELS if labels else 0), "batch_pack_results")

    def profile_enable(self, mode=1):
        """0 off, 1 hipEvents around each kernel group, 2 events + stages serialised on one stream"""
        self._chk(lib().hvo_profile_enable(self.h, int(mode)), "profile_enable")

    def profile_last(self):
        names = (C.c_char_p * 32)(); ms = (C.c_float * 32)()
        n = lib().hvo_profile_last(self.h, names, ms, 32)
        return {names[i].decode(): ms[i] for i in range(n)}


class BatchPipeline:
    """Consecutive batches with upload, run and download overlapped: `nctx` contexts on `nctx` host threads, each looping
    upload -> run -> download over its own resident batch, with one lock per stage so that only one context at a time
    uploads (the link), runs (the GPU) or downloads -- the three stages of different batches overlap, a stage never competes
    with itself.  (Different contexts may be driven from different threads, include/hvo.h.)  Measured on MI355X with pinned
    host buffers and int8 labels: 3 x 2048 frames sustain 72 % of the resident-batch rate, unlocked threads 50 %."""

    def __init__(self, nctx=3, batch=2048, stages=STAGE_ALL, make_context=None, **ctx_kw):
        import threading
        self.stages = stages; self.batch = batch
        self.ctxs = [make_context(batch) if make_context else Context(max_batch=batch, **ctx_kw) for _ in range(nctx)]
        self.locks = [threading.Lock() for _ in range(3)]

    def run(self, gray, depth, repeat=1, rounds=1, on_result=None, pinned=True):
        """every context processes `rounds` batches of gray/depth (cyclic `repeat`); on_result(ctx_index, results) is called
        with each downloaded batch (result arrays are reused by the next batch of that context).  Returns frames processed."""
        import threading
        def loop(i):
            c = self.ctxs[i]
            for _ in range(rounds):
                with self.locks[0]:
                    c.batch_upload(gray, depth, repeat=repeat)
                with self.locks[1]:
                    c.batch_run(self.stages)
                with self.locks[2]:
                    res = c.batch_download(self.stages, reuse=True, labels8=True, pinned=pinned)
                if on_result:
                    on_result(i, res)
        thr = [threading.Thread(target=loop, args=(i,)) for i in range(len(self.ctxs))]
        for t in thr: t.start()
        for t in thr: t.join()
        return len(self.ctxs) * rounds * len(gray) * repeat

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []


class Stream:
    """hvo_stream: the streamed-sequence mode (one Frame construction per camera image, src/Tracking.cc:262, with `depth`
    frames in flight and the last `depth` frames' results resident in HBM for frame-to-frame matching)."""

    def __init__(self, width=640, height=480, depth=4, stages=STAGE_ALL, dist5=(0, 0, 0, 0, 0), bf=40.0, params=None, seed=1, plane_dist_th=0.05, vp_th_angle=0.0, **kw):
        self.params = params if params is not None else default_params(**kw)
        sp = StreamParams(); sp.width = width; sp.height = height; sp.depth = depth; sp.stages = stages; sp.bf = bf
        sp.seed = seed; sp.plane_dist_th = plane_dist_th; sp.vp_th_angle = vp_th_angle; self.seed = seed
        for k in range(5):
            sp.dist5[k] = dist5[k]
        h = C.c_void_p()
        rc = lib().hvo_stream_create(C.byref(self.params), C.byref(sp), C.byref(h))
        if rc != HVO_OK:
            raise HvoError(rc, "hvo_stream_create")
        self.h = h; self.w = width; self.hgt = height; self.stages = stages; self.depth = depth
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        lib().hvo_stream_capacity(self.h, C.byref(a), C.byref(b), C.byref(c))
        self.kp_cap, self.kl_cap, self.pl_cap = a.value, b.value, c.value
        bb = np.zeros(4, np.float32); lib().hvo_stream_image_bounds(self.h, _p(bb)); self.bounds = bb      # mnMinX, mnMaxX, mnMinY, mnMaxY

    def close(self):
        if getattr(self, "h", None):
            lib().hvo_stream_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != HVO_OK:
            raise HvoError(rc, what + ": " + lib().hvo_stream_last_error(self.h).decode())

    def submit(self, gray, depth=None):
        assert gray.dtype == np.uint8 and gray.shape == (self.hgt, self.w) and gray.strides[1] == 1
        t = C.c_int64(-1)
        if depth is not None:
            assert depth.dtype == np.uint16 and depth.shape == (self.hgt, self.w) and depth.strides[1] == 2
            rc = lib().hvo_stream_submit(self.h, gray.ctypes.data, gray.strides[0], depth.ctypes.data, depth.strides[0], C.byref(t))
        else:
            rc = lib().hvo_stream_submit(self.h, gray.ctypes.data, gray.strides[0], None, 0, C.byref(t))
        self._chk(rc, "stream_submit")
        return t.value

    def poll(self, ticket):
        r = lib().hvo_stream_poll(self.h, ticket)
        if r < 0:
            raise HvoError(r, "stream_poll")
        return r == 1

    def collect(self, ticket, labels=True):
        fo = FrameOut(); r = {}
        tail_stages = self.stages & (STAGE_LINES3D | STAGE_VP | STAGE_PLANE_TAIL | STAGE_GRIDS)
        if tail_stages:                                   # the tail block is read before hvo_stream_collect releases the slot
            tb, tt = _tail_buffers(self.kp_cap, self.kl_cap, self.w, self.hgt)
            self._chk(lib().hvo_stream_collect_tail(self.h, ticket, C.byref(tt)), "stream_collect_tail")
        if self.stages & STAGE_ORB:
            kp = np.zeros(self.kp_cap, KEYPOINT_DT); desc = np.zeros((self.kp_cap, 32), np.uint8); kpu = np.zeros(self.kp_cap, KEYPOINT_DT)
            ur = np.zeros(self.kp_cap, np.float32); zd = np.zeros(self.kp_cap, np.float32)
            fo.kp = kp.ctypes.data; fo.desc = desc.ctypes.data; fo.kp_cap = self.kp_cap
        if self.stages & (STAGE_LSD | STAGE_LSD_CULL):
            kl = np.zeros(self.kl_cap, KEYLINE_DT); ldesc = np.zeros((self.kl_cap, 32), np.uint8); fn = np.zeros((self.kl_cap, 3))
            fo.kl = kl.ctypes.data; fo.ldesc = ldesc.ctypes.data; fo.linefn = fn.ctypes.data; fo.kl_cap = self.kl_cap
        if self.stages & STAGE_PLANES:
            planes = np.zeros(self.pl_cap, PLANE_DT); fo.planes = planes.ctypes.data; fo.pl_cap = self.pl_cap
            if labels:
                lab = np.empty((self.hgt, self.w), np.int32); fo.labels = lab.ctypes.data
        if self.stages & STAGE_ORB:
            self._chk(lib().hvo_stream_collect(self.h, ticket, C.byref(fo), _p(kpu), _p(ur), _p(zd)), "stream_collect")
            n = fo.n_kp
            r.update(kp=kp[:n], desc=desc[:n], kp_un=kpu[:n], uright=ur[:n], zdepth=zd[:n])
        else:
            self._chk(lib().hvo_stream_collect(self.h, ticket, C.byref(fo), None, None, None), "stream_collect")
        if self.stages & (STAGE_LSD | STAGE_LSD_CULL):
            n = fo.n_kl
            r.update(kl=kl[:n], ldesc=ldesc[:n], linefn=fn[:n])
        if self.stages & STAGE_PLANES:
            r["planes"] = planes[: fo.n_planes]
            if labels:
                r["labels"] = lab
        r["status"] = fo.status
        if tail_stages:
            r.update(_tail_result(tb, tt, fo.n_kl, fo.n_planes, tail_stages))
        return r

    def stage_ms(self, ticket):
        ms = np.zeros(3, np.float32)
        self._chk(lib().hvo_stream_stage_ms(self.h, ticket, _p(ms)), "stream_stage_ms")
        return {"orb": float(ms[0]), "lsd": float(ms[1]), "planes": float(ms[2])}

    def project_last(self, cur, last, cam, Tcw, Tlw, q_index, x3Dw, q_blocks, th, mono=False, t_occupied=None, q_desc=None,
                     th_high=100, check_orientation=True, want_uv=False):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) whole between two resident frames: projection prologue
        (src/ORBmatcher.cc:1364-1405) + search core on the device.  cam = (fx, fy, cx, cy, mbf, mb); Tcw / Tlw: 3 x 4 row-major.
        -> (nmatches, idx, dist[, uv])"""
        class Cam(C.Structure):
            _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "bf", "b")]
        q_index = np.ascontiguousarray(q_index, np.int32); nq = len(q_index)
        x3Dw = np.ascontiguousarray(x3Dw, np.float32).reshape(-1, 3); assert len(x3Dw) == nq
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(12); Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(12)
        q_blocks = np.ascontiguousarray(q_blocks, np.uint8)
        t_occupied = np.ascontiguousarray(t_occupied, np.uint8) if t_occupied is not None else None
        q_desc = np.ascontiguousarray(q_desc, np.uint8) if q_desc is not None else None
        mi = np.zeros(max(nq, 1), np.int32); md = np.zeros(max(nq, 1), np.int32); n = C.c_int(0)
        uv = np.zeros((max(nq, 1), 2), np.float32) if want_uv else None
        pp = lambda a: _p(a) if a is not None else None
        c = Cam(*[float(v) for v in cam])
        fn = lib().hvo_stream_project_last
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
        self._chk(fn(self.h, cur, last, C.byref(c), _p(Tcw), _p(Tlw), nq, _p(q_index), _p(x3Dw), _p(q_blocks), pp(q_desc), pp(t_occupied),
                     float(th), 1 if mono else 0, th_high, 1 if check_orientation else 0, _p(mi), _p(md), C.byref(n), pp(uv)), "stream_project_last")
        return (n.value, mi[:nq], md[:nq], uv[:nq]) if want_uv else (n.value, mi[:nq], md[:nq])

    def search_by_projection(self, cur, last, q_index, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_blocks,
                             t_occupied=None, q_desc=None, th_high=100, check_orientation=True):
        """ORBmatcher::SearchByProjection(Cur, Last) core between two resident frames -> (nmatches, idx, dist)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        q_index = np.ascontiguousarray(q_index, np.int32); nq = len(q_index)
        q_u, q_v, q_radius = map(f32, (q_u, q_v, q_radius))
        q_ur = f32(q_ur) if q_ur is not None else None
        q_min_level = np.ascontiguousarray(q_min_level, np.int32); q_max_level = np.ascontiguousarray(q_max_level, np.int32)
        q_blocks = np.ascontiguousarray(q_blocks, np.uint8)
        t_occupied = np.ascontiguousarray(t_occupied, np.uint8) if t_occupied is not None else None
        q_desc = np.ascontiguousarray(q_desc, np.uint8) if q_desc is not None else None
        mi = np.zeros(max(nq, 1), np.int32); md = np.zeros(max(nq, 1), np.int32); n = C.c_int(0)
        pp = lambda a: _p(a) if a is not None else None
        self._chk(lib().hvo_stream_search_by_projection(self.h, cur, last, nq, _p(q_index), pp(q_desc), _p(q_u), _p(q_v), _p(q_radius), _p(q_min_level),
                                                        _p(q_max_level), pp(q_ur), _p(q_blocks), pp(t_occupied), th_high, 1 if check_orientation else 0,
                                                        _p(mi), _p(md), C.byref(n)), "stream_search_by_projection")
        return n.value, mi[:nq], md[:nq]

    def set_readings(self, blur_float=False, lsd_8u=False):
        lib().hvo_stream_set_readings.argtypes = [C.c_void_p, C.c_uint]
        self._chk(lib().hvo_stream_set_readings(self.h, (READING_BLUR_FLOAT if blur_float else 0) | (READING_LSD_8U if lsd_8u else 0)), "stream_set_readings")

    def match_lines_geom(self, cur, last, desc_th=0.9, last_has_mapline=None):
        """LSDmatcher::SearchByGeomNApearance(Cur, Last) between two resident frames -> (lmatches, matches12, accepted)"""
        m = np.full(self.kl_cap, -1, np.int32); acc = np.zeros(self.kl_cap, np.uint8); n1 = C.c_int(0); n = C.c_int(0)
        hm = None if last_has_mapline is None else np.ascontiguousarray(last_has_mapline, np.uint8)
        self._chk(lib().hvo_stream_match_lines_geom(self.h, cur, last, desc_th, None if hm is None else _p(hm), _p(m), _p(acc), C.byref(n1), C.byref(n)), "stream_match_lines_geom")
        return n.value, m[: n1.value], acc[: n1.value]

    def search_lines_by_projection(self, cur, last, q_index, q_xyxy, th, q_blocks=None, t_occupied=None, q_desc=None):
        """LSDmatcher::SearchByProjection(Cur, Last, th) core between two resident frames -> (nmatches, match_idx, match_dist)"""
        q_index = np.ascontiguousarray(q_index, np.int32); nq = len(q_index)
        q_xyxy = np.ascontiguousarray(q_xyxy, np.float32).reshape(-1, 4)
        pp = lambda a, t: _p(np.ascontiguousarray(a, t)) if a is not None else None
        keep = [np.ascontiguousarray(a, t) if a is not None else None for a, t in ((q_desc, np.uint8), (q_blocks, np.uint8), (t_occupied, np.uint8))]
        mi = np.zeros(max(nq, 1), np.int32); md = np.zeros(max(nq, 1), np.int32); n = C.c_int(0)
        self._chk(lib().hvo_stream_search_lines_by_projection(self.h, cur, last, nq, _p(q_index), _p(q_xyxy), None if keep[0] is None else _p(keep[0]),
                                                              None if keep[1] is None else _p(keep[1]), None if keep[2] is None else _p(keep[2]), th,
                                                              _p(mi), _p(md), C.byref(n)), "stream_search_lines_by_projection")
        return n.value, mi[:nq], md[:nq]

    def match_lines(self, frm, to, mode=LINE_MATCH_NNR, th=50.0, nnratio=0.95):
        m = np.full(self.kl_cap, -1, np.int32); n1 = C.c_int(0); n = C.c_int(0)
        self._chk(lib().hvo_stream_match_lines(self.h, frm, to, mode, th, nnratio, _p(m), C.byref(n1), C.byref(n)), "stream_match_lines")
        return n.value, m[: n1.value]


# ---------------------------------------------------------------------------------------
# host-side mirrors of the reference's operator interfaces
# ---------------------------------------------------------------------------------------
class ORBextractor:
    """ORB_SLAM2::ORBextractor (include/ORBextractor.h:53-61).  operator()(image) -> (keypoints, descriptors)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0):
        self.ctx = Context(orb_nfeatures=nfeatures, orb_scale_factor=scaleFactor, orb_nlevels=nlevels,
                           orb_ini_th_fast=iniThFAST, orb_min_th_fast=minThFAST, device=device)

    def __call__(self, image, mask=None):
        return self.ctx.extract_orb(image)     # mask is ignored, as in the reference (ORBextractor.h:58)


class LINEextractor:
    """ORB_SLAM2::LINEextractor (include/LineExtractor.h:186-193)."""

    def __init__(self, numOctaves=1, scale=1.2, nLSDFeature=200, min_line_length=0, device=0):
        self.ctx = Context(lsd_num_octaves=numOctaves, lsd_scale=scale, lsd_nfeatures=nLSDFeature, device=device)

    def __call__(self, image, mask=None):
        return self.ctx.extract_lsd(image)


class PlaneDetection:
    """PlaneDetection (include/PlaneExtractor.h:36-56): readDepthImage + runPlaneDetection."""

    def __init__(self, K=None, depthMapFactor=1.0 / 5000.0, device=0):
        kw = dict(depth_map_factor=depthMapFactor, device=device)
        if K is not None:
            kw.update(fx=K[0][0], fy=K[1][1], cx=K[0][2], cy=K[1][2])
        self.ctx = Context(**kw)

    def run(self, depth_u16):
        return self.ctx.compute_planes(depth_u16)


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30      # src/ORBmatcher.cc:37-39

    def __init__(self, ctx=None):
        self.ctx = ctx or Context()

    def DescriptorDistance(self, a, b):
        return int(self.ctx.hamming_matrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32))[0, 0])


class LSDmatcher:
    TH_HIGH, TH_LOW = 80, 50                          # src/LSDmatcher.cpp:12-14

    def __init__(self, ctx=None):
        self.ctx = ctx or Context()

    def match(self, desc1, desc2, nnr):
        """LSDmatcher::match -> matchNNR (src/LSDmatcher.cpp:828-863, 803-826): (n, matches_12)"""
        return self.ctx.match_nnr(desc1, desc2, nnr)
