"""CPU-side checks of the C-ABI library: it is built, loads, and exports every symbol that
include/hvo.h declares.  No compute call is made here (no GPU in this container)."""
import ctypes
import os
import re

from conftest import ROOT, PKG_DIR


def _declared():
    hdr = open(os.path.join(ROOT, "include", "hvo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hvo_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_the_boundary():
    names = _declared()
    for n in ("hvo_extract_orb", "hvo_extract_lsd", "hvo_compute_planes", "hvo_hamming_knn2",
              "hvo_hamming_matrix", "hvo_extract_batch", "hvo_create", "hvo_destroy"):
        assert n in names


def test_library_exports_every_declared_symbol(hvo):
    path = os.path.join(PKG_DIR, "csrc", "libhvo.so")
    assert os.path.exists(path), "libhvo.so not built: run python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(hvo.EXPORTS) == _declared()
    lib.hvo_abi_version.restype = ctypes.c_int
    assert lib.hvo_abi_version() == 3


def test_default_params_match_tum3(hvo):
    p = hvo.default_params()
    assert (p.orb_nfeatures, p.orb_nlevels, p.orb_ini_th_fast, p.orb_min_th_fast) == (1000, 8, 20, 7)
    assert abs(p.orb_scale_factor - 1.2) < 1e-7 and p.lsd_nfeatures == 200 and p.lsd_num_octaves == 1
    assert abs(p.fx - 535.4) < 1e-4 and abs(p.cy - 247.6) < 1e-4
    assert abs(p.depth_map_factor - 1.0 / 5000.0) < 1e-10


def test_struct_layouts(hvo):
    assert hvo.KEYPOINT_DT.itemsize == 28      # cv::KeyPoint
    assert hvo.KEYLINE_DT.itemsize == 68       # cv::line_descriptor::KeyLine
    assert ctypes.sizeof(hvo.Params) == 15 * 4


def test_strerror(hvo):
    L = hvo.lib()
    assert L.hvo_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.hvo_strerror(-2)


def test_no_device_fails_loudly(hvo):
    """without a GPU hvo_create must fail -- there is no CPU fallback in the product path"""
    import torch
    if torch.cuda.is_available():
        return
    try:
        hvo.Context()
    except hvo.HvoError as e:
        assert e.status in (-2, -3)
    else:
        raise AssertionError("hvo_create succeeded without a GPU")


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle.py" not in txt and "import oracle" not in txt, f
