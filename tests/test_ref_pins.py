"""The oracle against the pieces of the REFERENCE that compile from their own files (oracle/_ref/libhvoref.so, built by
`make -C oracle ref` = __graft_entry__.build() when /root/reference is present; the prebuilt file travels to the GPU box):

    include/peac/DisjointSet.hpp   union by size / Find / getSetSize            -> oracle/peac.c ds_union, ds_find
    include/peac/AHCParamSet.hpp   T_mse, T_ang, T_dz with the default ParamSet -> oracle/peac.c T_mse_*, T_ang_init, T_dz
    src/lineIterator.cpp           ORB_SLAM2::LineIterator                      -> oracle/frame.c line_cells

These are the only parts of the hot path that the reference itself can vouch for here (everything else needs
OpenCV 3.2 / Eigen / PCL); they shrink what is ASSUMED, they do not turn "parity unpinned" green."""
import ctypes as C
import os
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libhvoref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libhvoref.so not built (needs /root/reference at build time)")
    L = C.CDLL(REF)
    L.ref_ds_create.restype = C.c_void_p; L.ref_ds_create.argtypes = [C.c_int]
    L.ref_ds_union.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ref_ds_find.argtypes = [C.c_void_p, C.c_int]; L.ref_ds_set_size.argtypes = [C.c_void_p, C.c_int]
    L.ref_ds_free.argtypes = [C.c_void_p]; L.ref_ds_free.restype = None
    L.ref_T_mse.restype = C.c_double; L.ref_T_mse.argtypes = [C.c_int, C.c_double]
    L.ref_T_ang.restype = C.c_double; L.ref_T_ang.argtypes = [C.c_int, C.c_double]
    L.ref_T_dz.restype = C.c_double; L.ref_T_dz.argtypes = [C.c_double]
    L.ref_line_iterator.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_void_p, C.c_int]
    return L


@pytest.fixture(scope="module")
def ol(orc):
    L = orc.lib()
    L.orc_ds_create.restype = C.c_void_p; L.orc_ds_create.argtypes = [C.c_int]
    L.orc_ds_union.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_ds_find.argtypes = [C.c_void_p, C.c_int]; L.orc_ds_set_size.argtypes = [C.c_void_p, C.c_int]
    L.orc_ds_free.argtypes = [C.c_void_p]; L.orc_ds_free.restype = None
    for f in ("orc_peac_T_mse_init", "orc_peac_T_mse_merge", "orc_peac_T_ang_init", "orc_peac_T_dz"):
        getattr(L, f).restype = C.c_double; getattr(L, f).argtypes = [C.c_double]
    L.orc_grid_line_cells.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_void_p, C.c_int]
    return L


def test_disjoint_set_matches_reference(ref, ol):
    """random union sequences: the returned root of every Union, every Find and every set size agree (union by
    size with the reference's tie rule: equal sizes -> x's root survives, DisjointSet.hpp:73-82)"""
    rng = np.random.default_rng(7)
    for n in (1, 2, 17, 3072):
        a, b = ref.ref_ds_create(n), ol.orc_ds_create(n)
        try:
            for _ in range(4 * n):
                x, y = int(rng.integers(n)), int(rng.integers(n))
                assert ref.ref_ds_union(a, x, y) == ol.orc_ds_union(b, x, y)
                q = int(rng.integers(n))
                assert ref.ref_ds_find(a, q) == ol.orc_ds_find(b, q)
                assert ref.ref_ds_set_size(a, q) == ol.orc_ds_set_size(b, q)
            for q in range(n):
                assert ref.ref_ds_find(a, q) == ol.orc_ds_find(b, q)
        finally:
            ref.ref_ds_free(a); ol.orc_ds_free(b)


def test_paramset_thresholds_match_reference(ref, ol):
    """T_mse (P_INIT / P_MERGING), T_ang(P_INIT) and T_dz over metre-scale and millimetre-scale depths, bit for bit"""
    zs = np.concatenate([np.linspace(0.0, 13.2, 331), np.linspace(0.0, 70000.0, 701), [499.999, 500.0, 500.001, 3999.9, 4000.0, 4000.1]])
    for z in zs:
        z = float(z)
        assert ref.ref_T_mse(0, z) == ol.orc_peac_T_mse_init(z)
        assert ref.ref_T_mse(1, z) == ol.orc_peac_T_mse_merge(z)
        assert ref.ref_T_mse(2, z) == ol.orc_peac_T_mse_merge(z)
        assert ref.ref_T_ang(0, z) == ol.orc_peac_T_ang_init(z)
        assert ref.ref_T_dz(z) == ol.orc_peac_T_dz(z)
        assert ref.ref_T_dz(-z) == ol.orc_peac_T_dz(-z)
    assert ref.ref_T_ang(1, 1.0) == np.cos(np.deg2rad(60.0)) and ref.ref_T_ang(2, 1.0) == np.cos(np.deg2rad(30.0))


def test_grid_line_iterator_matches_reference(ref, ol):
    """ORB_SLAM2::LineIterator: same cells in the same order, for random segments in grid coordinates (64 x 48),
    including steep / reversed / degenerate ones and end points outside the grid"""
    rng = np.random.default_rng(11)
    cap = 512
    ax, ay, bx, by = (np.zeros(cap, np.int32) for _ in range(4))
    segs = [(0, 0, 0, 0), (3.5, 2.5, 3.5, 2.5), (0, 0, 63.9, 47.9), (63.9, 0, 0, 47.9), (10, 40, 10, 2), (5.2, 7.9, 5.9, 30.1)]
    segs += [tuple(rng.uniform(-4, 70, 4)) for _ in range(2000)]
    for s in segs:
        s = [float(v) for v in s]
        n1 = ref.ref_line_iterator(*s, ax.ctypes.data, ay.ctypes.data, cap)
        n2 = ol.orc_grid_line_cells(*s, bx.ctypes.data, by.ctypes.data, cap)
        assert n1 == n2 and n1 <= cap
        assert np.array_equal(ax[:n1], bx[:n1]) and np.array_equal(ay[:n1], by[:n1])
