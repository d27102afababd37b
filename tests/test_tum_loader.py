"""HVO_TUM_DIR loader (BASELINE.json configs[4] on real data where it exists): the PNG decoder, the grey conversion of
Tracking::GrabImageRGBD_wh (reference src/Tracking.cc:240-252) and the association file (Examples/RGB-D/associations/fr1_desk.txt format)."""
import importlib
import os
import struct
import zlib

import numpy as np

from conftest import load_pkg


def _png(arr, filt):
    """a PNG of arr (uint8 [h,w,3] or uint16 [h,w]) with every row encoded by filter `filt` (0..4)"""
    h, w = arr.shape[:2]
    if arr.dtype == np.uint16:
        raw = np.stack([(arr >> 8).astype(np.uint8), (arr & 255).astype(np.uint8)], -1).reshape(h, -1); depth, ctype, bpp = 16, 0, 2
    else:
        raw = arr.reshape(h, -1); depth, ctype, bpp = 8, 2, 3
    rows = []; prev = np.zeros(raw.shape[1], np.int32)
    for y in range(h):
        cur = raw[y].astype(np.int32); enc = np.zeros_like(cur)
        for x in range(len(cur)):
            a = cur[x - bpp] if x >= bpp else 0; b = prev[x]; c = prev[x - bpp] if x >= bpp else 0
            if filt == 0: p = 0
            elif filt == 1: p = a
            elif filt == 2: p = b
            elif filt == 3: p = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            enc[x] = (cur[x] - p) & 255
        rows.append(bytes([filt]) + enc.astype(np.uint8).tobytes()); prev = cur
    def chunk(t, b): return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b"")


def test_png_decoder_all_filters_and_sequence(tmp_path):
    load_pkg()
    tum = importlib.import_module("hvo_amd.tum")
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (12, 17, 3), dtype=np.uint8)
    dep = rng.integers(0, 65536, (12, 17), dtype=np.uint16)
    for f in range(5):
        assert np.array_equal(tum.decode_png(_png(rgb, f)), rgb), f
        assert np.array_equal(tum.decode_png(_png(dep, f)), dep), f
    # grey: the reference applies CV_RGB2GRAY to imread's BGR data (Camera.RGB: 1): 0.299 on B, 0.114 on R, OpenCV's 14-bit fixed point
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)              # R, G, B order of the PNG
    assert tum.to_gray(px).tolist() == [[(255 * 1868 + 8192) >> 14, (255 * 9617 + 8192) >> 14, (255 * 4899 + 8192) >> 14, (30 * 4899 + 20 * 9617 + 10 * 1868 + 8192) >> 14]]
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines = []
    for k in range(3):
        (tmp_path / "rgb" / ("%d.png" % k)).write_bytes(_png(np.roll(rgb, k, 1), k))
        (tmp_path / "depth" / ("%d.png" % k)).write_bytes(_png(np.roll(dep, k, 1), 4 - k))
        lines.append("%d.0 rgb/%d.png %d.01 depth/%d.png" % (k, k, k, k))
    (tmp_path / "associations.txt").write_text("\n".join(lines) + "\n")
    g, d = tum.load_sequence(str(tmp_path))
    assert g.shape == (3, 12, 17) and g.dtype == np.uint8 and d.shape == (3, 12, 17) and d.dtype == np.uint16
    for k in range(3):
        assert np.array_equal(d[k], np.roll(dep, k, 1)) and np.array_equal(g[k], tum.to_gray(np.roll(rgb, k, 1)))
    # without an association file: rgb.txt + depth.txt matched by time stamp
    os.remove(tmp_path / "associations.txt")
    (tmp_path / "rgb.txt").write_text("# t file\n" + "".join("%d.0 rgb/%d.png\n" % (k, k) for k in range(3)))
    (tmp_path / "depth.txt").write_text("".join("%d.015 depth/%d.png\n" % (k, k) for k in range(3)))
    g2, d2 = tum.load_sequence(str(tmp_path), limit=2)
    assert len(g2) == 2 and np.array_equal(d2[1], d[1])


def test_png_photograph_filters_mixed_rows_and_alpha(tmp_path):
    """every row its own filter (what an encoder does on a photograph), a frame-sized image, grey + alpha, and time-stamp association"""
    load_pkg()
    tum = importlib.import_module("hvo_amd.tum")
    rng = np.random.default_rng(7)
    rgb = rng.integers(0, 256, (9, 23, 3), dtype=np.uint8)
    # rows with filters 0..4 in turn: built row by row with the reference encoder of this file
    h, w = rgb.shape[:2]
    rows = b""
    import zlib as _z
    for y in range(h):
        one = _png(rgb[:y + 1], y % 5)                     # encode rows 0..y with filter y % 5, take the last row's bytes
        body = _z.decompress(one[one.index(b"IDAT") + 4: one.index(b"IEND") - 8])
        rows += body[-(w * 3 + 1):]
    def chunk(t, b): return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b"")
    assert np.array_equal(tum.decode_png(png), rgb)
    ga = rng.integers(0, 256, (5, 7, 2), dtype=np.uint8)
    raw = b"".join(b"\x00" + ga[y].tobytes() for y in range(5))
    png4 = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 7, 5, 8, 4, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    assert np.array_equal(tum.to_gray(tum.decode_png(png4)), ga[..., 0])
    # associate(): the closest pair wins even when an earlier colour stamp is nearer to the same depth stamp than its own
    (tmp_path / "rgb.txt").write_text("# t file\n1.000 rgb/a.png\n1.030 rgb/b.png\n1.060 rgb/c.png\n")
    (tmp_path / "depth.txt").write_text("1.019 depth/x.png\n1.061 depth/y.png\n")
    assert tum.associate(str(tmp_path)) == [("rgb/b.png", "depth/x.png"), ("rgb/c.png", "depth/y.png")]
