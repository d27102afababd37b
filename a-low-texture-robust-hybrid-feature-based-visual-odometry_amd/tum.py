"""TUM RGB-D sequences for the streamed mode (BASELINE.json configs[4]: "TUM fr1_desk full sequence streamed").

`load_sequence(dir)` reads what the reference's RGB-D example reads (Examples/RGB-D/rgbd_tum.cc:47-96: an association file with
`t_rgb rgb/xxx.png t_depth depth/xxx.png` per line, colour image converted to grey in Tracking::GrabImageRGBD_wh, src/Tracking.cc:235-252,
depth kept as the raw 16-bit image).  No OpenCV / PIL in this image: the PNG decoder below is zlib + the five PNG filters, enough for the
dataset's files (8-bit RGB / grey, 16-bit grey, non-interlaced).  There is no dataset in the container or on the GPU box -- bench.py uses
this only when HVO_TUM_DIR points at one, and falls back to the synthetic sequence otherwise.
"""
import os
import struct
import zlib

import numpy as np


def decode_png(data):
    """PNG bytes -> uint8 (h, w[, c]) or uint16 (h, w) array.  Non-interlaced, bit depth 8 or 16, colour types 0 / 2 / 4 / 6."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos = 8; idat = []; w = h = depth = ctype = None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, inter = struct.unpack(">IIBBBBB", body)
            if inter != 0 or depth not in (8, 16) or ctype not in (0, 2, 4, 6):
                raise ValueError("unsupported PNG (interlace %d, depth %d, colour type %d)" % (inter, depth, ctype))
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = _unfilter(raw, h, w, bpp)
    if depth == 16:
        arr = out.reshape(h, w, ch, 2)
        arr = (arr[..., 0].astype(np.uint16) << 8) | arr[..., 1]
    else:
        arr = out.reshape(h, w, ch)
    return arr[:, :, 0] if ch == 1 else arr


def _unfilter(raw, h, w, bpp):
    """PNG filter reconstruction.  None / Sub / Up are running sums; Average and Paeth (what encoders pick for photographs) need the
    left, upper and upper-left pixels: every pixel of an anti-diagonal x + y = const has them on earlier diagonals, so the image is
    reconstructed diagonal by diagonal, w + h - 1 vectorised steps over all rows at once instead of w * h * bpp Python iterations
    (a 640x480 RGB frame: ~0.1 s instead of ~3 s; the 573 frames of fr1_desk load in a minute)."""
    f = raw[:, 0].astype(np.int32)
    line = raw[:, 1:].astype(np.int32).reshape(h, w, bpp)
    if not np.isin(f, (3, 4)).any():
        out = np.zeros((h, w, bpp), np.int32)
        prev = np.zeros((w, bpp), np.int32)
        for y in range(h):
            cur = line[y]
            if f[y] == 1: cur = np.cumsum(cur, axis=0) & 255
            elif f[y] == 2: cur = (cur + prev) & 255
            out[y] = cur; prev = cur
        return out.astype(np.uint8).reshape(h, w * bpp)
    # one guard row on top and one guard column on the left (zeros, as the specification's out-of-image neighbours)
    rec = np.zeros((h + 1, w + 1, bpp), np.int32)
    ys_all = np.arange(h)
    for dgl in range(w + h - 1):
        ys = ys_all[max(0, dgl - w + 1):min(h, dgl + 1)]
        xs = dgl - ys
        a = rec[ys + 1, xs]; b = rec[ys, xs + 1]; c = rec[ys, xs]
        fy = f[ys][:, None]
        pa = np.abs(b - c); pb = np.abs(a - c); pc = np.abs(a + b - 2 * c)
        paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        pred = np.where(fy == 1, a, np.where(fy == 2, b, np.where(fy == 3, (a + b) >> 1, np.where(fy == 4, paeth, 0))))
        rec[ys + 1, xs + 1] = (line[ys, xs] + pred) & 255
    return rec[1:, 1:].astype(np.uint8).reshape(h, w * bpp)


def to_gray(img, rgb_flag=1):
    """The grey image Tracking::GrabImageRGBD_wh hands to the Frame constructor (src/Tracking.cc:240-252; Camera.RGB: 1 in TUM1.yaml:29).  cv::imread delivers B, G, R;
    with Camera.RGB: 1 (TUM*.yaml) the reference calls cvtColor(.., CV_RGB2GRAY) on that BGR data, i.e. the weights 0.299 / 0.587 /
    0.114 land on B / G / R.  OpenCV's 8-bit path is fixed point: (c0 * 4899 + c1 * 9617 + c2 * 1868 + 8192) >> 14."""
    if img.ndim == 2:
        return img.astype(np.uint8)
    if img.shape[2] == 2:                                                  # grey + alpha (colour type 4): the alpha channel is dropped
        return img[..., 0].astype(np.uint8)
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))            # the PNG stores R, G, B (a fourth channel is alpha: ignored)
    c0, c2 = (b, r) if rgb_flag else (r, b)                                # channel order seen by cvtColor's "R" and "B" weights
    return ((c0 * 4899 + g * 9617 + c2 * 1868 + 8192) >> 14).astype(np.uint8)


def read_associations(path):
    out = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) >= 4 and not line.startswith("#"):
                out.append((p[1], p[3]))
    return out


def associate(root, max_dt=0.02):
    """rgb.txt + depth.txt -> pairs by nearest time stamp (the dataset's associate.py rule) when no association file is present"""
    def read(name):
        r = []
        with open(os.path.join(root, name)) as f:
            for line in f:
                if line.strip() and not line.startswith("#"):
                    t, fn = line.split()[:2]; r.append((float(t), fn))
        return r
    rgb, dep = read("rgb.txt"), read("depth.txt")
    # associate.py: all candidate pairs within max_dt, best |dt| first, each stamp used once; the result in time order
    tr = np.array([t for t, _ in rgb]); td = np.array([t for t, _ in dep])
    cand = []
    for i, t in enumerate(tr):
        lo, hi = np.searchsorted(td, t - max_dt), np.searchsorted(td, t + max_dt)
        cand += [(abs(td[j] - t), i, j) for j in range(lo, hi) if abs(td[j] - t) < max_dt]
    cand.sort()
    ui, uj, m = set(), set(), []
    for _, i, j in cand:
        if i not in ui and j not in uj:
            ui.add(i); uj.add(j); m.append((i, j))
    m.sort()
    return [(rgb[i][1], dep[j][1]) for i, j in m]


def load_sequence(root, limit=None, assoc=None):
    """-> (gray uint8 [n, h, w], depth uint16 [n, h, w]) of a TUM RGB-D sequence directory"""
    cands = [assoc] if assoc else [os.path.join(root, n) for n in ("associations.txt", "fr1_desk.txt", "associate.txt")]
    pairs = None
    for c in cands:
        if c and os.path.exists(c):
            pairs = read_associations(c); break
    if pairs is None:
        pairs = associate(root)
    if limit:
        pairs = pairs[:limit]
    if not pairs:
        raise ValueError("no RGB-D pairs under " + root)
    g = []; d = []
    for rf, df in pairs:
        with open(os.path.join(root, rf), "rb") as f:
            g.append(to_gray(decode_png(f.read())))
        with open(os.path.join(root, df), "rb") as f:
            dd = decode_png(f.read())
        if dd.dtype != np.uint16:
            raise ValueError("depth image is not 16-bit: " + df)
        d.append(dd)
    return np.stack(g), np.stack(d)
