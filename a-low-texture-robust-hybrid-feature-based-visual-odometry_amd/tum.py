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
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f = int(raw[y, 0]); line = raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        elif f == 1:                                   # Sub: a running sum per byte lane, modulo 256
            cur = line.copy().reshape(-1, bpp)
            cur = (np.cumsum(cur, axis=0) & 255).reshape(-1)
        else:                                          # Average / Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                if f == 3:
                    p = (a + b) >> 1
                else:
                    c = prev[x - bpp] if x >= bpp else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    if depth == 16:
        arr = out.reshape(h, w, ch, 2)
        arr = (arr[..., 0].astype(np.uint16) << 8) | arr[..., 1]
    else:
        arr = out.reshape(h, w, ch)
    return arr[:, :, 0] if ch == 1 else arr


def to_gray(img, rgb_flag=1):
    """The grey image Tracking::GrabImageRGBD_wh hands to the Frame constructor (src/Tracking.cc:240-252; Camera.RGB: 1 in TUM1.yaml:29).  cv::imread delivers B, G, R;
    with Camera.RGB: 1 (TUM*.yaml) the reference calls cvtColor(.., CV_RGB2GRAY) on that BGR data, i.e. the weights 0.299 / 0.587 /
    0.114 land on B / G / R.  OpenCV's 8-bit path is fixed point: (c0 * 4899 + c1 * 9617 + c2 * 1868 + 8192) >> 14."""
    if img.ndim == 2:
        return img.astype(np.uint8)
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))            # the PNG stores R, G, B
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
    td = np.array([t for t, _ in dep])
    pairs = []; used = set()
    for t, fn in rgb:
        j = int(np.argmin(np.abs(td - t)))
        if abs(td[j] - t) < max_dt and j not in used:
            used.add(j); pairs.append((fn, dep[j][1]))
    return pairs


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
