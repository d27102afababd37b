"""Deterministic synthetic RGB-D scenes.

No dataset ships with the reference (only association lists, Examples/RGB-D/associations/*.txt),
so benchmarks and parity tests run on generated frames with the TUM3 camera
(Examples/RGB-D/TUM3.yaml:8-11,34).  Everything that decides a pixel value is integer or
plain IEEE double arithmetic on a counter-based hash (splitmix64), so every host produces
the same bytes.

kinds:
  "lowtex"  stand-in for TUM fr3_structure_notexture_far: luminance ramp + 6 faint quads
  "std"     40 rotated rectangles / triangles with contrast 20..120 over the ramp
"""
import numpy as np

TUM3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_factor=5000.0)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _hash(seed, stream, n):
    """n 64-bit hashes for (seed, stream), vectorised."""
    with np.errstate(over="ignore"):
        base = _splitmix(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix(base + idx * np.uint64(0xD1342543DE82EF95))


def _randint(seed, stream, n, lo, hi):
    """integers in [lo, hi)"""
    return (lo + (_hash(seed, stream, n) >> np.uint64(33)).astype(np.int64) % (hi - lo)).astype(np.int64)


def intrinsics(w=640, h=480):
    s = w / 640.0
    return dict(fx=TUM3["fx"] * s, fy=TUM3["fy"] * s, cx=TUM3["cx"] * s, cy=TUM3["cy"] * s,
                depth_factor=TUM3["depth_factor"])


def _fill_convex(img, pts, delta):
    """add `delta` to pixels strictly inside the convex polygon pts (int vertices, any winding)"""
    h, w = img.shape
    xs = [p[0] for p in pts]; ys = [p[1] for p in pts]
    x0, x1 = max(min(xs), 0), min(max(xs), w - 1)
    y0, y1 = max(min(ys), 0), min(max(ys), h - 1)
    if x0 > x1 or y0 > y1:
        return
    yy, xx = np.mgrid[y0:y1 + 1, x0:x1 + 1]
    n = len(pts)
    area2 = sum(pts[i][0] * pts[(i + 1) % n][1] - pts[(i + 1) % n][0] * pts[i][1] for i in range(n))
    sgn = 1 if area2 > 0 else -1
    inside = np.ones(xx.shape, dtype=bool)
    for i in range(n):
        ax, ay = pts[i]; bx, by = pts[(i + 1) % n]
        cross = (bx - ax) * (yy - ay) - (by - ay) * (xx - ax)
        inside &= (cross * sgn) > 0
    img[y0:y1 + 1, x0:x1 + 1][inside] += delta


def make_gray(kind, seed, w=640, h=480):
    s = w / 640.0
    img = np.zeros((h, w), dtype=np.int32)
    img += (90 + (np.arange(h, dtype=np.int64) * 60) // max(h - 1, 1)).astype(np.int32)[:, None]
    if kind == "lowtex":
        nshape, cmin, cmax, noise = 6, 12, 13, 2
    elif kind == "std":
        nshape, cmin, cmax, noise = int(40 * s * s), 20, 121, 4
    else:
        raise ValueError(kind)
    r = _randint(seed, 1, nshape * 8, 0, 1 << 20)
    for i in range(nshape):
        q = r[8 * i: 8 * i + 8]
        cx, cy = int(q[0] % w), int(q[1] % h)
        big = kind == "lowtex"
        ext = int((90 if big else 25) * s) + int(q[2] % int((120 if big else 70) * s))
        ux = int(q[3] % (2 * ext + 1)) - ext
        uy = int(q[4] % (2 * ext + 1)) - ext
        if ux == 0 and uy == 0:
            ux = ext
        k = 6 + int(q[5] % 14)          # aspect 6/16 .. 19/16
        vx, vy = (-uy * k) // 16, (ux * k) // 16
        contrast = cmin + int(q[6] % (cmax - cmin))
        if q[7] & 1:
            contrast = -contrast
        if (q[7] >> 1) & 1 or big:
            pts = [(cx - ux - vx, cy - uy - vy), (cx + ux - vx, cy + uy - vy),
                   (cx + ux + vx, cy + uy + vy), (cx - ux + vx, cy - uy + vy)]
        else:
            pts = [(cx - ux - vx, cy - uy - vy), (cx + ux - vx, cy + uy - vy), (cx + vx, cy + vy)]
        _fill_convex(img, pts, contrast)
    nz = _randint(seed, 2, w * h, -noise, noise + 1).reshape(h, w)
    img = np.clip(img + nz, 0, 255)
    return img.astype(np.uint8)


def make_depth(seed, w=640, h=480, holes=True):
    """u16 depth (metres * 5000): floor + back wall + side wall + a slanted box, with
    z^2-proportional noise, a few rectangular dropouts and sparse zero pixels."""
    K = intrinsics(w, h)
    j = np.arange(w, dtype=np.float64)[None, :]
    i = np.arange(h, dtype=np.float64)[:, None]
    dx = (j - K["cx"]) / K["fx"]
    dy = (i - K["cy"]) / K["fy"]
    jit = _randint(seed, 10, 8, -50, 51).astype(np.float64) / 1000.0
    planes = [  # (nx, ny, nz, d): n.X = d
        (0.03 + jit[0] * 0.2, 0.97, 0.10, 1.20 + jit[1]),       # floor
        (0.05, 0.02 + jit[2] * 0.2, 1.0, 3.5 + jit[3] * 4),     # back wall
        (-0.96, 0.0, 0.28 + jit[4], 1.7 + jit[5]),              # left wall
    ]
    z = np.full((h, w), 1e9)
    for nx, ny, nz, d in planes:
        den = nx * dx + ny * dy + nz
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(den > 1e-9, d / den, 1e9)
        z = np.minimum(z, np.where(t > 0.2, t, 1e9))
    # a box face in front of the wall (creates depth discontinuities)
    bx0, by0 = int(w * 0.55) + int(jit[6] * 200), int(h * 0.30)
    bx1, by1 = bx0 + int(w * 0.22), by0 + int(h * 0.33)
    den = (0.25 * dx - 0.05 * dy + 1.0)
    zb = 2.1 / den
    box = np.zeros((h, w), dtype=bool)
    box[by0:by1, bx0:bx1] = True
    z = np.where(box & (zb < z), zb, z)
    z = np.where(z > 12.0, 0.0, z)
    # noise: sigma_z = 1.6e-3 z^2, Irwin-Hall(4) from integer uniforms (var = 4/12 each unit^2)
    u = _hash(seed, 11, w * h)
    s4 = ((u & np.uint64(0xFFFF)).astype(np.int64) + ((u >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
          + ((u >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64) + ((u >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64))
    g = (s4.astype(np.float64) - 2.0 * 65535.0) / (65536.0 * 0.5773502691896257)  # ~N(0,1)
    z = z + 1.6e-3 * z * z * g.reshape(h, w)
    d16 = np.clip(np.rint(z * K["depth_factor"]), 0, 65535).astype(np.uint16)
    if holes:
        r = _randint(seed, 12, 6 * 4, 0, 1 << 20)
        for k in range(6):
            x0 = int(r[4 * k] % w); y0 = int(r[4 * k + 1] % h)
            ww = 4 + int(r[4 * k + 2] % max(int(w * 0.06), 5)); hh = 4 + int(r[4 * k + 3] % max(int(h * 0.06), 5))
            d16[y0:y0 + hh, x0:x0 + ww] = 0
        sp = _hash(seed, 13, w * h).reshape(h, w)
        d16[(sp >> np.uint64(40)) % np.uint64(2000) == 0] = 0   # 0.05 % isolated zeros
    return d16


def make_frame(kind="std", seed=0x5EED0002, w=640, h=480):
    return make_gray(kind, seed, w, h), make_depth(seed, w, h)


def make_batch(kind, seed0, n, w=640, h=480):
    g = np.empty((n, h, w), dtype=np.uint8)
    d = np.empty((n, h, w), dtype=np.uint16)
    for k in range(n):
        g[k], d[k] = make_frame(kind, seed0 + k, w, h)
    return g, d


def make_sequence(kind, seed, n, w=640, h=480, max_step=4, pad=(192, 144)):
    """n frames of ONE scene seen through a window that drifts smoothly (<= max_step pixels per frame and axis): the
    stand-in for a camera sequence (`stream-573`).  Gray and depth are cut out of a larger scene;
    per-frame sensor noise is added to the gray image.  Returns (gray (n,h,w) u8, depth (n,h,w) u16, offsets (n,2) int)."""
    W, H = w + pad[0], h + pad[1]
    G = make_gray(kind, seed, W, H).astype(np.int32)
    D = make_depth(seed, W, H)
    r = _randint(seed, 20, 4, 0, 1 << 16).astype(np.float64) / 65536.0
    ox = np.zeros(n, np.int64); oy = np.zeros(n, np.int64)
    x = pad[0] / 2.0; y = pad[1] / 2.0
    for k in range(n):
        # a Lissajous-like drift whose per-frame step stays below max_step
        vx = max_step * 0.9 * np.sin(2 * np.pi * (k / 97.0 + r[0])); vy = max_step * 0.9 * np.cos(2 * np.pi * (k / 61.0 + r[1]))
        x = min(max(x + vx, 0.0), float(pad[0])); y = min(max(y + vy, 0.0), float(pad[1]))
        ox[k] = int(round(x)); oy[k] = int(round(y))
    g = np.empty((n, h, w), np.uint8); d = np.empty((n, h, w), np.uint16)
    for k in range(n):
        nz = _randint(seed + 7919 * (k + 1), 21, w * h, -2, 3).reshape(h, w)
        g[k] = np.clip(G[oy[k]:oy[k] + h, ox[k]:ox[k] + w] + nz, 0, 255).astype(np.uint8)
        d[k] = D[oy[k]:oy[k] + h, ox[k]:ox[k] + w]
    return g, d, np.stack([ox, oy], axis=1)
