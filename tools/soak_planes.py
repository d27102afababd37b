#!/usr/bin/env python3
"""Differential soak of the plane chain on synthetic depth built to produce TIES and edge cases: piecewise-planar scenes of a few random planes
(exactly planar before quantisation to the sensor's integer depth units), block-aligned and oblique boundaries, holes, depth beyond the range,
a noise level that is sometimes zero.  The lone-frame path (multi-head AHC, 512-thread flood) and batches (4 frames per wave / a wave per
frame) against the CPU oracle (test infrastructure).      python tools/soak_planes.py [scenes=200] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge


def scene(rng, w=640, h=480):
    fx, fy, cx, cy = 535.4, 539.2, 320.1, 247.6
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    rx, ry = (u - cx) / fx, (v - cy) / fy
    nplanes = int(rng.integers(1, 6))
    z = np.zeros((h, w))
    region = np.zeros((h, w), int)
    mode = int(rng.integers(0, 4))
    if mode == 3: nplanes = int(rng.choice([12, 24, 35, 48]))                            # a grid of many small planes (up to 48 of them; the library holds 64)
    if mode == 0: region = (u * nplanes // w)                                     # vertical strips
    elif mode == 1: region = ((u // 40) + (v // 40)) % nplanes                        # block-aligned checker (blocks are 10x10: boundaries on block edges)
    elif mode == 3:
        nx = int(np.ceil(np.sqrt(nplanes * 4 / 3))); ny = int(np.ceil(nplanes / nx)); region = np.minimum((v * ny // h) * nx + (u * nx // w), nplanes - 1)
    else:                                                                            # oblique half-planes
        for k in range(1, nplanes):
            a = rng.uniform(0, np.pi); c = rng.uniform(0.2, 0.8)
            region[(np.cos(a) * u / w + np.sin(a) * v / h) > c] = k
    for k in range(nplanes):
        n = rng.normal(0, 1, 3); n[2] = abs(n[2]) + 1.0; n /= np.linalg.norm(n)
        dist = rng.uniform(0.8, 4.5)
        zk = dist / (n[0] * rx + n[1] * ry + n[2])                                  # n . (rx z, ry z, z) = dist
        z[region == k] = zk[region == k]
    sigma = float(rng.choice([0.0, 0.0, 0.001, 0.004]))
    if sigma > 0: z = z + rng.normal(0, sigma, z.shape) * z * z
    d = np.clip(np.round(z * 5000.0), 0, 65535)
    q = int(rng.choice([1, 1, 5, 25]))                                              # coarser sensor steps: many equal depths
    d = (np.round(d / q) * q)
    if rng.uniform() < 0.5:                                                         # holes
        for _ in range(int(rng.integers(1, 8))):
            x0, y0 = int(rng.integers(0, w - 60)), int(rng.integers(0, h - 60)); d[y0:y0 + int(rng.integers(3, 60)), x0:x0 + int(rng.integers(3, 60))] = 0
    return np.clip(d, 0, 65535).astype(np.uint16)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xDEAF
    hvo = ge.package(); orc = ge.oracle()
    from test_peac_gpu import check
    rng = np.random.default_rng(seed)
    depths = [scene(rng) for _ in range(n)]
    ref = [orc.peac(d) for d in depths]
    bad = []
    def cmp(tag, i, lg, pg):
        try: check(lg, pg, ref[i][0], ref[i][1])
        except AssertionError as e: bad.append((tag, i, str(e)[:60]))
    ctx = hvo.Context()
    for i, d in enumerate(depths):
        lg, pg = ctx.compute_planes(d); cmp("lone", i, lg, pg)
    ctx.close()
    for B, env in ((8, {}), (40, {}), (40, {"HVO_PEAC_GL": "16"})):
        os.environ.update(env)
        ctx = hvo.Context(max_batch=B)
        for c0 in range(0, n, B):
            m = min(B, n - c0)
            dd = np.stack(depths[c0:c0 + m])
            ctx.batch_upload(np.zeros((m, 480, 640), np.uint8), dd); ctx.batch_run(hvo.STAGE_PLANES); res = ctx.batch_download(hvo.STAGE_PLANES)
            for b in range(m):
                if res[b]["status"] != 0: bad.append(("batch%d%s status %d" % (B, env, res[b]["status"]), c0 + b, ""))
                else: cmp("batch%d%s" % (B, env), c0 + b, res[b]["labels"], res[b]["planes"])
        ctx.close()
        for k in env: os.environ.pop(k, None)
    nplanes = [len(r[1]) for r in ref]
    print("RESULT planes soak scenes=%d seed=%#x planes per scene min/mean/max %d/%.1f/%d differing=%d %s" % (n, seed, min(nplanes), float(np.mean(nplanes)), max(nplanes), len(bad), bad[:8]))


if __name__ == "__main__":
    main()
