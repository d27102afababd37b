#!/usr/bin/env python3
"""Differential soak on content the synthetic scene generator does not make: 1/f^a noise images (corners and short edges at every scale, the
statistics of natural images), thresholded and posterised versions of them, and depth maps that are smooth random surfaces with planar
patches.  Every stage, lone frames and batches of 16 / 64, against the CPU oracle (test infrastructure).   python tools/soak_textures.py [frames=192] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge


def noise_1f(rng, w, h, a):
    fy = np.fft.fftfreq(h)[:, None]; fx = np.fft.rfftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy); f[0, 0] = 1.0
    spec = (rng.normal(size=(h, w // 2 + 1)) + 1j * rng.normal(size=(h, w // 2 + 1))) / f ** a
    spec[0, 0] = 0
    x = np.fft.irfft2(spec, s=(h, w))
    return (x - x.mean()) / (x.std() + 1e-12)


def frame(rng, w=640, h=480):
    a = float(rng.choice([0.6, 1.0, 1.5, 2.0]))
    x = noise_1f(rng, w, h, a)
    kind = int(rng.integers(0, 4))
    if kind == 0: g = 128 + x * float(rng.choice([20, 45, 80]))
    elif kind == 1: g = np.where(x > float(rng.normal(0, 0.5)), 200.0, 60.0)                 # two-level: long curved edges
    elif kind == 2: g = np.round((128 + x * 50) / 32) * 32                                    # posterised: plateaus with equal values
    else: g = 128 + x * 40 + 60 * (noise_1f(rng, w, h, 2.5) > 0.5)                            # texture over large blobs
    g = np.clip(np.round(g), 0, 255).astype(np.uint8)
    z = 2.0 + 0.6 * noise_1f(rng, w, h, 3.0)                                                  # a smooth surface, 0.5-4 m
    if rng.uniform() < 0.7:                                                                   # planar patches
        u, v = np.meshgrid(np.arange(w), np.arange(h))
        for _ in range(int(rng.integers(1, 4))):
            x0, y0 = int(rng.integers(0, w - 200)), int(rng.integers(0, h - 150)); ww, hh = int(rng.integers(120, 400)), int(rng.integers(100, 300))
            zz = rng.uniform(1.0, 3.5) + rng.uniform(-0.002, 0.002) * (u - x0) + rng.uniform(-0.002, 0.002) * (v - y0)
            z[y0:y0 + hh, x0:x0 + ww] = zz[y0:y0 + hh, x0:x0 + ww]
    d = np.clip(np.round(np.clip(z, 0.3, 9.0) * 5000.0), 0, 65535).astype(np.uint16)
    if rng.uniform() < 0.3: d[noise_1f(rng, w, h, 2.0) > 1.2] = 0                             # holes with ragged borders
    return g, d


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x7E87
    hvo = ge.package(); orc = ge.oracle()
    from test_lsd_gpu import check as check_lines
    from test_peac_gpu import check as check_planes
    from test_orb_gpu import check_orb
    rng = np.random.default_rng(seed)
    fr = [frame(rng) for _ in range(n)]
    orb = orc.Orb()
    ref = [(orb.extract(g), orc.line_extract(g), orc.peac(d)) for g, d in fr]
    bad = []
    def cmp(tag, i, kp, desc, kl, ld, fn, lab, pl):
        for what, f in (("orb", lambda: check_orb(kp, desc, *ref[i][0])), ("lines", lambda: check_lines(kl, ld, fn, *ref[i][1])), ("planes", lambda: check_planes(lab, pl, *ref[i][2]))):
            try: f()
            except AssertionError as e: bad.append((tag, i, what, str(e)[:50]))
    ctx = hvo.Context()
    for i, (g, d) in enumerate(fr):
        kp, desc = ctx.extract_orb(g); kl, ld, fn = ctx.extract_lsd(g); lab, pl = ctx.compute_planes(d)
        cmp("lone", i, kp, desc, kl, ld, fn, lab, pl)
    ctx.close()
    from test_tail_gpu import check_tail
    full = hvo.STAGE_ALL | hvo.STAGE_LINES3D | hvo.STAGE_VP | hvo.STAGE_PLANE_TAIL | hvo.STAGE_GRIDS
    for B in (16, 64):
        ctx = hvo.Context(max_batch=B)
        ctx.set_tail_params(seed=500)
        for c0 in range(0, n, B):
            m = min(B, n - c0)
            ctx.batch_upload(np.stack([fr[c0 + b][0] for b in range(m)]), np.stack([fr[c0 + b][1] for b in range(m)])); ctx.batch_run(full); res = ctx.batch_download(hvo.STAGE_ALL); ctx.batch_download_tail(full, res)
            for b in range(m):
                r = res[b]
                if r["status"] != 0: bad.append(("batch%d" % B, c0 + b, "status %d" % r["status"], ""))
                else:
                    cmp("batch%d" % B, c0 + b, r["kp"], r["desc"], r["kl"], r["ldesc"], r["linefn"], r["labels"], r["planes"])
                    try: check_tail(r, fr[c0 + b][1], orc, 500 + b, (0.0, 640.0, 0.0, 480.0))      # the rest of the Frame constructor
                    except AssertionError as e: bad.append(("batch%d" % B, c0 + b, "tail", str(e)[:50]))
        ctx.close()
    print("RESULT textures soak frames=%d seed=%#x key points %.0f, lines %.0f, planes %.1f per frame (means) differing=%d %s" % (
        n, seed, np.mean([len(r[0][0]) for r in ref]), np.mean([len(r[1][0]) for r in ref]), np.mean([len(r[2][1]) for r in ref]), len(bad), bad[:8]))


if __name__ == "__main__":
    main()
