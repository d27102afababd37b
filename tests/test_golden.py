"""Committed vectors (tests/golden/frontend_v1.npz, made by tests/golden/make_golden.py).

They are outputs of this repository's CPU oracle on seeded synthetic frames -- the reference has no
fixtures of its own for this path and cannot run here ("parity unpinned").  The CPU
test pins the oracle against drift; the GPU test checks the HIP path against the same committed bytes."""
import hashlib
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_v1.npz"), allow_pickle=False)
SEEDS = [int(s) for s in G["seeds"]]


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def check_orb(i, kp, desc):
    assert len(kp) == int(G["orb%d_n" % i])
    for f in kp.dtype.names:
        assert np.array_equal(kp[:64][f], G["orb%d_kp_head" % i][f]), f        # bit-exact, floats included
    assert np.array_equal(desc[:64], G["orb%d_desc_head" % i])
    assert np.array_equal(sha(np.stack([kp["x"], kp["y"], kp["octave"].astype(np.float32)])), G["orb%d_xy_octave_sha" % i])
    assert np.array_equal(sha(desc), G["orb%d_desc_sha" % i])


def check_lsd(i, kl, ldesc, fn):
    assert len(kl) == int(G["lsd%d_n" % i])
    assert np.array_equal(np.stack([kl["sx"], kl["sy"], kl["ex"], kl["ey"]], 1), G["lsd%d_endpoints" % i])
    assert np.array_equal(sha(ldesc), G["lsd%d_desc_sha" % i])
    assert np.allclose(fn[:16], G["lsd%d_linefn_head" % i], rtol=1e-12, atol=1e-12)


def check_peac(i, lab, pl):
    ref = G["peac%d_planes" % i]
    assert len(pl) == len(ref)
    assert np.array_equal(pl["n_points"], ref["n_points"]) and np.array_equal(pl["rid"], ref["rid"])
    for f in ("normal", "center", "mse"):
        assert np.allclose(pl[f], ref[f], rtol=1e-9, atol=1e-12), f
    assert np.array_equal(np.bincount((lab + 1).ravel(), minlength=8)[:8], G["peac%d_label_hist" % i])
    assert np.array_equal(sha(lab), G["peac%d_label_sha" % i])


@pytest.mark.parametrize("i", [0, 1])
def test_oracle_reproduces_golden(orc, synth, i):
    g = synth.make_gray("std", SEEDS[i]); d = synth.make_depth(SEEDS[i])
    check_orb(i, *orc.Orb().extract(g))
    check_lsd(i, *orc.line_extract(g))
    check_peac(i, *orc.peac(d))


def test_oracle_knn2_golden(orc, synth):
    orb = orc.Orb()
    _, d0 = orb.extract(synth.make_gray("std", SEEDS[0])); _, d1 = orb.extract(synth.make_gray("std", SEEDS[1]))
    idx, dist = orc.hamming_knn2(d0[:256], d1)
    assert np.array_equal(idx, G["knn2_idx"]) and np.array_equal(dist, G["knn2_dist"])


@pytest.mark.gpu
@pytest.mark.parametrize("i", [0, 1])
def test_hip_path_reproduces_golden(gpu_ctx, synth, i):
    g = synth.make_gray("std", SEEDS[i]); d = synth.make_depth(SEEDS[i])
    check_orb(i, *gpu_ctx.extract_orb(g))
    check_lsd(i, *gpu_ctx.extract_lsd(g))
    check_peac(i, *gpu_ctx.compute_planes(d))


@pytest.mark.gpu
def test_hip_knn2_golden(gpu_ctx, synth):
    _, d0 = gpu_ctx.extract_orb(synth.make_gray("std", SEEDS[0])); _, d1 = gpu_ctx.extract_orb(synth.make_gray("std", SEEDS[1]))
    idx, dist = gpu_ctx.hamming_knn2(d0[:256], d1)
    assert np.array_equal(idx, G["knn2_idx"]) and np.array_equal(dist, G["knn2_dist"])
