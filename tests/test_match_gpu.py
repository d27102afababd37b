"""GPU parity of the Hamming kernels (bit-exact integer work)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_desc(n, seed, dup=0):
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i in range(dup):               # force distance ties / collisions
        d[rng.integers(0, n)] = d[rng.integers(0, n)]
    return d


@pytest.mark.parametrize("nq,nt", [(1, 1), (3, 2), (200, 200), (1000, 1004), (2000, 2000), (65, 63)])
def test_knn2_and_matrix(gpu_ctx, orc, nq, nt):
    q = _rand_desc(nq, 1, dup=nq // 10); t = _rand_desc(nt, 2, dup=nt // 5)
    if nq > 2 and nt > 2:
        t[: min(nq, nt) // 2] = q[: min(nq, nt) // 2]          # exact matches (distance 0)
    assert np.array_equal(gpu_ctx.hamming_matrix(q, t), orc.hamming_matrix(q, t))
    ig, dg = gpu_ctx.hamming_knn2(q, t)
    io, do = orc.hamming_knn2(q, t)
    assert np.array_equal(ig, io) and np.array_equal(dg, do)


def test_knn2_ties_go_to_lower_index(gpu_ctx):
    q = np.zeros((1, 32), np.uint8)
    t = np.zeros((130, 32), np.uint8); t[:, 0] = 1      # all at distance 1
    idx, dist = gpu_ctx.hamming_knn2(q, t)
    assert idx.tolist() == [[0, 1]] and dist.tolist() == [[1, 1]]


def test_all_zero_vs_all_one(gpu_ctx):
    assert gpu_ctx.hamming_matrix(np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8))[0, 0] == 256


def test_empty_inputs(gpu_ctx):
    idx, dist = gpu_ctx.hamming_knn2(np.zeros((2, 32), np.uint8), np.zeros((0, 32), np.uint8))
    assert idx.tolist() == [[-1, -1], [-1, -1]]
    assert gpu_ctx.hamming_matrix(np.zeros((0, 32), np.uint8), np.zeros((5, 32), np.uint8)).shape == (0, 5)


def test_match_nnr(gpu_ctx, orc, hvo):
    d1 = _rand_desc(200, 5); d2 = _rand_desc(180, 6)
    d2[:60] = d1[:60]; d2[10, 0] ^= 0x0F
    for nnr in (0.6, 0.9, 0.95):
        ng, mg = hvo.LSDmatcher(gpu_ctx).match(d1, d2, nnr)
        no, mo = orc.match_nnr(d1, d2, nnr)
        assert ng == no and np.array_equal(mg, mo)
    assert hvo.ORBmatcher(gpu_ctx).DescriptorDistance(d1[0], d1[1]) == int(orc.hamming_matrix(d1[:1], d1[1:2])[0, 0])


@pytest.mark.parametrize("mutual", [False, True])
def test_frame_bf_match_parity(gpu_ctx, orc, synth, mutual):
    """LSDmatcher::FrameBFMatch / SearchDouble core (LSDmatcher.cpp:942-966, 902-939, lineDescriptorMAD 1110-1135)"""
    g1 = synth.make_gray("std", 0x5EED0002)
    g2 = np.roll(np.roll(g1, 2, axis=0), 3, axis=1)
    _, d1, _ = orc.line_extract(g1); _, d2, _ = orc.line_extract(g2)
    for th, ratio in ((50.0, 0.9), (80.0, 0.75)):
        no, mo = orc.frame_bf_match(d1, d2, th, ratio, mutual)
        ng, mg = gpu_ctx.frame_bf_match(d1, d2, th, ratio, mutual)
        assert ng == no and np.array_equal(mg, mo)
    assert orc.frame_bf_match(d1, d2, 80.0, 0.9, mutual)[0] > 10
    # fewer than two train descriptors: knnMatch(k=2) cannot rank -> no matches
    n, m = gpu_ctx.frame_bf_match(d1, d2[:1], 50.0, 0.9, mutual)
    assert n == 0 and np.all(m == -1)
    n, m = gpu_ctx.frame_bf_match(d1[:0], d2, 50.0, 0.9, mutual)
    assert n == 0 and len(m) == 0
