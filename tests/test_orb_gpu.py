"""GPU parity: HIP ORB path (through the C ABI) vs the CPU oracle on the same seeded inputs.
Bar (BASELINE.json north_star): bit-exact keypoint coordinates / octaves / responses and
descriptor bytes; float orientations within 1e-4 (they are expected to be bit-equal)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ANGLE_TOL = 1e-4     # degrees, north_star tolerance for float orientations/responses


def check_orb(kp_g, d_g, kp_o, d_o):
    assert len(kp_g) == len(kp_o), (len(kp_g), len(kp_o))
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kp_g[f], kp_o[f]), f
    assert np.max(np.abs(kp_g["angle"] - kp_o["angle"]), initial=0) <= ANGLE_TOL
    assert np.array_equal(d_g, d_o)


@pytest.mark.parametrize("kind,seed", [("std", 0x5EED0002), ("lowtex", 0x5EED0001), ("std", 7), ("std", 8)])
def test_orb_parity_640(gpu_ctx, orc, synth, kind, seed):
    g = synth.make_gray(kind, seed)
    kp_o, d_o = orc.Orb().extract(g)
    kp_g, d_g = gpu_ctx.extract_orb(g)
    check_orb(kp_g, d_g, kp_o, d_o)


def test_orb_parity_random_noise(gpu_ctx, orc):
    """dense-texture stress: thousands of FAST candidates per level, quadtree fully exercised"""
    rng = np.random.default_rng(123)
    g = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    g[100:300, 200:500] = (g[100:300, 200:500] // 8 + 100)      # a calmer region
    kp_o, d_o = orc.Orb().extract(g)
    kp_g, d_g = gpu_ctx.extract_orb(g)
    check_orb(kp_g, d_g, kp_o, d_o)


@pytest.mark.parametrize("hh,ww", [(200, 640), (160, 704)])
def test_orb_wide_image_several_initial_nodes(hvo, orc, synth, hh, ww):
    """DistributeOctTree starts from round(width / height) nodes (ORBextractor.cc:541-560): 4 and 5 of them here, on a textured and a
    noise image (initial nodes filtered by x, the first round walks them forwards)"""
    big = synth.make_gray("std", 21, 704, 480)
    rng = np.random.default_rng(5)
    for g in (np.ascontiguousarray(big[:hh, :ww]), rng.integers(0, 256, (hh, ww), dtype=np.uint8)):
        kp_o, d_o = orc.Orb().extract(g)
        ctx = hvo.Context()
        try:
            kp_g, d_g = ctx.extract_orb(g)
        finally:
            ctx.close()
        assert len(kp_o) > 300
        check_orb(kp_g, d_g, kp_o, d_o)


@pytest.mark.parametrize("nfeat", [40, 300, 5000])
def test_orb_quota_sweep(hvo, orc, synth, nfeat):
    """the quadtree's three ways out: the quota met inside the largest-first loop (40, 300 per frame) and every node down to one key
    before the quota is met (5000: the list stops growing, ORBextractor.cc:666); noise and a low-texture frame; a blank frame has no key point"""
    rng = np.random.default_rng(9)
    frames = [rng.integers(0, 256, (480, 640), dtype=np.uint8), synth.make_gray("lowtex", 0x5EED0001), np.full((480, 640), 90, np.uint8)]
    ctx = hvo.Context(orb_nfeatures=nfeat)
    o = orc.Orb(nfeatures=nfeat)
    try:
        for g in frames:
            kp_o, d_o = o.extract(g)
            kp_g, d_g = ctx.extract_orb(g)
            check_orb(kp_g, d_g, kp_o, d_o)
        assert len(kp_g) == 0
    finally:
        ctx.close()


def test_orb_parity_1280(hvo, orc, synth):
    """BASELINE config 3 geometry: 1280x960, 2000 features"""
    g = synth.make_gray("std", 0x5EED0003, 1280, 960)
    kp_o, d_o = orc.Orb(nfeatures=2000).extract(g)
    ctx = hvo.Context(orb_nfeatures=2000)
    try:
        kp_g, d_g = ctx.extract_orb(g)
    finally:
        ctx.close()
    check_orb(kp_g, d_g, kp_o, d_o)


def test_orb_odd_geometry_and_stride(gpu_ctx, orc, synth):
    """non-multiple-of-4 width (exercises the scalar tail of the blur column pass) and a strided view"""
    big = synth.make_gray("std", 11, 640, 480)
    g = np.ascontiguousarray(big[:397, :501])
    kp_o, d_o = orc.Orb().extract(g)
    kp_g, d_g = gpu_ctx.extract_orb(g)
    check_orb(kp_g, d_g, kp_o, d_o)


def _orb_plan(hvo, ctx):
    import ctypes
    out = (ctypes.c_int * 4)()
    L = hvo.lib(); L.hvo_debug_orb_plan.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert L.hvo_debug_orb_plan(ctx.h, out) == 0
    return list(out)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("hh,ww", [(480, 640), (397, 501), (479, 638), (240, 322), (200, 270)])
def test_orb_both_paths_every_geometry(hvo, orc, synth, monkeypatch, fused, hh, ww):
    """The fused per-level pass (orb_level.hip) and the separate resize / FAST / blur kernels are two formulations of the same levels:
    both are forced (HVO_ORB_FUSED) onto even, odd and small geometries, and WHICH one ran is asserted -- the separate kernels are the
    fallback for a geometry whose FAST cells do not fit the LDS tile, and must not rot behind a path that always wins."""
    monkeypatch.setenv("HVO_ORB_FUSED", fused)
    big = synth.make_gray("std", 11, 704, 480)
    g = np.ascontiguousarray(big[:hh, :ww])
    kp_o, d_o = orc.Orb().extract(g)
    ctx = hvo.Context()
    try:
        kp_g, d_g = ctx.extract_orb(g)
        plan = _orb_plan(hvo, ctx)
    finally:
        ctx.close()
    check_orb(kp_g, d_g, kp_o, d_o)
    # the camera geometries take the fused pass; the two small ones have FAST cells that do not fit the LDS tile and fall back (exactly)
    if fused == "0" or (hh, ww) in ((480, 640), (397, 501), (479, 638)): assert plan[0] == int(fused), plan
    else: assert plan[0] == 0, plan
    assert len(kp_o) > 20


@pytest.mark.parametrize("fused,sched", [("1", None), ("0", None), ("1", "5"), ("1", "7")])
def test_orb_chunked_batch(hvo, orc, synth, monkeypatch, fused, sched):
    """orb_run walks a batch chunk by chunk through scratch slabs that exist for one chunk (HVO_ORB_CHUNK = 3 of 8 frames: chunks of
    3, 3, 2): every frame of every chunk against the oracle, both ORB paths, and under the overlap policies that make the other stages
    wait for the LAST chunk's FAST (ev_fast is recorded there)."""
    monkeypatch.setenv("HVO_ORB_CHUNK", "3"); monkeypatch.setenv("HVO_ORB_FUSED", fused)
    if sched: monkeypatch.setenv("HVO_SCHED", sched)
    gray, depth = synth.make_batch("std", 0x5EED3000, 8)
    gray[5] = synth.make_gray("lowtex", 0x5EED0001)
    ctx = hvo.Context(max_batch=8)
    try:
        ctx.batch_upload(gray, depth)
        stages = hvo.STAGE_ALL if sched else hvo.STAGE_ORB
        for _ in range(2):
            ctx.batch_run(stages)
            res = ctx.batch_download(stages)
            plan = _orb_plan(hvo, ctx)
            assert plan[0] == int(fused) and plan[1] == 3 and plan[3] == 3, plan
            o = orc.Orb()
            for b in range(8):
                kp_o, d_o = o.extract(gray[b])
                assert res[b]["status"] == 0
                check_orb(res[b]["kp"], res[b]["desc"], kp_o, d_o)
            if sched:                                              # the stages that waited for FAST
                lo, po = orc.peac(depth[7]); assert np.array_equal(res[7]["labels"], lo)
                kl_o, dl_o, _ = orc.line_extract(gray[7]); assert np.array_equal(res[7]["ldesc"], dl_o)
    finally:
        ctx.close()


def test_orb_flat_image_gives_nothing(gpu_ctx):
    kp, d = gpu_ctx.extract_orb(np.full((480, 640), 128, np.uint8))
    assert len(kp) == 0 and d.shape == (0, 32)


def test_orb_empty_image_like_reference(gpu_ctx):
    kp, d = gpu_ctx.extract_orb(np.zeros((0, 0), np.uint8))     # ORBextractor.cc:1044
    assert len(kp) == 0


def test_orb_wrong_dtype_is_an_error(gpu_ctx, hvo):
    with pytest.raises(hvo.HvoError):
        gpu_ctx.extract_orb(np.zeros((480, 640), np.float32))   # ORBextractor.cc:1048 assert


def test_orb_batch_matches_single(hvo, orc, synth):
    gray, _ = synth.make_batch("std", 0x5EED1000, 6)
    ctx = hvo.Context(max_batch=6)
    try:
        ctx.batch_upload(gray)
        ctx.batch_run(hvo.STAGE_ORB)
        res = ctx.batch_download(hvo.STAGE_ORB)
    finally:
        ctx.close()
    o = orc.Orb()
    for b in range(6):
        kp_o, d_o = o.extract(gray[b])
        assert res[b]["status"] == 0
        check_orb(res[b]["kp"], res[b]["desc"], kp_o, d_o)


def test_orb_properties_full_size(gpu_ctx, synth):
    """size-independent properties at the benchmark geometry: determinism, level-major order,
    per-level quota, border margins, angle range"""
    g = synth.make_gray("std", 0x5EED0002)
    kp1, d1 = gpu_ctx.extract_orb(g)
    kp2, d2 = gpu_ctx.extract_orb(g)
    assert np.array_equal(kp1, kp2) and np.array_equal(d1, d2)
    assert np.all(np.diff(kp1["octave"]) >= 0)
    quota = [217, 181, 151, 126, 105, 87, 73, 60]
    for l in range(8):
        assert np.sum(kp1["octave"] == l) <= quota[l] + 3
    s = 1.2 ** kp1["octave"]
    assert np.all(kp1["x"] / s >= 15.9) and np.all(kp1["y"] / s >= 15.9)
    assert np.all((kp1["angle"] >= 0) & (kp1["angle"] < 360))


SLAB_SCRIPT = r"""
import importlib, sys
import numpy as np
import torch
torch.cuda.init()                      # torch's HIP runtime first: a process that loads libhvo.so before torch cannot initialise torch.cuda
sys.path.insert(0, %r)
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth"); hd = importlib.import_module("hvo_amd.dist")
g, d = synth.make_batch("std", 0x5EED1000, 3)
ctx = hvo.Context(max_batch=3)
ctx.batch_upload(g, d); ctx.batch_run(hvo.STAGE_ALL)
res = ctx.batch_download(hvo.STAGE_ALL)
kc, lc, pc, sb = ctx.slab_layout()
assert sb == hd.slab_layout(hvo, kc, lc, pc)["size"]
slabs = hd.device_slabs(ctx, 3)
assert slabs.is_cuda and tuple(slabs.shape) == (3, sb)
back = hd.unpack_results(hvo, slabs.cpu().numpy(), kc, lc, pc)
for a, b in zip(back, res):
    assert a["status"] == b["status"]
    for k in ("kp", "desc", "kl", "ldesc", "linefn", "planes"):
        assert np.array_equal(a[k], b[k]), k
    assert len(a["kp"]) > 500 and len(a["kl"]) > 20 and len(a["planes"]) >= 3
# the slab that also carries the label image (HVO_SLAB_LABELS): same records, plus membershipImg bit for bit
kc2, lc2, pc2, sb2, lo2 = ctx.slab_layout(labels=True)
L2 = hd.slab_layout(hvo, kc2, lc2, pc2, label_shape=(480, 640))
assert sb2 == L2["size"] and lo2 == L2["labels"][0] == sb and sb2 == sb + 480 * 640
slabs2 = hd.device_slabs(ctx, 3, labels=True)
back2 = hd.unpack_results(hvo, slabs2.cpu().numpy(), kc2, lc2, pc2, label_shape=(480, 640))
for a, b in zip(back2, res):
    for k in ("kp", "desc", "kl", "ldesc", "linefn", "planes", "labels"):
        assert np.array_equal(a[k], b[k]), k
    assert (a["labels"] >= 0).mean() > 0.3
ctx.close()
print("slabs ok")
"""


def test_device_result_slabs_match_download():
    """hvo_batch_pack_results: the device-resident slabs (what the multi-GPU gather ships, written into a torch CUDA tensor)
    hold exactly what hvo_batch_download returns.  Own process: torch.cuda has to be initialised before libhvo.so is loaded."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", SLAB_SCRIPT % root], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "slabs ok" in p.stdout, p.stderr[-3000:]


NCCL_SCRIPT = r"""
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.init()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "%d"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
sys.path.insert(0, %r)
import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth"); hd = importlib.import_module("hvo_amd.dist")
g, d = synth.make_batch("std", 0x5EED1000, 2)
ctx = hvo.Context(max_batch=2)
ctx.batch_upload(g, d); ctx.batch_run(hvo.STAGE_ALL)
seen, sb = hd.gather_device_slabs(ctx, 2)               # device tensor -> all_gather_into_tensor (RCCL), world size 1
assert seen == 1 and sb == ctx.slab_layout()[3], (seen, sb)
seen, sb = hd.gather_device_slabs(ctx, 2, labels=True)  # the slabs that carry the label image too
assert seen == 1 and sb == ctx.slab_layout(labels=True)[3] == ctx.slab_layout()[3] + 480 * 640, (seen, sb)
ctx.close()
dist.destroy_process_group()
print("nccl gather ok")
"""

ORDER_SCRIPT = r"""
import importlib, sys
sys.path.insert(0, %r)
import __graft_entry__ as ge
hvo = ge.package(); hd = importlib.import_module("hvo_amd.dist")
ctx = hvo.Context(max_batch=1)                          # loads libhvo.so; torch has not been imported
try:
    hd.device_slabs(ctx, 1)
except RuntimeError as e:
    assert "before torch" in str(e), e
    print("order rule ok")
ctx.close()
"""


def test_rccl_gather_world1():
    """the one collective of the path (dist.gather_device_slabs: packed device slabs -> all_gather_into_tensor) through backend
    nccl (= RCCL) on hardware, world size 1: the code path an 8-GPU node runs, minus the peers"""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-c", NCCL_SCRIPT % (port, root)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "nccl gather ok" in p.stdout, p.stderr[-3000:]


def test_torch_load_order_rule_is_reported():
    """libhvo.so before torch: the package says so instead of letting torch.cuda fail obscurely"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", ORDER_SCRIPT % root], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "order rule ok" in p.stdout, p.stderr[-3000:]
